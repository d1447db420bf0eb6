// rt_wide.h — traversal of the 8-wide tree with quantised inner boxes (trees that are read through the caches: rt_wide.hip builds it
// from the binary tree).
//
// Why: a ray through the binary tree of the 5 M-triangle mesh makes 37 DEPENDENT 64-B node fetches (depth 25) and the closest-hit
// kernel sits at a fifth of the vector ALU's time — it waits for memory.  Only a LEAF's box has reference arithmetic to reproduce
// (Core/Acceleration/SingleObjectBvhNode.cs:10-16: the object is asked only if its own AxisAlignedBoundingBox.Hit passes); inner
// boxes merely have to be conservative (BinaryBvhNode.cs:13-37 returns what its children return), and any-hit has no box test at
// all in the reference (BinaryBvhNode.cs:41).  So inner nodes hold EIGHT children whose boxes are 8-bit offsets in the node's own
// frame, rounded outward: one 104-B record per visit instead of three 64-B ones, a third of the dependent fetches, and a node array
// of a third the size (5 M triangles: 320 MB -> 94 MB, inside the 256 MB Infinity Cache).  The leaf's exact box sits in front
// of a copy of its primitive in a leaf-ordered table: one fetch gives the reference's box test and the intersector's operands.
//
// Node record (104 B; WideNode in rt_wide.hip writes it):
//   +0   org.x, org.y, org.z, ex << 23 | imask      frame origin = min corner; e = biased exponent of the axis scale 2^(e-127), in a float's exponent field
//   +16  child_base, leaf_base, ey << 23 | lmask, ez << 23      first inner child / first leaf record; imask, lmask: which slots hold an inner node / a leaf
//   +32  lo.x[0..7] hi.x[0..7] lo.x[0..7]           one byte per slot: plane = org + q * 2^(e-127), rounded outward (an empty slot: lo = 255, hi = 0 and in neither mask)
//   +56  the same for y, +80 for z
// The low planes stand twice so that the ray's direction picks its (near planes, far planes) pair by ADDRESS: a ray along +axis reads
// 16 B at +0 of the block (lo, hi), a ray against it at +8 (hi, lo) — the trick of the LDS image (rt_kernels.h: LdsTree); the 80-B
// record of round 3's first form held each plane once and a visit chose with twelve v_cndmask.
// The k-th inner child (slots in ascending order) is node child_base + k, the k-th leaf is record leaf_base + k.
// Slots are assigned by octant (Ylitie, Karras, Laine 2017): a child on the (-x, -y, -z) side of the node sits in slot 0, on
// (+x, +y, +z) in slot 7, so visiting the hit slots by descending (slot XOR (7 - ray octant)) is front to back with no sorting.
//
// Box arithmetic of a visit (both kinds of ray): t = q * (2^e * inv) + ((org - o) * inv -+ pad * |inv|), one cvt and one FMA per
// plane, no selects.  `pad` makes it conservative against the reference's own rounding of the leaf test (closest hit: 1e-6 of the farthest
// distance from the ray origin to anything in the scene — the FMA form and the reference's (plane - o) * inv differ by at most
// 3.6e-7 of that, see wide_ray_closest) or is the any-hit pad of rt_device.h (any_box); inv is clamped (make_wide_ray) so that an
// axis-parallel ray gives +-huge, not inf - inf.
#pragma once
#include "rt_device.h"

namespace rtd {

constexpr uint32_t WIDE_NODE_BYTES = 104;
constexpr uint32_t WIDE_AXIS_BYTES = 24, WIDE_HEAD_BYTES = 32;
constexpr uint32_t WIDE_MAX_NODES = 1u << 24;      // a visit forms node * stride with v_mul_u32_u24 and addresses the record by a 32-bit offset
constexpr uint32_t WIDE_LEAF_BOX_BYTES = 32;     // (min.xyz, object) (max.xyz, object code): then the primitive's record when the tree holds one type (sc.wide_inline)

struct WideTree {
    static constexpr bool wide = true;
    static constexpr bool list = false;
    static constexpr bool vote = true;
    uint2* stack; int stride;                    // this lane's column of 64-bit entries: (base, hits | imask << 8)
    RT_FN void push(int sp, uint2 v) const { stack[sp * stride] = v; }
    RT_FN uint2 pop(int sp) const { return stack[sp * stride]; }
};

// (WideRay — what a node visit needs of the ray — is declared in rt_device.h)
// `lim` = WIDE_INV_LIMIT / |d|.  A component of 1/d beyond it belongs to a ray that is parallel to that axis for every purpose (it
// moves less than 1e-12 of its length along it): its slab distances are then +-(distance to the plane) * lim, at least
// pad * lim = 1e6 scene sizes / |d| on the far side — beyond any hit — instead of inf - inf or an overflow of q * S (a scene 1e17
// units across times 1/d = 1e24).
constexpr float WIDE_INV_LIMIT = ANY_INV_LIMIT;
// Any-hit rays enter children and test leaves in SLOT order (RT_WIDE_ANY_ORDERED 0): whether a ray is occluded does not depend on the
// order, most shadow rays are not occluded and have to see everything their box pad lets through anyway, and the two mask
// permutations of an ordered visit are 30 of its 230 vector instructions.  Closest-hit rays go front to back.
#ifndef RT_WIDE_ANY_ORDERED
#define RT_WIDE_ANY_ORDERED 0
#endif
template <bool ORDERED = true>
RT_FN WideRay make_wide_ray(const Ray& r, f3 inv, float pad, float lim) {
    WideRay w; w.o = r.o;
    w.inv = mk3(__builtin_fminf(__builtin_fmaxf(inv.x, -lim), lim), __builtin_fminf(__builtin_fmaxf(inv.y, -lim), lim), __builtin_fminf(__builtin_fmaxf(inv.z, -lim), lim));
    w.pt = mk3(pad * fabsf(w.inv.x), pad * fabsf(w.inv.y), pad * fabsf(w.inv.z));
    const uint32_t gx = __builtin_signbitf(inv.x) ? 1u : 0u, gy = __builtin_signbitf(inv.y) ? 1u : 0u, gz = __builtin_signbitf(inv.z) ? 1u : 0u;
    w.ax = WIDE_HEAD_BYTES + 8u * gx; w.ay = WIDE_HEAD_BYTES + WIDE_AXIS_BYTES + 8u * gy; w.az = WIDE_HEAD_BYTES + 2u * WIDE_AXIS_BYTES + 8u * gz;
    w.octinv = ORDERED ? 7u ^ (gx | (gy << 1) | (gz << 2)) : 0u;
    return w;
}
// Closest hit.  The reference's leaf test computes fl(fl(plane - o) * inv): at most 2 * 2^-24 |plane - o| away from the real number.
// The visit's fl(q * S + fl(fl(fl(org - o) * inv) -+ pt)) is at most 4 * 2^-24 D away (D = the farthest any point of the scene is from the
// ray origin), so planes moved outward by 1e-6 D (2.8 times the sum) can never cull a leaf whose own test would pass.
RT_FN WideRay wide_ray_closest(const SceneDev& sc, const Ray& r, f3 inv_exact) {
    const f3 c = mk3(sc.bounds_centre[0], sc.bounds_centre[1], sc.bounds_centre[2]);
    const float far_side = __builtin_amdgcn_sqrtf(distance_sq(r.o, c)) + sc.bounds_radius;
    return make_wide_ray(r, inv_exact, (1e-6f * far_side) + 1e-30f, WIDE_INV_LIMIT * __builtin_amdgcn_rsqf(dot(r.d, r.d)));
}
// Any hit: the pad of any_box (rt_device.h) — two looks, first with none.
RT_FN WideRay wide_ray_any(const SceneDev& sc, const WideTree&, const Ray& r, float kMax, float cap, int mode, bool& again) {
    const f3 c = mk3(sc.bounds_centre[0], sc.bounds_centre[1], sc.bounds_centre[2]);
    const float len = __builtin_amdgcn_sqrtf(dot(r.d, r.d));
    const float far_side = __builtin_amdgcn_sqrtf(distance_sq(r.o, c)) + sc.bounds_radius;
    const float range = kMax * len;
    const float reach = (range < far_side ? range : far_side) + sc.max_object_radius;
    float round = 0.0f, rm = 3e38f;
    if (sc.min_round_radius < 3e38f) {
        rm = sc.min_round_radius;
        round = __builtin_fminf((__builtin_amdgcn_sqrtf((rm * rm) + (1.1e-6f * reach * reach)) - rm) * 1.02f, cap * rm);
    }
    again = mode == 0 && round > (sc.any_again * rm);
    const f3 inv = mk3(rcp_fast(r.d.x), rcp_fast(r.d.y), rcp_fast(r.d.z));
    const float lim = WIDE_INV_LIMIT * rcp_fast(len);
    if (again) return make_wide_ray<RT_WIDE_ANY_ORDERED>(r, inv, 0.0f, lim);
    // (+ the FMA form's own rounding against far_side, as for closest hit: a visit places a plane within 2.4e-7 far_side)
    return make_wide_ray<RT_WIDE_ANY_ORDERED>(r, inv, sc.any_pad_scale * (round + (1e-6f * reach) + 1e-5f + (2.4e-7f * __builtin_fmaxf(__builtin_fmaxf(fabsf(r.o.x), fabsf(r.o.y)), fabsf(r.o.z))) + (4e-7f * far_side)), lim);
}

RT_FN uint32_t lane_id_() { return threadIdx.x & 63u; }
RT_FN float ubyte_f(uint32_t w, int i) { return (float)((w >> (8 * i)) & 0xFFu); }      // v_cvt_f32_ubyte<i>

// position p of an 8-bit mask -> position p ^ x
RT_FN uint32_t xor_permute8(uint32_t m, uint32_t x) {
    if (x & 4u) m = ((m & 0x0Fu) << 4) | ((m >> 4) & 0x0Fu);
    if (x & 2u) m = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
    if (x & 1u) m = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
    return m;
}

// the same for two masks side by side (bits 0-7 and 8-15)
#ifndef RT_WIDE_ADDC
#define RT_WIDE_ADDC 1
#endif
#ifndef RT_WIDE_JOINT_PERMUTE
#define RT_WIDE_JOINT_PERMUTE 0
#endif
RT_FN uint32_t xor_permute8x2(uint32_t m, uint32_t x) {
    if (x & 4u) m = ((m & 0x0F0Fu) << 4) | ((m >> 4) & 0x0F0Fu);
    if (x & 2u) m = ((m & 0x3333u) << 2) | ((m >> 2) & 0x3333u);
    if (x & 1u) m = ((m & 0x5555u) << 1) | ((m >> 1) & 0x5555u);
    return m;
}

struct __attribute__((aligned(8))) WideQuad { uint32_t x, y, z, w; };      // 16 B of a node record (records and their axis blocks are 8-B aligned: one global_load_dwordx4 all the same)
// One node: which of its (up to) eight child boxes the ray's [kMin, kMax] meets.  `hits` in slot order; the caller splits it by imask / lmask.
struct WideVisit { uint32_t hits, imask, lmask, child_base, leaf_base; };
RT_FN WideVisit wide_visit(const SceneDev& sc, const WideRay& wr, uint32_t node, float kMin, float kMax) {
    const char* base = (const char*)sc.wide_nodes;
    const uint32_t off = __umul24(node, sc.wide_stride);      // (rt_lib.hip: fewer than 2^24 nodes, the array below 4 GB)
    const WideQuad h0 = *(const WideQuad*)(base + off), h1 = *(const WideQuad*)(base + (off + 16u));
    const WideQuad rx = *(const WideQuad*)(base + (off + wr.ax)), ry = *(const WideQuad*)(base + (off + wr.ay)), rz = *(const WideQuad*)(base + (off + wr.az));      // (near[0..3], near[4..7], far[0..3], far[4..7])
    __builtin_amdgcn_sched_barrier(0);           // the five loads in flight before the first wait
    const float sx = __uint_as_float(h0.w & 0x7F800000u), sy = __uint_as_float(h1.z & 0x7F800000u), sz = __uint_as_float(h1.w);
    const float Sx = sx * wr.inv.x, Sy = sy * wr.inv.y, Sz = sz * wr.inv.z;
    const float Ax = (__uint_as_float(h0.x) - wr.o.x) * wr.inv.x, Ay = (__uint_as_float(h0.y) - wr.o.y) * wr.inv.y, Az = (__uint_as_float(h0.z) - wr.o.z) * wr.inv.z;
    const v2f Bx = (v2f){Ax, Ax} + (v2f){-wr.pt.x, wr.pt.x}, By = (v2f){Ay, Ay} + (v2f){-wr.pt.y, wr.pt.y}, Bz = (v2f){Az, Az} + (v2f){-wr.pt.z, wr.pt.z};      // (near, far) addends: one v_pk_add_f32 each (neg_lo)
    const v2f SSx = {Sx, Sx}, SSy = {Sy, Sy}, SSz = {Sz, Sz};
    const uint32_t nx[2] = {rx.x, rx.y}, fx[2] = {rx.z, rx.w}, ny[2] = {ry.x, ry.y}, fy[2] = {ry.z, ry.w}, nz[2] = {rz.x, rz.y}, fz[2] = {rz.z, rz.w};
    uint32_t hits = 0;
#pragma unroll
    for (int c = 7; c >= 0; c--) {                   // slot 7 first: `hits` is shifted up as the slots below it arrive
        const int w = c >> 2, b = c & 3;
        const v2f qx = {ubyte_f(nx[w], b), ubyte_f(fx[w], b)}, qy = {ubyte_f(ny[w], b), ubyte_f(fy[w], b)}, qz = {ubyte_f(nz[w], b), ubyte_f(fz[w], b)};
        const v2f tx = __builtin_elementwise_fma(qx, SSx, Bx), ty = __builtin_elementwise_fma(qy, SSy, By), tz = __builtin_elementwise_fma(qz, SSz, Bz);
        const float en = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(tx.x, kMin), ty.x), tz.x);      // a NaN distance (0 * inf cannot happen with the clamp; inf - inf can) never replaces a bound
        const float ex = __builtin_fminf(__builtin_fminf(__builtin_fminf(tx.y, kMax), ty.y), tz.y);
        // closed interval: a box met on a face or an edge is entered.  hits = 2 * hits + !(ex < en): compare into vcc, add with carry
        // (the compiler's own form is compare, select a constant, shift / or: 2.5 instead of 2 instructions per box)
#if RT_WIDE_ADDC
        asm("v_cmp_nlt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %3, %3, vcc" : "=v"(hits) : "v"(ex), "v"(en), "v"(hits) : "vcc");
#else
        hits = (hits + hits) + (!(ex < en) ? 1u : 0u);
#endif
    }
    WideVisit v;
    v.imask = h0.w & 0xFFu; v.lmask = h1.z & 0xFFu; v.child_base = h1.x; v.leaf_base = h1.y;
    v.hits = hits & (v.imask | v.lmask);
    return v;
}

// The lane's traversal state.  G = the group of inner children still to enter (hits by PRIORITY position = slot ^ octinv in bits 0-7,
// imask in bits 8-15; the nodes themselves are child_base + rank among the inner slots); P = the leaves of the last node still to
// test (hits by priority position in bits 0-7, lmask in bits 8-15).  The newest group lives in registers, older ones in the lane's LDS column:
// entering a child while siblings remain pushes the group, an exhausted group pops one — at most one entry per level.
struct WideCursor { uint32_t gbase, gbits, pbase, pbits; int sp; };
RT_FN WideCursor wide_idle() { WideCursor c; c.gbase = 0; c.gbits = 0; c.pbase = 0; c.pbits = 0; c.sp = 0; return c; }
RT_FN WideCursor wide_start(const WideRay& wr) {      // the root as a one-child group: node 0 in slot 0 (octinv = 0 for a ray that takes its children in slot order)
    WideCursor c = wide_idle(); c.gbits = (1u << wr.octinv) | (1u << 8); return c;
}
RT_FN bool wide_can_step(const WideCursor& c) { return (c.gbits & 0xFFu) != 0u && (c.pbits & 0xFFu) == 0u; }
RT_FN bool wide_has_leaf(const WideCursor& c) { return (c.pbits & 0xFFu) != 0u; }
RT_FN bool wide_busy(const WideCursor& c) { return ((c.gbits | c.pbits) & 0xFFu) != 0u; }
RT_FN void wide_stop(WideCursor& c) { c.gbits = 0; c.pbits = 0; c.sp = 0; }

// Enter the nearest child still in G and test its children.  Returns the number of boxes tested.
template <bool ORDERED = true>
RT_FN uint32_t wide_step(const SceneDev& sc, const WideTree& tree, const WideRay& wr, float kMin, float kMax, WideCursor& c) {
    const uint32_t pi = 31u - (uint32_t)__builtin_clz(c.gbits & 0xFFu);
    const uint32_t slot = ORDERED ? pi ^ wr.octinv : pi, imask = (c.gbits >> 8) & 0xFFu;
    const uint32_t node = c.gbase + (uint32_t)__builtin_popcount(imask & ((1u << slot) - 1u));
    const uint32_t rest = c.gbits & ~(1u << pi);                 // the siblings still to enter
    const WideVisit v = wide_visit(sc, wr, node, kMin, kMax);
    uint32_t ph = v.hits & v.lmask, inner = v.hits & v.imask, gh = inner;
    if (ORDERED) {
#if RT_WIDE_JOINT_PERMUTE
        const uint32_t both = xor_permute8x2(ph | (inner << 8), wr.octinv);      // both masks in one pass
        ph = both & 0xFFu; gh = both >> 8;
#else
        ph = xor_permute8(ph, wr.octinv);
#endif
    }
    c.pbase = v.leaf_base; c.pbits = ph | (v.lmask << 8);
    if (inner) {                                                  // descend: the siblings wait in the lane's column
        if (rest & 0xFFu) { tree.push(c.sp, make_uint2(c.gbase, rest)); c.sp++; }
#if !RT_WIDE_JOINT_PERMUTE
        if (ORDERED) gh = xor_permute8(inner, wr.octinv);
#endif
        c.gbase = v.child_base; c.gbits = gh | (v.imask << 8);
    } else if (rest & 0xFFu) c.gbits = rest;                      // nothing below: on with the siblings
    else if (c.sp) { c.sp--; const uint2 e = tree.pop(c.sp); c.gbase = e.x; c.gbits = e.y; }
    else c.gbits = 0;
    return (uint32_t)__builtin_popcount(v.imask | v.lmask);
}
// the next leaf of P (the nearest slot first): its record's index
template <bool ORDERED = true>
RT_FN uint32_t wide_take_leaf(WideCursor& c, const WideRay& wr) {
    const uint32_t pi = 31u - (uint32_t)__builtin_clz(c.pbits & 0xFFu);
    const uint32_t slot = ORDERED ? pi ^ wr.octinv : pi, lmask = (c.pbits >> 8) & 0xFFu;
    c.pbits &= ~(1u << pi);
    return c.pbase + (uint32_t)__builtin_popcount(lmask & ((1u << slot) - 1u));
}

// the wave's vote (rt_device.h: step_vote): lanes that can enter a node against lanes that must test a leaf first
#ifndef RT_WIDE_VOTE_A
#define RT_WIDE_VOTE_A 1
#define RT_WIDE_VOTE_B 3
#endif
RT_FN bool wide_vote(const WideCursor& c) {
    const int n_step = __popcll(__ballot(wide_can_step(c)));
    const int n_blocked = __popcll(__ballot(wide_has_leaf(c)));
    return n_step != 0 && n_step * RT_WIDE_VOTE_A >= n_blocked * RT_WIDE_VOTE_B;
}

// A leaf.  Closest hit: the object's own box first, in the reference's arithmetic (AxisAlignedBoundingBox.Hit through slab():
// SingleObjectBvhNode.cs:12-16), then Hittable.TryHit.  The record's primitive copy (when the tree holds one type) arrives with the box.
template <class C>
RT_FN void wide_leaf_closest(const SceneDev& sc, uint32_t leaf, const Ray& r, f3 inv_exact, float kMin, Closest& best, TravCount& cnt) {
    const char* lp = sc.wide_leaves + (size_t)leaf * sc.wide_leaf_stride;
    const float4 b0 = *(const float4*)lp, b1 = *(const float4*)(lp + 16);
    const uint32_t obj = __float_as_uint(b0.w), code = __float_as_uint(b1.w);
    float entry, exit_;
    cnt.nodes++;
    if (RT_HAS_PRIM(C, RT_HIT_SPHERE) && sc.wide_inline == RT_HIT_SPHERE) {
        const float4 s = *(const float4*)(lp + WIDE_LEAF_BOX_BYTES);
        slab(mk3(b0.x, b0.y, b0.z), mk3(b1.x, b1.y, b1.z), r.o, inv_exact, kMin, entry, exit_);
        if (!slab_hit(entry, exit_, best.k)) { cnt.box_miss++; return; }
        float k; cnt.prims++;
        if (try_sphere(s, r, kMin, best.k, k)) { consider(best, k, obj, 0.0f); cnt.prim_hit++; }
        return;
    }
    if ((RT_HAS_PRIM(C, RT_HIT_TRIANGLE) || RT_HAS_PRIM(C, RT_HIT_QUAD)) && (sc.wide_inline == RT_HIT_TRIANGLE || sc.wide_inline == RT_HIT_QUAD)) {
        const float4* q = (const float4*)(lp + WIDE_LEAF_BOX_BYTES);
        slab(mk3(b0.x, b0.y, b0.z), mk3(b1.x, b1.y, b1.z), r.o, inv_exact, kMin, entry, exit_);
        if (!slab_hit(entry, exit_, best.k)) { cnt.box_miss++; return; }
        float k, u, v, nd; cnt.prims++;
        const bool hit = (RT_HAS_PRIM(C, RT_HIT_TRIANGLE) && (!RT_HAS_PRIM(C, RT_HIT_QUAD) || sc.wide_inline == RT_HIT_TRIANGLE)) ? try_quad<true>(q, r, kMin, best.k, k, u, v, nd) : try_quad<false>(q, r, kMin, best.k, k, u, v, nd);
        if (hit) { consider(best, k, obj, 0.0f); cnt.prim_hit++; }
        return;
    }
    slab(mk3(b0.x, b0.y, b0.z), mk3(b1.x, b1.y, b1.z), r.o, inv_exact, kMin, entry, exit_);
    if (!slab_hit(entry, exit_, best.k)) { cnt.box_miss++; return; }
    float k; cnt.prims++;
    if (try_object_solid<C>(sc, code >> OBJ_TYPE_SHIFT, code & OBJ_INDEX_MASK, r, kMin, best.k, k)) { consider(best, k, obj, 0.0f); cnt.prim_hit++; }
}
// Any hit: Hittable.FastTryHit, no box test (BinaryBvhNode.cs:41).
template <class C>
RT_FN bool wide_leaf_any(const SceneDev& sc, uint32_t leaf, const Ray& r, float kMin, float kMax) {
    const char* lp = sc.wide_leaves + (size_t)leaf * sc.wide_leaf_stride;
    if (RT_HAS_PRIM(C, RT_HIT_SPHERE) && sc.wide_inline == RT_HIT_SPHERE) { float k; return try_sphere(*(const float4*)(lp + WIDE_LEAF_BOX_BYTES), r, kMin, kMax, k); }
    if ((RT_HAS_PRIM(C, RT_HIT_TRIANGLE) || RT_HAS_PRIM(C, RT_HIT_QUAD)) && (sc.wide_inline == RT_HIT_TRIANGLE || sc.wide_inline == RT_HIT_QUAD)) {
        const float4* q = (const float4*)(lp + WIDE_LEAF_BOX_BYTES);
        float k, u, v, nd;
        return (RT_HAS_PRIM(C, RT_HIT_TRIANGLE) && (!RT_HAS_PRIM(C, RT_HIT_QUAD) || sc.wide_inline == RT_HIT_TRIANGLE)) ? try_quad<true>(q, r, kMin, kMax, k, u, v, nd) : try_quad<false>(q, r, kMin, kMax, k, u, v, nd);
    }
    const uint32_t code = __float_as_uint(((const float4*)(lp + 16))->w);
    return fast_object_solid<C>(sc, code >> OBJ_TYPE_SHIFT, code & OBJ_INDEX_MASK, r, kMin, kMax);
}

// the tree part of closest_hit (rt_device.h) for a WideTree
template <class C>
RT_FN void closest_wide(const SceneDev& sc, const WideTree& tree, const Ray& r, float kMin, Closest& best, TravCount& cnt) {
    const f3 inv = splat(1.0f) / r.d;
    const WideRay wr = wide_ray_closest(sc, r, inv);
    WideCursor c = wide_start(wr);
    for (;;) {
#ifdef RT_WIDE_PROF      // (one-off measurement builds: lane slots of the step / leaf rounds against the lanes that use them, in the AabbMisses / HittableIntersections counters)
        while (wide_vote(c)) {
            if (RT_WIDE_PROF == 1) { const unsigned long long in = __ballot(true); const uint32_t nu = (uint32_t)__popcll(__ballot(wide_can_step(c))); if (lane_id_() == (uint32_t)__builtin_ctzll(in)) { cnt.box_miss += 64u; cnt.prim_hit += nu; } }
            if (wide_can_step(c)) cnt.nodes += wide_step(sc, tree, wr, kMin, best.k, c);
        }
        if (RT_WIDE_PROF == 2) { const unsigned long long in = __ballot(true); const uint32_t nu = (uint32_t)__popcll(__ballot(wide_has_leaf(c))); if (lane_id_() == (uint32_t)__builtin_ctzll(in)) { cnt.box_miss += 64u; cnt.prim_hit += nu; } }
        if (wide_has_leaf(c)) { TravCount t2 = trav_count_zero(); wide_leaf_closest<C>(sc, wide_take_leaf(c, wr), r, inv, kMin, best, t2); cnt.nodes += t2.nodes; cnt.prims += t2.prims; }
#else
        while (wide_vote(c)) if (wide_can_step(c)) cnt.nodes += wide_step(sc, tree, wr, kMin, best.k, c);
        if (wide_has_leaf(c)) wide_leaf_closest<C>(sc, wide_take_leaf(c, wr), r, inv, kMin, best, cnt);
#endif
        if (!wide_busy(c)) break;
    }
}
// the tree part of any_hit for a WideTree: one look with the pad of `wr`
template <class C>
RT_FN bool any_wide(const SceneDev& sc, const WideTree& tree, const WideRay& wr, const Ray& r, float kMin, float kMax, TravCount& cnt) {
    WideCursor c = wide_start(wr);
    for (;;) {
        while (wide_vote(c)) if (wide_can_step(c)) cnt.nodes += wide_step<RT_WIDE_ANY_ORDERED>(sc, tree, wr, kMin, kMax, c);
        if (wide_has_leaf(c)) { cnt.prims++; if (wide_leaf_any<C>(sc, wide_take_leaf<RT_WIDE_ANY_ORDERED>(c, wr), r, kMin, kMax)) return true; }
        if (!wide_busy(c)) return false;
    }
}
// ---- the names k_shadow_persistent uses for either wide tree (rt_ldsw.h has the LdswTreeT overloads)
RT_FN WideRay wide_ray_idle(const WideTree&) { Ray r; r.o = mk3(0, 0, 0); r.d = mk3(0, 0, 1); return make_wide_ray<RT_WIDE_ANY_ORDERED>(r, mk3(0, 0, 1), 0.0f, 1.0f); }
RT_FN uint32_t wide_step_any(const SceneDev& sc, const WideTree& tree, const WideRay& wr, float kMin, float kMax, WideCursor& c) { return wide_step<RT_WIDE_ANY_ORDERED>(sc, tree, wr, kMin, kMax, c); }
template <class C> RT_FN bool wide_next_leaf_any(const SceneDev& sc, const WideTree&, WideCursor& c, const WideRay& wr, const Ray& r, float kMin, float kMax) {
    return wide_leaf_any<C>(sc, wide_take_leaf<RT_WIDE_ANY_ORDERED>(c, wr), r, kMin, kMax);
}

} // namespace rtd
