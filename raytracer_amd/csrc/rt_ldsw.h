// rt_ldsw.h — a W-wide tree (W = 4 or 8) with full-precision inner boxes, held in LDS (small all-sphere scenes: Night Spheres).
//
// Why: with the tree in LDS nothing waits for memory; the binary traversal (LdsTree, rt_device.h) spends its vector instructions on the
// bookkeeping around a node — address, child select, push / pop, leaf parking: about 45 of the 70 issue slots of a two-box visit.
// Only a LEAF's box has reference arithmetic to reproduce (Core/Acceleration/SingleObjectBvhNode.cs:10-16); inner boxes only have to
// be conservative (BinaryBvhNode.cs:13-37 returns what its children return) and any-hit has no box test at all in the reference
// (BinaryBvhNode.cs:41).  So W children are tested per visit with ONE fused multiply-add per plane, t = plane * (1/d) + (-(o -+ pad)/d)
// (the any-hit form of AnyBox, here for closest hit too), and the bookkeeping is paid once per W boxes: the group cursor of rt_wide.h
// (a node's hit children are one stack entry, not up to W).  A leaf's own box is tested when the leaf is taken, in the reference's
// arithmetic (slab()), in front of its sphere.
//
// LDS image (rt_lib.hip: build_ldsw writes it; every workgroup copies it):
//   nodes[n]      LDSW_NODE_BYTES each:
//                   +0   u32  child_base: the first inner child's offset in 16-B units
//                   +4   u32  imask | lmask << 8      which slots hold an inner node / a leaf
//                   +16  u32 x W  object id of the leaf in each slot
//                   +16 + 4 W  per axis (x, y, z) 3 W floats: min[W] max[W] min[W] — the repeated min block turns the ray's sign into a byte
//                        offset: read at +0 (near, far) = (min, max) for a ray along +axis, at +4W bytes (max, min) along -axis
//                 the k-th inner child (slots ascending) is the record at child_base + k * LDSW_NODE_UNITS
//   spheres[n_objects]   float4 by object id
//   boxes[n_objects]     6 floats by object id (min.xyz, max.xyz): the object's exact box (closest-hit kernels only: the any-hit kernels copy the image up to here)
// Slots are assigned by quadrant / octant (Ylitie, Karras, Laine 2017; W = 4: by the two axes along which the scene is largest), so
// visiting the hit slots by descending (slot XOR octinv) is front to back with no sorting.
#pragma once
#include <type_traits>
#include "rt_wide.h"

namespace rtd {

#ifndef RT_LDSW
#define RT_LDSW 4
#endif
#ifndef RT_LDSW_DEFAULT
#define RT_LDSW_DEFAULT 1          // 1: the any-hit kernels of scenes whose tree fits LDS in both forms walk the W-wide one unless RT_HIP_LDS_WIDE=0 (=2: the closest-hit kernels too)
#endif
#ifndef RT_QUAD_ORDERED
#define RT_QUAD_ORDERED 1           // any-hit rays enter a node's children front to back in trees read through the caches (an occluder near the origin ends the ray at once), in slot order in LDS
#endif
constexpr int LDSW = RT_LDSW;
static_assert(LDSW == 4 || LDSW == 8, "slots per node of the W-wide float tree");
constexpr uint32_t LDSW_AXIS_BYTES = 12u * LDSW, LDSW_HEAD_BYTES = 16u + 4u * LDSW, LDSW_NODE_BYTES = LDSW_HEAD_BYTES + 3u * LDSW_AXIS_BYTES, LDSW_NODE_UNITS = LDSW_NODE_BYTES / 16u;
constexpr uint32_t LDSW_BOX_BYTES = 24u;
constexpr uint32_t LDSW_MAX_NODES_LDS = 65536u / LDSW_NODE_UNITS;      // in LDS a node is named by its offset in 16-B units in 16 bits (a stack entry is 32 bits)

// BIG = false: the image lives in LDS (every workgroup copies it), a stack entry packs group base (16 bits) | hits | imask into 32 bits, lanes
//       step until they hold a leaf (rays of about equal cost: no votes, as for LdsTree).
// BIG = true:  the image is read through the caches (trees of up to 2^28 nodes), a stack entry is 64 bits, the wave votes as for WideTree.
template <bool BIG>
struct LdswTreeT {
    static constexpr bool wide = true;
    static constexpr bool list = false;
    static constexpr bool vote = BIG;
    typedef typename std::conditional<BIG, uint2, uint32_t>::type Entry;
    const char* nodes; const float4* sph; const float2* boxes;
    Entry* stack; int stride;                    // this lane's column of group entries
    RT_FN void push(int sp, uint32_t gbase, uint32_t gbits) const {
        if constexpr (BIG) stack[sp * stride] = make_uint2(gbase, gbits); else stack[sp * stride] = gbase | (gbits << 16);
    }
    RT_FN void pop(int sp, uint32_t& gbase, uint32_t& gbits) const {
        if constexpr (BIG) { const uint2 e = stack[sp * stride]; gbase = e.x; gbits = e.y; } else { const uint32_t e = stack[sp * stride]; gbase = e & 0xFFFFu; gbits = e >> 16; }
    }
    // LDS: carve the copied image into its tables, the stack columns behind it
    RT_FN static LdswTreeT carve(const void* base, uint32_t node_bytes, uint32_t n_objects, uint32_t copied_bytes, int tid, int block) {
        LdswTreeT t;
        t.nodes = (const char*)base;
        t.sph = (const float4*)((const char*)base + node_bytes);
        t.boxes = (const float2*)(t.sph + n_objects);
        t.stack = (Entry*)((const char*)base + copied_bytes) + tid; t.stride = block;
        return t;
    }
    // read through the caches: the image stays where it is, the stack columns are the workgroup's dynamic LDS
    RT_FN static LdswTreeT global(const void* image, uint32_t node_bytes, uint32_t n_objects, void* lds, int tid, int block) {
        LdswTreeT t;
        t.nodes = (const char*)image;
        t.sph = (const float4*)((const char*)image + node_bytes);
        t.boxes = (const float2*)(t.sph + n_objects);
        t.stack = (Entry*)lds + tid; t.stride = block;
        return t;
    }
};
typedef LdswTreeT<false> LdswTree;
typedef LdswTreeT<true> QuadTree;
typedef uint32_t LdswEntry;

// what a node visit needs of the ray: per axis 1/d (clamped), the two addends of the near and the far plane, the byte offset of the near block
// ORDERED: the hit slots of a node are entered front to back (closest hit).  Any-hit rays enter them in slot order: in a tree this small
// the order saves less than the two mask permutations per visit cost (as for the binary LDS tree, whose any-hit walk is unordered too).
struct LdswRay { f3 inv, an, af; uint32_t ox, oy, oz, octinv; };
struct LdswAxis { float in, an, af; uint32_t off; bool neg; };      // (one axis at a time, no local arrays: those ended up in the kernel's private segment)
RT_FN LdswAxis ldsw_axis(float d, float o, float inv, float pad, float lim) {
    LdswAxis a;
    a.neg = __builtin_signbitf(d);
    const float sp = a.neg ? -pad : pad;
    a.in = __builtin_fminf(__builtin_fmaxf(inv, -lim), lim);
    a.an = -(o + sp) * a.in;                         // near plane: the min plane moved by -pad for a ray along +axis, the max plane moved by +pad otherwise
    a.af = -(o - sp) * a.in;
    a.off = a.neg ? 4u * LDSW : 0u;
    return a;
}
template <bool ORDERED>
RT_FN LdswRay make_ldsw_ray(const SceneDev& sc, const Ray& r, f3 inv, float pad, float lim) {
    const LdswAxis x = ldsw_axis(r.d.x, r.o.x, inv.x, pad, lim), y = ldsw_axis(r.d.y, r.o.y, inv.y, pad, lim), z = ldsw_axis(r.d.z, r.o.z, inv.z, pad, lim);
    LdswRay q;
    q.inv = mk3(x.in, y.in, z.in); q.an = mk3(x.an, y.an, z.an); q.af = mk3(x.af, y.af, z.af);
    q.ox = x.off; q.oy = y.off; q.oz = z.off;
    const uint32_t neg = (x.neg ? 1u : 0u) | (y.neg ? 2u : 0u) | (z.neg ? 4u : 0u);
    if (!ORDERED) q.octinv = 0u;
    else if (LDSW == 8) q.octinv = 7u ^ neg;
    else q.octinv = 3u ^ (((neg >> sc.ldsw_axis_a) & 1u) | (((neg >> sc.ldsw_axis_b) & 1u) << 1));
    return q;
}
// Closest hit: planes moved outward by 1e-6 of the farthest the scene reaches from the ray origin (see wide_ray_closest: the FMA form and
// the reference's fl(fl(plane - o) * inv) of the leaf test differ by at most 3.6e-7 of that), so a visit never culls a leaf whose own test would pass.
RT_FN LdswRay ldsw_ray_closest(const SceneDev& sc, const Ray& r, f3 inv_exact) {
    const f3 c = mk3(sc.bounds_centre[0], sc.bounds_centre[1], sc.bounds_centre[2]);
    const float far_side = __builtin_amdgcn_sqrtf(distance_sq(r.o, c)) + sc.bounds_radius;
    return make_ldsw_ray<true>(sc, r, inv_exact, (1e-6f * far_side) + 1e-30f, WIDE_INV_LIMIT * __builtin_amdgcn_rsqf(dot(r.d, r.d)));
}
// Any hit: the pad of any_box (rt_device.h) — two looks, first with none.
template <bool BIG>
RT_FN LdswRay wide_ray_any(const SceneDev& sc, const LdswTreeT<BIG>&, const Ray& r, float kMax, float cap, int mode, bool& again) {
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
    if (again) return make_ldsw_ray<BIG && RT_QUAD_ORDERED>(sc, r, inv, 0.0f, lim);
    return make_ldsw_ray<BIG && RT_QUAD_ORDERED>(sc, r, inv, sc.any_pad_scale * (round + (1e-6f * reach) + 1e-5f + (2.4e-7f * __builtin_fmaxf(__builtin_fmaxf(fabsf(r.o.x), fabsf(r.o.y)), fabsf(r.o.z)))), lim);
}

RT_FN uint32_t xor_permute_w(uint32_t m, uint32_t x) {      // position p of a W-bit mask -> position p ^ x
    if (LDSW == 8 && (x & 4u)) m = ((m & 0x0Fu) << 4) | ((m >> 4) & 0x0Fu);
    if (x & 2u) m = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
    if (x & 1u) m = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
    return m;
}

// One node: which of its W child boxes the ray's [kMin, kMax] meets (slot order).  `head` returns the node's first two words.
template <bool BIG>
RT_FN uint32_t ldsw_visit(const LdswTreeT<BIG>& t, const LdswRay& q, uint32_t node, float kMin, float kMax, uint2& head) {
    const char* n = t.nodes + ((size_t)node << 4);
    head = *(const uint2*)n;
    const char* ax = n + LDSW_HEAD_BYTES + q.ox;
    const char* ay = n + LDSW_HEAD_BYTES + LDSW_AXIS_BYTES + q.oy;
    const char* az = n + LDSW_HEAD_BYTES + 2u * LDSW_AXIS_BYTES + q.oz;
    const v2f ix = {q.inv.x, q.inv.x}, iy = {q.inv.y, q.inv.y}, iz = {q.inv.z, q.inv.z};
    const v2f anx = {q.an.x, q.an.x}, any_ = {q.an.y, q.an.y}, anz = {q.an.z, q.an.z};
    const v2f afx = {q.af.x, q.af.x}, afy = {q.af.y, q.af.y}, afz = {q.af.z, q.af.z};
    uint32_t hits = 0;
#pragma unroll
    for (int g = 0; g < LDSW / 4; g++) {
        const float4 nx = *(const float4*)(ax + 16 * g), fx = *(const float4*)(ax + 4 * LDSW + 16 * g);
        const float4 ny = *(const float4*)(ay + 16 * g), fy = *(const float4*)(ay + 4 * LDSW + 16 * g);
        const float4 nz = *(const float4*)(az + 16 * g), fz = *(const float4*)(az + 4 * LDSW + 16 * g);
        const v2f nx0 = __builtin_elementwise_fma((v2f){nx.x, nx.y}, ix, anx), nx1 = __builtin_elementwise_fma((v2f){nx.z, nx.w}, ix, anx);
        const v2f ny0 = __builtin_elementwise_fma((v2f){ny.x, ny.y}, iy, any_), ny1 = __builtin_elementwise_fma((v2f){ny.z, ny.w}, iy, any_);
        const v2f nz0 = __builtin_elementwise_fma((v2f){nz.x, nz.y}, iz, anz), nz1 = __builtin_elementwise_fma((v2f){nz.z, nz.w}, iz, anz);
        const v2f fx0 = __builtin_elementwise_fma((v2f){fx.x, fx.y}, ix, afx), fx1 = __builtin_elementwise_fma((v2f){fx.z, fx.w}, ix, afx);
        const v2f fy0 = __builtin_elementwise_fma((v2f){fy.x, fy.y}, iy, afy), fy1 = __builtin_elementwise_fma((v2f){fy.z, fy.w}, iy, afy);
        const v2f fz0 = __builtin_elementwise_fma((v2f){fz.x, fz.y}, iz, afz), fz1 = __builtin_elementwise_fma((v2f){fz.z, fz.w}, iz, afz);
        const float tn[4][3] = {{nx0.x, ny0.x, nz0.x}, {nx0.y, ny0.y, nz0.y}, {nx1.x, ny1.x, nz1.x}, {nx1.y, ny1.y, nz1.y}};
        const float tf[4][3] = {{fx0.x, fy0.x, fz0.x}, {fx0.y, fy0.y, fz0.y}, {fx1.x, fy1.x, fz1.x}, {fx1.y, fy1.y, fz1.y}};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // (a NaN distance — inf - inf on an axis the ray is parallel to — never replaces a bound: maxNum / minNum)
            const float en = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(tn[j][0], tn[j][1]), tn[j][2]), kMin);
            const float ex = __builtin_fminf(__builtin_fminf(__builtin_fminf(tf[j][0], tf[j][1]), tf[j][2]), kMax);
            hits |= !(ex < en) ? (1u << (4 * g + j)) : 0u;                                   // closed interval: a box met on a face or an edge is entered
        }
    }
    return hits;
}

// the traversal state is rt_wide.h's WideCursor; gbase = the group's first node, pbase = the node whose leaves are in P
RT_FN WideCursor ldsw_start(const LdswRay& q) { WideCursor c = wide_idle(); c.gbits = (1u << q.octinv) | (1u << 8); return c; }      // the root as a one-child group: node 0 in slot 0

// Enter the next child still in G (ORDERED: the nearest) and test its children.  Returns the number of boxes tested.
template <bool ORDERED, bool BIG>
RT_FN uint32_t ldsw_step(const LdswTreeT<BIG>& tree, const LdswRay& q, float kMin, float kMax, WideCursor& c) {
    const uint32_t pi = 31u - (uint32_t)__builtin_clz(c.gbits & 0xFFu);
    const uint32_t slot = ORDERED ? pi ^ q.octinv : pi, gmask = (c.gbits >> 8) & 0xFFu;
    const uint32_t node = c.gbase + __umul24((uint32_t)__builtin_popcount(gmask & ((1u << slot) - 1u)), LDSW_NODE_UNITS);
    const uint32_t rest = c.gbits & ~(1u << pi);                 // the siblings still to enter
    uint2 head;
    const uint32_t hits = ldsw_visit(tree, q, node, kMin, kMax, head);
    const uint32_t imask = head.y & 0xFFu, lmask = (head.y >> 8) & 0xFFu;
    c.pbase = node; c.pbits = (ORDERED ? xor_permute_w(hits & lmask, q.octinv) : hits & lmask) | (lmask << 8);
    const uint32_t inner = hits & imask;
    if (inner) {                                                  // descend: the siblings wait in the lane's column
        if (rest & 0xFFu) { tree.push(c.sp, c.gbase, rest); c.sp++; }
        c.gbase = head.x; c.gbits = (ORDERED ? xor_permute_w(inner, q.octinv) : inner) | (imask << 8);
    } else if (rest & 0xFFu) c.gbits = rest;                      // nothing below: on with the siblings
    else if (c.sp) { c.sp--; tree.pop(c.sp, c.gbase, c.gbits); }
    else c.gbits = 0;
    return (uint32_t)__builtin_popcount(imask | lmask);
}
// the next leaf of P: its object id
template <bool ORDERED, bool BIG>
RT_FN uint32_t ldsw_take_leaf(const LdswTreeT<BIG>& tree, WideCursor& c, const LdswRay& q) {
    const uint32_t pi = 31u - (uint32_t)__builtin_clz(c.pbits & 0xFFu);
    const uint32_t slot = ORDERED ? pi ^ q.octinv : pi;
    c.pbits &= ~(1u << pi);
    return *(const uint32_t*)(tree.nodes + ((size_t)c.pbase << 4) + 16u + 4u * slot);
}

// the tree part of closest_hit (rt_device.h).  Lanes step until they hold a leaf (rays of about equal cost: no votes, as for LdsTree).
template <class C, bool BIG>
RT_FN void closest_wide(const SceneDev& sc, const LdswTreeT<BIG>& tree, const Ray& r, float kMin, Closest& best, TravCount& cnt) {
    const f3 inv = splat(1.0f) / r.d;
    const LdswRay q = ldsw_ray_closest(sc, r, inv);
    WideCursor c = ldsw_start(q);
    for (;;) {
        if constexpr (BIG) { while (wide_vote(c)) if (wide_can_step(c)) cnt.nodes += ldsw_step<true>(tree, q, kMin, best.k, c); }
        else while (wide_can_step(c)) cnt.nodes += ldsw_step<true>(tree, q, kMin, best.k, c);
        if (wide_has_leaf(c)) {
            // the object's own box first, in the reference's arithmetic (AxisAlignedBoundingBox.Hit through slab(): SingleObjectBvhNode.cs:12-16), then Hittable.TryHit
            const uint32_t obj = ldsw_take_leaf<true>(tree, c, q);
            const float2 b0 = tree.boxes[3 * obj], b1 = tree.boxes[3 * obj + 1], b2 = tree.boxes[3 * obj + 2];
            float entry, exit_;
            cnt.nodes++;
            slab(mk3(b0.x, b0.y, b1.x), mk3(b1.y, b2.x, b2.y), r.o, inv, kMin, entry, exit_);
            if (!slab_hit(entry, exit_, best.k)) cnt.box_miss++;
            else {
                float k; cnt.prims++;
                if (try_sphere(tree.sph[obj], r, kMin, best.k, k)) { consider(best, k, obj, 0.0f); cnt.prim_hit++; }
            }
        }
        if (!wide_busy(c)) break;
    }
}
// the tree part of any_hit: one look with the pad of `q`.  Hittable.FastTryHit, no box test at the leaf (BinaryBvhNode.cs:41).
template <class C, bool BIG>
RT_FN bool any_wide(const SceneDev& sc, const LdswTreeT<BIG>& tree, const LdswRay& q, const Ray& r, float kMin, float kMax, TravCount& cnt) {
    WideCursor c = ldsw_start(q);
    for (;;) {
        if constexpr (BIG) { while (wide_vote(c)) if (wide_can_step(c)) cnt.nodes += ldsw_step<RT_QUAD_ORDERED>(tree, q, kMin, kMax, c); }
        else while (wide_can_step(c)) cnt.nodes += ldsw_step<false>(tree, q, kMin, kMax, c);
        if (wide_has_leaf(c)) { float k; cnt.prims++; if (try_sphere(tree.sph[ldsw_take_leaf<BIG && RT_QUAD_ORDERED>(tree, c, q)], r, kMin, kMax, k)) return true; }
        if (!wide_busy(c)) return false;
    }
}

// ---- the names k_shadow_persistent uses for either wide tree (rt_wide.h has the WideTree overloads)
template <bool BIG> RT_FN LdswRay wide_ray_idle(const LdswTreeT<BIG>&) { LdswRay q; q.inv = mk3(0, 0, 1); q.an = mk3(0, 0, 0); q.af = mk3(0, 0, 0); q.ox = 0; q.oy = 0; q.oz = 0; q.octinv = 0; return q; }
RT_FN WideCursor wide_start(const LdswRay& q) { return ldsw_start(q); }
template <bool BIG> RT_FN uint32_t wide_step_any(const SceneDev&, const LdswTreeT<BIG>& tree, const LdswRay& q, float kMin, float kMax, WideCursor& c) { return ldsw_step<BIG && RT_QUAD_ORDERED>(tree, q, kMin, kMax, c); }
template <class C, bool BIG> RT_FN bool wide_next_leaf_any(const SceneDev&, const LdswTreeT<BIG>& tree, WideCursor& c, const LdswRay& q, const Ray& r, float kMin, float kMax) {
    float k; return try_sphere(tree.sph[ldsw_take_leaf<BIG && RT_QUAD_ORDERED>(tree, c, q)], r, kMin, kMax, k);
}

} // namespace rtd
