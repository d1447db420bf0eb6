// rt_kernels.h — the wavefront trace kernels (gfx950).
//
// One batch = P paths = (pixels) x (passes in the batch).  The paths alive at a bounce are DENSE: their state sits at
// queue positions 0..n-1 of a ping-pong pair of float4 arrays, and the survivors of a bounce are compacted into the other
// half — every state access is a coalesced 16-B-per-lane load or store, whatever fraction of the batch is still alive.
// Per bounce:
//   k_closest : position j -> closest hit (BVH traversal, per-lane stack in LDS)            RenderJob.cs:581-617
//   k_shade   : hit/miss -> sky, scatter, affine colour terms, shadow requests; survivors
//               and shadow requests get their output positions from one block scan          RenderJob.cs:413-451, materials
//   k_shadow  : one thread per shaded hit, LightSampleCountHint x |lights| any-hit rays      Material.cs:30-45, SimpleLightBase.cs:48-66
// A path that ends writes its radiance to out[path id]; k_accumulate adds those into the image sums in pass order
// (RenderJob.cs:626-647) and k_resolve makes RGB24 (RenderJob.cs:646-654).
//
// The reference walks the hits backwards and lets each material transform the "future" colour
// (RenderJob.cs:461-487).  Every builtin CalculateColour is affine in that colour, final = a*future + c, so
// the kernels carry a throughput T and add T*c as they go: L = sum_k T_k c_k + T_n * terminal, T_{k+1} = T_k a_k.
#pragma once
#include "rt_device.h"
#include "rt_wide.h"
#include "rt_ldsw.h"

namespace rtd {

constexpr int BLOCK = 256;

enum StatSlot : int {
    ST_RAYS = 0, ST_SHADOW_RAYS, ST_NODES, ST_PRIMS, ST_SH_NODES, ST_SH_PRIMS, ST_SKY, ST_LIMIT, ST_SCATTER, ST_ABSORBED, ST_SAMPLES,
    ST_BOX_MISS, ST_PRIM_HIT,        // AabbMisses, HittableIntersections (HittableMisses = ST_PRIMS - ST_PRIM_HIT)
    ST_NORMAL_WARN, ST_UV_WARN,      // the warn-only GraphicsValidator rules (RenderJob.cs:588-607)
    ST_DEPTH0, ST_COUNT = ST_DEPTH0 + RT_DEPTH_HIST_SIZE
};

// Path state at queue position j (3 x float4 + uint2 = 56 B):
//   g0 = (origin.xyz, dir.x)  g1 = (dir.y, dir.z, T.r, T.g)  g2 = (T.b, L.r, L.g, L.b)  ids = (path id, previous material | first material << 16)
// Shadow request at shadow-queue position p (3 x float4, + 1 for Phong):
//   h0 = (hit point.xyz, normal.x)  h1 = (normal.y, normal.z, W.r, W.g)  h2 = (W.b, path id, dst, hit object) as bits   h3 = (incoming dir.xyz, -)
//   dst = position in the NEXT state whose L receives the light sum, or RT_NO_HIT when the path ended (then out[path id] does)
struct RenderDev {
    float4* st[2][3];
    uint2*  ids[2];           // (path id, pack_mats(previous hit's material, first hit's material)): what the refractive materials ask about the path
    float2* hit;              // (K, object id bits) by queue position
    float*  hit_aux;          // HitRecord.ShaderData (media only) by queue position
    float4* out;              // final radiance by path id
    float4* sh[4];
    uint32_t* qcount;         // [max_depth + 2] alive paths per bounce
    uint32_t* sqcount;        // [max_depth + 2] shadow requests per bounce
    unsigned long long* stats;// [ST_COUNT]
    uint32_t width, height, n_pixels;
    uint32_t tiled_slots, tiles_per_row, tile_magic, tile_shift;      // slot_pixel: the first tiled_slots slots of a pass go tile by tile (0: row order throughout), tiles of 2^tile_shift pixels a side; tile_magic = ceil(2^32 / tiles_per_row)
    uint32_t pass_shift, group_paths;                     // path_pass_pixel: the batch's passes go in groups of 2^pass_shift, group_paths = n_pixels << pass_shift
    uint32_t n_paths;         // pixels * passes in this batch
    uint64_t first_pass; uint32_t pass_stride;    // pass of path p = first_pass + (its pass in the batch, path_pass_pixel) * pass_stride
    uint64_t seed;
    float k_min, k_max;
    uint32_t max_depth, light_samples;
    uint32_t compact_requests;        // shadow requests of surviving paths are 16 B and point at the survivor's state (req_head): no Phong material, no debug visualisation
    uint32_t misses_end_in_closest;   // the closest-hit kernels finish the paths that hit nothing (finish_miss): tree in LDS, no debug visualisation
    uint32_t obj_shade_lds_bytes;     // k_shade copies the table of object records (sc.obj_shade) into LDS first: this many bytes, 0 = reads it through the caches
    uint32_t skip_mute_rays;          // the shadow kernels first ask whether a light can change the sum at all (more than two rays per request, or a light with a cutoff)
    uint32_t implicit_primary;        // depth 0 has no k_raygen pass: the first k_closest launch makes the camera ray of a path from its number (primary_ray); k_shade makes it again (1) or reads the 32 B k_closest left (2)
    float any_cap;            // any_box: growth limit in units of the smallest round radius (inf = RT_RENDER_EXACT_OCCLUSION)
    uint32_t vis;             // rt_debug_vis (k_shade<C, true> only)
    unsigned long long* validate;     // RT_RENDER_VALIDATE: [n_objects][2] closest hits per object that fail the normal rule / the UV rule (k_shade<C, VIS, true> only), else null
};

// material ids are < 0xFFFF (rt_scene_create), 0xFFFF = none yet
RT_FN uint32_t pack_mats(uint32_t prev, uint32_t first) { return (prev == RT_NO_HIT ? 0xFFFFu : prev) | ((first == RT_NO_HIT ? 0xFFFFu : first) << 16); }
RT_FN uint32_t unpack_mat(uint32_t half) { half &= 0xFFFFu; return half == 0xFFFFu ? RT_NO_HIT : half; }

RT_FN uint32_t lane_id() { return threadIdx.x & 63u; }

RT_FN unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
RT_FN void stat_add(unsigned long long* stats, int slot, unsigned long long v) {
    v = wave_sum(v);
    if (lane_id() == 0 && v) atomicAdd(stats + slot, v);
}
// Three counters of a traversal kernel, one atomic per counter per WORKGROUP: every wave of every block adding to the same three
// words costs 25 k same-address atomics per launch, which the L2 serialises (k_closest -4 %).  Every thread of the block must call it.
template <int N>
RT_FN void stat_add_block(unsigned long long* stats, const int (&slot)[N], const unsigned long long (&v)[N]) {
    __shared__ unsigned long long red[N][16];
    const uint32_t wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < N; i++) { const unsigned long long w = wave_sum(v[i]); if (lane_id() == 0) red[i][wave] = w; }
    __syncthreads();
    // thread i adds counter i: every index below is a compile-time constant, so slot[] and v[] stay in registers
#pragma unroll
    for (int i = 0; i < N; i++)
        if ((int)threadIdx.x == i) {
            unsigned long long t = 0;
            for (uint32_t w = 0; w < n_waves; w++) t += red[i][w];
            if (t) atomicAdd(stats + slot[i], t);
        }
}
RT_FN void stat_add3_block(unsigned long long* stats, int s0, int s1, int s2, uint32_t v0, uint32_t v1, uint32_t v2) {
    const int slot[3] = {s0, s1, s2}; const unsigned long long v[3] = {v0, v1, v2};
    stat_add_block<3>(stats, slot, v);
}
// (+ the paths a closest-hit kernel finished itself because they hit nothing: finish_miss below)
RT_FN void stat_add_closest(unsigned long long* stats, uint32_t rays, const TravCount& c, uint32_t sky, uint32_t depth) {
    const int slot[8] = {ST_RAYS, ST_NODES, ST_PRIMS, ST_BOX_MISS, ST_PRIM_HIT, ST_SKY, ST_SAMPLES, ST_DEPTH0 + (int)min(depth > 0 ? depth - 1 : 0u, (uint32_t)RT_DEPTH_HIST_SIZE - 1)};
    const unsigned long long v[8] = {rays, c.nodes, c.prims, c.box_miss, c.prim_hit, sky, sky, depth > 0 ? sky : 0u};
    stat_add_block<8>(stats, slot, v);
}

// A path whose ray hit nothing ends here (RenderJob.cs:444-450: the sky's colour times the throughput, on top of what the path has
// collected), inside the closest-hit kernel that found it out: the kernel holds the ray and two thirds of the throughput already,
// reads the rest of the state (24 B) and writes the path's radiance — and k_shade never loads the 56 B of a path that missed
// (a third of all paths it used to process on Night Spheres: k_shade 56.4 -> 49.9, k_closest 47.7 -> 49.7 ms per step).  Only for
// trees in LDS: where the tree is read through the caches the extra registers cost the traversal more than k_shade saves
// (Cornell k_closest 23.8 -> 30.2 for -5.0 in k_shade, 5 M triangles 84.4 -> 89.6 for -1.8).  Not in debug-visualisation
// renders, whose misses have their own colours (k_shade<CfgAll, true> keeps them).  g1w = the path's second state word (dir.y, dir.z, T.r, T.g).
RT_FN void finish_miss(const SceneDev& sc, const RenderDev& rd, uint32_t depth, uint32_t j, f3 dir, float4 g1w, bool primary) {
    const float4 g2w = primary ? make_float4(1.0f, 0.0f, 0.0f, 0.0f) : rd.st[depth & 1][2][j];      // (a primary path: throughput 1, nothing collected, path number = position)
    const uint32_t id = primary ? j : rd.ids[depth & 1][j].x;
    const col3 T = mkc(g1w.z, g1w.w, g2w.x);
    col3 L = mkc(g2w.y, g2w.z, g2w.w);
    L = L + (T * sky_colour(sc, dir));
    rd.out[id] = make_float4(L.r, L.g, L.b, 0.0f);
}

// A shadow request (what k_shade leaves for the shadow kernels about a hit that samples the lights), in one of two layouts:
//   full (48 B):    sh[0] = (point.xyz, normal.x)  sh[1] = (normal.yz, W.rg)  sh[2] = (W.b, path id, survivor position or RT_NO_HIT, object)
//   compact (16 B): sh[0] = (normal.xyz, survivor position).  When the path survives the bounce its next state already holds the
//                   rest: the new ray starts at the hit point, the new throughput T * a IS the light weight W of the Standard and
//                   refractive materials (StandardMaterial.cs:74-80: future and light sum share the albedo factor), and the path id
//                   sits beside it.  k_shade then writes 16 instead of 48 B per request (a fifth of all it writes on Night Spheres).
// Compact needs every lit hit to survive (depth < max depth: launch-uniform), W == T * a (no Phong material in the scene) and no
// debug visualisation (EstimatedLight weighs with 1): rd.compact_requests says the last two.
RT_FN bool requests_are_compact(const RenderDev& rd, uint32_t depth) { return rd.compact_requests && depth < rd.max_depth; }
struct ReqHead { f3 wp, nrm; uint32_t id, dst; float obj_bits; };      // dst: compact requests only (kept across the traversals: one register instead of a reload)
RT_FN ReqHead req_head(const RenderDev& rd, uint32_t depth, uint32_t p, bool compact) {
    ReqHead q;
    if (compact) {
        const float4 h = rd.sh[0][p];
        const uint32_t dst = __float_as_uint(h.w);
        const float4 s0 = rd.st[(depth & 1) ^ 1][0][dst];
        q.wp = mk3(s0.x, s0.y, s0.z); q.nrm = mk3(h.x, h.y, h.z); q.id = rd.ids[(depth & 1) ^ 1][dst].x; q.dst = dst; q.obj_bits = 0.0f;
    } else {
        const float4 h0 = rd.sh[0][p], h1 = rd.sh[1][p], h2 = rd.sh[2][p];
        q.wp = mk3(h0.x, h0.y, h0.z); q.nrm = mk3(h0.w, h1.x, h1.y); q.id = __float_as_uint(h2.y); q.dst = 0; q.obj_bits = h2.w;
    }
    return q;
}
// the rest of a request, read when its rays are done: the weight of the light sum, where the light goes (the survivor's state, or
// the finished path's radiance), the path id
RT_FN void req_tail(const RenderDev& rd, uint32_t depth, uint32_t p, bool compact, uint32_t head_dst, col3& W, uint32_t& dst, uint32_t& id) {
    if (compact) {
        dst = head_dst; id = 0;
        const float4 e1 = rd.st[(depth & 1) ^ 1][1][dst], e2 = rd.st[(depth & 1) ^ 1][2][dst];
        W = mkc(e1.z, e1.w, e2.x);
    } else {
        const float4 e1 = rd.sh[1][p], e2 = rd.sh[2][p];
        W = mkc(e1.z, e1.w, e2.x); dst = __float_as_uint(e2.z); id = __float_as_uint(e2.y);
    }
}

// Path number <-> (pass, pixel).  A batch's passes go in groups of g = 2^pass_shift (the largest power of two that divides their
// number, at most 16: rt_lib.hip); inside a group the paths are numbered slot by slot and, for one slot, pass by pass:
//     id = group * (n_pixels * g) + slot * g + pass-in-group
// and slot s of the image is not pixel s either: for the first rd.tiled_slots slots the slots walk the image tile by tile, in Z order
// inside a tile (32 x 32 pixels where the width is a multiple of 32, else 8 x 8, rows that fill no tile in row order; rt_lib.hip).  With 16 passes in a group a wave's 64
// camera rays are 16 samples each of a 2 x 2 block of pixels, neighbouring waves neighbouring blocks of a tile, where the
// reference's loop nest (pass by pass, row by row) would make them a 64 x 1 strip of one pass: the rays of a wave stay together in
// the tree, end on the same object, ask for the same light from the same place — and since survivors are compacted in order the
// queues of every later bounce inherit the neighbourhood.  Nothing else knows about the order: the RNG is keyed by (pixel, pass) and
// k_accumulate adds a pixel's passes in pass order, so the image is the same bit for bit (RT_HIP_TILES=0: row order,
// RT_HIP_PASS_GROUP=0: pass-major numbering; A/B runs and tests).  Measured, both together: Night Spheres 3680 -> 3800 Msamples/s,
// 5 M triangles 1030 -> 1142 (short runs), Cornell and 100 k spheres + 64 lights unchanged.
RT_FN uint32_t even_bits(uint32_t v) {      // bits 0, 2, 4, 6, 8 of v side by side
    v &= 0x555u; v = (v | (v >> 1)) & 0x333u;
    return ((v | (v >> 2)) & 0x0Fu) | ((v >> 4) & 0x10u);
}
RT_FN uint32_t slot_pixel(const RenderDev& rd, uint32_t slot) {
    if (slot >= rd.tiled_slots) return slot;
    const uint32_t sh = rd.tile_shift, in = slot & ((1u << (2u * sh)) - 1u), tile = slot >> (2u * sh);
    const uint32_t ty = __umulhi(tile, rd.tile_magic), tx = tile - ty * rd.tiles_per_row;      // (exact: the host checks n_tiles * tiles_per_row < 2^32)
    return ((ty << sh) | even_bits(in >> 1)) * rd.width + ((tx << sh) | even_bits(in));       // Z order inside the tile: 4, 16, 64, ... consecutive slots are squares
}
RT_FN void path_pass_pixel(const RenderDev& rd, uint32_t id, uint32_t& pass, uint32_t& pix) {
    const uint32_t k = id / rd.group_paths, l = id - k * rd.group_paths;
    pass = (k << rd.pass_shift) | (l & ((1u << rd.pass_shift) - 1u));
    pix = slot_pixel(rd, l >> rd.pass_shift);
}
RT_FN RngKey path_key(const RenderDev& rd, uint32_t id) {
    uint32_t b, pix; path_pass_pixel(rd, id, b, pix);
    return make_key(rd.seed, pix, rd.first_pass + (uint64_t)b * rd.pass_stride);
}

// RenderPixelWithVisualisations (None branch), RenderJob.cs:135-151 + Camera.GetRay: the camera ray of path `id` of the batch
// (pass and pixel: path_pass_pixel).  A function of the path's number alone, so depth 0 needs no k_raygen
// pass that writes 64 B of state per path for the first closest-hit launch to read back: with rd.implicit_primary that launch
// makes the rays itself.  What the first k_shade launch needs of them it either makes again (rd.implicit_primary == 1) or reads
// (== 2: the closest-hit launch leaves the 32 B of origin and direction; throughput 1, radiance 0 and the path number are
// implied).  Reading is for trees in LDS, whose closest-hit kernel is bound by arithmetic and hides the stores (Night Spheres,
// since the per-object records made k_shade arithmetic-bound too: 38.8 -> 37.3 ms per step for +0.7 in k_closest; before them,
// while k_shade waited on its material gathers, making the ray again was free and reading cost 1 ms); where the tree is read
// through the caches the stores cost the traversal more than k_shade saves (5 M triangles k_closest 85.1 -> 86.8 ms per batch).
#ifndef RT_PRIMARY_STORE
#define RT_PRIMARY_STORE 1
#endif
RT_FN Ray primary_ray(const SceneDev& sc, const RenderDev& rd, uint32_t id) {
    uint32_t b, pix; path_pass_pixel(rd, id, b, pix);
    uint32_t x = pix % rd.width, y = pix / rd.width;                                   // Decompress2DIndex, MathUtils.cs:38-43
    RngKey key = make_key(rd.seed, pix, rd.first_pass + (uint64_t)b * rd.pass_stride);
    float4 u = rng_draw(key, 0, SLOT_CAMERA, 0);
    float s = (float)x, t = (float)y;
    s += plus_minus_one(u.x) * 0.5f;
    t += plus_minus_one(u.y) * 0.5f;
    return camera_ray(sc.camera, s / (float)rd.width, t / (float)rd.height, u.z, u.w);
}

#ifdef RT_KERNELS_COMMON      // the kernels that do not depend on the scene variant: defined by rt_lib.hip only
// the stored form of depth 0 (debug visualisations, RT_HIP_RAYGEN=1)
__global__ void __launch_bounds__(BLOCK) k_raygen(SceneDev sc, RenderDev rd) {
    for (uint32_t id = blockIdx.x * BLOCK + threadIdx.x; id < rd.n_paths; id += gridDim.x * BLOCK) {
        Ray r = primary_ray(sc, rd, id);
        rd.st[0][0][id] = make_float4(r.o.x, r.o.y, r.o.z, r.d.x);
        rd.st[0][1][id] = make_float4(r.d.y, r.d.z, 1.0f, 1.0f);
        rd.st[0][2][id] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
        rd.ids[0][id] = make_uint2(id, pack_mats(RT_NO_HIT, RT_NO_HIT));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) rd.qcount[0] = rd.n_paths;
}

#endif

#define RT_ANY_GROWTH_CAP 0.5f         // any_box without RT_RENDER_EXACT_OCCLUSION
extern __shared__ int lds_stack[];   // HbmTree: int [stack_depth][block] ; LdsTree: scene image, then ushort [stack_depth][block]

// Traversal kernels come in two forms: TB = BLOCK reading the tree through the cache hierarchy, and (small all-sphere scenes)
// TB = a larger workgroup that first copies the scene's LDS image and then never touches the vector cache for nodes or spheres.
#ifndef RT_LDS_BLOCK_SHADOW
#define RT_LDS_BLOCK_SHADOW 1024
#endif
constexpr int LDS_BLOCK_CLOSEST = 1024, LDS_BLOCK_SHADOW = RT_LDS_BLOCK_SHADOW;
// the closest-hit kernels over the W-wide LDS tree: 79 VGPRs as they come (8 waves per SIMD would spill 28-39 of them), so two workgroups
// of 768 threads per CU = 6 waves per SIMD
#ifndef RT_LDSW_CLOSEST_WAVES
#define RT_LDSW_CLOSEST_WAVES 6
#endif
#ifndef RT_LDSW_BLOCK_CLOSEST
#define RT_LDSW_BLOCK_CLOSEST 768
#endif
constexpr int LDSW_BLOCK_CLOSEST = RT_LDSW_BLOCK_CLOSEST;
constexpr int lds_block_closest(int tree) { return tree == 3 ? LDSW_BLOCK_CLOSEST : LDS_BLOCK_CLOSEST; }
#ifndef RT_LDS_BLOCK_RAYS
#define RT_LDS_BLOCK_RAYS 768
#endif
constexpr int LDS_BLOCK_RAYS = RT_LDS_BLOCK_RAYS;      // k_shadow_rays with the tree in LDS: 81 VGPRs let 6 waves/SIMD in, two blocks of 12 waves per CU
// where a traversal kernel reads the tree from: the binary tree through the caches, the binary tree in LDS, the 8-wide tree through the caches
enum : int { TREE_HBM = 0, TREE_LDS = 1, TREE_WIDE = 2, TREE_LDSW = 3, TREE_QUAD = 4 };      // LDSW: the W-wide float tree in LDS, QUAD: the same read through the caches (rt_ldsw.h)
constexpr bool tree_in_lds(int tree) { return tree == TREE_LDS || tree == TREE_LDSW; }
template <int TREE, int TB, bool CLOSEST = true>
RT_FN auto make_tree(const SceneDev& sc) {
    if constexpr (TREE == TREE_LDSW) {
        uint4* dst = (uint4*)lds_stack;
        const uint32_t bytes = CLOSEST ? sc.ldsw_all_bytes : sc.ldsw_any_bytes;      // (any-hit has no leaf box test: its kernels leave the boxes out)
        for (uint32_t i = threadIdx.x; i < bytes / 16; i += TB) dst[i] = sc.ldsw_image[i];
        __syncthreads();
        return LdswTree::carve(lds_stack, sc.ldsw_node_bytes, sc.n_objects, bytes, (int)threadIdx.x, TB);
    } else if constexpr (TREE == TREE_QUAD) {
        return QuadTree::global(sc.quad_image, sc.quad_node_bytes, sc.n_objects, lds_stack, (int)threadIdx.x, TB);
    } else if constexpr (TREE == TREE_WIDE) {
        WideTree t; t.stack = (uint2*)lds_stack + threadIdx.x; t.stride = TB;
        return t;
    } else if constexpr (TREE == TREE_LDS) {
        uint4* dst = (uint4*)lds_stack;
        const uint32_t n16 = sc.lds_image_bytes / 16;
        for (uint32_t i = threadIdx.x; i < n16; i += TB) dst[i] = sc.lds_image[i];
        __syncthreads();
        return LdsTree::carve(lds_stack, sc.n_nodes, sc.n_objects, sc.lds_image_bytes, (int)threadIdx.x, TB);
    } else {
        HbmTree t; t.stack = lds_stack + threadIdx.x; t.stride = TB;
        return t;
    }
}

// (Plain grid-stride loop: where rays cost about the same — Night Spheres — it beats k_closest_persistent below, 15.0 vs 17.1 ms per 184 M rays.)
// PRIMARY: the launch of depth 0 when the render keeps no stored state for it (rd.implicit_primary): the camera rays are made here
// (primary_ray).  A kernel of its own so that the ray set-up does not cost the bounce launches registers (LDS tree: 57 -> 68 VGPRs
// would leave room for one 1024-thread workgroup per CU instead of two); held to the same 8 waves per SIMD.
template <class C, int TREE, bool PRIMARY = false>
__global__ void __launch_bounds__(tree_in_lds(TREE) ? lds_block_closest(TREE) : BLOCK, TREE == TREE_LDS && PRIMARY ? 8 : TREE == TREE_LDSW ? RT_LDSW_CLOSEST_WAVES : 1) k_closest(SceneDev sc, RenderDev rd, uint32_t depth) {
    constexpr bool LDS = tree_in_lds(TREE);
    constexpr int TB = LDS ? lds_block_closest(TREE) : BLOCK;
    const auto tree = make_tree<TREE, TB>(sc);
    constexpr bool primary = PRIMARY;
    const uint32_t n = primary ? rd.n_paths : rd.qcount[depth];
    if (primary && blockIdx.x == 0 && threadIdx.x == 0) rd.qcount[0] = n;      // (k_shade reads it)
    const float4* g0 = rd.st[depth & 1][0]; const float4* g1 = rd.st[depth & 1][1]; const uint2* g3 = rd.ids[depth & 1];
    uint32_t rays = 0, sky = 0;
    const bool finish = LDS && rd.misses_end_in_closest;       // a ray that hits nothing ends its path here (finish_miss)
    TravCount cnt = trav_count_zero();
    for (uint32_t j = blockIdx.x * TB + threadIdx.x; j < n; j += gridDim.x * TB) {
        Ray r; float4 b;
        if (primary) {
            r = primary_ray(sc, rd, j); b = make_float4(r.d.y, r.d.z, 1.0f, 1.0f);
            if (LDS && RT_PRIMARY_STORE) { rd.st[0][0][j] = make_float4(r.o.x, r.o.y, r.o.z, r.d.x); rd.st[0][1][j] = b; }      // the ray, for k_shade (32 of a stored state's 64 B; rd.implicit_primary == 2)
        }
        else { const float4 a = g0[j]; b = g1[j]; r.o = mk3(a.x, a.y, a.z); r.d = mk3(a.w, b.x, b.y); }
        RngKey key = make_key(0, 0, 0);
        if constexpr (RT_HAS_PRIM(C, RT_HIT_MEDIUM)) key = path_key(rd, primary ? j : g3[j].x);     // only a medium draws random numbers
        Closest c = closest_hit<C>(sc, tree, r, rd.k_min, rd.k_max, key, depth, cnt);
        rd.hit[j] = make_float2(c.k, __uint_as_float(c.obj));
        if constexpr (RT_HAS_PRIM(C, RT_HIT_MEDIUM)) rd.hit_aux[j] = c.aux;
        rays++;
        if (c.obj == RT_NO_HIT && finish) { finish_miss(sc, rd, depth, j, r.d, b, primary); sky++; }
    }
    stat_add_closest(rd.stats, rays, cnt, sky, depth);
}

// Persistent-lane closest hit.  In scenes where the cost of a ray has a heavy tail (a bounce ray skimming a 5 M-triangle terrain
// visits thousands of nodes, its 63 wave-mates a few dozen) a lane must not wait for its wave.  Every WAVE owns a contiguous run of
// the queue; the wave runs the postponed-leaf loop, and after each leaf round it votes (ballot + popcount): once fewer than
// REFILL_BELOW lanes are still traversing and at least REFILL_MIN_TAKERS are idle, the idle lanes store their results and take the
// run's next rays — consecutive positions, so a refill is a coalesced load and the rays in flight stay neighbours.  All control
// flow is wave-uniform and the result of a ray does not depend on the schedule.  Measured: 5 M triangles 790 -> 1014
// Msamples/s; on Night Spheres, whose tree is in LDS, the refill rounds cost what the idle lanes recover (15.0 vs 15.2 ms), and
// on the 10-object Cornell box they lose, so the host picks per scene.
#ifndef RT_REFILL_BELOW
#define RT_REFILL_BELOW 32
#define RT_REFILL_MIN_TAKERS 8
#endif
#ifndef RT_CP_WAVES
#define RT_CP_WAVES 1
#endif
#ifndef RT_CPW_WAVES
#define RT_CPW_WAVES 1      // waves per SIMD k_closest_persistent is held to when it walks the 8-wide tree
#endif
constexpr int REFILL_BELOW = RT_REFILL_BELOW, REFILL_MIN_TAKERS = RT_REFILL_MIN_TAKERS;

template <class C, int TREE>
__global__ void __launch_bounds__(TREE == TREE_LDS ? LDS_BLOCK_CLOSEST : BLOCK, TREE == TREE_WIDE ? RT_CPW_WAVES : RT_CP_WAVES) k_closest_persistent(SceneDev sc, RenderDev rd, uint32_t depth) {
    constexpr bool LDS = TREE == TREE_LDS, WIDE = TREE == TREE_WIDE;
    constexpr int TB = LDS ? LDS_BLOCK_CLOSEST : BLOCK;
    const auto tree = make_tree<TREE, TB>(sc);
    const uint32_t n = rd.qcount[depth];
    const float4* g0 = rd.st[depth & 1][0]; const float4* g1 = rd.st[depth & 1][1]; const uint2* g3 = rd.ids[depth & 1];
    const float kMin = rd.k_min, kMax = rd.k_max;
    // every wave owns a contiguous run of the queue and hands its next rays to whichever lanes are idle: a refill then loads
    // consecutive positions (coalesced), and the rays a wave works on at one time stay neighbours in the image
    const uint32_t n_waves = gridDim.x * (TB / 64), wave_id = blockIdx.x * (TB / 64) + (threadIdx.x >> 6);
    const uint32_t per_wave = (n + n_waves - 1) / n_waves;
    uint32_t next = min(n, wave_id * per_wave);                                  // wave-uniform cursor
    const uint32_t end = min(n, next + per_wave);
    const unsigned long long lt_mask = (1ull << lane_id()) - 1ull;
    uint32_t my_j = 0;
    uint32_t rays = 0, sky = 0;
    const bool finish = LDS && rd.misses_end_in_closest;       // a ray that hits nothing ends its path here (finish_miss)
    TravCount cnt = trav_count_zero();
    bool active = false;                 // this lane holds a ray (possibly finished and waiting for its store)
    Ray r; r.o = mk3(0, 0, 0); r.d = mk3(0, 0, 1);
    ClosestBox box = closest_box(r);                                             // binary trees
    int cur = TRAV_DONE, pend = TRAV_DONE; Stk sp = stk_empty();
    f3 inv = mk3(0, 0, 1); WideRay wr = make_wide_ray(r, inv, 0.0f, 1.0f); WideCursor wc = wide_idle();      // the wide tree
    Closest best; best.k = kMax; best.obj = RT_NO_HIT; best.aux = 0.0f;
    auto lane_busy = [&]() { if constexpr (WIDE) return wide_busy(wc); else return cur != TRAV_DONE || pend != TRAV_DONE; };
    for (;;) {
        // ---- store / fetch for every lane that is not traversing
        const unsigned long long idle = __ballot(!lane_busy());
        if (!lane_busy()) {
            if (active) {
                if constexpr (RT_HAS_PRIM(C, RT_HIT_MEDIUM)) {
                    if (!ray_has_nan(r)) {
                        RngKey key = path_key(rd, g3[my_j].x);        // only a medium draws random numbers
                        for (uint32_t i = 0; i < sc.n_media; i++) {
                            uint32_t obj = sc.media[i], code = sc.obj_code[obj];
                            float k, aux; cnt.prims++;
                            if (try_medium<C>(sc, code & OBJ_INDEX_MASK, obj, r, kMin, best.k, key, depth, k, aux)) { consider(best, k, obj, aux); cnt.prim_hit++; }
                        }
                    }
                    rd.hit_aux[my_j] = best.aux;
                }
                // GraphicsValidator.CheckValueRange, Core/Debugging/GraphicsValidator.cs:69 via Core/RenderJob.cs:609-614
                if (best.obj != RT_NO_HIT && !((best.k >= kMin) && (best.k <= kMax))) best.obj = RT_NO_HIT;
                rd.hit[my_j] = make_float2(best.k, __uint_as_float(best.obj));
                if (best.obj == RT_NO_HIT && finish) { finish_miss(sc, rd, depth, my_j, r.d, g1[my_j], false); sky++; }
                active = false;
            }
            const uint32_t j = next + (uint32_t)__popcll(idle & lt_mask);
            if (j < end) {
                float4 a = g0[j], b = g1[j];
                r.o = mk3(a.x, a.y, a.z); r.d = mk3(a.w, b.x, b.y);
                my_j = j; rays++;
                best.k = kMax; best.obj = RT_NO_HIT; best.aux = 0.0f;
                sp = stk_empty(); active = true;
                if (!ray_has_nan(r)) {
                    for (uint32_t i = 0; i < sc.n_unbounded; i++) {
                        uint32_t obj = sc.unbounded[i], code = sc.obj_code[obj];
                        float k; cnt.prims++;
                        if (try_object_solid<C>(sc, code >> OBJ_TYPE_SHIFT, code & OBJ_INDEX_MASK, r, kMin, best.k, k)) { consider(best, k, obj, 0.0f); cnt.prim_hit++; }
                    }
                    if (sc.n_nodes) {
                        if constexpr (WIDE) { inv = splat(1.0f) / r.d; wr = wide_ray_closest(sc, r, inv); wc = wide_start(wr); }
                        else { box = closest_box(r); cur = 0; }
                    }
                }
            }
        }
        next = min(end, next + (uint32_t)__popcll(idle));
        if (!__any(lane_busy())) {
            if (!__any(active) && next >= end) break;     // nothing traversing, nothing to store, nothing to fetch
            continue;                                    // (rays that were finished at fetch time: store them, fetch again)
        }
        // ---- traverse until too few lanes are busy
        for (;;) {
            if constexpr (WIDE) {
#ifdef RT_WIDE_PROF
                while (wide_vote(wc)) {
                    if (RT_WIDE_PROF == 1) { const uint32_t nu = (uint32_t)__popcll(__ballot(wide_can_step(wc))); if (lane_id() == 0) { cnt.box_miss += 64u; cnt.prim_hit += nu; } }
                    if (wide_can_step(wc)) cnt.nodes += wide_step(sc, tree, wr, kMin, best.k, wc);
                }
                if (RT_WIDE_PROF == 2) { const uint32_t nu = (uint32_t)__popcll(__ballot(wide_has_leaf(wc))); if (lane_id() == 0) { cnt.box_miss += 64u; cnt.prim_hit += nu; } }
                if (wide_has_leaf(wc)) { TravCount t2 = trav_count_zero(); wide_leaf_closest<C>(sc, wide_take_leaf(wc, wr), r, inv, kMin, best, t2); cnt.nodes += t2.nodes; cnt.prims += t2.prims; }
#else
                while (wide_vote(wc)) if (wide_can_step(wc)) cnt.nodes += wide_step(sc, tree, wr, kMin, best.k, wc);
                if (wide_has_leaf(wc)) wide_leaf_closest<C>(sc, wide_take_leaf(wc, wr), r, inv, kMin, best, cnt);
#endif
            } else {
                if constexpr (decltype(tree)::vote) { while (step_vote(cur, pend)) if (cur >= 0) { cnt.nodes += 2; node_step<false>(sc, tree, box, kMin, best.k, cur, pend, sp, cnt.box_miss); } }
                else while (cur >= 0) { cnt.nodes += 2; node_step<false>(sc, tree, box, kMin, best.k, cur, pend, sp, cnt.box_miss); }
                if (pend != TRAV_DONE) {
                    uint32_t obj = (uint32_t)~pend; float k; cnt.prims++;
                    if (tree.template leaf_try<C>(sc, obj, r, kMin, best.k, k)) { consider(best, k, obj, 0.0f); cnt.prim_hit++; }
                    leaf_advance(tree, cur, pend, sp);
                }
            }
            const int n_busy = __popcll(__ballot(lane_busy()));
            if (n_busy == 0) break;
            if (n_busy < REFILL_BELOW && next < end && 64 - n_busy >= REFILL_MIN_TAKERS) break;
        }
    }
    stat_add_closest(rd.stats, rays, cnt, sky, depth);
}

// Output positions.  A same-address returning atomic is serviced about every 10 ns chip-wide, so a reservation per wave
// (260 k per 16.6 M paths) made the kernel 8x slower than its memory traffic.  Whether a path survives and whether it asks
// for light samples follows from its hit alone, so a block first reads the hit ids of SHADE_ITEMS x BLOCK positions, counts,
// reserves both output ranges with ONE atomic each, and then shades; positions inside the tile keep the input order.
constexpr int SHADE_ITEMS = 8;
#ifndef RT_SHADE_MIN_WAVES
#define RT_SHADE_MIN_WAVES 4
#endif
constexpr int SHADE_MIN_WAVES = RT_SHADE_MIN_WAVES;
#ifndef RT_SHADE_DENSE
#define RT_SHADE_DENSE 1
#endif

// RenderJob.UniqueColourFromMaterialHash, RenderJob.cs:317-328 (see rt_hip.h for the stand-in hash)
RT_FN col3 material_hash_colour(uint32_t material, bool alternate) {
    uint32_t hash = (material + 1u) * 2654435761u;
    int o = alternate ? 1 : 0;
    return mkc((float)((hash >> (8 * (o + 0))) & 0xFFu) / 255.0f, (float)((hash >> (8 * (o + 1))) & 0xFFu) / 255.0f, (float)((hash >> (8 * (o + 2))) & 0xFFu) / 255.0f);
}
// the bounce-depth picture of a finished path, RenderJob.cs:303-305
RT_FN col3 bounce_depth_colour(int maxDepthReached, uint32_t maxAllowedDepth) {
    if (maxDepthReached < 0) return mkc(1, 0, 1);
    float val = 1 - powf(E_F, -((float)(maxDepthReached * maxDepthReached) * sqrtf((float)maxAllowedDepth)));
    return (mkc(1, 1, 1) * (1.0f - val)) + ((mkc(0, 0, 1) * 0.02f) * val);
}

// `sum /= samples` at the end of Material.CalculateSimpleColourFromLights (Material.cs:30-45), :42: Colour / float divides every channel
template <bool EXACT> RT_FN col3 over_samples(col3 sum, float S) { if constexpr (EXACT) return sum / S; else return sum * rcp_fast(S); }

// VIS = RenderOptions.DebugVisualisation != None (RenderJob.cs:148-309): the same kernel decides a debug colour from the first hit
// instead of shading it (the host runs one bounce then, except for BounceDepth which runs the whole loop without colours).
// VALIDATE = RT_RENDER_VALIDATE (GraphicsValidator in full, RenderJob.cs:588-607): UVs are computed for every hit, and both warn-only
// rules are also counted per object (rd.validate), as GraphicsValidator.Errors files them.  Everything else is the same arithmetic.
template <class C, bool VIS = false, bool VALIDATE = false>
__global__ void __launch_bounds__(BLOCK, VIS || VALIDATE ? 1 : SHADE_MIN_WAVES) k_shade(SceneDev sc, RenderDev rd, uint32_t depth) {
    const uint32_t n = rd.qcount[depth];
    const int cur = depth & 1, nxt = cur ^ 1;
    uint32_t sky = 0, limit = 0, scat = 0, samples = 0, hist_miss = 0, n_warn = 0, uv_warn = 0;      // per thread: far below 2^32 (a thread shades a few thousand paths per launch)
    __shared__ uint32_t wcnt[3][SHADE_ITEMS][BLOCK / 64];      // per item and wave: survivors, light requests, hits
    __shared__ uint32_t bases[3];                              // first output position of the tile in either queue; hits in the tile
    constexpr uint32_t TILE = SHADE_ITEMS * BLOCK;
    // Dense shading (trees in LDS, whose closest-hit kernel already finished the paths that missed): a third of the tile's positions
    // are holes then, and shading position by position would leave a third of the lanes idle in a kernel that is bound by its
    // arithmetic.  The hits' positions are written to a list in tile order, and the lanes shade list entry k = the k-th hit —
    // whose output positions are base + k in both queues (every hit survives below the depth limit and asks for light if there
    // is any: no Volumetric material in these variants) — in as many rounds as the hits need, not SHADE_ITEMS.  Same paths, same
    // output order, same bits.
    __shared__ unsigned short hit_list[TILE];
    const bool dense = !VIS && !RT_HAS_MAT(C, RT_MAT_VOLUMETRIC) && rd.misses_end_in_closest && RT_SHADE_DENSE;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const bool can_survive = depth < rd.max_depth;
    const bool primary = !VIS && rd.implicit_primary && depth == 0;
    // a table of object records small enough (a few hundred objects) is copied into the workgroup's LDS: the gather per hit is then an LDS read
    const bool rec_lds = !VIS && rd.obj_shade_lds_bytes != 0;
    if (rec_lds) {
        uint4* dst = (uint4*)lds_stack; const uint4* src = (const uint4*)sc.obj_shade;
        for (uint32_t i = threadIdx.x; i < rd.obj_shade_lds_bytes / 16; i += BLOCK) dst[i] = src[i];
        __syncthreads();
    }
    for (uint32_t tile = blockIdx.x * TILE; tile < n; tile += gridDim.x * TILE) {     // uniform per block
      uint32_t m_surv = 0, m_shad = 0, m_hit = 0;
#pragma unroll
      for (int item = 0; item < SHADE_ITEMS; item++) {
          const uint32_t j = tile + (uint32_t)item * BLOCK + threadIdx.x;
          bool hit = false, lit = false;
          if (j < n) {
              uint32_t obj = __float_as_uint(rd.hit[j].y);
              hit = obj != RT_NO_HIT;
              lit = hit && sc.n_lights > 0;
              if constexpr (VIS) lit = lit && rd.vis == RT_VIS_ESTIMATED_LIGHT;          // the only visualisation that samples lights, for every material
              else if constexpr (RT_HAS_MAT(C, RT_MAT_VOLUMETRIC)) { if (hit) lit = lit && sc.materials[sc.obj_material[obj]].type != RT_MAT_VOLUMETRIC; }
          }
          const bool surv = hit && can_survive;
          unsigned long long bs = __ballot(surv), bl = __ballot(lit), bh = __ballot(hit);
          if (lane == 0) { wcnt[0][item][wave] = (uint32_t)__popcll(bs); wcnt[1][item][wave] = (uint32_t)__popcll(bl); wcnt[2][item][wave] = (uint32_t)__popcll(bh); }
          m_surv |= (surv ? 1u : 0u) << item; m_shad |= (lit ? 1u : 0u) << item; m_hit |= (hit ? 1u : 0u) << item;
      }
      __syncthreads();
      if (threadIdx.x < 3) {
          uint32_t tot = 0;
          for (int i = 0; i < SHADE_ITEMS; i++) for (int w = 0; w < BLOCK / 64; w++) tot += wcnt[threadIdx.x][i][w];
          bases[threadIdx.x] = threadIdx.x == 2 ? tot : tot ? atomicAdd((threadIdx.x == 0 ? rd.qcount + depth + 1 : rd.sqcount + depth), tot) : 0u;
      }
      if (dense) {                                        // the tile's hits, in tile order
          uint32_t row = 0;
#pragma unroll
          for (int item = 0; item < SHADE_ITEMS; item++) {
              const bool hit = (m_hit >> item) & 1u;
              uint32_t rank = row;
              for (uint32_t w = 0; w < BLOCK / 64; w++) { const uint32_t ch = wcnt[2][item][w]; if (w < wave) rank += ch; row += ch; }
              rank += (uint32_t)__popcll(__ballot(hit) & lt_mask);
              if (hit) hit_list[rank] = (unsigned short)((uint32_t)item * BLOCK + threadIdx.x);
          }
      }
      __syncthreads();
      uint32_t row_s = bases[0], row_l = bases[1];
      const uint32_t n_hits = bases[2];
      const int n_rounds = dense ? (int)((n_hits + BLOCK - 1) / BLOCK) : SHADE_ITEMS;
#pragma unroll 1
      for (int item = 0; item < n_rounds; item++) {
        uint32_t j, p0, p1; bool survive, shadow, valid;
        if (dense) {
            const uint32_t k = (uint32_t)item * BLOCK + threadIdx.x;
            valid = k < n_hits;
            j = tile + (valid ? (uint32_t)hit_list[k] : 0u);
            survive = valid && can_survive; shadow = valid && sc.n_lights > 0;
            p0 = row_s + k; p1 = row_l + k;
        } else {
            j = tile + (uint32_t)item * BLOCK + threadIdx.x;
            survive = (m_surv >> item) & 1u; shadow = (m_shad >> item) & 1u;
            // this path's output positions: rows before, waves before in the row, lanes before in the wave
            p0 = row_s; p1 = row_l;
            for (uint32_t w = 0; w < BLOCK / 64; w++) {
                uint32_t cs = wcnt[0][item][w], cl = wcnt[1][item][w];
                if (w < wave) { p0 += cs; p1 += cl; }
                row_s += cs; row_l += cl;
            }
            p0 += (uint32_t)__popcll(__ballot(survive) & lt_mask);
            p1 += (uint32_t)__popcll(__ballot(shadow) & lt_mask);
            // (a path whose ray hit nothing was finished by the closest-hit kernel — finish_miss — and its state is not even loaded here;
            //  debug visualisations colour their misses themselves)
            valid = j < n && (VIS || !rd.misses_end_in_closest || ((m_hit >> item) & 1u));
        }
        if (valid) {
            Ray r; col3 T, L; uint2 a3;
            if (primary) {                                                               // nothing stored for depth 0: the camera ray again (primary_ray)
                if (rd.implicit_primary == 2u) { const float4 a0 = rd.st[cur][0][j], a1 = rd.st[cur][1][j]; r.o = mk3(a0.x, a0.y, a0.z); r.d = mk3(a0.w, a1.x, a1.y); }
                else r = primary_ray(sc, rd, j);
                T = mkc(1, 1, 1); L = mkc(0, 0, 0); a3 = make_uint2(j, pack_mats(RT_NO_HIT, RT_NO_HIT));
            } else {
                const float4 a0 = rd.st[cur][0][j], a1 = rd.st[cur][1][j], a2 = rd.st[cur][2][j]; a3 = rd.ids[cur][j];
                r.o = mk3(a0.x, a0.y, a0.z); r.d = mk3(a0.w, a1.x, a1.y);
                T = mkc(a1.z, a1.w, a2.x);
                L = mkc(a2.y, a2.z, a2.w);
            }
            const float2 hk = rd.hit[j];
            const uint32_t id = a3.x;
            Closest c; c.k = hk.x; c.obj = __float_as_uint(hk.y); c.aux = 0.0f;
            if constexpr (RT_HAS_PRIM(C, RT_HIT_MEDIUM)) { if (rd.hit_aux) c.aux = rd.hit_aux[j]; }      // (the debug-visualisation kernel is the generic one on every scene; a render of a scene without media has no hit_aux)
            if (c.obj == RT_NO_HIT) {                                                    // RenderJob.cs:444-450
                if constexpr (VIS) { L = rd.vis == RT_VIS_BOUNCE_DEPTH ? bounce_depth_colour((int)depth - 1, rd.max_depth) : mkc(0, 0, 0); }   // NoColour / :303-305
                else L = L + (T * sky_colour(sc, r.d));
                sky++; samples++;
                if (depth > 0) hist_miss++;                                              // RenderJob.cs:458: maxDepthReached = depth - 1
                rd.out[id] = make_float4(L.r, L.g, L.b, 0.0f);
            } else {
                RngKey key = path_key(rd, id);
                // The hit's record (small scenes whose textures are all solid colours: sc.obj_shade, one 64-B line per object): what the
                // object's code, its primitive (a sphere or an infinite plane), its material and the material's two textures say,
                // resolved per object by the host — one gather instead of the chains object -> primitive and object -> material ->
                // textures.  Same numbers either way.
                const bool flat = !VIS && sc.obj_shade != nullptr;
                float4 s0 = make_float4(0, 0, 0, 0), s1 = s0, s2 = s0, s3 = s0;
                if (flat && rec_lds) { const float4* rec = (const float4*)lds_stack + 4 * c.obj; s0 = rec[0]; s1 = rec[1]; s2 = rec[2]; s3 = rec[3]; }
                else if (flat) { const float4* rec = sc.obj_shade + 4 * (size_t)c.obj; s0 = rec[0]; s1 = rec[1]; s2 = rec[2]; s3 = rec[3]; }
                HitRec h = finalize_hit<C, VIS || VALIDATE>(sc, r, c, key, depth, flat && __float_as_uint(s3.z) != 0u, s2, __float_as_uint(s3.x), __float_as_uint(s3.y));
                if (fabsf(length_sq(h.n) - 1.0f) > 0.01f) {                              // GraphicsValidator.CheckVectorNormalized: !(Abs(LengthSquared - 1) > 0.01)
                    n_warn++;
                    if constexpr (VALIDATE) atomicAdd(rd.validate + 2 * (size_t)c.obj, 1ull);        // RecordError(NormalsWrongMagnitude, obj), RenderJob.cs:597
                }
                if constexpr (VIS || VALIDATE) {
                    if (!((h.u >= 0.0f) && (h.u <= 1.0f) && (h.v >= 0.0f) && (h.v <= 1.0f))) {     // CheckUVCoordValid
                        uv_warn++;
                        if constexpr (VALIDATE) atomicAdd(rd.validate + 2 * (size_t)c.obj + 1, 1ull);  // RecordError(UVInvalid, obj), RenderJob.cs:606
                    }
                }
                const rt_material* m = sc.materials + h.mat;
                const uint32_t mtype = flat ? (__float_as_uint(s1.w) & 0xFFu) : m->type, mflags = flat ? (__float_as_uint(s1.w) >> 8) : m->flags;
                const float mf0 = flat ? s0.w : m->f[0];
                auto tex0 = [&]() { return flat ? mkc(s0.x, s0.y, s0.z) : texture_colour<C>(sc, m->tex[0], h.wp); };
                auto tex1 = [&]() { return flat ? mkc(s1.x, s1.y, s1.z) : texture_colour<C>(sc, m->tex[1], h.wp); };
                uint32_t prev = unpack_mat(a3.y), first = unpack_mat(a3.y >> 16);
                float4 u = rng_draw(key, depth, SLOT_SCATTER, 0);
                Ray nr; nr.o = h.wp; nr.d = r.d;
                col3 a = mkc(1, 1, 1);        // multiplier of the future colour
                col3 W = mkc(0, 0, 0);        // weight of the light sum
                switch (mtype) {
                    case RT_MAT_STANDARD: if constexpr (RT_HAS_MAT(C, RT_MAT_STANDARD)) {                                            // StandardMaterial.cs:64-80 (Scatter :64-72, CalculateColour :74-80)
                        f3 diffuse = on_unit_sphere(u.x, u.y, u.z);
                        if (dot(diffuse, h.n) < 0) diffuse = diffuse * -1.0f;
                        f3 refl = reflect(r.d, h.n);
                        nr.d = normalize(normalize(lerp3(refl, diffuse, mf0)));
                        a = tex0();
                        L = L + (T * tex1());
                        W = T * a;
                    } break;
                    case RT_MAT_PHONG: if constexpr (RT_HAS_MAT(C, RT_MAT_PHONG)) {                                               // PhongMaterial.cs:26-85 (Scatter :26-37, CalculateColour :39-85)
                        f3 dir = on_unit_sphere(u.x, u.y, u.z);
                        if (dot(dir, h.n) < 0) dir = dir * -1.0f;
                        nr.d = normalize(dir);
                        col3 spec = mkc(m->f[0], m->f[1], m->f[2]), diff = mkc(m->f[3], m->f[4], m->f[5]), amb = mkc(m->f[6], m->f[7], m->f[8]);
                        f3 rl = reflect(r.d, -h.n);
                        float sw = powf(fabsf(dot(r.d, rl)), powf(2.0f, m->f[9]));
                        L = L + (T * amb);
                        a = diff + (spec * sw);
                        W = T;
                        if (shadow) rd.sh[3][p1] = make_float4(r.d.x, r.d.y, r.d.z, 0.0f);
                    } break;
                    case RT_MAT_VOLUMETRIC: if constexpr (RT_HAS_MAT(C, RT_MAT_VOLUMETRIC)) {                                          // ConstantDensityMedium.cs:102-129
                        nr.d = normalize(on_unit_sphere(u.x, u.y, u.z));
                        float pw = m->f[3] * h.shader;
                        a = mkc(powf(m->f[0], pw), powf(m->f[1], pw), powf(m->f[2], pw));
                    } break;
                    default: if constexpr (RT_HAS_MAT(C, RT_MAT_REFRACTIVE) || RT_HAS_MAT(C, RT_MAT_EMISSIVE_REFRACTIVE)) {            // RefractiveMaterial.cs:37-103, EmissiveRefractiveMaterial.cs:43-64
                        f3 ud = normalize(r.d);
                        float cosT = minf_net(dot(-ud, h.n), 1.0f);
                        float sinT = sqrtf(1.0f - (cosT * cosT));
                        bool entering = (depth == 0) || (prev != h.mat);
                        float eta = entering ? 1.0f : mf0, etaP = entering ? mf0 : 1.0f;
                        float ratio = eta / etaP;
                        bool cannot = ratio * sinT > 1.0f;
                        float r0 = (eta - etaP) / (eta + etaP);
                        r0 *= r0;
                        float reflectance = r0 + ((1 - r0) * powf(1 - cosT, 5.0f));
                        if (reflectance > u.w) cannot = true;
                        f3 out;
                        if (cannot) out = reflect(ud, h.n);
                        else if (!(mflags & RT_MAT_FLAG_ALTERNATE_REFRACTION)) {
                            f3 perp = ratio * (ud + (cosT * h.n));
                            f3 par = -sqrtf(fabsf(1.0f - length_sq(perp))) * h.n;
                            out = perp + par;
                        } else {
                            float nd = dot(h.n, ud);
                            out = normalize((sqrtf((1 - powf(ratio, 2.0f)) * (1 - powf(nd, 2.0f))) * h.n) + (ratio * (ud - (nd * h.n))));
                        }
                        nr.d = normalize(out);
                        a = tex0();
                        W = T * a;
                        if (mtype == RT_MAT_EMISSIVE_REFRACTIVE) {
                            bool emit = (mflags & RT_MAT_FLAG_DIRECT_EMISSION) || !(depth == 0 || (depth == 1 && first == h.mat));
                            if (emit) L = L + (T * tex1());
                        }
                    } break;
                }
                scat++;                                                                   // RenderJob.cs:439
                if constexpr (VIS) {
                    uint32_t pass_, pix; path_pass_pixel(rd, id, pass_, pix);
                    const uint32_t px = pix % rd.width, py = pix / rd.width;
                    float z = h.k - rd.k_min;
                    switch (rd.vis) {
                        case RT_VIS_NORMALS: { f3 n = (h.n + splat(1.0f)) / 2.0f; L = mkc(n.x, n.y, n.z); } break;
                        case RT_VIS_FACE_DIRECTION: L = h.outside ? mkc(0, 1, 0) : mkc(1, 0, 0); break;
                        case RT_VIS_DISTANCE_CLOSE: { float v = powf(E_F, -(0.01f * z * z)); L = mkc(v, v, v); } break;
                        case RT_VIS_DISTANCE_MID:   { float v = powf(E_F, -(0.0001f * z * z)); L = mkc(v, v, v); } break;
                        case RT_VIS_DISTANCE_FAR:   { float v = powf(E_F, -(.000001f * z * z)); L = mkc(v, v, v); } break;
                        case RT_VIS_UV_COORDS: L = mkc(h.u, h.v, 1); break;
                        case RT_VIS_PIXEL_COORD_TEXTURE: L = material_hash_colour(h.mat, sinf((float)px / 2.0f) * sinf((float)py / 2.0f) < 0); break;
                        case RT_VIS_WORLD_COORD_TEXTURE: L = material_hash_colour(h.mat, sinf(h.wp.x * 40.0f) * sinf(h.wp.y * 40.0f) * sinf(h.wp.z * 40.0f) < 0); break;
                        case RT_VIS_LOCAL_COORD_TEXTURE: L = material_hash_colour(h.mat, sinf(h.lp.x * 40.0f) * sinf(h.lp.y * 40.0f) * sinf(h.lp.z * 40.0f) < 0); break;
                        case RT_VIS_SCATTER_DIRECTION: { f3 n = (nr.d + splat(1.0f)) / 2.0f; L = mkc(n.x, n.y, n.z); } break;
                        case RT_VIS_ESTIMATED_LIGHT: L = mkc(0, 0, 0); W = mkc(1, 1, 1); break;       // k_shadow adds sum / samples
                        default: L = bounce_depth_colour((int)depth, rd.max_depth); break;            // BounceDepth: used if the path ends here (bounce limit)
                    }
                }
                if (shadow && !VIS && requests_are_compact(rd, depth)) rd.sh[0][p1] = make_float4(h.n.x, h.n.y, h.n.z, __uint_as_float(p0));
                else if (shadow) {
                    rd.sh[0][p1] = make_float4(h.wp.x, h.wp.y, h.wp.z, h.n.x);
                    rd.sh[1][p1] = make_float4(h.n.y, h.n.z, W.r, W.g);
                    rd.sh[2][p1] = make_float4(W.b, __uint_as_float(id), __uint_as_float(survive ? p0 : RT_NO_HIT), __uint_as_float(c.obj));
                }
                T = T * a;
                if (survive) {
                    rd.st[nxt][0][p0] = make_float4(nr.o.x, nr.o.y, nr.o.z, nr.d.x);
                    rd.st[nxt][1][p0] = make_float4(nr.d.y, nr.d.z, T.r, T.g);
                    rd.st[nxt][2][p0] = make_float4(T.b, L.r, L.g, L.b);
                    rd.ids[nxt][p0] = make_uint2(id, pack_mats(h.mat, depth == 0 ? h.mat : first));
                } else {                                                                  // RenderJob.cs:416-421: the next depth exceeds the limit, future = NoColour
                    limit++; samples++;                                                   // maxDepthReached = depth
                    rd.out[id] = make_float4(L.r, L.g, L.b, 0.0f);
                }
            }
        }
      }
      __syncthreads();       // wcnt / bases are reused by the next tile
    }
    {   // (RenderJob.cs:458: a miss at depth d ends a path of maxDepthReached d-1, the bounce limit one of depth d; a primary miss is not counted)
        const int slot[8] = {ST_SKY, ST_LIMIT, ST_SCATTER, ST_SAMPLES, ST_DEPTH0 + (int)min(depth > 0 ? depth - 1 : 0u, (uint32_t)RT_DEPTH_HIST_SIZE - 1),
                             ST_DEPTH0 + (int)min(depth, (uint32_t)RT_DEPTH_HIST_SIZE - 1), ST_NORMAL_WARN, ST_UV_WARN};
        const unsigned long long v[8] = {sky, limit, scat, samples, depth > 0 ? hist_miss : 0u, limit, n_warn, uv_warn};
        stat_add_block<8>(rd.stats, slot, v);
    }
}

// Material.CalculateSimpleColourFromLights (Material.cs:30-45) -> SimpleLightBase.CalculateLight (SimpleLightBase.cs:48-66).
// EXACT: light sampling in the reference's correctly rounded arithmetic instead of the 1-ulp hardware forms (rt_device.h).
// waves per SIMD the LDS-tree shadow kernel is held to.  6 (80 VGPRs) spills 15 registers around the traversal, 7 more: measured on
// Night Spheres 1080p per 256-pass step 5 waves 74.5 ms, 6 waves 80.7 ms, 7 waves 87.2 ms (profiles/r02_d_ab.log)
#ifndef RT_SHADOW_LDS_WAVES
#define RT_SHADOW_LDS_WAVES 4
#endif
#ifndef RT_SKIP_MUTE_RAYS
#define RT_SKIP_MUTE_RAYS 1
#endif
// waves per SIMD k_shadow is held to when the tree is read from HBM, per variant (none specialised: holding the Cornell variant to
// 4 waves = 128 VGPRs instead of its 157 spills 22 of them and is slower, 56.9 vs 53.5 ms per batch)
template <class C> struct ShadowTune { static constexpr int min_waves = 1; };
template <class C, int TREE, bool EXACT = false>
__global__ void __launch_bounds__(tree_in_lds(TREE) ? LDS_BLOCK_SHADOW : BLOCK, EXACT ? 1 : tree_in_lds(TREE) ? RT_SHADOW_LDS_WAVES : ShadowTune<C>::min_waves) k_shadow(SceneDev sc, RenderDev rd, uint32_t depth) {
    constexpr bool LDS = tree_in_lds(TREE);
    constexpr int TB = LDS ? LDS_BLOCK_SHADOW : BLOCK;
    const auto tree = make_tree<TREE, TB, false>(sc);
    const uint32_t n = rd.sqcount[depth];
    const bool compact = requests_are_compact(rd, depth);
    uint32_t rays = 0;                                // per thread: at most a few thousand requests x rays x nodes
    TravCount cnt = trav_count_zero();
    float4* next_l = rd.st[(depth & 1) ^ 1][2];
    for (uint32_t p = blockIdx.x * TB + threadIdx.x; p < n; p += gridDim.x * TB) {
        const ReqHead rq = req_head(rd, depth, p, compact);
        const uint32_t rq_dst = rq.dst;
        const f3 wp = rq.wp, nrm = rq.nrm;                          // (the weight W and the destination are read after the traversals: 4 registers fewer across them)
        RngKey key = path_key(rd, rq.id);
        const rt_material* m = nullptr;
        bool phong = false;
        f3 in = mk3(0, 0, 0);
        if constexpr (RT_HAS_MAT(C, RT_MAT_PHONG)) {
            if (!compact) {                                         // (compact requests: no Phong material in the scene)
                m = sc.materials + sc.obj_material[__float_as_uint(rq.obj_bits)];
                phong = m->type == RT_MAT_PHONG && rd.vis != RT_VIS_ESTIMATED_LIGHT;      // EstimatedLight is the plain light sum whatever the material (RenderJob.cs:234-250)
                if (phong) { float4 h3 = rd.sh[3][p]; in = mk3(h3.x, h3.y, h3.z); }
            }
        }
        col3 sum = mkc(0, 0, 0), spec_sum = mkc(0, 0, 0);
        for (uint32_t s = 0; s < rd.light_samples; s++) {
            for (uint32_t li = 0; li < sc.n_lights; li++) {
                const rt_light* l = sc.lights + li;
                uint32_t sub = s * sc.n_lights + li;
                ShadowRay sr = light_shadow_ray<C, EXACT>(sc, l, wp, nrm, key, depth, SLOT_LIGHT, sub);
                rays++;
                col3 c = mkc(0, 0, 0);
                if (!phong) {
                    // What the light adds if nothing is in the way is known before the ray is traced.  When the running sum does not
                    // change by it (a light beyond its cutoff, or one so dim next to what is already there that the addition rounds
                    // back — the far lights of a many-light scene), occluded and unoccluded give the same bits: the ray is not traced.
                    // (Only the yes/no is kept across the traversal, the contribution is worked out again afterwards: registers.  The host
                    //  asks for the check where it can pay: more than two rays per request, or a light with a cutoff.)
                    bool mute = false;
                    if (RT_SKIP_MUTE_RAYS && rd.skip_mute_rays) { const col3 t = sum + light_contribution<EXACT>(l, sr.kMax); mute = t.r == sum.r && t.g == sum.g && t.b == sum.b; }
                    if (!mute && !any_hit<C>(sc, tree, sr.r, sr.kMin, sr.kMax, rd.any_cap, cnt)) sum = sum + light_contribution<EXACT>(l, sr.kMax);
                    continue;
                }
                if (!any_hit<C>(sc, tree, sr.r, sr.kMin, sr.kMax, rd.any_cap, cnt)) c = light_contribution<EXACT>(l, sr.kMax);
                if constexpr (RT_HAS_MAT(C, RT_MAT_PHONG)) {                              // PhongMaterial.cs:50-65
                    ShadowRay s2 = light_shadow_ray<C, EXACT>(sc, l, wp, nrm, key, depth, SLOT_LIGHT_RAW, sub);
                    rays++;
                    col3 c2 = any_hit<C>(sc, tree, s2.r, s2.kMin, s2.kMax, rd.any_cap, cnt) ? mkc(0, 0, 0) : mkc(l->colour[0], l->colour[1], l->colour[2]);
                    f3 rl = reflect(s2.r.d, -nrm);
                    float sd = fabsf(dot(in, rl));
                    spec_sum = spec_sum + (c2 * powf(sd, powf(2.0f, m->f[9])));
                    float dd = fabsf(dot(sr.r.d, nrm));
                    sum = sum + (c * dd);
                }
            }
        }
        const float S = (float)rd.light_samples;
        sum = over_samples<EXACT>(sum, S);
        col3 W; uint32_t dst, id;
        req_tail(rd, depth, p, compact, rq_dst, W, dst, id);
        col3 add;
        if (!phong) add = W * sum;
        else { spec_sum = over_samples<EXACT>(spec_sum, S); add = W * ((sum * mkc(m->f[3], m->f[4], m->f[5])) + (spec_sum * mkc(m->f[0], m->f[1], m->f[2]))); }
        float4* tgt = dst != RT_NO_HIT ? next_l + dst : rd.out + id;
        float4 t = *tgt;
        if (dst != RT_NO_HIT) { t.y += add.r; t.z += add.g; t.w += add.b; } else { t.x += add.r; t.y += add.g; t.z += add.b; }
        *tgt = t;
    }
    stat_add3_block(rd.stats, ST_SHADOW_RAYS, ST_SH_NODES, ST_SH_PRIMS, rays, cnt.nodes, cnt.prims);
}

// One lane per SHADOW RAY instead of per request ("1 thread per (hit, light, light-sample)").  A block pass takes
// G = TB / R whole requests (R = lights x samples rays each); thread t traces ray t mod R of request t / R, so the R rays of a
// request — same origin, R lights — sit in neighbouring lanes and walk the same top of the tree (with 64 lights a wave IS one
// request: its node fetches hit the same cache lines), and no lane carries a request's running sums through 64 traversals.
// The contributions meet in LDS and the request's first lane adds them in ray order, 0 + c0 + c1 + ...: the same additions in
// the same order as k_shadow, so the same bits.  Not for Phong materials (their second ray per light folds two sums): the
// host launches it only for scenes without them, and only when R <= TB.
// waves per SIMD the ray-lane kernel is held to when the tree is read through the caches, per variant.  The Cornell variant sits at
// 127 VGPRs: one register more and it runs 3 waves instead of 4 (its shadow kernel 47.9 -> 56 ms per batch) — pinned, so that a
// change elsewhere spills a register outside the loops instead of losing the wave.
template <class C> struct ShadowRaysTune { static constexpr int min_waves = 1; };
template <class C, int TREE, bool EXACT = false>
__global__ void __launch_bounds__(TREE == TREE_LDS ? LDS_BLOCK_RAYS : BLOCK, TREE == TREE_LDS ? (EXACT ? 1 : 6) : (EXACT ? 1 : ShadowRaysTune<C>::min_waves)) k_shadow_rays(SceneDev sc, RenderDev rd, uint32_t depth) {
    constexpr bool LDS = TREE == TREE_LDS;
    constexpr int TB = LDS ? LDS_BLOCK_RAYS : BLOCK;
    const auto tree = make_tree<TREE, TB>(sc);
    __shared__ float contrib[3][TB];
    const uint32_t n = rd.sqcount[depth];
    const uint32_t n_lights = sc.n_lights, R = rd.light_samples * n_lights;
    const bool compact = requests_are_compact(rd, depth);
    const uint32_t G = (uint32_t)TB / R;                                          // whole requests per block pass (the host guarantees 1 <= R <= TB)
    const uint32_t my_req = threadIdx.x / R, q = threadIdx.x - my_req * R;        // q = sample * n_lights + light: the order k_shadow walks
    const uint32_t li = q % n_lights;
    uint32_t rays = 0;
    TravCount cnt = trav_count_zero();
    float4* next_l = rd.st[(depth & 1) ^ 1][2];
    for (uint32_t base = blockIdx.x * G; base < n; base += gridDim.x * G) {       // (block-uniform: every thread reaches both barriers)
        const uint32_t p = base + my_req;
        const bool have = my_req < G && p < n;
        col3 c = mkc(0, 0, 0);
        uint32_t rq_dst = 0;
        if (have) {
            const ReqHead rq = req_head(rd, depth, p, compact);
            rq_dst = rq.dst;
            const f3 wp = rq.wp, nrm = rq.nrm;
            const RngKey key = path_key(rd, rq.id);
            const rt_light* l = sc.lights + li;
            const ShadowRay sr = light_shadow_ray<C, EXACT>(sc, l, wp, nrm, key, depth, SLOT_LIGHT, q);
            rays++;
            // (a light that adds exactly nothing — beyond its cutoff, SimpleLightBase.cs:56 — needs no ray: the zero is the same; what the
            //  request lanes skip on top of that, k_shadow, needs the running sum, which ray lanes do not have)
            bool mute = false;
            if (RT_SKIP_MUTE_RAYS && rd.skip_mute_rays) { const col3 u = light_contribution<EXACT>(l, sr.kMax); mute = u.r == 0.0f && u.g == 0.0f && u.b == 0.0f; }
            if (!mute && !any_hit<C>(sc, tree, sr.r, sr.kMin, sr.kMax, rd.any_cap, cnt)) c = light_contribution<EXACT>(l, sr.kMax);
        }
        contrib[0][threadIdx.x] = c.r; contrib[1][threadIdx.x] = c.g; contrib[2][threadIdx.x] = c.b;
        __syncthreads();
        if (have && q == 0) {
            col3 sum = mkc(0, 0, 0);
            for (uint32_t i = 0; i < R; i++) sum = sum + mkc(contrib[0][threadIdx.x + i], contrib[1][threadIdx.x + i], contrib[2][threadIdx.x + i]);
            sum = over_samples<EXACT>(sum, (float)rd.light_samples);
            col3 W; uint32_t dst, id;
            req_tail(rd, depth, p, compact, rq_dst, W, dst, id);
            const col3 add = W * sum;
            float4* tgt = dst != RT_NO_HIT ? next_l + dst : rd.out + id;
            float4 t = *tgt;
            if (dst != RT_NO_HIT) { t.y += add.r; t.z += add.g; t.w += add.b; } else { t.x += add.r; t.y += add.g; t.z += add.b; }
            *tgt = t;
        }
        __syncthreads();
    }
    stat_add3_block(rd.stats, ST_SHADOW_RAYS, ST_SH_NODES, ST_SH_PRIMS, rays, cnt.nodes, cnt.prims);
}
// Persistent-lane k_shadow (same reasoning as k_closest_persistent).  A request's rays are traced one after the other by its
// lane, and an occluded ray ends at its first hit: with 64 lights the lane whose rays all reach their light sets the pace of the
// wave 64 times per request.  Here a lane that has finished a ray accounts for it, draws the next one of its request (or writes
// the request's sum and takes the next request) while the rest of the wave keeps traversing; the votes are those of
// k_closest_persistent.  Per request the rays, their order and the order of the additions are the same as in k_shadow: same bits.
// waves per SIMD the kernel is held to, per variant (specialised next to the variants: 96 VGPRs instead of 99 for "spheres" is
// 5 waves instead of 4 and 33.5 -> 36.8 Msamples/s on the 100 k spheres + 64 lights scene; the others would only spill)
template <class C> struct ShadowPersistentTune { static constexpr int min_waves = 1; };
template <class C> struct ShadowPersistentWideTune { static constexpr int min_waves = 1; };
#ifndef RT_SPQ_WAVES
#define RT_SPQ_WAVES 1
#endif
#ifndef RT_SPW_KEEP_CONTRIBUTION
#define RT_SPW_KEEP_CONTRIBUTION 1
#endif
#ifdef RT_SPW_PROF
constexpr uint32_t SPW_COUNTS = 0;      // (measurement builds: the three counters hold rounds and lanes, see below)
#else
constexpr uint32_t SPW_COUNTS = 1;
#endif
template <class C, int TREE, bool EXACT = false>
__global__ void __launch_bounds__(TREE == TREE_LDS ? LDS_BLOCK_SHADOW : BLOCK, EXACT ? 1 : TREE == TREE_QUAD ? RT_SPQ_WAVES : TREE == TREE_WIDE ? ShadowPersistentWideTune<C>::min_waves : ShadowPersistentTune<C>::min_waves) k_shadow_persistent(SceneDev sc, RenderDev rd, uint32_t depth) {
    constexpr bool WIDE = TREE == TREE_WIDE || TREE == TREE_QUAD;
    constexpr int TB = TREE == TREE_LDS ? LDS_BLOCK_SHADOW : BLOCK;
    const auto tree = make_tree<TREE, TB>(sc);
    const uint32_t n = rd.sqcount[depth];
    const uint32_t n_lights = sc.n_lights, per_request = rd.light_samples * n_lights;
    const bool compact = requests_are_compact(rd, depth);
    // every wave owns a contiguous run of the request queue (as in k_closest_persistent)
    const uint32_t n_waves = gridDim.x * (TB / 64), wave_id = blockIdx.x * (TB / 64) + (threadIdx.x >> 6);
    const uint32_t per_wave = (n + n_waves - 1) / n_waves;
    uint32_t next = min(n, wave_id * per_wave);
    const uint32_t end = min(n, next + per_wave);
    const unsigned long long lt_mask = (1ull << lane_id()) - 1ull;
    float4* next_l = rd.st[(depth & 1) ^ 1][2];
    uint32_t rays = 0, nodes = 0, prims = 0;
    uint32_t p = 0, q = 0, li = 0, rq_dst = 0;      // q: index of the request's ray in flight, li = q mod n_lights
    bool have = false, flying = false, occluded = false, second = false;       // second: the Phong material's unattenuated ray of the same light
    f3 wp = mk3(0, 0, 0), nrm = mk3(0, 0, 1), in = mk3(0, 0, 0);
    RngKey key = path_key(rd, 0);
    const rt_material* m = nullptr; bool phong = false;
    col3 sum = mkc(0, 0, 0), spec_sum = mkc(0, 0, 0), cdd = mkc(0, 0, 0), unocc = mkc(0, 0, 0);      // unocc: what the ray in flight adds if it reaches its light (known since the mute check)
    Ray r; r.o = mk3(0, 0, 0); r.d = mk3(0, 0, 1);
    bool again = false;                  // the ray in flight gets a second look through grown boxes if it comes out unoccluded (any_box)
    AnyBox box = make_any_box(r, 0.0f, 1.0f);                                          // binary trees
    uint32_t unused = 0;
    float kMin = 0, kMax = 0;
    int cur = TRAV_DONE, pend = TRAV_DONE; Stk sp = stk_empty();
    auto wr = [&] { if constexpr (WIDE) return wide_ray_idle(tree); else return 0; }(); WideCursor wc = wide_idle();      // the wide trees
    auto lane_busy = [&]() { if constexpr (WIDE) return wide_busy(wc); else return cur != TRAV_DONE || pend != TRAV_DONE; };
    int spins = 0;
    if (per_request == 0) return;                                               // (the host launches no shadow kernel without lights)
    for (;;) {
        bool fetched = false;
#ifdef RT_SPW_PROF
        if (WIDE && RT_SPW_PROF == 2) { if (__any(!lane_busy()) && lane_id() == (uint32_t)__builtin_ctzll(__ballot(true))) rays++; if (!lane_busy()) nodes++; }
#endif
        if (!lane_busy()) {
            int look = -1;                                                      // the ray enters the tree: 0 = first look, 1 = second (ONE place below sets up the traversal: assigned in two, the ray's box lived in the private segment)
            if (flying && again && !occluded) look = 1;                         // ---- unoccluded in the boxes as they are: look again in the grown ones
            else if (flying) {                                                // ---- account for the finished ray
                flying = false;
                const rt_light* l = sc.lights + li;
                bool advance = true;
                if (!second) {
                    col3 c = mkc(0, 0, 0);
                    if (!occluded) c = RT_SPW_KEEP_CONTRIBUTION && RT_SKIP_MUTE_RAYS && rd.skip_mute_rays && !phong ? unocc : light_contribution<EXACT>(l, kMax);      // (kept from the mute check below: the same function of the same operands)
                    if (!phong) sum = sum + c;
                    else if constexpr (RT_HAS_MAT(C, RT_MAT_PHONG)) { cdd = c * fabsf(dot(r.d, nrm)); second = true; advance = false; }
                } else if constexpr (RT_HAS_MAT(C, RT_MAT_PHONG)) {              // PhongMaterial.cs:50-65
                    col3 c2 = occluded ? mkc(0, 0, 0) : mkc(l->colour[0], l->colour[1], l->colour[2]);
                    f3 rl = reflect(r.d, -nrm);
                    float sd = fabsf(dot(in, rl));
                    spec_sum = spec_sum + (c2 * powf(sd, powf(2.0f, m->f[9])));
                    sum = sum + cdd;
                    second = false;
                }
                if (advance) { q++; li++; if (li == n_lights) li = 0; }
                if (q == per_request) {                                         // ---- the request is complete: add its light to the path
                    const float S = (float)rd.light_samples;
                    sum = over_samples<EXACT>(sum, S);
                    col3 W; uint32_t dst, id;
                    req_tail(rd, depth, p, compact, rq_dst, W, dst, id);
                    col3 add;
                    if (!phong) add = W * sum;
                    else { spec_sum = over_samples<EXACT>(spec_sum, S); add = W * ((sum * mkc(m->f[3], m->f[4], m->f[5])) + (spec_sum * mkc(m->f[0], m->f[1], m->f[2]))); }
                    float4* tgt = dst != RT_NO_HIT ? next_l + dst : rd.out + id;
                    float4 t = *tgt;
                    if (dst != RT_NO_HIT) { t.y += add.r; t.z += add.g; t.w += add.b; } else { t.x += add.r; t.y += add.g; t.z += add.b; }
                    *tgt = t;
                    have = false;
                }
            }
            const uint32_t p_new = next + (uint32_t)__popcll(__ballot(!have) & lt_mask);      // (the idle lanes are the active ones here)
            if (!have && p_new < end) {                                         // ---- next request
                p = p_new; fetched = true;
                const ReqHead rq = req_head(rd, depth, p, compact);
                rq_dst = rq.dst;
                wp = rq.wp; nrm = rq.nrm;
                key = path_key(rd, rq.id);
                if constexpr (RT_HAS_MAT(C, RT_MAT_PHONG)) {
                    phong = false;
                    if (!compact) {                                             // (compact requests: no Phong material in the scene)
                        m = sc.materials + sc.obj_material[__float_as_uint(rq.obj_bits)];
                        phong = m->type == RT_MAT_PHONG && rd.vis != RT_VIS_ESTIMATED_LIGHT;
                        if (phong) { float4 h3 = rd.sh[3][p]; in = mk3(h3.x, h3.y, h3.z); }
                    }
                }
                sum = mkc(0, 0, 0); spec_sum = mkc(0, 0, 0);
                q = 0; li = 0; second = false; have = true;
            }
            if (have && !flying) {                                              // ---- next ray of the request
                ShadowRay sr = light_shadow_ray<C, EXACT>(sc, sc.lights + li, wp, nrm, key, depth, second ? SLOT_LIGHT_RAW : SLOT_LIGHT, q);
                r = sr.r; kMin = sr.kMin; kMax = sr.kMax;
#ifndef RT_SPW_PROF
                rays++;
#endif
                flying = true; occluded = false;
                if (RT_SKIP_MUTE_RAYS && rd.skip_mute_rays && !phong) {         // a ray whose light cannot change the sum is not traced (see k_shadow): accounted as if occluded, the same bits
                    const col3 unoccluded = light_contribution<EXACT>(sc.lights + li, kMax), t = sum + unoccluded;
                    occluded = t.r == sum.r && t.g == sum.g && t.b == sum.b;
                    if (RT_SPW_KEEP_CONTRIBUTION) unocc = unoccluded;
                }
                for (uint32_t i = 0; i < sc.n_unbounded && !occluded; i++) {
                    uint32_t code = sc.obj_code[sc.unbounded[i]]; prims += SPW_COUNTS;
                    occluded = fast_object_solid<C>(sc, code >> OBJ_TYPE_SHIFT, code & OBJ_INDEX_MASK, r, kMin, kMax);
                }
                if constexpr (RT_HAS_PRIM(C, RT_HIT_MEDIUM)) {
                    for (uint32_t i = 0; i < sc.n_media && !occluded; i++) {
                        rt_medium md = sc.media_rec[sc.obj_code[sc.media[i]] & OBJ_INDEX_MASK]; prims += SPW_COUNTS;
                        occluded = fast_object_solid<C>(sc, md.boundary_type, md.boundary_index, r, kMin, kMax);      // ConstantDensityMedium.cs:77
                    }
                }
                again = false;
                if (!occluded && sc.n_nodes) look = 0;
            }
            if (look >= 0) {
                if constexpr (WIDE) { wr = wide_ray_any(sc, tree, r, kMax, rd.any_cap, look, again); wc = wide_start(wr); }
                else { box = any_box(sc, r, kMax, rd.any_cap, look, again); cur = 0; sp = stk_empty(); }
            }
        }
        next += (uint32_t)__popcll(__ballot(fetched));
        if (!__any(lane_busy())) {
            if (!__any(flying) && next >= end) break;
            continue;                                                           // (rays decided without the tree: account, draw again)
        }
        // many lanes whose ray was decided without the tree (muted lights, the ground plane): serve them again before anybody traverses
        if (RT_SKIP_MUTE_RAYS && spins < 3 && __popcll(__ballot(flying && !lane_busy())) >= REFILL_MIN_TAKERS) { spins++; continue; }
        spins = 0;
        for (;;) {
            if constexpr (WIDE) {
#ifdef RT_SPW_PROF      // (one-off measurement builds: rounds of the wave against the lanes that use them, in the three counters.  1: step rounds, lane steps, leaf rounds; 2: refill rounds, lanes refilled, lane leaves)
                while (wide_vote(wc)) {
                    if (RT_SPW_PROF == 1 && lane_id() == (uint32_t)__builtin_ctzll(__ballot(true))) rays++;
                    if (wide_can_step(wc)) { const uint32_t nb = wide_step_any(sc, tree, wr, kMin, kMax, wc); nodes += RT_SPW_PROF == 1 ? 1u : 0u * nb; }
                }
                if (RT_SPW_PROF == 1 && __any(wide_has_leaf(wc)) && lane_id() == (uint32_t)__builtin_ctzll(__ballot(true))) prims++;
                if (wide_has_leaf(wc)) {
                    if (RT_SPW_PROF == 2) prims++;
                    if (wide_next_leaf_any<C>(sc, tree, wc, wr, r, kMin, kMax)) { occluded = true; wide_stop(wc); }
                }
#else
                while (wide_vote(wc)) if (wide_can_step(wc)) nodes += wide_step_any(sc, tree, wr, kMin, kMax, wc);
#endif
#ifndef RT_SPW_PROF
                if (wide_has_leaf(wc)) {
                    prims++;
                    if (wide_next_leaf_any<C>(sc, tree, wc, wr, r, kMin, kMax)) { occluded = true; wide_stop(wc); }
                }
#endif
            } else {
                while (step_vote(cur, pend)) if (cur >= 0) { nodes += 2; node_step<true>(sc, tree, box, kMin, kMax, cur, pend, sp, unused); }
                if (pend != TRAV_DONE) {
                    prims++;
                    if (tree.template leaf_fast<C>(sc, (uint32_t)~pend, r, kMin, kMax)) { occluded = true; cur = TRAV_DONE; pend = TRAV_DONE; }
                    else leaf_advance(tree, cur, pend, sp);
                }
            }
            const bool busy = lane_busy();
            const int n_busy = __popcll(__ballot(busy));
            if (n_busy == 0) break;
            if (n_busy < REFILL_BELOW && __popcll(__ballot(!busy && (flying || next < end))) >= REFILL_MIN_TAKERS) break;
        }
    }
    stat_add3_block(rd.stats, ST_SHADOW_RAYS, ST_SH_NODES, ST_SH_PRIMS, rays, nodes, prims);
}

#ifdef RT_KERNELS_COMMON
// UpdateBuffers' accumulation, RenderJob.cs:626-647: flip y, sum += sample — the passes of the batch are added in pass order.
// The g = 2^pass_shift passes of a slot are neighbours in rd.out (path_pass_pixel), so a workgroup reads BLOCK consecutive paths at a
// time — the g passes of BLOCK / g slots, one coalesced 4-KB read — through LDS, and the first BLOCK / g threads add their slot's g
// values in order; the running sums stay in those threads' registers from one group of passes to the next.
__global__ void __launch_bounds__(BLOCK) k_accumulate(RenderDev rd, float* accum, uint32_t n_passes) {
    __shared__ float4 stage[BLOCK];
    const uint32_t g = 1u << rd.pass_shift, slots_per_chunk = BLOCK >> rd.pass_shift, n_groups = n_passes >> rd.pass_shift;
    const uint32_t n_chunks = (rd.n_pixels + slots_per_chunk - 1) / slots_per_chunk;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint32_t slot = chunk * slots_per_chunk + threadIdx.x;
        const bool mine = threadIdx.x < slots_per_chunk && slot < rd.n_pixels;
        float* dst = nullptr; float r = 0, gr = 0, b = 0;
        if (mine) {
            const uint32_t pix = slot_pixel(rd, slot), x = pix % rd.width, y = pix / rd.width, yy = rd.height - y - 1;
            dst = accum + 3 * ((size_t)x + (size_t)yy * rd.width);
            r = dst[0]; gr = dst[1]; b = dst[2];
        }
        const uint32_t first = chunk * BLOCK;                                   // = chunk's first slot * g: where its paths start inside a group
        for (uint32_t k = 0; k < n_groups; k++) {
            if (first + threadIdx.x < rd.group_paths) stage[threadIdx.x] = rd.out[(size_t)k * rd.group_paths + first + threadIdx.x];
            __syncthreads();
            if (mine) {
#pragma unroll 8
                for (uint32_t p = 0; p < g; p++) { const float4 v = stage[(threadIdx.x << rd.pass_shift) + p]; r += v.x; gr += v.y; b += v.z; }
            }
            __syncthreads();
        }
        if (mine) { dst[0] = r; dst[1] = gr; dst[2] = b; }
    }
}

// n_devices > 1: the shards' colour sums, added in shard order — ((s0 + s1) + s2) + ... whatever order the copies arrived in — into
// the combined buffer (rt_lib.hip: combine_shards): every source is read once, the result written once
struct ShardSums { const float* p[RT_MAX_DEVICES]; uint32_t n; };
__global__ void __launch_bounds__(BLOCK) k_add_shards(float* dst, ShardSums src, size_t n_values) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n_values; i += (size_t)gridDim.x * BLOCK) {
        float s = src.p[0][i];
        for (uint32_t g = 1; g < src.n; g++) s += src.p[g][i];
        dst[i] = s;
    }
}

// RenderJob.cs:646-654 ; Colour.Clamp01 / Sqrt / (Rgb24), Colour.cs:197-234
__global__ void __launch_bounds__(BLOCK) k_resolve(const float* accum, uint32_t sample_count, size_t n_values, uint8_t* out) {
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n_values; i += (size_t)gridDim.x * BLOCK) {
        float c = accum[i] / (float)(int)sample_count;
        c = (c > 0.0f) ? c : 0.0f;
        c = (c < 1.0f) ? c : 1.0f;
        c = sqrtf(c);
        out[i] = (uint8_t)(255 * c);
    }
}

#endif

// Parity hooks: one thread per caller-supplied ray, same device functions as the render kernels.
template <class C, int TREE>
__global__ void __launch_bounds__(tree_in_lds(TREE) ? lds_block_closest(TREE) : BLOCK) k_trace_closest(SceneDev sc, const rt_ray* rays, uint64_t n, float kMin, float kMax, uint64_t seed, rt_hit* hits) {
    constexpr int TB = tree_in_lds(TREE) ? lds_block_closest(TREE) : BLOCK;
    const auto tree = make_tree<TREE, TB>(sc);
    for (uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TB) {
        Ray r; r.o = mk3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]); r.d = mk3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]);
        RngKey key = make_key(seed, (uint32_t)i, 0);
        TravCount cnt = trav_count_zero();
        Closest c = closest_hit<C>(sc, tree, r, kMin, kMax, key, 0, cnt);
        rt_hit h = {};
        h.object = RT_NO_HIT; h.material = RT_NO_HIT;
        if (c.obj != RT_NO_HIT) {
            HitRec hr = finalize_hit<C, true>(sc, r, c, key, 0);
            h.k = hr.k; h.object = hr.obj; h.material = hr.mat;
            h.normal[0] = hr.n.x; h.normal[1] = hr.n.y; h.normal[2] = hr.n.z;
            h.uv[0] = hr.u; h.uv[1] = hr.v;
            h.world_point[0] = hr.wp.x; h.world_point[1] = hr.wp.y; h.world_point[2] = hr.wp.z;
            h.outside_face = hr.outside ? 1u : 0u;
        }
        hits[i] = h;
    }
}

template <class C, int TREE>
__global__ void __launch_bounds__(tree_in_lds(TREE) ? LDS_BLOCK_SHADOW : BLOCK) k_trace_any(SceneDev sc, const rt_ray* rays, const float* kMin, const float* kMax, uint64_t n, uint8_t* occluded) {
    constexpr int TB = tree_in_lds(TREE) ? LDS_BLOCK_SHADOW : BLOCK;
    const auto tree = make_tree<TREE, TB, false>(sc);
    for (uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TB) {
        Ray r; r.o = mk3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]); r.d = mk3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]);
        TravCount cnt = trav_count_zero();
        occluded[i] = any_hit<C>(sc, tree, r, kMin[i], kMax[i], __builtin_inff(), cnt) ? 1 : 0;
    }
}


// ---- kernel variants (Cfg in rt_device.h).  The host (rt_lib.hip) picks the first one whose masks cover the scene. ----
constexpr uint32_t pbit(int t) { return 1u << t; }
using CfgSpheres = Cfg<pbit(RT_HIT_SPHERE) | pbit(RT_HIT_INFINITE_PLANE),
                       pbit(RT_MAT_STANDARD) | pbit(RT_MAT_REFRACTIVE) | pbit(RT_MAT_EMISSIVE_REFRACTIVE), pbit(RT_TEX_SOLID),
                       pbit(RT_LIGHT_POINT) | pbit(RT_LIGHT_DIFFUSE_SPHERE)>;                       // Night Spheres, RtInAWeekend, many-lights
using CfgMesh    = Cfg<pbit(RT_HIT_TRIANGLE) | pbit(RT_HIT_INFINITE_PLANE), pbit(RT_MAT_STANDARD), pbit(RT_TEX_SOLID), pbit(RT_LIGHT_POINT)>;
using CfgBoxes   = Cfg<pbit(RT_HIT_SPHERE) | pbit(RT_HIT_BOX) | pbit(RT_HIT_XY_PLANE) | pbit(RT_HIT_XZ_PLANE) | pbit(RT_HIT_YZ_PLANE) | pbit(RT_HIT_QUAD),
                       pbit(RT_MAT_STANDARD) | pbit(RT_MAT_REFRACTIVE) | pbit(RT_MAT_EMISSIVE_REFRACTIVE), pbit(RT_TEX_SOLID),
                       pbit(RT_LIGHT_POINT) | pbit(RT_LIGHT_DIFFUSE_SPHERE)>;                       // Cornell box
template <> struct ShadowPersistentTune<CfgSpheres> { static constexpr int min_waves = 5; };
#ifndef RT_SPW_WAVES
#define RT_SPW_WAVES 1
#endif
template <> struct ShadowPersistentWideTune<CfgSpheres> { static constexpr int min_waves = RT_SPW_WAVES; };
template <> struct ShadowRaysTune<CfgBoxes> { static constexpr int min_waves = 4; };
template <> struct ShadowPersistentTune<CfgBoxes> { static constexpr int min_waves = 4; };       // (128 VGPRs as it was; 129 with the check for lights that cannot matter)
#ifndef RT_MESH_RAYS_WAVES
#define RT_MESH_RAYS_WAVES 5
#endif
template <> struct ShadowRaysTune<CfgMesh> { static constexpr int min_waves = RT_MESH_RAYS_WAVES; };
using CfgAll     = Cfg<(1u << RT_HIT_TYPE_COUNT) - 1, 0x1Fu, 0x7u, 0x7u>;

struct KernelSet {
    const char* name;
    uint32_t prims, mats, texs, lights;
    void (*closest)(SceneDev, RenderDev, uint32_t);
    void (*shade)(SceneDev, RenderDev, uint32_t);
    void (*shadow)(SceneDev, RenderDev, uint32_t);
    void (*trace_closest)(SceneDev, const rt_ray*, uint64_t, float, float, uint64_t, rt_hit*);
    void (*trace_any)(SceneDev, const rt_ray*, const float*, const float*, uint64_t, uint8_t*);
    // LdsTree forms (null when the variant has none)
    void (*closest_persistent)(SceneDev, RenderDev, uint32_t);     // heavy-tailed scenes (see k_closest_persistent)
    void (*shadow_persistent)(SceneDev, RenderDev, uint32_t);
    void (*shadow_exact)(SceneDev, RenderDev, uint32_t);           // light sampling in the reference's correctly rounded arithmetic (RT_RENDER_EXACT_LIGHT_SAMPLING)
    void (*shadow_persistent_exact)(SceneDev, RenderDev, uint32_t);
    void (*shadow_lds_exact)(SceneDev, RenderDev, uint32_t);
    void (*closest_lds)(SceneDev, RenderDev, uint32_t);
    void (*shadow_lds)(SceneDev, RenderDev, uint32_t);
    void (*shadow_rays)(SceneDev, RenderDev, uint32_t);            // one lane per shadow ray (k_shadow_rays)
    void (*shadow_rays_exact)(SceneDev, RenderDev, uint32_t);
    void (*shadow_rays_lds)(SceneDev, RenderDev, uint32_t);
    void (*shadow_rays_lds_exact)(SceneDev, RenderDev, uint32_t);
    void (*trace_closest_lds)(SceneDev, const rt_ray*, uint64_t, float, float, uint64_t, rt_hit*);
    void (*trace_any_lds)(SceneDev, const rt_ray*, const float*, const float*, uint64_t, uint8_t*);
    void (*closest_primary)(SceneDev, RenderDev, uint32_t);        // depth 0 from the path number (k_closest<.., PRIMARY>)
    void (*closest_primary_lds)(SceneDev, RenderDev, uint32_t);
    // the same kernels over the 8-wide tree (rt_wide.h), for trees that are read through the caches
    struct Wide {
        void (*closest)(SceneDev, RenderDev, uint32_t);
        void (*closest_primary)(SceneDev, RenderDev, uint32_t);
        void (*closest_persistent)(SceneDev, RenderDev, uint32_t);
        void (*shadow)(SceneDev, RenderDev, uint32_t);
        void (*shadow_exact)(SceneDev, RenderDev, uint32_t);
        void (*shadow_rays)(SceneDev, RenderDev, uint32_t);
        void (*shadow_rays_exact)(SceneDev, RenderDev, uint32_t);
        void (*shadow_persistent)(SceneDev, RenderDev, uint32_t);
        void (*shadow_persistent_exact)(SceneDev, RenderDev, uint32_t);
        void (*trace_closest)(SceneDev, const rt_ray*, uint64_t, float, float, uint64_t, rt_hit*);
        void (*trace_any)(SceneDev, const rt_ray*, const float*, const float*, uint64_t, uint8_t*);
    } wide;
    // the W-wide tree in LDS (rt_ldsw.h; null when the variant has no LDS forms)
    struct Ldsw {
        void (*closest)(SceneDev, RenderDev, uint32_t);
        void (*closest_primary)(SceneDev, RenderDev, uint32_t);
        void (*shadow)(SceneDev, RenderDev, uint32_t);
        void (*shadow_exact)(SceneDev, RenderDev, uint32_t);
        void (*trace_closest)(SceneDev, const rt_ray*, uint64_t, float, float, uint64_t, rt_hit*);
        void (*trace_any)(SceneDev, const rt_ray*, const float*, const float*, uint64_t, uint8_t*);
    } ldsw;
    // the any-hit kernels over the W-wide float tree read through the caches (QuadTree, rt_ldsw.h; null when the variant knows no spheres)
    struct Quad {
        void (*shadow)(SceneDev, RenderDev, uint32_t);
        void (*shadow_exact)(SceneDev, RenderDev, uint32_t);
        void (*shadow_rays)(SceneDev, RenderDev, uint32_t);
        void (*shadow_rays_exact)(SceneDev, RenderDev, uint32_t);
        void (*shadow_persistent)(SceneDev, RenderDev, uint32_t);
        void (*shadow_persistent_exact)(SceneDev, RenderDev, uint32_t);
        void (*trace_any)(SceneDev, const rt_ray*, const float*, const float*, uint64_t, uint8_t*);
    } quad;
};
template <class C, bool WITH_LDS> constexpr KernelSet make_kernel_set(const char* name) {
    KernelSet k{name, C::prims, C::mats, C::texs, C::lights, k_closest<C, false>, k_shade<C>, k_shadow<C, false>, k_trace_closest<C, false>, k_trace_any<C, false>,
                k_closest_persistent<C, false>, k_shadow_persistent<C, false>, k_shadow<C, false, true>, k_shadow_persistent<C, false, true>, nullptr,
                nullptr, nullptr, k_shadow_rays<C, false>, k_shadow_rays<C, false, true>, nullptr, nullptr, nullptr, nullptr, k_closest<C, false, true>, nullptr,
                {k_closest<C, TREE_WIDE>, k_closest<C, TREE_WIDE, true>, k_closest_persistent<C, TREE_WIDE>, k_shadow<C, TREE_WIDE>, k_shadow<C, TREE_WIDE, true>, k_shadow_rays<C, TREE_WIDE>, k_shadow_rays<C, TREE_WIDE, true>,
                 k_shadow_persistent<C, TREE_WIDE>, k_shadow_persistent<C, TREE_WIDE, true>, k_trace_closest<C, TREE_WIDE>, k_trace_any<C, TREE_WIDE>}};
    if constexpr (WITH_LDS) { k.closest_lds = k_closest<C, true>; k.closest_primary_lds = k_closest<C, true, true>; k.shadow_lds = k_shadow<C, true>; k.shadow_rays_lds = k_shadow_rays<C, true>; k.shadow_rays_lds_exact = k_shadow_rays<C, true, true>; k.shadow_lds_exact = k_shadow<C, true, true>; k.trace_closest_lds = k_trace_closest<C, true>; k.trace_any_lds = k_trace_any<C, true>;
                               k.quad = {k_shadow<C, TREE_QUAD>, k_shadow<C, TREE_QUAD, true>, k_shadow_rays<C, TREE_QUAD>, k_shadow_rays<C, TREE_QUAD, true>, k_shadow_persistent<C, TREE_QUAD>, k_shadow_persistent<C, TREE_QUAD, true>, k_trace_any<C, TREE_QUAD>};
                               k.ldsw = {k_closest<C, TREE_LDSW>, k_closest<C, TREE_LDSW, true>, k_shadow<C, TREE_LDSW>, k_shadow<C, TREE_LDSW, true>, k_trace_closest<C, TREE_LDSW>, k_trace_any<C, TREE_LDSW>}; }
    return k;
}


// One translation unit per variant (rt_variant_*.hip) so that they compile in parallel; rt_lib.hip picks among them.
extern const KernelSet ks_spheres, ks_mesh, ks_boxes, ks_all;
extern void (*const k_shade_vis)(SceneDev, RenderDev, uint32_t);      // k_shade<CfgAll, true>: debug visualisations, any scene
extern void (*const k_shade_validate)(SceneDev, RenderDev, uint32_t);      // k_shade<CfgAll, false, true>: RT_RENDER_VALIDATE, any scene
extern void (*const k_shade_vis_validate)(SceneDev, RenderDev, uint32_t);  // k_shade<CfgAll, true, true>

} // namespace rtd
