// rt_device.h — device-side data layout, float math with the reference's operation order, the counter
// RNG, ray/primitive intersectors, materials, lights and skyboxes for the gfx950 trace kernels.
//
// Arithmetic contract: IEEE binary32, one rounding per C# operator (the library is built with
// -ffp-contract=off and correctly rounded divide/sqrt), operands in the order the reference writes
// them, so K and the hit/miss decisions are reproducible bit for bit against a CPU restatement.
// Reference paths are relative to the reference root (Core/ = Ararem.RayTracer.Core/, Impl/ = Ararem.RayTracer.Impl/).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rt_hip.h"

#define RT_FN __device__ __forceinline__

namespace rtd {

constexpr float PI_F = 3.14159265358979323846f;
constexpr float E_F  = 2.71828182845904523536f;

// ------------------------------------------------------------------------------------------------
// Vector3 (System.Numerics) in registers
// ------------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
RT_FN f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_FN f3 splat(float a) { return mk3(a, a, a); }
RT_FN f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_FN f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_FN f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_FN f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_FN f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_FN f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_FN f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
RT_FN f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_FN float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_FN f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_FN float length_sq(f3 a) { return dot(a, a); }
RT_FN float length(f3 a) { return sqrtf(dot(a, a)); }
RT_FN f3 normalize(f3 a) { return a / length(a); }
RT_FN float distance_sq(f3 a, f3 b) { f3 d = a - b; return dot(d, d); }
RT_FN float distance(f3 a, f3 b) { return sqrtf(distance_sq(a, b)); }
RT_FN f3 lerp3(f3 a, f3 b, float t) { return (a * (1.0f - t)) + (b * t); }
RT_FN f3 reflect(f3 v, f3 n) { float d = dot(v, n); return v - (n * (2.0f * d)); }
RT_FN float lerpf(float a, float b, float t) { return a + (t * (b - a)); }
// MathF.Max / MathF.Min (NaN-propagating)
RT_FN float maxf_net(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : (a > b ? a : b); }
RT_FN float minf_net(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : (a < b ? a : b); }
RT_FN bool is_nan(float a) { return a != a; }
RT_FN bool is_inf(float a) { return __builtin_isinf(a); }
RT_FN f3 point_at(f3 o, f3 d, float k) { return o + (k * d); }                      // Core/Ray.cs:22

// ------------------------------------------------------------------------------------------------
// Counter RNG replacing Random.Shared (Core/RandUtils.cs:17): Philox4x32-10 keyed by
// (seed, pixel, pass, depth, slot, sub).  Results do not depend on traversal order, batch size or the
// device a pass is rendered on.
// ------------------------------------------------------------------------------------------------
enum : uint32_t { SLOT_CAMERA = 0, SLOT_MEDIUM = 1, SLOT_SCATTER = 2, SLOT_LIGHT = 16, SLOT_LIGHT_RAW = 2048 + 16 };

struct RngKey { uint32_t k0, k1, pixel, pass_lo; };

RT_FN RngKey make_key(uint64_t seed, uint32_t pixel, uint64_t pass) {
    RngKey k; k.k0 = (uint32_t)seed; k.k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(pass >> 32); k.pixel = pixel; k.pass_lo = (uint32_t)pass; return k;
}
RT_FN float u01(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }      // Random.NextSingle(): 24 bits in [0,1)
RT_FN float4 rng_draw(const RngKey& key, uint32_t depth, uint32_t slot, uint32_t sub) {
    uint32_t c0 = key.pixel, c1 = key.pass_lo, c2 = (depth << 16) | slot, c3 = sub;
    uint32_t k0 = key.k0, k1 = key.k1;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_float4(u01(c0), u01(c1), u01(c2), u01(c3));
}
RT_FN float plus_minus_one(float u) { return (u * 2) - 1; }                           // RandUtils.cs:29
RT_FN f3 on_unit_sphere(float a, float b, float c) {                                   // RandUtils.cs:33-37 (normalised cube sample, biased as in the reference)
    return normalize(mk3(plus_minus_one(a), plus_minus_one(b), plus_minus_one(c)));
}

// ------------------------------------------------------------------------------------------------
// Scene tables in HBM.  Records are float4-aligned so one or two 16-B loads fetch a primitive.
// ------------------------------------------------------------------------------------------------
// BVH2 node, 64 B: both children's boxes + both child codes -> one 64-B line per visit.
//   q0 = (minA.x minB.x minA.y minB.y)  q1 = (minA.z minB.z maxA.x maxB.x)  q2 = (maxA.y maxB.y maxA.z maxB.z)
//   (A/B interleaved so that one packed-f32 instruction works on both boxes)
//   q3 = (childA, childB, -, -) as int bits: >= 0 inner node index, < 0 leaf holding object ~index, CHILD_EMPTY = none.
// Leaves hold exactly one object and the leaf's box is that object's Hittable.BoundingVolume, so testing it
// is the reference's SingleObjectBvhNode box test (Core/Acceleration/SingleObjectBvhNode.cs:12-16).
struct Node { float4 q0, q1, q2; int4 q3; };
constexpr int CHILD_EMPTY = (int)0x80000000;
constexpr int TRAV_DONE = CHILD_EMPTY;      // traversal cursor: nothing left (never a valid leaf code)

constexpr uint32_t OBJ_TYPE_SHIFT = 28;
constexpr uint32_t OBJ_INDEX_MASK = (1u << OBJ_TYPE_SHIFT) - 1;

struct SceneDev {
    const Node*    nodes;        uint32_t n_nodes;
    uint32_t any_near_first;     // any-hit traversal visits the nearer child first (trees of >= 1024 nodes)
    const float4* obj_shade;     // per object, 64 B: (albedo or tint rgb, f[0]) (emission rgb, type | flags << 8) (sphere centre + radius, or plane normal) (obj_code, material, -, -): scenes of solid-colour textures, few enough objects for the table to stay cached; else null
    const Node*    list;         uint32_t n_list_pairs;  // scenes of a few objects: every bounded object, two per record, instead of the tree (see list_closest)
    const uint32_t* obj_code;    // (type << 28) | index-in-type-table, per object
    uint32_t leaf_code_affine, leaf_code_base;   // every object in the tree: obj_code[obj] == leaf_code_base + obj (see leaf_code)
    uint32_t all_code_affine;    // ... and every other object too: finalize_hit computes the code of whatever was hit
    uint32_t uniform_material;   // the one material every object has (a mesh), or RT_NO_HIT: finalize_hit need not look it up per object, and the material's loads become scalar
    const uint32_t* obj_material;
    uint32_t n_objects;
    const uint32_t* unbounded;   uint32_t n_unbounded;   // objects with a non-finite AABB (InfinitePlane.cs:28), tested for every ray
    const uint32_t* media;       uint32_t n_media;       // ConstantDensityMedium objects, tested last
    const float4* spheres;       // 1 x float4 : centre.xyz, radius
    const float4* quads;         // 4 x float4 : origin.xyz,- | normal.xyz,- | inv_col1 | inv_col2   (triangles share the layout)
    const float4* triangles;
    const float4* inf_planes;    // 2 x float4 : point.xyz, negPointDotNormal | normal.xyz, -
    const float4* axis_planes[3];// 2 x float4 : a_lo a_hi b_lo b_hi | c - - -       [0]=XY [1]=XZ [2]=YZ
    const float4* disks;         // 4 x float4 : centre.xyz, radius | normal.xyz, negCentreDotNorm | inv_col1 | inv_col2
    const float4* boxes;         // 6 x float4 : world_to_box[12] | box_to_world_rows[9], pad
    const float4* capsules;      // 4 x float4 : p1.xyz, radius | p2.xyz, - | uvm[0..3] | uvm[4..7] ; (5th) uvm[8]
    const float4* cylinders;     // 2 x float4 : p1.xyz, radius | p2.xyz, -
    const rt_medium* media_rec;
    const rt_material* materials;
    const rt_texture* textures;
    const rt_light* lights;      uint32_t n_lights;
    float bounds_centre[3], bounds_radius, max_object_radius;    // sphere around every bounded object; largest half-diagonal of one object's box (any_box)
    float min_round_radius;                                      // smallest radius of a sphere / capsule / cylinder in the tree or list, FLT_MAX without any (any_box)
    float any_again;                                             // a second look is due when the sphere term of the pad exceeds this many smallest radii (RT_HIP_ANY_AGAIN)
    float any_pad_scale;                                         // 1; RT_HIP_ANY_PAD_SCALE overrides (A/B knob: 0 = boxes as they are, the culling of round 1)
    rt_camera camera;
    rt_skybox skybox;
    // small all-sphere scenes: a copy of the tree laid out for LDS (see LdsTree), null otherwise
    const uint4* lds_image;      uint32_t lds_image_bytes;   // q0[n_nodes] q1[n_nodes] q2[n_nodes] children[n_nodes] (int2, padded to 16 B) spheres[n_objects]
    // big trees: the 8-wide tree with quantised inner boxes (rt_wide.h), null otherwise
    const uint4* wide_nodes;     uint32_t wide_stride;       // bytes from one node record to the next
    const char* wide_leaves;     uint32_t wide_leaf_stride;  // leaf-ordered records: the object's exact box, its id and code, then (wide_inline) a copy of its primitive
    uint32_t wide_inline;        // the hittable type whose record follows every leaf's box (the tree holds that type only), or RT_NO_HIT
    // small all-sphere scenes, alternative to lds_image: the W-wide tree with full-precision boxes laid out for LDS (rt_ldsw.h), null otherwise
    const uint4* ldsw_image;     uint32_t ldsw_node_bytes;   // nodes, then spheres[n_objects], then boxes[n_objects] (2 x float4 each)
    uint32_t ldsw_any_bytes, ldsw_all_bytes;                 // what an any-hit kernel copies (nodes + spheres) / a closest-hit kernel (+ boxes)
    uint32_t ldsw_axis_a, ldsw_axis_b;                       // W = 4: the two axes the quadrant slots are laid out along
    // all-sphere trees that are read through the caches: the same W-wide form for their any-hit kernels (QuadTree, rt_ldsw.h), null otherwise
    const uint4* quad_image;     uint32_t quad_node_bytes;   // nodes, then spheres[n_objects], then boxes[n_objects]
};

struct Ray { f3 o, d; };

// ------------------------------------------------------------------------------------------------
// AxisAlignedBoundingBox.Hit, Core/Acceleration/AxisAlignedBoundingBox.cs:63-95.
// The reference folds kMin/kMax through the three axes with `?:` selects and fails when kMax <= kMin after
// any axis; because the running kMin only grows and kMax only shrinks this equals the single final test.
// slab() returns the entry bound (kMin folded in) and the exit bound WITHOUT kMax, so the caller can apply
// the current kMax at the time of use:  hit <=> !(sel_min(exit, kMax) <= entry).
// ------------------------------------------------------------------------------------------------
RT_FN void slab(f3 bmin, f3 bmax, f3 o, f3 invD, float kMin, float& entry, float& exit_) {
    float t0 = (bmin.x - o.x) * invD.x, t1 = (bmax.x - o.x) * invD.x;
    if (invD.x < 0.0f) { float s = t0; t0 = t1; t1 = s; }
    float lo = t0 > kMin ? t0 : kMin;
    float hi = t1;
    t0 = (bmin.y - o.y) * invD.y; t1 = (bmax.y - o.y) * invD.y;
    if (invD.y < 0.0f) { float s = t0; t0 = t1; t1 = s; }
    lo = t0 > lo ? t0 : lo;
    hi = is_nan(hi) ? t1 : (t1 < hi ? t1 : hi);
    t0 = (bmin.z - o.z) * invD.z; t1 = (bmax.z - o.z) * invD.z;
    if (invD.z < 0.0f) { float s = t0; t0 = t1; t1 = s; }
    lo = t0 > lo ? t0 : lo;
    hi = is_nan(hi) ? t1 : (t1 < hi ? t1 : hi);
    entry = lo; exit_ = hi;
}
RT_FN bool slab_hit(float entry, float exit_, float kMax) {
    float hi = exit_ < kMax ? exit_ : kMax;     // a NaN exit (every axis degenerate) leaves kMax, as the reference's selects do
    return !(hi <= entry);
}

// Both boxes of one node at once.  NodePlanes = the six planes of child boxes A and B in the order the ray meets them (per axis the
// near plane is the box's min plane when the ray runs in +axis direction, its max plane otherwise: the reference's swap of t0 / t1
// on invD < 0, done on the operands instead of on the products — the same subtract and the same multiply per plane, so the
// distances are bit-identical to AxisAlignedBoundingBox.Hit).  The reference's `?:` folds are exactly maxNum / minNum chains (a NaN
// plane distance never replaces a bound), i.e. v_max3_f32 / v_min3_f32; the A/B pair goes through packed-f32 instructions.
typedef float v2f __attribute__((ext_vector_type(2)));
struct NodePlanes { v2f nx, fx, ny, fy, nz, fz; };
RT_FN void fold_planes(v2f nx, v2f fx, v2f ny, v2f fy, v2f nz, v2f fz, float kMin, float& enA, float& exA, float& enB, float& exB) {
    enA = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx.x, kMin), ny.x), nz.x);
    enB = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(nx.y, kMin), ny.y), nz.y);
    exA = __builtin_fminf(__builtin_fminf(fx.x, fy.x), fz.x);
    exB = __builtin_fminf(__builtin_fminf(fx.y, fy.y), fz.y);
}

// Compile-time description of what a scene contains: the kernels are instantiated for a few common mixes so that the
// intersectors, materials, textures and lights a scene does not use cost no registers and no instruction cache.
template <uint32_t PRIMS, uint32_t MATS, uint32_t TEXS, uint32_t LIGHTS>
struct Cfg { static constexpr uint32_t prims = PRIMS, mats = MATS, texs = TEXS, lights = LIGHTS; };
#define RT_HAS_PRIM(C, t)  ((C::prims >> (t)) & 1u)
#define RT_HAS_MAT(C, t)   ((C::mats >> (t)) & 1u)
#define RT_HAS_TEX(C, t)   ((C::texs >> (t)) & 1u)
#define RT_HAS_LIGHT(C, t) ((C::lights >> (t)) & 1u)

// ------------------------------------------------------------------------------------------------
// Intersectors.  try_* = Hittable.TryHit's decision and K; fast_* = Hittable.FastTryHit.
// ------------------------------------------------------------------------------------------------
// Sphere.cs:35-57 / :99-121
RT_FN bool try_sphere(float4 s, Ray r, float kMin, float kMax, float& k) {
    f3 oc = r.o - mk3(s.x, s.y, s.z);
    float a = length_sq(r.d);
    float halfB = dot(oc, r.d);
    float c = length_sq(oc) - (s.w * s.w);
    float disc = (halfB * halfB) - (a * c);
    if (disc < 0) return false;
    float sqrtD = sqrtf(disc);
    float root = (-halfB - sqrtD) / a;
    if ((root < kMin) || (kMax < root) || is_nan(root)) {
        root = (-halfB + sqrtD) / a;
        if ((root < kMin) || (kMax < root) || is_nan(root)) return false;
    }
    k = root;
    return true;
}

// Quad.cs:82-115 ; Triangle (NEW) replaces the [0,1]^2 test with u>=0, v>=0, u+v<=1
template <bool TRI>
RT_FN bool try_quad(const float4* q, Ray r, float kMin, float kMax, float& k, float& u, float& v, float& nd) {
    float4 q0 = q[0], q1 = q[1];
    f3 origin = mk3(q0.x, q0.y, q0.z), normal = mk3(q1.x, q1.y, q1.z);
    nd = dot(normal, r.d);
    if (fabsf(nd) <= 0.001f) return false;
    float t = -dot(normal, r.o - origin) / nd;
    if ((t < kMin) || (t > kMax)) return false;
    f3 lp = point_at(r.o, r.d, t) - origin;
    float4 c1 = q[2], c2 = q[3];
    u = (lp.x * c1.x) + (lp.y * c1.y) + (lp.z * c1.z) + c1.w;
    v = (lp.x * c2.x) + (lp.y * c2.y) + (lp.z * c2.z) + c2.w;
    if (TRI) { if (!(u >= 0) || !(v >= 0) || !(u + v <= 1)) return false; }
    else     { if (u < 0 || u > 1 || is_nan(u) || v < 0 || v > 1 || is_nan(v)) return false; }
    k = t;
    return true;
}

// InfinitePlane.cs:37-55 (TryHit) — a NaN t passes the range test there and is turned into a miss by the
// validator (Core/RenderJob.cs:609-614); :63-82 (FastTryHit) rejects NaN through its >=,<= form.
RT_FN bool try_inf_plane(const float4* p, Ray r, float kMin, float kMax, float& k, float& nd) {
    float4 p0 = p[0], p1 = p[1];
    f3 normal = mk3(p1.x, p1.y, p1.z);
    nd = dot(r.d, normal);
    if (fabsf(nd) <= 0.001f) return false;
    float t = -(dot(r.o, normal) + p0.w) / nd;
    if ((t < kMin) || (t > kMax)) return false;
    if (is_nan(t)) return false;
    k = t;
    return true;
}

// XYPlane.cs:53-67, XZPlane.cs:53-66, YZPlane.cs:50-63.  axis: 0 = XY, 1 = XZ, 2 = YZ
RT_FN bool axis_plane_core(const float4* p, int axis, Ray r, float kMin, float kMax, float& k, float& a, float& b) {
    float4 p0 = p[0]; float C = p[1].x;
    float oC = axis == 0 ? r.o.z : axis == 1 ? r.o.y : r.o.x;
    float dC = axis == 0 ? r.d.z : axis == 1 ? r.d.y : r.d.x;
    float t = (C - oC) / dC;
    if ((dC == 0.0f) || is_nan(t)) return false;
    if ((t < kMin) || (t > kMax)) return false;
    f3 wp = point_at(r.o, r.d, t);
    a = axis == 2 ? wp.y : wp.x;
    b = axis == 0 ? wp.y : wp.z;
    k = t;
    (void)p0;
    return true;
}
RT_FN bool try_axis_plane(const float4* p, int axis, Ray r, float kMin, float kMax, float& k) {
    float a, b;
    if (!axis_plane_core(p, axis, r, kMin, kMax, k, a, b)) return false;
    float4 p0 = p[0];
    if ((a < p0.x) || (a > p0.y) || (b < p0.z) || (b > p0.w)) return false;
    return true;
}
RT_FN bool fast_axis_plane(const float4* p, int axis, Ray r, float kMin, float kMax) {
    float k, a, b;
    if (!axis_plane_core(p, axis, r, kMin, kMax, k, a, b)) return false;
    float4 p0 = p[0];
    if (axis == 2) return (a >= p0.x) && (a <= p0.y) && (b >= p0.z) && (b <= p0.w);     // YZPlane.cs:89
    return !(a < p0.x) && !(a > p0.y) && !(b < p0.z) && !(b > p0.w);                   // XYPlane.cs:118-121, XZPlane.cs:91
}

// Disk.cs:58-81 / :101-123
RT_FN bool try_disk(const float4* d, Ray r, float kMin, float kMax, float& k, float& nd, bool fast) {
    float4 d0 = d[0], d1 = d[1];
    f3 centre = mk3(d0.x, d0.y, d0.z), normal = mk3(d1.x, d1.y, d1.z);
    nd = dot(r.d, normal);
    if (fabsf(nd) <= 0.001f) return false;
    float t = -(dot(r.o, normal) + d1.w) / nd;
    if ((t < kMin) || (t > kMax)) return false;
    f3 wp = point_at(r.o, r.d, t);
    float ds = distance_sq(centre, wp), r2 = d0.w * d0.w;
    if (fast) { if (!(ds <= r2)) return false; } else { if (ds > r2) return false; }
    k = t;
    return true;
}

// Box.cs:122-145 (TryHit: Transform / TransformNormal) and :191-216 (FastTryHit: Vector4.Transform, w = 0 / 1)
struct BoxSolve { f3 t1, s; };
RT_FN bool box_core(const float4* b, Ray r, float kMin, float kMax, bool fast, float& k, BoxSolve& bs) {
    float4 m0 = b[0], m1 = b[1], m2 = b[2];   // M11 M12 M13 M21 | M22 M23 M31 M32 | M33 M41 M42 M43
    float M11 = m0.x, M12 = m0.y, M13 = m0.z, M21 = m0.w, M22 = m1.x, M23 = m1.y, M31 = m1.z, M32 = m1.w, M33 = m2.x, M41 = m2.y, M42 = m2.z, M43 = m2.w;
    f3 rd, ro;
    if (fast) {
        rd = mk3((r.d.x * M11) + (r.d.y * M21) + (r.d.z * M31) + (0.0f * M41), (r.d.x * M12) + (r.d.y * M22) + (r.d.z * M32) + (0.0f * M42), (r.d.x * M13) + (r.d.y * M23) + (r.d.z * M33) + (0.0f * M43));
        ro = mk3((r.o.x * M11) + (r.o.y * M21) + (r.o.z * M31) + (1.0f * M41), (r.o.x * M12) + (r.o.y * M22) + (r.o.z * M32) + (1.0f * M42), (r.o.x * M13) + (r.o.y * M23) + (r.o.z * M33) + (1.0f * M43));
    } else {
        rd = mk3((r.d.x * M11) + (r.d.y * M21) + (r.d.z * M31), (r.d.x * M12) + (r.d.y * M22) + (r.d.z * M32), (r.d.x * M13) + (r.d.y * M23) + (r.d.z * M33));
        ro = mk3((r.o.x * M11) + (r.o.y * M21) + (r.o.z * M31) + M41, (r.o.x * M12) + (r.o.y * M22) + (r.o.z * M32) + M42, (r.o.x * M13) + (r.o.y * M23) + (r.o.z * M33) + M43);
    }
    f3 m = splat(1.0f) / rd;
    if (is_nan(m.x) || is_nan(m.y) || is_nan(m.z) || is_inf(m.x) || is_inf(m.y) || is_inf(m.z)) return false;
    f3 s = mk3(rd.x < 0.0f ? 1.0f : -1.0f, rd.y < 0.0f ? 1.0f : -1.0f, rd.z < 0.0f ? 1.0f : -1.0f);
    f3 t1 = m * (-ro + (s / 2.0f));
    f3 t2 = m * (-ro - (s / 2.0f));
    float kNear = maxf_net(maxf_net(t1.x, t1.y), t1.z);
    float kFar = minf_net(minf_net(t2.x, t2.y), t2.z);
    if ((kNear > kFar) || (kFar < 0.0f)) return false;
    float kk = kNear;
    if ((kk < kMin) || (kMax < kk)) {
        kk = kFar;
        if ((kk < kMin) || (kMax < kk)) return false;
    }
    if (is_nan(kk)) return false;
    k = kk; bs.t1 = t1; bs.s = s;
    return true;
}

// Capsule.cs:77-112 / :147-182.  ba, baba, r2, r2*baba are the constructor's cached products (:51-54).
RT_FN float capsule_k(f3 P1, f3 P2, float radius, Ray r) {
    f3 ba = P2 - P1;
    float baba = dot(ba, ba), r2 = radius * radius, r2baba = r2 * baba;
    float k = -1.0f;
    f3 oa = r.o - P1;
    float bard = dot(ba, r.d), baoa = dot(ba, oa), rdoa = dot(r.d, oa), oaoa = dot(oa, oa);
    float a = baba - (bard * bard);
    float b = (baba * rdoa) - (baoa * bard);
    float c = (baba * oaoa) - (baoa * baoa) - r2baba;
    float h = (b * b) - (a * c);
    if (h >= 0.0f) {
        float t = (-b - sqrtf(h)) / a;
        float y = baoa + (t * bard);
        if ((y > 0.0f) && (y < baba)) k = t;
        else {
            f3 oc = y <= 0.0f ? oa : r.o - P2;
            b = dot(r.d, oc);
            c = dot(oc, oc) - r2;
            h = (b * b) - c;
            if (h > 0.0f) k = -b - sqrtf(h);
        }
    }
    return k;
}
RT_FN bool try_capsule(const float4* c, Ray r, float kMin, float kMax, float& k, bool fast) {
    float4 c0 = c[0], c1 = c[1];
    float kk = capsule_k(mk3(c0.x, c0.y, c0.z), mk3(c1.x, c1.y, c1.z), c0.w, r);
    if (fast) { if (!((kk != -1.0f) && (kk >= kMin) && (kk <= kMax))) return false; }
    else      { if ((kk == -1.0f) || (kk < kMin) || (kk > kMax)) return false; if (is_nan(kk)) return false; /* validator, RenderJob.cs:609 */ }
    k = kk;
    return true;
}

// Cylinder.cs:55-98 / :109-142
RT_FN bool try_cylinder(const float4* c, Ray r, float kMin, float kMax, float& k, f3& normal, bool fast) {
    float4 c0 = c[0], c1 = c[1];
    f3 P1 = mk3(c0.x, c0.y, c0.z), P2 = mk3(c1.x, c1.y, c1.z);
    float radius = c0.w;
    f3 ba = P2 - P1;
    float baba = dot(ba, ba), r2baba = radius * radius * baba;
    f3 oc = r.o - P1;
    float bard = dot(ba, r.d), baoc = dot(ba, oc);
    float k2 = baba - (bard * bard);
    float k1 = (baba * dot(oc, r.d)) - (baoc * bard);
    float k0 = (baba * dot(oc, oc)) - (baoc * baoc) - r2baba;
    float h = (k1 * k1) - (k2 * k0);
    if (h < 0.0f) return false;
    h = sqrtf(h);
    float kk = (-k1 - h) / k2;
    float y = baoc + (kk * bard);
    if ((y > 0.0f) && (y < baba)) {
        normal = ((oc + (kk * r.d)) - ((ba * y) / baba)) / radius;
    } else {
        kk = ((y < 0.0f ? 0.0f : baba) - baoc) / bard;
        if (fabsf(k1 + (k2 * kk)) < h) {
            float sg = y > 0 ? 1.0f : (y < 0 ? -1.0f : 0.0f);
            normal = (ba * sg) / baba;
        } else return false;
    }
    if (fast) return (kk != -1.0f) && (kk >= kMin) && (kk <= kMax);
    if ((kk == -1.0f) || !(kk >= kMin) || !(kk <= kMax)) return false;
    k = kk;
    return true;
}

// One object, TryHit semantics.
template <class C>
RT_FN bool try_object_solid(const SceneDev& sc, uint32_t type, uint32_t idx, Ray r, float kMin, float kMax, float& k) {
    float u, v, nd; f3 n; BoxSolve bs;
    switch (type) {
        case RT_HIT_SPHERE:         if constexpr (RT_HAS_PRIM(C, RT_HIT_SPHERE)) return try_sphere(sc.spheres[idx], r, kMin, kMax, k); break;
        case RT_HIT_TRIANGLE:       if constexpr (RT_HAS_PRIM(C, RT_HIT_TRIANGLE)) return try_quad<true>(sc.triangles + 4 * (size_t)idx, r, kMin, kMax, k, u, v, nd); break;
        case RT_HIT_QUAD:           if constexpr (RT_HAS_PRIM(C, RT_HIT_QUAD)) return try_quad<false>(sc.quads + 4 * (size_t)idx, r, kMin, kMax, k, u, v, nd); break;
        case RT_HIT_INFINITE_PLANE: if constexpr (RT_HAS_PRIM(C, RT_HIT_INFINITE_PLANE)) return try_inf_plane(sc.inf_planes + 2 * (size_t)idx, r, kMin, kMax, k, nd); break;
        case RT_HIT_XY_PLANE:       if constexpr (RT_HAS_PRIM(C, RT_HIT_XY_PLANE)) return try_axis_plane(sc.axis_planes[0] + 2 * (size_t)idx, 0, r, kMin, kMax, k); break;
        case RT_HIT_XZ_PLANE:       if constexpr (RT_HAS_PRIM(C, RT_HIT_XZ_PLANE)) return try_axis_plane(sc.axis_planes[1] + 2 * (size_t)idx, 1, r, kMin, kMax, k); break;
        case RT_HIT_YZ_PLANE:       if constexpr (RT_HAS_PRIM(C, RT_HIT_YZ_PLANE)) return try_axis_plane(sc.axis_planes[2] + 2 * (size_t)idx, 2, r, kMin, kMax, k); break;
        case RT_HIT_DISK:           if constexpr (RT_HAS_PRIM(C, RT_HIT_DISK)) return try_disk(sc.disks + 4 * (size_t)idx, r, kMin, kMax, k, nd, false); break;
        case RT_HIT_BOX:            if constexpr (RT_HAS_PRIM(C, RT_HIT_BOX)) return box_core(sc.boxes + 6 * (size_t)idx, r, kMin, kMax, false, k, bs); break;
        case RT_HIT_CAPSULE:        if constexpr (RT_HAS_PRIM(C, RT_HIT_CAPSULE)) return try_capsule(sc.capsules + 5 * (size_t)idx, r, kMin, kMax, k, false); break;
        case RT_HIT_CYLINDER:       if constexpr (RT_HAS_PRIM(C, RT_HIT_CYLINDER)) return try_cylinder(sc.cylinders + 2 * (size_t)idx, r, kMin, kMax, k, n, false); break;
        default: break;
    }
    return false;
}
// FastTryHit semantics
template <class C>
RT_FN bool fast_object_solid(const SceneDev& sc, uint32_t type, uint32_t idx, Ray r, float kMin, float kMax) {
    float k, u, v, nd; f3 n; BoxSolve bs;
    switch (type) {
        case RT_HIT_SPHERE:         if constexpr (RT_HAS_PRIM(C, RT_HIT_SPHERE)) return try_sphere(sc.spheres[idx], r, kMin, kMax, k); break;
        case RT_HIT_TRIANGLE:       if constexpr (RT_HAS_PRIM(C, RT_HIT_TRIANGLE)) return try_quad<true>(sc.triangles + 4 * (size_t)idx, r, kMin, kMax, k, u, v, nd); break;
        case RT_HIT_QUAD:           if constexpr (RT_HAS_PRIM(C, RT_HIT_QUAD)) return try_quad<false>(sc.quads + 4 * (size_t)idx, r, kMin, kMax, k, u, v, nd); break;
        case RT_HIT_INFINITE_PLANE: if constexpr (RT_HAS_PRIM(C, RT_HIT_INFINITE_PLANE)) return try_inf_plane(sc.inf_planes + 2 * (size_t)idx, r, kMin, kMax, k, nd); break;
        case RT_HIT_XY_PLANE:       if constexpr (RT_HAS_PRIM(C, RT_HIT_XY_PLANE)) return fast_axis_plane(sc.axis_planes[0] + 2 * (size_t)idx, 0, r, kMin, kMax); break;
        case RT_HIT_XZ_PLANE:       if constexpr (RT_HAS_PRIM(C, RT_HIT_XZ_PLANE)) return fast_axis_plane(sc.axis_planes[1] + 2 * (size_t)idx, 1, r, kMin, kMax); break;
        case RT_HIT_YZ_PLANE:       if constexpr (RT_HAS_PRIM(C, RT_HIT_YZ_PLANE)) return fast_axis_plane(sc.axis_planes[2] + 2 * (size_t)idx, 2, r, kMin, kMax); break;
        case RT_HIT_DISK:           if constexpr (RT_HAS_PRIM(C, RT_HIT_DISK)) return try_disk(sc.disks + 4 * (size_t)idx, r, kMin, kMax, k, nd, true); break;
        case RT_HIT_BOX:            if constexpr (RT_HAS_PRIM(C, RT_HIT_BOX)) return box_core(sc.boxes + 6 * (size_t)idx, r, kMin, kMax, true, k, bs); break;
        case RT_HIT_CAPSULE:        if constexpr (RT_HAS_PRIM(C, RT_HIT_CAPSULE)) return try_capsule(sc.capsules + 5 * (size_t)idx, r, kMin, kMax, k, true); break;
        case RT_HIT_CYLINDER:       if constexpr (RT_HAS_PRIM(C, RT_HIT_CYLINDER)) return try_cylinder(sc.cylinders + 2 * (size_t)idx, r, kMin, kMax, k, n, true); break;
        default: break;
    }
    return false;
}

// ConstantDensityMedium.TryHit, ConstantDensityMedium.cs:46-68.  K = entry.K + dist, ShaderData = dist.
template <class C>
RT_FN bool try_medium(const SceneDev& sc, uint32_t midx, uint32_t self_obj, Ray r, float kMin, float kMax, const RngKey& key, uint32_t depth, float& k, float& aux) {
    rt_medium m = sc.media_rec[midx];
    float kEntry, kExit;
    if (!try_object_solid<C>(sc, m.boundary_type, m.boundary_index, r, kMin, kMax, kEntry)) return false;
    if (!try_object_solid<C>(sc, m.boundary_type, m.boundary_index, r, kEntry + 0.001f, kMax, kExit)) return false;
    float between = kExit - kEntry;
    float4 u = rng_draw(key, depth, SLOT_MEDIUM, self_obj);
    float dist = m.neg_inv_density * logf(u.x);
    if (dist > between) return false;
    k = kEntry + dist; aux = dist;
    return true;
}

// ------------------------------------------------------------------------------------------------
// Closest hit / any hit over the whole scene.  `stack` is this lane's LDS column: entry i lives at
// stack[i * stride] so that the 64 lanes of a wave hit 64 different banks.
// Result rule: smallest K wins; on an exact tie the larger object index wins (the reference returns
// whichever its implementation-defined tree visits last, Core/Acceleration/BinaryBvhNode.cs:28-30).
// ------------------------------------------------------------------------------------------------
struct Closest { float k; uint32_t obj; float aux; };
// nodes / prims: box tests and primitive tests (V and T of SURVEY 8d).  box_miss / prim_hit: closest-hit traversal only, the
// reference's AabbMisses (BinaryBvhNode.cs:18, SingleObjectBvhNode.cs:15) and HittableIntersections (SingleObjectBvhNode.cs:23);
// HittableMisses (:28) = prims - prim_hit.  (FastTryHit keeps no counters in the reference.)
struct TravCount { uint32_t nodes, prims, box_miss, prim_hit; };
RT_FN TravCount trav_count_zero() { TravCount c; c.nodes = 0; c.prims = 0; c.box_miss = 0; c.prim_hit = 0; return c; }

// Any-hit traversal culls by boxes, the reference's AnyIntersectionFast does not (BinaryBvhNode.cs:41 asks every object).  The
// answers agree as long as no intersector accepts a point outside its object's box — true in exact arithmetic, not in binary32: a
// sphere test at distance D from the ray origin carries an error of up to 18 * 2^-24 * D^2 = 1.07e-6 D^2 in its discriminant
// (cancellation in halfB^2 - a c), so a ray that passes a sphere of radius r at up to sqrt(r^2 + 1.07e-6 D^2) can be reported as a
// hit (observed: shadow rays from ground points 2 000 units away in the Demo scene, 1 unit outside the box).  So the any-hit box
// test sees every box larger by
//   pad = sqrt(rmin^2 + 1.1e-6 reach^2) - rmin   the sphere bound at the ray's reach (rmin = the smallest sphere / capsule /
//         cylinder radius of the scene; a larger radius errs less); nothing in scenes of planar primitives only, whose error is
//         linear and whose reference boxes are padded by 1e-4 already (Quad.cs:43, XYPlane.cs:27)
//       + 1e-6 reach + 1e-5 + 2.4e-7 |origin|     the rounding of the reported root and of the FMA plane form
//   reach = the ray's range, capped by the far side of the scene, plus the largest object's half-diagonal.
// The ray's range needs no widening: the point o + t d of a reported root t lies within `pad` of its object, hence inside the
// padded box at that very t, so the box's interval and [kMin, kMax] share t.
// Night Spheres: pad = 1.5e-3 on spheres of 0.1 and up (nothing).  100 k spheres of 0.05 .. 0.25, lights up to 100 units away:
// 0.07 at full reach — see the two looks of any_box.
//
// A ray prepared for node box tests.  ClosestBox is the reference's arithmetic (AxisAlignedBoundingBox.Hit, operand for operand: a
// leaf's box test decides whether its object is asked at all, so its bits matter).  AnyBox is the any-hit form: padded boxes, and —
// no reference arithmetic to reproduce here, BinaryBvhNode.FastTryHit has no box test — one fused multiply-add per plane,
// t = plane * inv + addend: 6 packed FMAs per node instead of 6 packed subtracts and 6 packed multiplies.  The FMA form places a
// plane with an absolute error of ~2^-23 |origin| (the rounding of the precomputed addend), which pad0 absorbs; reciprocals are
// v_rcp_f32 (1 ulp is plenty here) clamped to +-1e12 / |d| (make_any_box), so an axis-parallel ray gives +-huge, not inf - inf = NaN.
// o8x/y/z: 8 where the ray runs against the axis, else 0 — the byte offset that makes LdsTree's node layout hand out near/far planes.
struct ClosestBox {
    f3 o, inv; uint32_t o8x, o8y, o8z;
    RT_FN void fold(const NodePlanes& p, float kMin, float& enA, float& exA, float& enB, float& exB) const {
        const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z}, ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
        fold_planes((p.nx - ox) * ix, (p.fx - ox) * ix, (p.ny - oy) * iy, (p.fy - oy) * iy, (p.nz - oz) * iz, (p.fz - oz) * iz, kMin, enA, exA, enB, exB);
    }
};
RT_FN ClosestBox closest_box(const Ray& r) {
    ClosestBox b; b.o = r.o; b.inv = splat(1.0f) / r.d;
    b.o8x = b.inv.x < 0.0f ? 8u : 0u; b.o8y = b.inv.y < 0.0f ? 8u : 0u; b.o8z = b.inv.z < 0.0f ? 8u : 0u;
    return b;
}
struct AnyBox {
    f3 inv, an, af;                   // per axis: t_near = near plane * inv + an, t_far = far plane * inv + af
    uint32_t o8x, o8y, o8z;
    RT_FN void fold(const NodePlanes& p, float kMin, float& enA, float& exA, float& enB, float& exB) const {
        const v2f ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
        const v2f anx = {an.x, an.x}, any = {an.y, an.y}, anz = {an.z, an.z}, afx = {af.x, af.x}, afy = {af.y, af.y}, afz = {af.z, af.z};
        fold_planes(__builtin_elementwise_fma(p.nx, ix, anx), __builtin_elementwise_fma(p.fx, ix, afx), __builtin_elementwise_fma(p.ny, iy, any),
                    __builtin_elementwise_fma(p.fy, iy, afy), __builtin_elementwise_fma(p.nz, iz, anz), __builtin_elementwise_fma(p.fz, iz, afz), kMin, enA, exA, enB, exB);
    }
};
RT_FN AnyBox any_box(const SceneDev& sc, const Ray& r, float kMax, float cap, int mode, bool& again);

RT_FN void consider(Closest& best, float k, uint32_t obj, float aux) {
    if (best.obj != RT_NO_HIT && k == best.k && best.obj > obj) return;
    best.k = k; best.obj = obj; best.aux = aux;
}

RT_FN bool ray_has_nan(Ray r) { return is_nan(r.o.x) || is_nan(r.o.y) || is_nan(r.o.z) || is_nan(r.d.x) || is_nan(r.d.y) || is_nan(r.d.z); }

// Where a traversal reads the tree from and where it keeps its stack.
//   HbmTree: nodes and primitives come through the vector cache (L1 -> L2 -> Infinity Cache -> HBM); 32-bit stack column in LDS.
//   LdsTree: the whole tree and the sphere table live in the workgroup's LDS (scenes of a few hundred objects): a divergent
//            64-lane node fetch costs LDS cycles instead of 64 cache-line lookups in the texture path; 16-bit stack column.
#ifndef RT_TRAV_VOTE
#define RT_TRAV_VOTE 1
#endif
#ifndef RT_VOTE_A
#define RT_VOTE_A 1
#define RT_VOTE_B 3
#endif
#ifndef RT_LDS_VOTE
#define RT_LDS_VOTE 0
#endif
// the 64-B node record (q0 q1 q2 of struct Node) -> near / far planes for this ray
template <class Box>
RT_FN void planes_of(float4 q0, float4 q1, float4 q2, const Box& box, NodePlanes& p) {
    const bool gx = box.o8x != 0u, gy = box.o8y != 0u, gz = box.o8z != 0u;
    const v2f mnx = {q0.x, q0.y}, mny = {q0.z, q0.w}, mnz = {q1.x, q1.y}, mxx = {q1.z, q1.w}, mxy = {q2.x, q2.y}, mxz = {q2.z, q2.w};
    p.nx = gx ? mxx : mnx; p.fx = gx ? mnx : mxx;
    p.ny = gy ? mxy : mny; p.fy = gy ? mny : mxy;
    p.nz = gz ? mxz : mnz; p.fz = gz ? mnz : mxz;
}
struct HbmTree {
    static constexpr bool wide = false;
    static constexpr bool vote = RT_TRAV_VOTE;
    static constexpr bool any_order = true;
    static constexpr bool list = true;       // scenes of a handful of objects are walked as a list (list_closest)
    static constexpr bool may_have_empty_child = true;       // the root of a one-object tree
    int* stack; int stride;
    template <class Box> RT_FN void planes(const SceneDev& sc, int i, const Box& box, NodePlanes& p, int2& ch) const {
        const Node* np = sc.nodes + i;
        const float4 q0 = np->q0, q1 = np->q1, q2 = np->q2; ch = *(const int2*)&np->q3;
        __builtin_amdgcn_sched_barrier(0);       // all four loads in flight before the first wait (left alone the scheduler issues the children's load after waiting for the planes: two memory round trips per node, 100 k spheres k_shadow 656 vs 518 ms)
        planes_of(q0, q1, q2, box, p);
    }
    RT_FN void push(int sp, int v) const { stack[sp * stride] = v; }
    RT_FN int pop(int sp) const { return stack[sp * stride]; }
    template <class C> RT_FN bool leaf_try(const SceneDev& sc, uint32_t obj, Ray r, float kMin, float kMax, float& k) const;
    template <class C> RT_FN bool leaf_fast(const SceneDev& sc, uint32_t obj, Ray r, float kMin, float kMax) const;
};
// (type, index) of a leaf's object.  When every object in the tree has the same type and its table index is its object id minus a
// constant (a mesh, a field of spheres: the way bulk scenes are flattened) the code is computed instead of loaded: one dependent,
// uncached 4-B fetch less between the node and the primitive record.
RT_FN uint32_t leaf_code(const SceneDev& sc, uint32_t obj) { return sc.leaf_code_affine ? sc.leaf_code_base + obj : sc.obj_code[obj]; }
template <class C> RT_FN bool HbmTree::leaf_try(const SceneDev& sc, uint32_t obj, Ray r, float kMin, float kMax, float& k) const {
    uint32_t code = leaf_code(sc, obj);
    return try_object_solid<C>(sc, code >> OBJ_TYPE_SHIFT, code & OBJ_INDEX_MASK, r, kMin, kMax, k);
}
template <class C> RT_FN bool HbmTree::leaf_fast(const SceneDev& sc, uint32_t obj, Ray r, float kMin, float kMax) const {
    uint32_t code = leaf_code(sc, obj);
    return fast_object_solid<C>(sc, code >> OBJ_TYPE_SHIFT, code & OBJ_INDEX_MASK, r, kMin, kMax);
}
// LDS image (built by rt_scene_create): n_nodes records of LDS_NODE_BYTES, then the sphere table (float4 per object id).
//   node record: per axis x, y, z 24 B = (min A, min B, max A, max B, min A, min B), then (child A, child B).
// The repeated min pair makes the ray's sign a byte offset instead of twelve selects: at +0 the lane reads (min pair, max pair) =
// (near, far) for a ray along +axis, at +8 it reads (max pair, min pair) = (near, far) for a ray along -axis — one ds_read2_b64 per
// axis.  An inner child code is index * 5, the record's offset in 16-B units.  The stack column holds 16-bit entries: an inner code (< 32768) as it is, a leaf ~object as the negative number it is
// (ds_read_i16 sign-extends it back), so push and pop are one LDS instruction each, no encoding.
constexpr uint32_t LDS_NODE_BYTES = 80;
struct LdsTree {
    static constexpr bool wide = false;
    static constexpr bool any_order = false;
    static constexpr bool list = false;      // (the host never pairs the LDS tree with an object list: keeps these kernels' registers)
    static constexpr bool may_have_empty_child = false;      // rt_scene_create puts a tree in LDS only if it has two objects or more: every node is full
    static constexpr bool vote = RT_LDS_VOTE;      // rays of equal cost: the votes only cost instructions here
    const char* nodes; const float4* sph;
    short* stack; int stride;
    template <class Box> RT_FN void planes(const SceneDev&, int i, const Box& box, NodePlanes& p, int2& ch) const {
        const char* n = nodes + ((uint32_t)i << 4);              // i = the record's offset in 16-B units (rt_scene_create writes inner child codes that way; the root is 0)
        const char* ax = n + box.o8x; const char* ay = n + 24 + box.o8y; const char* az = n + 48 + box.o8z;
        p.nx = *(const v2f*)ax; p.fx = *(const v2f*)(ax + 8);
        p.ny = *(const v2f*)ay; p.fy = *(const v2f*)(ay + 8);
        p.nz = *(const v2f*)az; p.fz = *(const v2f*)(az + 8);
        ch = *(const int2*)(n + 72);
    }
    RT_FN void push(int sp, int v) const { stack[sp * stride] = (short)v; }
    RT_FN int pop(int sp) const { return (int)stack[sp * stride]; }
    template <class C> RT_FN bool leaf_try(const SceneDev&, uint32_t obj, Ray r, float kMin, float kMax, float& k) const { return try_sphere(sph[obj], r, kMin, kMax, k); }
    template <class C> RT_FN bool leaf_fast(const SceneDev&, uint32_t obj, Ray r, float kMin, float kMax) const { float k; return try_sphere(sph[obj], r, kMin, kMax, k); }
    // carve the LDS image (copied verbatim from SceneDev::lds_image) into its tables
    RT_FN static LdsTree carve(const void* base, uint32_t n_nodes, uint32_t n_objects, uint32_t image_bytes, int tid, int block) {
        LdsTree t;
        t.nodes = (const char*)base;
        t.sph = (const float4*)((const char*)base + (size_t)n_nodes * LDS_NODE_BYTES);
        t.stack = (short*)((const char*)base + image_bytes) + tid; t.stride = block;
        (void)n_objects;
        return t;
    }
};

// Traversal: `cur` is an inner node (>= 0: test both child boxes, go to the nearer hit child, push the other) or a leaf
// (< 0: run the object's intersector).  While-while with one postponed leaf: a lane that reaches a leaf parks it and keeps
// descending until it meets a second one, so the 64 lanes of a wave do box tests together and then intersect together
// instead of paying for both kinds of work on every iteration (closest hit does not depend on the order of the tests).  A leaf's box is tested when its parent is
// visited, with the best K known then — the reference's SingleObjectBvhNode test (SingleObjectBvhNode.cs:12-16); a K found later
// only shrinks the range the intersector accepts, so the result is the same.
// One inner-node step: test both child boxes against [kMin, kMax] (kMax = the best K so far for closest hit), continue with
// the nearer hit child, push the other, pop when neither is hit; a leaf met while no leaf is parked is parked in `pend` and
// the descent goes on.  ANY = any-hit rule: closed box interval, no near/far ordering.
// The traversal stack: its newest entry lives in a register, the rest in the lane's LDS column.  A pop hands out the register at
// once and refills it from LDS for the NEXT pop, so the LDS round trip is off the node-to-node critical path (RT_STACK_TOP=0:
// every entry in LDS, the layout of round 1).  Slot sp of the column holds the entry below the register; slot 0 is never read back.
#ifndef RT_STACK_TOP
#define RT_STACK_TOP 1
#endif
struct Stk { int sp, top; };
RT_FN Stk stk_empty() { Stk s; s.sp = 0; s.top = TRAV_DONE; return s; }
template <class Tree> RT_FN void stk_push(const Tree& tree, Stk& s, int v) {
    if (RT_STACK_TOP) { tree.push(s.sp, s.top); s.top = v; } else tree.push(s.sp, v);
    s.sp++;
}
template <class Tree> RT_FN int stk_pop(const Tree& tree, Stk& s) {      // TRAV_DONE when empty
    if (!s.sp) return TRAV_DONE;
    s.sp--;
    if (RT_STACK_TOP) { const int v = s.top; s.top = tree.pop(s.sp); return v; }
    return tree.pop(s.sp);
}
template <bool ANY, class Tree, class Box>
RT_FN void node_step(const SceneDev& sc, const Tree& tree, const Box& box, float kMin, float kMax, int& cur, int& pend, Stk& sp, uint32_t& box_miss) {
    NodePlanes pl; int2 ch;
    tree.planes(sc, cur, box, pl, ch);
    float enA, exA, enB, exB;
    box.fold(pl, kMin, enA, exA, enB, exB);
    bool hA, hB;
    if (ANY) {     // a hit exactly on a (grown) box face must never be culled: exit >= entry
        const float hiA = __builtin_fminf(exA, kMax), hiB = __builtin_fminf(exB, kMax);
        hA = (!Tree::may_have_empty_child || ch.x != CHILD_EMPTY) && !(hiA < enA);
        hB = (!Tree::may_have_empty_child || ch.y != CHILD_EMPTY) && !(hiB < enB);
    } else {
        hA = (!Tree::may_have_empty_child || ch.x != CHILD_EMPTY) && slab_hit(enA, exA, kMax);
        hB = (!Tree::may_have_empty_child || ch.y != CHILD_EMPTY) && slab_hit(enB, exB, kMax);
        box_miss += (hA ? 0u : 1u) + (hB ? 0u : 1u);
    }
    int next = ch.x; bool pop = false;
    if (hA && hB) {
        // nearer entry first (affects only the amount of work).  For any-hit rays it pays in big scenes, where an occluder close to the
        // origin ends the ray at once (100 k spheres + 64 lights: 66 -> 53 box tests per shadow ray, k_shadow 659 -> 521 ms), and costs
        // its two selects in small ones (Cornell +5 %), so the host sets the flag by tree size.
        const bool bFirst = (!ANY || (Tree::any_order && sc.any_near_first)) && (enB < enA);
        stk_push(tree, sp, bFirst ? ch.x : ch.y);
        next = bFirst ? ch.y : ch.x;
    } else if (hB) next = ch.y;
    else if (!hA) pop = true;
    if (!pop && next < 0 && pend == TRAV_DONE) { pend = next; pop = true; }   // park one leaf, keep descending
    if (pop) {
        next = stk_pop(tree, sp);
        if (next < 0 && next != TRAV_DONE && pend == TRAV_DONE) {          // the popped entry is a leaf and the parking slot is free
            pend = next;
            next = stk_pop(tree, sp);
        }
    }
    cur = next;                                                            // invariant on return: a leaf in `cur` implies a leaf in `pend`
}
// After the parked leaf has been tested: a second leaf waiting in `cur` takes its place and the descent resumes from the stack.
template <class Tree>
RT_FN void leaf_advance(const Tree& tree, int& cur, int& pend, Stk& sp) {
    pend = TRAV_DONE;
    if (cur < 0 && cur != TRAV_DONE) {
        pend = cur;
        cur = stk_pop(tree, sp);
    }
}

// The wave's vote on what to do next (HbmTree scenes).  The plain rule "box tests until every lane holds a leaf" lets one lane
// that descends 30 levels without meeting a leaf hold 63 lanes that each sit on two: on the 5 M-triangle mesh 9 of 64 lanes
// were busy on average.  So the lanes that can step are counted against the lanes that cannot move until a leaf of theirs is
// tested, and the majority decides; the leaf round that follows serves every lane with a parked leaf, the steppers too.
// The order of a ray's own leaf tests is unchanged, and testing one earlier only shrinks kMax earlier: same result.
RT_FN bool step_vote(int cur, int pend) {
    const int n_step = __popcll(__ballot(cur >= 0));
    const int n_blocked = __popcll(__ballot(cur < 0 && pend != TRAV_DONE));
    return n_step != 0 && n_step * RT_VOTE_A >= n_blocked * RT_VOTE_B;
}

// Scenes of a handful of objects (the Cornell box: 10, the Demo scene: 16).  A tree saves nothing there and costs the wave its
// coherence: 64 lanes at 64 different leaves run 5 different intersectors one after the other (Cornell: 26 of 64 lanes busy).  The
// list walks EVERY bounded object in scene order, the same object for all lanes at the same time: the object's type is a scalar, so
// each intersector runs once per object with whatever lanes passed its box test, and there is no stack.  Two objects per 64-B record
// (the node layout, both children leaves) so the packed two-box test is reused.  This is also literally what the reference's
// AnyIntersectionFast does (every object, RenderJob.cs:575).  Cornell box: k_closest 44.5 -> 22.4, k_shadow 59.1 -> 39.0 ms per
// batch (513 -> 732 Msamples/s).  RT_HIP_LIST=<max objects> (default 16, 0 = never) is the A/B knob.
template <class C>
RT_FN void list_closest(const SceneDev& sc, const Ray& r, float kMin, Closest& best, TravCount& cnt) {
    const ClosestBox box = closest_box(r);
    for (uint32_t p = 0; p < sc.n_list_pairs; p++) {
        const Node* np = sc.list + p;
        float enA, exA, enB, exB; NodePlanes pl;
        planes_of(np->q0, np->q1, np->q2, box, pl);
        box.fold(pl, kMin, enA, exA, enB, exB);
        const int cA = np->q3.x, cB = np->q3.y;
        cnt.nodes += 2;
        if (slab_hit(enA, exA, best.k)) {
            const uint32_t obj = (uint32_t)~cA, code = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.obj_code[obj]);
            float k; cnt.prims++;
            if (try_object_solid<C>(sc, code >> OBJ_TYPE_SHIFT, code & OBJ_INDEX_MASK, r, kMin, best.k, k)) { consider(best, k, obj, 0.0f); cnt.prim_hit++; }
        } else cnt.box_miss++;
        if (cB != CHILD_EMPTY && slab_hit(enB, exB, best.k)) {
            const uint32_t obj = (uint32_t)~cB, code = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.obj_code[obj]);
            float k; cnt.prims++;
            if (try_object_solid<C>(sc, code >> OBJ_TYPE_SHIFT, code & OBJ_INDEX_MASK, r, kMin, best.k, k)) { consider(best, k, obj, 0.0f); cnt.prim_hit++; }
        } else cnt.box_miss++;
    }
}
template <class C>
RT_FN bool list_any(const SceneDev& sc, const Ray& r, const AnyBox& box, float kMin, float kMax, TravCount& cnt) {
    bool occluded = false;
    for (uint32_t p = 0; p < sc.n_list_pairs; p++) {
        if (!__any(!occluded)) break;
        const Node* np = sc.list + p;
        float enA, exA, enB, exB; NodePlanes pl;
        planes_of(np->q0, np->q1, np->q2, box, pl);
        box.fold(pl, kMin, enA, exA, enB, exB);
        const int cA = np->q3.x, cB = np->q3.y;
        const float hiA = __builtin_fminf(exA, kMax), hiB = __builtin_fminf(exB, kMax);      // closed interval: a hit on a box face is never culled
        if (!occluded) cnt.nodes += 2;
        if (!occluded && !(hiA < enA)) {
            const uint32_t code = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.obj_code[(uint32_t)~cA]);
            cnt.prims++;
            occluded = fast_object_solid<C>(sc, code >> OBJ_TYPE_SHIFT, code & OBJ_INDEX_MASK, r, kMin, kMax);
        }
        if (!occluded && cB != CHILD_EMPTY && !(hiB < enB)) {
            const uint32_t code = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.obj_code[(uint32_t)~cB]);
            cnt.prims++;
            occluded = fast_object_solid<C>(sc, code >> OBJ_TYPE_SHIFT, code & OBJ_INDEX_MASK, r, kMin, kMax);
        }
    }
    return occluded;
}

RT_FN float rcp_fast(float a);
// The box test of a ray that sees every box larger by `pad` on every side.  lim = ANY_INV_LIMIT / |d| bounds 1/d: a component
// beyond it belongs to a ray that is parallel to that axis for every purpose (it moves less than 1e-12 of its length along it), and
// its slab distances become +-(distance to the plane) * lim — on the far side at least pad * lim, beyond any hit — instead of
// overflowing plane * inv in a scene 1e17 units across (where +-1e30, the limit of rounds 1 and 2, gave inf - inf: found by
// tests/test_gpu_wide.py on the builders' geometric-spread case, two axis-parallel rays of 30 000 answered "not occluded").
constexpr float ANY_INV_LIMIT = 1e12f;
RT_FN AnyBox make_any_box(const Ray& r, float pad, float lim) {
    AnyBox b;
    const float dd[3] = {r.d.x, r.d.y, r.d.z}, oo[3] = {r.o.x, r.o.y, r.o.z};
    float in_[3], an_[3], af_[3]; uint32_t o8[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const bool neg = __builtin_signbitf(dd[i]);
        const float sp = neg ? -pad : pad;
        in_[i] = __builtin_fminf(__builtin_fmaxf(rcp_fast(dd[i]), -lim), lim);
        an_[i] = -(oo[i] + sp) * in_[i];                                                        // near plane: the min plane moved by -pad for a ray along +axis, the max plane moved by +pad otherwise
        af_[i] = -(oo[i] - sp) * in_[i];
        o8[i] = neg ? 8u : 0u;
    }
    b.inv = mk3(in_[0], in_[1], in_[2]); b.an = mk3(an_[0], an_[1], an_[2]); b.af = mk3(af_[0], af_[1], af_[2]);
    b.o8x = o8[0]; b.o8y = o8[1]; b.o8z = o8[2];
    return b;
}
// Two looks.  Padded boxes only ever ADD candidates, and a hit found in the boxes as they are is a hit of the reference's scan too
// (the same FastTryHit said yes).  So when the pad is large against the objects (a ray reaching 100 units through 0.05-unit
// spheres; a shadow ray from a ground point 10 000 units out, whose sphere tests are noise the size of the whole scene) the ray is
// first traced through the boxes as they are — in a dense scene it is occluded within a few units and done — and only a ray that
// comes out unoccluded is traced again through the padded boxes, where the object that answers yes from outside its box is found.
// When the pad is small (<= 0.1 rmin) the padded boxes are the only look.
//   cap: the sphere term of the pad is limited to cap * rmin.  Infinite under RT_RENDER_EXACT_OCCLUSION and in rt_trace_any: every
//        answer is the reference scan's, at the price of horizon rays that ask every object of the scene (as the reference does
//        for every ray).  0.5 otherwise: exact up to reach = 1 100 rmin by the worst-case bound, ~2 500 rmin by the error observed;
//        beyond that a noise hit of the reference outside 1.5 radii is missed (pixels at the horizon of an infinite ground plane —
//        the 100 k-sphere scene renders 20x faster for it).
//   mode: 0 = first look (sets `again` when a second one is due), 1 = the second look.
RT_FN AnyBox any_box(const SceneDev& sc, const Ray& r, float kMax, float cap, int mode, bool& again) {
    const f3 c = mk3(sc.bounds_centre[0], sc.bounds_centre[1], sc.bounds_centre[2]);
    const float len = __builtin_amdgcn_sqrtf(dot(r.d, r.d));                                     // 1 for every ray the renderer makes; callers of rt_trace_any may pass any length
    const float far_side = __builtin_amdgcn_sqrtf(distance_sq(r.o, c)) + sc.bounds_radius;      // no bounded object is farther from the origin
    const float range = kMax * len;
    const float reach = (range < far_side ? range : far_side) + sc.max_object_radius;            // world units (a NaN kMax leaves far_side): no centre of a reported hit is farther
    float round = 0.0f, rm = 3e38f;
    if (sc.min_round_radius < 3e38f) {
        rm = sc.min_round_radius;
        round = __builtin_fminf((__builtin_amdgcn_sqrtf((rm * rm) + (1.1e-6f * reach * reach)) - rm) * 1.02f, cap * rm);
    }
    again = mode == 0 && round > (sc.any_again * rm);
    const float lim = ANY_INV_LIMIT * rcp_fast(len);
    if (again) return make_any_box(r, 0.0f, lim);
    // + the rounding of the reported root (1e-6 reach), the reference's own margins, and the plane error of the FMA form
    return make_any_box(r, sc.any_pad_scale * (round + (1e-6f * reach) + 1e-5f + (2.4e-7f * __builtin_fmaxf(__builtin_fmaxf(fabsf(r.o.x), fabsf(r.o.y)), fabsf(r.o.z)))), lim);
}

// (rt_wide.h: the 8-wide tree with quantised inner boxes)
struct WideTree;
struct WideRay {              // what a node visit of the wide tree needs of the ray
    f3 o, inv, pt;            // origin, 1/d clamped, pad * |1/d| per axis
    uint32_t ax, ay, az;      // byte offset in a node record of the axis's (near planes, far planes) pair: 8 further along when the ray runs against the axis
    uint32_t octinv;          // 7 - octant (0 for a ray that takes a node's children in slot order)
};
template <class C> RT_FN void closest_wide(const SceneDev& sc, const WideTree& tree, const Ray& r, float kMin, Closest& best, TravCount& cnt);
template <class C> RT_FN bool any_wide(const SceneDev& sc, const WideTree& tree, const WideRay& wr, const Ray& r, float kMin, float kMax, TravCount& cnt);
RT_FN WideRay wide_ray_any(const SceneDev& sc, const WideTree& tree, const Ray& r, float kMax, float cap, int mode, bool& again);
// (rt_ldsw.h: the W-wide tree in LDS has overloads of the same three names for its own tree and ray types; the calls below depend on
//  the tree's type, so they are resolved where the kernels are instantiated)

template <class C, class Tree>
RT_FN Closest closest_hit(const SceneDev& sc, const Tree& tree, Ray r, float kMin, float kMax, const RngKey& key, uint32_t depth, TravCount& cnt) {
    Closest best; best.k = kMax; best.obj = RT_NO_HIT; best.aux = 0.0f;
    if (ray_has_nan(r)) return best;
    for (uint32_t i = 0; i < sc.n_unbounded; i++) {
        uint32_t obj = sc.unbounded[i], code = sc.obj_code[obj];
        float k; cnt.prims++;
        if (try_object_solid<C>(sc, code >> OBJ_TYPE_SHIFT, code & OBJ_INDEX_MASK, r, kMin, best.k, k)) { consider(best, k, obj, 0.0f); cnt.prim_hit++; }
    }
    if constexpr (Tree::wide) { if (sc.n_nodes) closest_wide<C>(sc, tree, r, kMin, best, cnt); }
    else if (Tree::list && sc.n_list_pairs) list_closest<C>(sc, r, kMin, best, cnt);
    else if (sc.n_nodes) {
        const ClosestBox box = closest_box(r);
        Stk sp = stk_empty(); int cur = 0, pend = TRAV_DONE;
        for (;;) {
            if constexpr (Tree::vote) { while (step_vote(cur, pend)) if (cur >= 0) { cnt.nodes += 2; node_step<false>(sc, tree, box, kMin, best.k, cur, pend, sp, cnt.box_miss); } }
            else while (cur >= 0) { cnt.nodes += 2; node_step<false>(sc, tree, box, kMin, best.k, cur, pend, sp, cnt.box_miss); }   // every lane does box tests here
            if (pend != TRAV_DONE) {                               // leaves: most lanes arrive here with one parked
                uint32_t obj = (uint32_t)~pend; float k; cnt.prims++;
                if (tree.template leaf_try<C>(sc, obj, r, kMin, best.k, k)) { consider(best, k, obj, 0.0f); cnt.prim_hit++; }
            }
            if (cur == TRAV_DONE) break;
            leaf_advance(tree, cur, pend, sp);
        }
    }
    if constexpr (RT_HAS_PRIM(C, RT_HIT_MEDIUM)) {
        for (uint32_t i = 0; i < sc.n_media; i++) {
            uint32_t obj = sc.media[i], code = sc.obj_code[obj];
            float k, aux; cnt.prims++;
            // SingleObjectBvhNode box test (the medium's BoundingVolume is its boundary's, ConstantDensityMedium.cs:80) is implied:
            // the boundary intersector only returns points inside that box.
            if (try_medium<C>(sc, code & OBJ_INDEX_MASK, obj, r, kMin, best.k, key, depth, k, aux)) { consider(best, k, obj, aux); cnt.prim_hit++; }
        }
    }
    // GraphicsValidator.CheckValueRange, Core/Debugging/GraphicsValidator.cs:69 via Core/RenderJob.cs:609-614
    if (best.obj != RT_NO_HIT && !((best.k >= kMin) && (best.k <= kMax))) best.obj = RT_NO_HIT;
    return best;
}

// RenderJob.AnyIntersectionFast, Core/RenderJob.cs:575.  The reference scans every object's FastTryHit; culling by
// conservative boxes returns the same boolean.
template <class C, class Tree>
RT_FN bool any_hit(const SceneDev& sc, const Tree& tree, Ray r, float kMin, float kMax, float cap, TravCount& cnt) {
    for (uint32_t i = 0; i < sc.n_unbounded; i++) {
        uint32_t code = sc.obj_code[sc.unbounded[i]]; cnt.prims++;
        if (fast_object_solid<C>(sc, code >> OBJ_TYPE_SHIFT, code & OBJ_INDEX_MASK, r, kMin, kMax)) return true;
    }
    if constexpr (RT_HAS_PRIM(C, RT_HIT_MEDIUM)) {
        for (uint32_t i = 0; i < sc.n_media; i++) {
            rt_medium m = sc.media_rec[sc.obj_code[sc.media[i]] & OBJ_INDEX_MASK]; cnt.prims++;
            if (fast_object_solid<C>(sc, m.boundary_type, m.boundary_index, r, kMin, kMax)) return true;      // ConstantDensityMedium.cs:77
        }
    }
    if (!(Tree::list && sc.n_list_pairs) && !sc.n_nodes) return false;
    bool again;
    if constexpr (Tree::wide) {
        for (int look = 0;; look = 1) {                          // once, or twice (see any_box)
            const auto wr = wide_ray_any(sc, tree, r, kMax, cap, look, again);
            if (any_wide<C>(sc, tree, wr, r, kMin, kMax, cnt)) return true;
            if (!again) return false;
        }
    } else {
    AnyBox box = any_box(sc, r, kMax, cap, 0, again);
    for (;;) {                                                   // once, or twice (see any_box)
        if (Tree::list && sc.n_list_pairs) { if (list_any<C>(sc, r, box, kMin, kMax, cnt)) return true; }
        else {
            Stk sp = stk_empty(); int cur = 0, pend = TRAV_DONE;
            uint32_t unused = 0;
            for (;;) {
                if constexpr (Tree::vote) { while (step_vote(cur, pend)) if (cur >= 0) { cnt.nodes += 2; node_step<true>(sc, tree, box, kMin, kMax, cur, pend, sp, unused); } }
                else while (cur >= 0) { cnt.nodes += 2; node_step<true>(sc, tree, box, kMin, kMax, cur, pend, sp, unused); }
                if (pend != TRAV_DONE) {
                    cnt.prims++;
                    if (tree.template leaf_fast<C>(sc, (uint32_t)~pend, r, kMin, kMax)) return true;
                }
                if (cur == TRAV_DONE) break;
                leaf_advance(tree, cur, pend, sp);
            }
        }
        if (!again) return false;
        box = any_box(sc, r, kMax, cap, 1, again);
    }
    }
}

// ------------------------------------------------------------------------------------------------
// HitRecord reconstruction for the winning object (the rest of each TryHit body).
// ------------------------------------------------------------------------------------------------
struct HitRec { f3 wp, n, lp; float k, u, v, shader; uint32_t obj, mat; bool outside; };    // lp (HitRecord.LocalPoint) only with WANT_UV

RT_FN void sphere_uv(f3 p, float& u, float& v) {                                       // Sphere.cs:78-93
    float theta = acosf(-p.y);
    float phi = atan2f(-p.z, p.x) + PI_F;
    u = phi / (2 * PI_F);
    v = theta / PI_F;
}

template <class C, bool WANT_UV>
RT_FN HitRec finalize_hit(const SceneDev& sc, Ray r, Closest c, const RngKey& key, uint32_t depth,
                          bool have_rec = false, float4 rec_geo = make_float4(0, 0, 0, 0), uint32_t rec_code = 0, uint32_t rec_mat = 0) {
    // have_rec: the caller has the object's record of sc.obj_shade (code, material, and the 16 B this function needs of the primitive
    // when it wants no UV: a sphere, a plane's or quad's or triangle's or disk's normal, an axis plane's offset): nothing is fetched here then
    HitRec h; h.k = c.k; h.obj = c.obj; h.u = 0; h.v = 0; h.shader = 0; h.outside = true;
    h.mat = have_rec ? rec_mat : sc.uniform_material != RT_NO_HIT ? sc.uniform_material : sc.obj_material[c.obj];
    uint32_t code = have_rec ? rec_code : sc.all_code_affine ? sc.leaf_code_base + c.obj : sc.obj_code[c.obj], type = code >> OBJ_TYPE_SHIFT; size_t idx = code & OBJ_INDEX_MASK;
    h.wp = point_at(r.o, r.d, c.k);
    h.n = mk3(0, 0, 0); h.lp = h.wp;
    switch (type) {
        case RT_HIT_SPHERE: if constexpr (RT_HAS_PRIM(C, RT_HIT_SPHERE)) {                                                            // Sphere.cs:59-70
            float4 s = have_rec ? rec_geo : sc.spheres[idx];
            f3 outN = normalize(h.wp - mk3(s.x, s.y, s.z));
            bool inside = dot(r.d, outN) > 0.0f;
            h.n = inside ? -outN : outN; h.outside = !inside;
            if (WANT_UV) { sphere_uv(outN, h.u, h.v); h.lp = h.wp - mk3(s.x, s.y, s.z); }
        } break;
        case RT_HIT_QUAD: case RT_HIT_TRIANGLE: if constexpr (RT_HAS_PRIM(C, RT_HIT_QUAD) || RT_HAS_PRIM(C, RT_HIT_TRIANGLE)) {                                        // Quad.cs:100-119
            const float4* q = (type == RT_HIT_QUAD ? sc.quads : sc.triangles) + 4 * idx;
            float4 q1 = have_rec ? rec_geo : q[1];
            h.n = mk3(q1.x, q1.y, q1.z);
            h.outside = dot(h.n, r.d) < 0;
            if (WANT_UV) {
                float4 q0 = q[0];
                f3 lp = h.wp - mk3(q0.x, q0.y, q0.z); float4 c1 = q[2], c2 = q[3];
                h.u = (lp.x * c1.x) + (lp.y * c1.y) + (lp.z * c1.z) + c1.w;
                h.v = (lp.x * c2.x) + (lp.y * c2.y) + (lp.z * c2.z) + c2.w;
                h.lp = lp;
            }
        } break;
        case RT_HIT_INFINITE_PLANE: if constexpr (RT_HAS_PRIM(C, RT_HIT_INFINITE_PLANE)) {                                                    // InfinitePlane.cs:55-60
            float4 p1 = have_rec ? rec_geo : sc.inf_planes[2 * idx + 1];
            h.n = mk3(p1.x, p1.y, p1.z);
            h.outside = dot(r.d, h.n) < 0;
            if (WANT_UV) { float4 p0 = sc.inf_planes[2 * idx]; h.lp = h.wp - mk3(p0.x, p0.y, p0.z); }
        } break;
        case RT_HIT_XY_PLANE: case RT_HIT_XZ_PLANE: case RT_HIT_YZ_PLANE: if constexpr (RT_HAS_PRIM(C, RT_HIT_XY_PLANE) || RT_HAS_PRIM(C, RT_HIT_XZ_PLANE) || RT_HAS_PRIM(C, RT_HIT_YZ_PLANE)) {              // XYPlane.cs:69-101
            int axis = type - RT_HIT_XY_PLANE;
            const float4* p = sc.axis_planes[axis] + 2 * idx;
            float planeC = have_rec ? rec_geo.x : p[1].x;
            float oC = axis == 0 ? r.o.z : axis == 1 ? r.o.y : r.o.x;
            f3 unit = axis == 0 ? mk3(0, 0, 1) : axis == 1 ? mk3(0, 1, 0) : mk3(1, 0, 0);
            h.n = oC < planeC ? -unit : unit;
            if (WANT_UV) {
                float4 p0 = p[0];
                float a = axis == 2 ? h.wp.y : h.wp.x, b = axis == 0 ? h.wp.y : h.wp.z;
                h.u = (a - p0.x) / (p0.y - p0.x); h.v = (b - p0.z) / (p0.w - p0.z);
                float ca = (p0.x + p0.y) / 2.0f, cb = (p0.z + p0.w) / 2.0f;                // XYPlane.cs:28 centre
                h.lp = h.wp - (axis == 0 ? mk3(ca, cb, planeC) : axis == 1 ? mk3(ca, planeC, cb) : mk3(planeC, ca, cb));
            }
        } break;
        case RT_HIT_DISK: if constexpr (RT_HAS_PRIM(C, RT_HIT_DISK)) {                                                              // Disk.cs:83-97
            const float4* d = sc.disks + 4 * idx;
            float4 d1 = have_rec ? rec_geo : d[1];
            h.n = mk3(d1.x, d1.y, d1.z);
            h.outside = dot(r.d, h.n) < 0;
            if (WANT_UV) {
                float4 d0 = d[0];
                f3 lp = h.wp - mk3(d0.x, d0.y, d0.z); float4 c1 = d[2], c2 = d[3];
                float u = (lp.x * c1.x) + (lp.y * c1.y) + (lp.z * c1.z) + c1.w;
                float v = (lp.x * c2.x) + (lp.y * c2.y) + (lp.z * c2.z) + c2.w;
                if (u < 0 || u > 1 || is_nan(u) || v < 0 || v > 1 || is_nan(v)) {
                    u = (u > 0.0f) ? u : 0.0f; u = (u < 1.0f) ? u : 1.0f;
                    v = (v > 0.0f) ? v : 0.0f; v = (v < 1.0f) ? v : 1.0f;
                }
                h.u = u; h.v = v; h.lp = lp;
            }
        } break;
        case RT_HIT_BOX: if constexpr (RT_HAS_PRIM(C, RT_HIT_BOX)) {                                                               // Box.cs:147-182
            const float4* b = sc.boxes + 6 * idx;
            float k; BoxSolve bs;
            box_core(b, r, c.k, c.k, false, k, bs);     // same t1/s as the accepted test; the range only selects near/far
            float4 r0 = b[3], r1 = b[4], r2 = b[5];     // rows: r0.xyz | r0.w r1.x r1.y | r1.z r1.w r2.x
            f3 nrm;
            if ((bs.t1.x > bs.t1.y) && (bs.t1.x > bs.t1.z)) nrm = mk3(r0.x, r0.y, r0.z) * bs.s.x;
            else if (bs.t1.y > bs.t1.z) nrm = mk3(r0.w, r1.x, r1.y) * bs.s.y;
            else nrm = mk3(r1.z, r1.w, r2.x) * bs.s.z;
            h.n = normalize(nrm);
            h.outside = dot(r.d, nrm) < 0.0f;
            if (WANT_UV) {                                                                // LocalPoint = Transform(worldPoint, WorldToBoxTransform), Box.cs:172
                float4 m0 = b[0], m1 = b[1], m2 = b[2];
                h.lp = mk3((h.wp.x * m0.x) + (h.wp.y * m0.w) + (h.wp.z * m1.z) + m2.y, (h.wp.x * m0.y) + (h.wp.y * m1.x) + (h.wp.z * m1.w) + m2.z,
                           (h.wp.x * m0.z) + (h.wp.y * m1.y) + (h.wp.z * m2.x) + m2.w);
            }
        } break;
        case RT_HIT_CAPSULE: if constexpr (RT_HAS_PRIM(C, RT_HIT_CAPSULE)) {                                                           // Capsule.cs:114-140, CapNormal :186-191
            const float4* cp = sc.capsules + 5 * idx;
            float4 c0 = cp[0], c1 = cp[1];
            f3 P1 = mk3(c0.x, c0.y, c0.z), P2 = mk3(c1.x, c1.y, c1.z); float radius = c0.w;
            f3 ba = P2 - P1; float baba = dot(ba, ba);
            f3 pa = h.wp - P1;
            float hh = dot(pa, ba) / baba; hh = hh < 0.0f ? 0.0f : (hh > 1.0f ? 1.0f : hh);
            h.n = (pa - (hh * ba)) / radius;
            h.outside = !(dot(r.d, h.n) > 0.0f);
            if (WANT_UV) {
                float4 m0 = cp[2], m1 = cp[3]; float m8 = cp[4].x;
                f3 q = mk3((pa.x * m0.x) + (pa.y * m0.w) + (pa.z * m1.z) + 0.0f, (pa.x * m0.y) + (pa.y * m1.x) + (pa.z * m1.w) + 0.0f, (pa.x * m0.z) + (pa.y * m1.y) + (pa.z * m8) + 0.0f);
                float rawU = atan2f(q.y, q.x);
                h.u = (rawU + PI_F) / 6.2831855f;
                float lo = -radius, hi = distance(P1, P2) + radius;
                h.v = (q.z - lo) / (hi - lo);
                h.lp = h.wp - lerp3(P1, P2, 0.5f);
            }
        } break;
        case RT_HIT_CYLINDER: if constexpr (RT_HAS_PRIM(C, RT_HIT_CYLINDER)) {                                                          // Cylinder.cs:100-104
            float k; f3 nrm = mk3(0, 0, 0);
            try_cylinder(sc.cylinders + 2 * idx, r, c.k, c.k, k, nrm, false);
            h.n = nrm; h.outside = !(dot(r.d, nrm) > 0.0f);
            if (WANT_UV) { float4 c0 = sc.cylinders[2 * idx], c1 = sc.cylinders[2 * idx + 1]; h.lp = h.wp - lerp3(mk3(c0.x, c0.y, c0.z), mk3(c1.x, c1.y, c1.z), 0.5f); }
        } break;
        case RT_HIT_MEDIUM: if constexpr (RT_HAS_PRIM(C, RT_HIT_MEDIUM)) {                                                            // ConstantDensityMedium.cs:61-67
            float4 u = rng_draw(key, depth, SLOT_MEDIUM, c.obj);
            h.wp = point_at(r.o, r.d, c.aux);         // sic: measured from the ray origin
            h.n = on_unit_sphere(u.y, u.z, u.w);
            h.shader = c.aux; h.outside = true; h.lp = h.wp;
        } break;
        default: break;
    }
    return h;
}

// ------------------------------------------------------------------------------------------------
// Textures, skyboxes
// ------------------------------------------------------------------------------------------------
struct col3 { float r, g, b; };
RT_FN col3 mkc(float r, float g, float b) { col3 c; c.r = r; c.g = g; c.b = b; return c; }
RT_FN col3 operator+(col3 a, col3 b) { return mkc(a.r + b.r, a.g + b.g, a.b + b.b); }
RT_FN col3 operator*(col3 a, col3 b) { return mkc(a.r * b.r, a.g * b.g, a.b * b.b); }
RT_FN col3 operator*(col3 a, float s) { return mkc(a.r * s, a.g * s, a.b * s); }
RT_FN col3 operator/(col3 a, float s) { return mkc(a.r / s, a.g / s, a.b / s); }

// MarbleTexture's noise: SharpNoise's gradient table is not part of the reference tree (parity unpinned);
// a libnoise-style gradient noise over hashed gradients, 5 octaves, lacunarity 3, persistence .5, linear interpolation.
RT_FN uint32_t hash3(int32_t x, int32_t y, int32_t z, int32_t seed) {
    uint32_t h = (uint32_t)x * 1619u + (uint32_t)y * 31337u + (uint32_t)z * 6971u + (uint32_t)seed * 1013u;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
RT_FN float grad_noise(float fx, float fy, float fz, int32_t ix, int32_t iy, int32_t iz, int32_t seed) {
    uint32_t h = hash3(ix, iy, iz, seed);
    float gx = (float)((int)(h & 1023u) - 512) * (1.0f / 512.0f);
    float gy = (float)((int)((h >> 10) & 1023u) - 512) * (1.0f / 512.0f);
    float gz = (float)((int)((h >> 20) & 1023u) - 512) * (1.0f / 512.0f);
    float dx = fx - (float)ix, dy = fy - (float)iy, dz = fz - (float)iz;
    return ((gx * dx) + (gy * dy) + (gz * dz)) * 1.2f;
}
RT_FN float coherent_noise(float x, float y, float z, int32_t seed) {
    float flx = floorf(x), fly = floorf(y), flz = floorf(z);
    int32_t x0 = (int32_t)flx, y0 = (int32_t)fly, z0 = (int32_t)flz;
    int32_t x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    float xs = x - flx, ys = y - fly, zs = z - flz;
    float n0 = grad_noise(x, y, z, x0, y0, z0, seed), n1 = grad_noise(x, y, z, x1, y0, z0, seed);
    float ix0 = lerpf(n0, n1, xs);
    n0 = grad_noise(x, y, z, x0, y1, z0, seed); n1 = grad_noise(x, y, z, x1, y1, z0, seed);
    float ix1 = lerpf(n0, n1, xs);
    float iy0 = lerpf(ix0, ix1, ys);
    n0 = grad_noise(x, y, z, x0, y0, z1, seed); n1 = grad_noise(x, y, z, x1, y0, z1, seed);
    ix0 = lerpf(n0, n1, xs);
    n0 = grad_noise(x, y, z, x0, y1, z1, seed); n1 = grad_noise(x, y, z, x1, y1, z1, seed);
    ix1 = lerpf(n0, n1, xs);
    float iy1 = lerpf(ix0, ix1, ys);
    return lerpf(iy0, iy1, zs);
}
// libnoise's Perlin::GetValue: octaves of coherent noise, seed + octave each, frequency x lacunarity, amplitude x persistence
RT_FN float perlin_noise(float x, float y, float z, float frequency, float lacunarity, float persistence, int octaves, int32_t seed) {
    float value = 0.0f, amplitude = 1.0f;
    x = x * frequency; y = y * frequency; z = z * frequency;
    for (int o = 0; o < octaves; o++) {
        value = value + (coherent_noise(x, y, z, seed + o) * amplitude);
        x = x * lacunarity; y = y * lacunarity; z = z * lacunarity;
        amplitude = amplitude * persistence;
    }
    return value;
}
RT_FN float marble_noise(float x, float y, float z) { return perlin_noise(x, y, z, 1.0f, 3.0f, 0.5f, 5, 0); }      // MarbleTexture.cs:11-17

template <class C>
RT_FN col3 texture_colour(const SceneDev& sc, uint32_t tex, f3 wp) {
    const rt_texture* t = sc.textures + tex;
    if (!(RT_HAS_TEX(C, RT_TEX_MARBLE) || RT_HAS_TEX(C, RT_TEX_GREYSCALE_PERLIN)) || t->type == RT_TEX_SOLID) return mkc(t->f[0], t->f[1], t->f[2]);   // SolidColourTexture.cs:14
    if (RT_HAS_TEX(C, RT_TEX_GREYSCALE_PERLIN) && t->type == RT_TEX_GREYSCALE_PERLIN) {                                  // GreyscaleNoiseTexture.cs:20
        float v = perlin_noise(wp.x, wp.y, wp.z, t->f[0], t->f[1], t->f[2], (int)t->f[3], (int32_t)t->f[4]);
        return mkc(v, v, v);
    }
    float scale = t->f[0], noiseScale = t->f[1], noiseStrength = t->f[2], dropoff = t->f[3];   // MarbleTexture.cs:38-48
    float x = wp.x / scale, y = wp.y / scale, z = wp.z / scale;
    float tt = x + y + z;
    tt += marble_noise(x / noiseScale, y / noiseScale, z / noiseScale) * noiseStrength;
    float val = sinf(tt);
    val = (0.5f * val) + 0.5f;
    val = powf(val, dropoff);
    return (mkc(0, 0, 0) * (1.0f - val)) + (mkc(1, 1, 0.95f) * val);                    // Colour.Lerp, Core/Colour.cs:214
}

RT_FN col3 sky_colour(const SceneDev& sc, f3 dir) {
    if (sc.skybox.type == RT_SKY_SINGLE_COLOUR) return mkc(sc.skybox.colour[0], sc.skybox.colour[1], sc.skybox.colour[2]);
    float t = 0.5f * (dir.y + 1);                                                        // DefaultSkyBox.cs:12-13
    return mkc((1 - t) + (0.5f * t), (1 - t) + (0.7f * t), (1 - t) + (1.0f * t));
}

// ------------------------------------------------------------------------------------------------
// Lights
// ------------------------------------------------------------------------------------------------
struct ShadowRay { Ray r; float kMin, kMax; };

// Light sampling arithmetic, two modes (template parameter EXACT of the shadow kernels; rt_render_opts.flags / RT_HIP_EXACT_LIGHTS):
//   EXACT = false (default): traversal and the intersectors are correctly rounded (they decide hit ids and K, which the parity
//     check compares bit for bit); what PRODUCES a shadow ray and weights its light is sampling arithmetic whose last bit only
//     moves a sample by an ulp, so it uses the hardware's 1-ulp v_rsq_f32 / v_sqrt_f32 / v_rcp_f32 / v_exp_f32 instead of the
//     ~12-instruction correctly rounded sequences (a normalise is 4 instructions instead of 45).  A shadow ray that grazes an
//     occluder within that ulp may fall the other way: the image then differs from the reference arithmetic by one light sample
//     in a few pixels per thousand (tests bound it).
//   EXACT = true: every operator as the reference writes it (Normalize = v / Sqrt(Dot), Distance, x / AttenuationRadius,
//     MathF.Pow(MathF.E, y), sum / samples), correctly rounded: the shadow rays are bit-identical to the CPU restatement's.
RT_FN f3 normalize_fast(f3 a) { return a * __builtin_amdgcn_rsqf(dot(a, a)); }
RT_FN float length_fast(f3 a) { return __builtin_amdgcn_sqrtf(dot(a, a)); }
RT_FN float rcp_fast(float a) { return __builtin_amdgcn_rcpf(a); }
RT_FN float exp_e_fast(float y) { return __builtin_amdgcn_exp2f(y * 1.44269502f); }     // MathF.Pow(MathF.E, y): log2((float)e) rounds to the same float as log2(e)
template <bool EXACT> RT_FN float exp_e(float y) { if constexpr (EXACT) return powf(E_F, y); else return exp_e_fast(y); }
template <bool EXACT> RT_FN float div_by(float a, float b) { if constexpr (EXACT) return a / b; else return a * rcp_fast(b); }

template <bool EXACT>
RT_FN ShadowRay default_shadow_ray(f3 hitPos, f3 lightPos) {                             // SimpleLightBase.DefaultGetShadowRayForHit, SimpleLightBase.cs:68-77
    ShadowRay s; s.r.o = hitPos; s.kMin = 0.01f;
    if constexpr (EXACT) {
        s.r.d = normalize(lightPos - hitPos);                                            // Ray.FromPoints (Core/Ray.cs:26) -> the Ray ctor normalises (:17)
        s.kMax = distance(hitPos, lightPos) - 0.01f;                                     // Vector3.Distance(hitPos, lightPos) - kMin
    } else {
        f3 v = lightPos - hitPos;
        float d2 = dot(v, v), inv = __builtin_amdgcn_rsqf(d2);
        s.r.d = v * inv;
        s.kMax = (d2 * inv) - 0.01f;
    }
    return s;
}
template <bool EXACT>
RT_FN float attenuation(const rt_light* l, float x) {                                    // the DistanceAttenuation factories, SimpleLightBase.cs:83-132
    switch (l->atten_kind) {
        case RT_ATTEN_CONSTANT: return l->atten_p[0];
        case RT_ATTEN_LINEAR: return 1 - x;
        case RT_ATTEN_POWER: return powf(1 - x, l->atten_p[0]);
        case RT_ATTEN_POWER_STAY_HIGH: return 1 - powf(x, l->atten_p[0]);
        case RT_ATTEN_LOGISTIC: return div_by<EXACT>(1.0f, 1 + exp_e<EXACT>(l->atten_p[1] * (x - l->atten_p[0])));
        case RT_ATTEN_LOGISTIC_LKX: return div_by<EXACT>(l->atten_p[0], 1.0f + exp_e<EXACT>(-l->atten_p[1] * (x - l->atten_p[2])));
        default: return exp_e<EXACT>(l->atten_p[0] * -x);
    }
}
// what SimpleLightBase.CalculateLight (SimpleLightBase.cs:48-66) does after the occlusion test, :57-65: cutoff, attenuation of kMax / AttenuationRadius, clamp at 0
template <bool EXACT>
RT_FN col3 light_contribution(const rt_light* l, float kMax) {
    if (kMax > l->cutoff_radius) return mkc(0, 0, 0);
    float scale = attenuation<EXACT>(l, div_by<EXACT>(kMax, l->attenuation_radius));
    scale = maxf_net(scale, 0);
    return mkc(l->colour[0], l->colour[1], l->colour[2]) * mkc(scale, scale, scale);
}
// GetShadowRayForHit: PointLight.cs:9, DiffuseSphereLight.cs:88-142, DiffuseShapedLight.cs:69-105
template <class C, bool EXACT>
RT_FN ShadowRay light_shadow_ray(const SceneDev& sc, const rt_light* l, f3 wp, f3 normal, const RngKey& key, uint32_t depth, uint32_t slotBase, uint32_t sub) {
    f3 pos = mk3(l->position[0], l->position[1], l->position[2]);
    // (the host picks a kernel variant whose Cfg covers every light type in the scene, so a type outside C::lights cannot occur)
    if (l->type == RT_LIGHT_POINT || C::lights == (1u << RT_LIGHT_POINT)) return default_shadow_ray<EXACT>(wp, pos);
    if (RT_HAS_LIGHT(C, RT_LIGHT_DIFFUSE_SPHERE) && (l->type == RT_LIGHT_DIFFUSE_SPHERE || !RT_HAS_LIGHT(C, RT_LIGHT_DIFFUSE_SHAPED))) {
        float R = l->diffusion_radius;
        if ((EXACT ? distance(wp, pos) : length_fast(wp - pos)) <= R * 1.01f) {
            for (uint32_t i = 0; i < 1000; i++) {
                float4 u = rng_draw(key, depth, slotBase + i, sub);
                f3 randDir = on_unit_sphere(u.x, u.y, u.z);
                if (dot(randDir, normal) < 0) randDir = -randDir;
                f3 oc = wp - pos;
                float a = length_sq(randDir);
                float halfB = dot(oc, randDir);
                float c = length_sq(oc) - (R * R);
                float disc = (halfB * halfB) - (a * c);
                if (disc < 0) continue;
                float sqrtD = sqrtf(disc);
                float k = (-halfB - sqrtD) / a;
                if (is_nan(k)) { k = (-halfB + sqrtD) / a; if (is_nan(k)) continue; }
                ShadowRay s; s.r.o = wp; s.r.d = EXACT ? normalize(randDir) : normalize_fast(randDir); s.kMin = 0.001f; s.kMax = k;   // new Ray(hit, randDir) normalises
                return s;
            }
            return default_shadow_ray<EXACT>(wp, pos);
        }
        float4 u = rng_draw(key, depth, slotBase, sub);
        f3 randDir;
        if constexpr (EXACT) {
            randDir = on_unit_sphere(u.x, u.y, u.z);                                     // RandomOnUnitSphere
            if (dot(randDir, normalize(wp - pos)) < 0) randDir = -randDir;               // DiffuseSphereLight.cs: Dot(randDir, Normalize(hit - centre))
        } else {
            randDir = normalize_fast(mk3(plus_minus_one(u.x), plus_minus_one(u.y), plus_minus_one(u.z)));
            if (dot(randDir, wp - pos) < 0) randDir = -randDir;                          // the sign of the same dot product
        }
        return default_shadow_ray<EXACT>(wp, pos + (randDir * R));
    }
    // DiffuseShapedLight
    if constexpr (!RT_HAS_LIGHT(C, RT_LIGHT_DIFFUSE_SHAPED)) return default_shadow_ray<EXACT>(wp, pos);
    f3 mn = mk3(l->shape_aabb_min[0], l->shape_aabb_min[1], l->shape_aabb_min[2]), mx = mk3(l->shape_aabb_max[0], l->shape_aabb_max[1], l->shape_aabb_max[2]);
    for (uint32_t i = 0; i < 1000; i++) {
        float4 u = rng_draw(key, depth, slotBase + i, sub);
        f3 p2 = mk3(lerpf(mn.x, mx.x, u.x), lerpf(mn.y, mx.y, u.y), lerpf(mn.z, mx.z, u.z));
        Ray ry; ry.o = wp; ry.d = normalize(p2 - wp);
        float k;
        if (try_object_solid<C>(sc, l->shape_type, l->shape_index, ry, 0.0001f, __builtin_inff(), k)) {
            ShadowRay s; s.r = ry; s.kMin = 0.001f; s.kMax = k - 0.001f; return s;
        }
    }
    ShadowRay s; s.r.o = wp; s.r.d = normalize(splat(1.0f)); s.kMin = 0.0f; s.kMax = __builtin_inff();
    return s;
}

// Camera.GetRay, Core/Camera.cs:35-42 with RandUtils.RandomInUnitCircle, Core/RandUtils.cs:53-59
RT_FN Ray camera_ray(const rt_camera& cam, float u, float v, float thetaU, float rU) {
    f3 U = mk3(cam.u[0], cam.u[1], cam.u[2]), V = mk3(cam.v[0], cam.v[1], cam.v[2]);
    f3 origin = mk3(cam.look_from[0], cam.look_from[1], cam.look_from[2]);
    // A pinhole (LensRadius 0, every builtin scene): the reference still draws the point on the lens and multiplies it by 0, so the
    // offset is (+-0, +-0, +-0) and LookFrom + offset is LookFrom bit for bit — unless a component of LookFrom is -0 (-0 + +0 = +0)
    // or the basis is not finite (0 * inf): then, and for a real lens, the arithmetic is done.  (Wave-uniform: the sincos and the
    // square root are a third of what making a camera ray costs, and depth 0 makes it twice, rt_kernels.h: primary_ray.)
    const bool pinhole = cam.lens_radius == 0.0f && isfinite(U.x) && isfinite(U.y) && isfinite(U.z) && isfinite(V.x) && isfinite(V.y) && isfinite(V.z) &&
                         !(origin.x == 0.0f && __builtin_signbitf(origin.x)) && !(origin.y == 0.0f && __builtin_signbitf(origin.y)) && !(origin.z == 0.0f && __builtin_signbitf(origin.z));
    if (!pinhole) {
        float theta = 0 + (((2 * PI_F) - 0) * thetaU);
        float rr = sqrtf(rU);
        float sx = sinf(theta), cy = cosf(theta);
        float randX = (rr * sx) * cam.lens_radius, randY = (rr * cy) * cam.lens_radius;
        f3 offset = (U * randX) + (V * randY);
        origin = origin + offset;
    }
    f3 llc = mk3(cam.lower_left_corner[0], cam.lower_left_corner[1], cam.lower_left_corner[2]);
    f3 hor = mk3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]), ver = mk3(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    f3 direction = (llc + (u * hor) + (v * ver)) - origin;
    Ray r; r.o = origin; r.d = normalize(normalize(direction));      // GetRay normalises, then the Ray ctor normalises again (Core/Ray.cs:17)
    return r;
}

} // namespace rtd
