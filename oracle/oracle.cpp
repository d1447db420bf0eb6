/*
 * oracle.cpp — CPU restatement of Ararem.RayTracer's per-sample trace loop.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the HIP library never links or calls it.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * root; Core/ = Ararem.RayTracer.Core/, Impl/ = Ararem.RayTracer.Impl/).  The reference is C# on
 * .NET 7 and cannot be built in this environment (no dotnet/mono; LibArarem submodule is empty),
 * so this restatement is pinned by the only artefacts the reference holds for the path:
 *   (1) Renders/Demo Scene/image.png sky rows  -> camera + y-flip + DefaultSkyBox + tonemap, bit-exact
 *   (2) the UV table in Impl/Hittables/Sphere.cs:83-85
 *   (3) the capsule basis/UV ranges in Impl/Hittables/Capsule.cs:126-130
 *   (4) the GraphicsValidator invariants (Core/Debugging/GraphicsValidator.cs:57-69)
 * Everything else (hit ids, K, converged images) is "parity vs this restatement".
 * MarbleTexture's SharpNoise Perlin is an un-vendored third-party dependency (SharpNoise 0.12.1.1):
 * its gradient table is not in the tree, so marble is PARITY UNPINNED (see marble_noise below).
 *
 * Deliberate, documented deviations (the reference result is not reproducible there):
 *   D1  Random.Shared (unseeded xoshiro, Core/RandUtils.cs:17) -> Philox4x32-10 keyed by
 *       (seed, pixel, pass, depth, slot, sub); the device uses the identical keying.
 *   D2  exact ties in K between two objects: the reference returns whichever its BVH visits last
 *       (Core/Acceleration/BinaryBvhNode.cs:28-30) and its BVH topology is implementation-defined
 *       (inconsistent comparer, BvhTree.cs:186); here the larger object index wins.
 *   D3  ConstantDensityMedium objects are tested after all other objects (a legal reference
 *       visiting order), because their result depends on kMax at the time of the test
 *       (Impl/Hittables/ConstantDensityMedium.cs:49-51).
 *   D4  a ray with a NaN component is a miss.
 *
 * Compile with -ffp-contract=off: RyuJIT never fuses a*b+c, so neither may we.
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <vector>
#include <memory>
#include <algorithm>
#include <thread>
#include <atomic>
#include <limits>
#include <string>

#include "../include/rt_hip.h"

namespace {

constexpr float PI_F = 3.14159265358979323846f;   // MathF.PI
constexpr float E_F  = 2.71828182845904523536f;   // MathF.E
constexpr float INF  = std::numeric_limits<float>::infinity();

// ---------------------------------------------------------------------------------------------
// System.Numerics.Vector3 semantics (.NET 7 software definitions; SURVEY §8a footer)
// ---------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
inline V3 v3(float a) { return {a, a, a}; }
inline V3 v3(const float* p) { return {p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }            // Vector3 unary minus = Zero - value? .NET: negates each lane
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }   // true division per lane
inline float Dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 Cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float LengthSquared(V3 a) { return Dot(a, a); }
inline float Length(V3 a) { return std::sqrt(LengthSquared(a)); }
inline V3 Normalize(V3 a) { return a / Length(a); }
inline float DistanceSquared(V3 a, V3 b) { V3 d = a - b; return Dot(d, d); }
inline float Distance(V3 a, V3 b) { return std::sqrt(DistanceSquared(a, b)); }
inline V3 Lerp(V3 a, V3 b, float t) { return (a * (1.0f - t)) + (b * t); }
inline V3 Reflect(V3 v, V3 n) { float d = Dot(v, n); return v - (n * (2.0f * d)); }
inline V3 Min3(V3 a, V3 b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }
inline V3 Max3(V3 a, V3 b) { return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }
// MathF.Max / MathF.Min propagate NaN
inline float MaxF(float a, float b) { if (std::isnan(a) || std::isnan(b)) return NAN; return a > b ? a : b; }
inline float MinF(float a, float b) { if (std::isnan(a) || std::isnan(b)) return NAN; return a < b ? a : b; }

// Core/Colour.cs:152-234
struct Colour { float r, g, b; };
inline Colour col(float v) { return {v, v, v}; }
inline Colour col(const float* p) { return {p[0], p[1], p[2]}; }
inline Colour operator+(Colour a, Colour b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Colour operator*(Colour a, Colour b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Colour operator*(Colour a, float s) { return a * col(s); }                     // Colour.cs:173
inline Colour operator/(Colour a, Colour b) { return {a.r / b.r, a.g / b.g, a.b / b.b}; }
inline Colour operator/(Colour a, float s) { return a / col(s); }                     // Colour.cs:165
inline Colour ColourLerp(Colour a, Colour b, float t) { return (a * (1.0f - t)) + (b * t); }  // Colour.cs:214
const Colour Black{0, 0, 0};

// Core/Ray.cs:14-26 — the constructor normalises, always.
struct Ray {
    V3 o, d;
    Ray() : o{0, 0, 0}, d{0, 0, 0} {}
    Ray(V3 origin, V3 direction) : o(origin), d(Normalize(direction)) {}
    static Ray Raw(V3 origin, V3 direction) { Ray r; r.o = origin; r.d = direction; return r; }  // parity hooks: rays used as given
    V3 PointAt(float k) const { return o + (k * d); }
    static Ray FromPoints(V3 p1, V3 p2) { return Ray(p1, p2 - p1); }
};

// ---------------------------------------------------------------------------------------------
// D1: counter RNG replacing Random.Shared.  Philox4x32-10; identical constants and keying on the device.
// ---------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
    for (int i = 0; i < 10; i++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
// Random.NextSingle(): 24 random bits * 2^-24, in [0,1)
inline float u01(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }

enum Slot : uint32_t { SLOT_CAMERA = 0, SLOT_MEDIUM = 1, SLOT_SCATTER = 2, SLOT_LIGHT = 16, SLOT_LIGHT_RAW = 2048 + 16 };

struct SampleKey {
    uint64_t seed = 0;
    uint32_t pixel = 0;
    uint64_t pass = 0;
    uint32_t depth = 0;
    struct F4 { float x, y, z, w; };
    F4 draw(uint32_t slot, uint32_t sub) const {
        U4 c{pixel, (uint32_t)pass, (depth << 16) | slot, sub};
        U4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(pass >> 32));
        return {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
    }
};
// Core/RandUtils.cs:29 RandomPlusMinusOne, :33 RandomInUnitCube, :37 RandomOnUnitSphere (biased, on purpose)
inline float PlusMinusOne(float u) { return (u * 2) - 1; }
inline V3 OnUnitSphere(float a, float b, float c) { return Normalize(V3{PlusMinusOne(a), PlusMinusOne(b), PlusMinusOne(c)}); }

// ---------------------------------------------------------------------------------------------
// Core/Acceleration/AxisAlignedBoundingBox.cs
// ---------------------------------------------------------------------------------------------
struct Aabb {
    V3 Min, Max;
    // :27-96 (variant 3, the live code)
    bool Hit(const Ray& ray, float kMin, float kMax) const {
        V3 invD = v3(1.0f) / ray.d;
        {
            float t0 = (Min.x - ray.o.x) * invD.x, t1 = (Max.x - ray.o.x) * invD.x;
            if (invD.x < 0.0f) std::swap(t0, t1);
            kMin = t0 > kMin ? t0 : kMin;
            kMax = t1 < kMax ? t1 : kMax;
            if (kMax <= kMin) return false;
        }
        {
            float t0 = (Min.y - ray.o.y) * invD.y, t1 = (Max.y - ray.o.y) * invD.y;
            if (invD.y < 0.0f) std::swap(t0, t1);
            kMin = t0 > kMin ? t0 : kMin;
            kMax = t1 < kMax ? t1 : kMax;
            if (kMax <= kMin) return false;
        }
        {
            float t0 = (Min.z - ray.o.z) * invD.z, t1 = (Max.z - ray.o.z) * invD.z;
            if (invD.z < 0.0f) std::swap(t0, t1);
            kMin = t0 > kMin ? t0 : kMin;
            kMax = t1 < kMax ? t1 : kMax;
            if (kMax <= kMin) return false;
        }
        return true;
    }
    static Aabb Encompass(const Aabb& a, const Aabb& b) {                 // :123-132
        V3 mn = v3(INF), mx = v3(-INF);
        mn = Min3(mn, a.Min); mx = Max3(mx, a.Max);
        mn = Min3(mn, b.Min); mx = Max3(mx, b.Max);
        return {mn, mx};
    }
};

// ---------------------------------------------------------------------------------------------
// Core/HitRecord.cs:19-20
// ---------------------------------------------------------------------------------------------
struct Hittable;
struct HitRecord {
    Ray IncomingRay;
    V3 WorldPoint, LocalPoint, Normal;
    float K = 0;
    bool OutsideFace = false;
    float U = 0, V = 0;
    const Hittable* hittable = nullptr;
    uint32_t Material = 0;          // material id (identity)
    float ShaderData = 0;           // only ConstantDensityMedium sets it (a boxed float)
    bool valid = false;
};

// Core/Base Type Definitions/Hittable.cs
struct Hittable {
    uint32_t material = 0;
    Aabb BoundingVolume{};
    virtual ~Hittable() {}
    virtual HitRecord TryHit(const Ray& ray, float kMin, float kMax, const SampleKey& key, uint32_t self) const = 0;
    virtual bool FastTryHit(const Ray& ray, float kMin, float kMax) const = 0;
    virtual bool IsMedium() const { return false; }
};

inline HitRecord Miss() { return HitRecord{}; }
inline HitRecord MakeHit(const Ray& ray, V3 wp, V3 lp, V3 n, float k, bool outside, float u, float v, const Hittable* h, float shader = 0) {
    HitRecord r; r.IncomingRay = ray; r.WorldPoint = wp; r.LocalPoint = lp; r.Normal = n; r.K = k; r.OutsideFace = outside;
    r.U = u; r.V = v; r.hittable = h; r.Material = h->material; r.ShaderData = shader; r.valid = true; return r;
}

// Impl/Hittables/Sphere.cs
struct Sphere : Hittable {
    V3 Centre; float Radius; float radiusSqr;
    Sphere(const rt_sphere& s) : Centre(v3(s.centre)), Radius(s.radius), radiusSqr(s.radius * s.radius) {}    // :17-22
    static void GetSphereUV(V3 p, float& u, float& v) {                                                        // :78-93
        float theta = std::acos(-p.y);
        float phi = std::atan2(-p.z, p.x) + PI_F;
        u = phi / (2 * PI_F);
        v = theta / PI_F;
    }
    bool Root(const Ray& ray, float kMin, float kMax, float& root) const {                                     // :35-57 == :99-121
        V3 oc = ray.o - Centre;
        float a = LengthSquared(ray.d);
        float halfB = Dot(oc, ray.d);
        float c = LengthSquared(oc) - radiusSqr;
        float discriminant = (halfB * halfB) - (a * c);
        if (discriminant < 0) return false;
        float sqrtD = std::sqrt(discriminant);
        root = (-halfB - sqrtD) / a;
        if ((root < kMin) || (kMax < root) || std::isnan(root)) {
            root = (-halfB + sqrtD) / a;
            if ((root < kMin) || (kMax < root) || std::isnan(root)) return false;
        }
        return true;
    }
    HitRecord TryHit(const Ray& ray, float kMin, float kMax, const SampleKey&, uint32_t) const override {     // :35-71
        float k;
        if (!Root(ray, kMin, kMax, k)) return Miss();
        V3 worldPoint = ray.PointAt(k);
        V3 localPoint = worldPoint - Centre;
        V3 outwardNormal = Normalize(localPoint);
        bool inside = Dot(ray.d, outwardNormal) > 0.0f;
        V3 normal = inside ? -outwardNormal : outwardNormal;
        float u, v; GetSphereUV(outwardNormal, u, v);
        return MakeHit(ray, worldPoint, localPoint, normal, k, !inside, u, v, this);
    }
    bool FastTryHit(const Ray& ray, float kMin, float kMax) const override { float k; return Root(ray, kMin, kMax, k); }  // :99-124
};

// Impl/Hittables/Quad.cs  (and the NEW Triangle, same construction with a barycentric inside test)
struct Quad : Hittable {
    V3 Origin, Normal; float c1[4], c2[4]; bool triangle;
    Quad(const float* origin, const float* normal, const float* ic1, const float* ic2, bool tri) : Origin(v3(origin)), Normal(v3(normal)), triangle(tri) {
        memcpy(c1, ic1, 16); memcpy(c2, ic2, 16);
    }
    bool Solve(const Ray& ray, float kMin, float kMax, float& t, V3& worldPoint, V3& localPoint, float& u, float& v, float& normDotDir) const { // :82-115
        normDotDir = Dot(Normal, ray.d);
        if (std::fabs(normDotDir) <= 0.001f) return false;
        t = -Dot(Normal, ray.o - Origin) / normDotDir;
        if ((t < kMin) || (t > kMax)) return false;
        worldPoint = ray.PointAt(t);
        localPoint = worldPoint - Origin;
        u = (localPoint.x * c1[0]) + (localPoint.y * c1[1]) + (localPoint.z * c1[2]) + c1[3];
        v = (localPoint.x * c2[0]) + (localPoint.y * c2[1]) + (localPoint.z * c2[2]) + c2[3];
        if (triangle) {
            // NEW: u >= 0, v >= 0, u + v <= 1 (NaN fails)
            if (!(u >= 0) || !(v >= 0) || !(u + v <= 1)) return false;
        } else {
            if (u < 0 || u > 1 || std::isnan(u) || v < 0 || v > 1 || std::isnan(v)) return false;
        }
        return true;
    }
    HitRecord TryHit(const Ray& ray, float kMin, float kMax, const SampleKey&, uint32_t) const override {     // :82-120
        float t, u, v, nd; V3 wp, lp;
        if (!Solve(ray, kMin, kMax, t, wp, lp, u, v, nd)) return Miss();
        return MakeHit(ray, wp, lp, Normal, t, nd < 0, u, v, this);
    }
    bool FastTryHit(const Ray& ray, float kMin, float kMax) const override {                                   // :122-151
        float t, u, v, nd; V3 wp, lp; return Solve(ray, kMin, kMax, t, wp, lp, u, v, nd);
    }
};

// Impl/Hittables/InfinitePlane.cs
struct InfinitePlane : Hittable {
    V3 Point, Normal; float negPointDotNormal;
    InfinitePlane(const rt_infinite_plane& p) : Point(v3(p.point)), Normal(v3(p.normal)), negPointDotNormal(p.neg_point_dot_normal) {}
    HitRecord TryHit(const Ray& ray, float kMin, float kMax, const SampleKey&, uint32_t) const override {     // :37-61
        float normDotDir = Dot(ray.d, Normal);
        if (std::fabs(normDotDir) <= 0.001f) return Miss();
        float t = -(Dot(ray.o, Normal) + negPointDotNormal) / normDotDir;
        if ((t < kMin) || (t > kMax)) return Miss();
        V3 worldPoint = ray.PointAt(t);
        V3 localPoint = worldPoint - Point;
        return MakeHit(ray, worldPoint, localPoint, Normal, t, normDotDir < 0, 0, 0, this);
    }
    bool FastTryHit(const Ray& ray, float kMin, float kMax) const override {                                   // :63-82
        float normDotDir = Dot(ray.d, Normal);
        if (std::fabs(normDotDir) <= 0.001f) return false;
        float t = -(Dot(ray.o, Normal) + negPointDotNormal) / normDotDir;
        return (t >= kMin) && (t <= kMax);
    }
};

// Impl/Hittables/XYPlane.cs, XZPlane.cs, YZPlane.cs  (axis: 0 = XY, 1 = XZ, 2 = YZ)
struct AxisPlane : Hittable {
    int axis; float aLo, aHi, bLo, bHi, C;
    AxisPlane(const rt_axis_plane& p, int ax) : axis(ax), aLo(p.a_lo), aHi(p.a_hi), bLo(p.b_lo), bHi(p.b_hi), C(p.c) {}
    HitRecord TryHit(const Ray& ray, float kMin, float kMax, const SampleKey&, uint32_t) const override {
        float oC = axis == 0 ? ray.o.z : axis == 1 ? ray.o.y : ray.o.x;
        float dC = axis == 0 ? ray.d.z : axis == 1 ? ray.d.y : ray.d.x;
        float k = (C - oC) / dC;                                              // XYPlane.cs:56
        if ((dC == 0.0f) || std::isnan(k)) return Miss();                     // :59
        if ((k < kMin) || (k > kMax)) return Miss();                          // :60
        V3 wp = ray.PointAt(k);
        float a = axis == 2 ? wp.y : wp.x;                                    // XY: x,y   XZ: x,z   YZ: y,z
        float b = axis == 0 ? wp.y : wp.z;
        if ((a < aLo) || (a > aHi) || (b < bLo) || (b > bHi)) return Miss();  // :66
        V3 centre = axis == 0 ? V3{(aLo + aHi) / 2.0f, (bLo + bHi) / 2.0f, C}  // XYPlane.cs:28
                  : axis == 1 ? V3{(aLo + aHi) / 2.0f, C, (bLo + bHi) / 2.0f}  // XZPlane.cs:28
                              : V3{C, (aLo + aHi) / 2.0f, (bLo + bHi) / 2.0f}; // YZPlane.cs:65
        V3 lp = wp - centre;
        float u = (a - aLo) / (aHi - aLo), v = (b - bLo) / (bHi - bLo);       // :72
        V3 unit = axis == 0 ? V3{0, 0, 1} : axis == 1 ? V3{0, 1, 0} : V3{1, 0, 0};
        V3 n = oC < C ? -unit : unit;                                         // :98
        return MakeHit(ray, wp, lp, n, k, true, u, v, this);
    }
    bool FastTryHit(const Ray& ray, float kMin, float kMax) const override {
        float oC = axis == 0 ? ray.o.z : axis == 1 ? ray.o.y : ray.o.x;
        float dC = axis == 0 ? ray.d.z : axis == 1 ? ray.d.y : ray.d.x;
        float k = (C - oC) / dC;
        if ((dC == 0.0f) || std::isnan(k)) return false;
        if ((k < kMin) || (k > kMax)) return false;
        V3 wp = ray.PointAt(k);
        float a = axis == 2 ? wp.y : wp.x;
        float b = axis == 0 ? wp.y : wp.z;
        if (axis == 2) return (a >= aLo) && (a <= aHi) && (b >= bLo) && (b <= bHi);   // YZPlane.cs:89 (NaN -> false)
        return !(a < aLo) && !(a > aHi) && !(b < bLo) && !(b > bHi);                  // XYPlane.cs:118-121, XZPlane.cs:91
    }
};

// Impl/Hittables/Disk.cs
struct Disk : Hittable {
    V3 Centre, Normal; float radiusSqr, negCentreDotNorm; float c1[4], c2[4];
    Disk(const rt_disk& d) : Centre(v3(d.centre)), Normal(v3(d.normal)), radiusSqr(d.radius * d.radius), negCentreDotNorm(d.neg_centre_dot_norm) {
        memcpy(c1, d.inv_col1, 16); memcpy(c2, d.inv_col2, 16);
    }
    HitRecord TryHit(const Ray& ray, float kMin, float kMax, const SampleKey&, uint32_t) const override {     // :58-98
        float normDotDir = Dot(ray.d, Normal);
        if (std::fabs(normDotDir) <= 0.001f) return Miss();
        float t = -(Dot(ray.o, Normal) + negCentreDotNorm) / normDotDir;
        if ((t < kMin) || (t > kMax)) return Miss();
        V3 wp = ray.PointAt(t);
        if (DistanceSquared(Centre, wp) > radiusSqr) return Miss();
        V3 lp = wp - Centre;
        float u = (lp.x * c1[0]) + (lp.y * c1[1]) + (lp.z * c1[2]) + c1[3];
        float v = (lp.x * c2[0]) + (lp.y * c2[1]) + (lp.z * c2[2]) + c2[3];
        if (u < 0 || u > 1 || std::isnan(u) || v < 0 || v > 1 || std::isnan(v)) {
            // Vector2.Clamp(uv, Zero, One) = Min(Max(uv, 0), 1) lane-wise with ?: (NaN stays as the other operand rules give)
            u = (u > 0.0f) ? u : 0.0f; u = (u < 1.0f) ? u : 1.0f;
            v = (v > 0.0f) ? v : 0.0f; v = (v < 1.0f) ? v : 1.0f;
        }
        return MakeHit(ray, wp, lp, Normal, t, normDotDir < 0, u, v, this);
    }
    bool FastTryHit(const Ray& ray, float kMin, float kMax) const override {                                   // :101-123
        float normDotDir = Dot(ray.d, Normal);
        if (std::fabs(normDotDir) <= 0.001f) return false;
        float t = -(Dot(ray.o, Normal) + negCentreDotNorm) / normDotDir;
        if ((t < kMin) || (t > kMax)) return false;
        V3 wp = ray.PointAt(t);
        return DistanceSquared(Centre, wp) <= radiusSqr;
    }
};

// Impl/Hittables/Box.cs
struct Box : Hittable {
    float w2b[12]; V3 row1, row2, row3;
    Box(const rt_box& b) { memcpy(w2b, b.world_to_box, 48); row1 = v3(b.box_to_world_rows); row2 = v3(b.box_to_world_rows + 3); row3 = v3(b.box_to_world_rows + 6); }
    V3 TransformNormal(V3 n) const { return {(n.x * w2b[0]) + (n.y * w2b[3]) + (n.z * w2b[6]), (n.x * w2b[1]) + (n.y * w2b[4]) + (n.z * w2b[7]), (n.x * w2b[2]) + (n.y * w2b[5]) + (n.z * w2b[8])}; }
    V3 Transform(V3 p) const { return {(p.x * w2b[0]) + (p.y * w2b[3]) + (p.z * w2b[6]) + w2b[9], (p.x * w2b[1]) + (p.y * w2b[4]) + (p.z * w2b[7]) + w2b[10], (p.x * w2b[2]) + (p.y * w2b[5]) + (p.z * w2b[8]) + w2b[11]}; }
    // Vector4.Transform(new Vector4(v, w), M).xyz — Box.cs:191-194
    V3 Transform4(V3 p, float w) const { return {(p.x * w2b[0]) + (p.y * w2b[3]) + (p.z * w2b[6]) + (w * w2b[9]), (p.x * w2b[1]) + (p.y * w2b[4]) + (p.z * w2b[7]) + (w * w2b[10]), (p.x * w2b[2]) + (p.y * w2b[5]) + (p.z * w2b[8]) + (w * w2b[11])}; }
    static bool Core(V3 ro, V3 rd, float kMin, float kMax, float& k, V3& t1, V3& s) {                          // :126-145
        V3 m = v3(1.0f) / rd;
        if (std::isnan(m.x) || std::isnan(m.y) || std::isnan(m.z) || std::isinf(m.x) || std::isinf(m.y) || std::isinf(m.z)) return false;
        s = {rd.x < 0.0f ? 1.0f : -1.0f, rd.y < 0.0f ? 1.0f : -1.0f, rd.z < 0.0f ? 1.0f : -1.0f};
        t1 = m * (-ro + (s / 2.0f));
        V3 t2 = m * (-ro - (s / 2.0f));
        float kNear = MaxF(MaxF(t1.x, t1.y), t1.z);
        float kFar = MinF(MinF(t2.x, t2.y), t2.z);
        if ((kNear > kFar) || (kFar < 0.0f)) return false;
        k = kNear;
        if ((k < kMin) || (kMax < k)) {
            k = kFar;
            if ((k < kMin) || (kMax < k)) return false;
        }
        if (std::isnan(k)) return false;
        return true;
    }
    HitRecord TryHit(const Ray& ray, float kMin, float kMax, const SampleKey&, uint32_t) const override {     // :96-183
        V3 rd = TransformNormal(ray.d);
        V3 ro = Transform(ray.o);
        float k; V3 t1, s;
        if (!Core(ro, rd, kMin, kMax, k, t1, s)) return Miss();
        V3 normal;
        if ((t1.x > t1.y) && (t1.x > t1.z)) normal = row1 * s.x;
        else if (t1.y > t1.z) normal = row2 * s.y;
        else normal = row3 * s.z;
        V3 wp = ray.PointAt(k);
        V3 lp = Transform(wp);
        return MakeHit(ray, wp, lp, Normalize(normal), k, Dot(ray.d, normal) < 0.0f, 0, 0, this);
    }
    bool FastTryHit(const Ray& ray, float kMin, float kMax) const override {                                   // :186-219
        V3 rd = Transform4(ray.d, 0.0f);
        V3 ro = Transform4(ray.o, 1.0f);
        float k; V3 t1, s;
        return Core(ro, rd, kMin, kMax, k, t1, s);
    }
};

// Impl/Hittables/Capsule.cs
struct Capsule : Hittable {
    V3 P1, P2, centre, ba; float Radius, baba, r2baba, r2; float uvm[9];
    Capsule(const rt_capsule& c) : P1(v3(c.p1)), P2(v3(c.p2)), Radius(c.radius) {                              // :35-55
        centre = Lerp(P1, P2, 0.5f);
        ba = P2 - P1; baba = Dot(ba, ba); r2 = Radius * Radius; r2baba = r2 * baba;
        memcpy(uvm, c.uv_matrix, 36);
    }
    float K(const Ray& ray) const {                                                                            // :77-107
        float k = -1.0f;
        V3 oa = ray.o - P1;
        float bard = Dot(ba, ray.d), baoa = Dot(ba, oa), rdoa = Dot(ray.d, oa), oaoa = Dot(oa, oa);
        float a = baba - (bard * bard);
        float b = (baba * rdoa) - (baoa * bard);
        float c = (baba * oaoa) - (baoa * baoa) - r2baba;
        float h = (b * b) - (a * c);
        if (h >= 0.0) {
            float t = (-b - std::sqrt(h)) / a;
            float y = baoa + (t * bard);
            if ((y > 0.0) && (y < baba)) k = t;
            else {
                V3 oc = y <= 0.0 ? oa : ray.o - P2;
                b = Dot(ray.d, oc);
                c = Dot(oc, oc) - r2;
                h = (b * b) - c;
                if (h > 0.0) k = -b - std::sqrt(h);
            }
        }
        return k;
    }
    HitRecord TryHit(const Ray& ray, float kMin, float kMax, const SampleKey&, uint32_t) const override {     // :71-141
        float k = K(ray);
        if ((k == -1.0f) || (k < kMin) || (k > kMax)) return Miss();
        V3 wp = ray.PointAt(k);
        V3 lp = wp - centre;
        V3 pa = wp - P1;                                                                                        // CapNormal :186-191
        float hh = Dot(pa, ba) / baba; hh = hh < 0.0f ? 0.0f : (hh > 1.0f ? 1.0f : hh);
        V3 outNormal = (pa - (hh * ba)) / Radius;
        V3 q0 = wp - P1;                                                                                        // Transform(q0, uvMatrix): row 4 is zero (:48)
        V3 q = {(q0.x * uvm[0]) + (q0.y * uvm[3]) + (q0.z * uvm[6]) + 0.0f, (q0.x * uvm[1]) + (q0.y * uvm[4]) + (q0.z * uvm[7]) + 0.0f, (q0.x * uvm[2]) + (q0.y * uvm[5]) + (q0.z * uvm[8]) + 0.0f};
        float rawU = std::atan2(q.y, q.x), rawV = q.z;
        float u = (rawU + PI_F) / 6.2831855f;
        float lo = -Radius, hi = Distance(P1, P2) + Radius;
        float v = (rawV - lo) / (hi - lo);                                                                      // MathUtils.InverseLerp, Core/MathUtils.cs:20
        bool inside = Dot(ray.d, outNormal) > 0.0f;
        return MakeHit(ray, wp, lp, outNormal, k, !inside, u, v, this);
    }
    bool FastTryHit(const Ray& ray, float kMin, float kMax) const override {                                   // :145-183
        float k = K(ray);
        return (k != -1.0f) && (k >= kMin) && (k <= kMax);
    }
};

// Impl/Hittables/Cylinder.cs
struct Cylinder : Hittable {
    V3 P1, P2, centre, ba; float Radius, baba, r2baba;
    Cylinder(const rt_cylinder& c) : P1(v3(c.p1)), P2(v3(c.p2)), Radius(c.radius) {
        centre = Lerp(P1, P2, 0.5f); ba = P2 - P1; baba = Dot(ba, ba); r2baba = Radius * Radius * baba;
    }
    HitRecord TryHit(const Ray& ray, float kMin, float kMax, const SampleKey&, uint32_t) const override {     // :55-106
        float k; V3 normal;
        V3 oc = ray.o - P1;
        float bard = Dot(ba, ray.d), baoc = Dot(ba, oc);
        float k2 = baba - (bard * bard);
        float k1 = (baba * Dot(oc, ray.d)) - (baoc * bard);
        float k0 = (baba * Dot(oc, oc)) - (baoc * baoc) - r2baba;
        float h = (k1 * k1) - (k2 * k0);
        if (h < 0.0) return Miss();
        h = std::sqrt(h);
        k = (-k1 - h) / k2;
        float y = baoc + (k * bard);
        if ((y > 0.0) && (y < baba)) {
            normal = ((oc + (k * ray.d)) - ((ba * y) / baba)) / Radius;
        } else {
            k = ((y < 0.0f ? 0.0f : baba) - baoc) / bard;
            if (std::fabs(k1 + (k2 * k)) < h) {
                float sg = y > 0 ? 1.0f : (y < 0 ? -1.0f : 0.0f);      // MathF.Sign (throws on NaN in .NET; NaN -> 0 here)
                normal = (ba * sg) / baba;
            } else return Miss();
        }
        if ((k == -1.0f) || !(k >= kMin) || !(k <= kMax)) return Miss();
        V3 wp = ray.PointAt(k);
        V3 lp = wp - centre;
        bool inside = Dot(ray.d, normal) > 0.0f;
        return MakeHit(ray, wp, lp, normal, k, !inside, 0, 0, this);
    }
    bool FastTryHit(const Ray& ray, float kMin, float kMax) const override {                                   // :109-142
        V3 oc = ray.o - P1;
        float bard = Dot(ba, ray.d), baoc = Dot(ba, oc);
        float k2 = baba - (bard * bard);
        float k1 = (baba * Dot(oc, ray.d)) - (baoc * bard);
        float k0 = (baba * Dot(oc, oc)) - (baoc * baoc) - r2baba;
        float h = (k1 * k1) - (k2 * k0);
        if (h < 0.0) return false;
        h = std::sqrt(h);
        float k = (-k1 - h) / k2;
        float y = baoc + (k * bard);
        if (!(y > 0.0) || !(y < baba)) {
            k = ((y < 0.0f ? 0.0f : baba) - baoc) / bard;
            if (!(std::fabs(k1 + (k2 * k)) < h)) return false;
        }
        return (k != -1.0f) && (k >= kMin) && (k <= kMax);
    }
};

// Impl/Hittables/ConstantDensityMedium.cs
struct Medium : Hittable {
    std::unique_ptr<Hittable> Boundary; float negInvDensity;
    bool IsMedium() const override { return true; }
    HitRecord TryHit(const Ray& ray, float kMin, float kMax, const SampleKey& key, uint32_t self) const override {   // :46-68
        HitRecord entryHit = Boundary->TryHit(ray, kMin, kMax, key, self);
        if (!entryHit.valid) return Miss();
        HitRecord exitHit = Boundary->TryHit(ray, entryHit.K + 0.001f, kMax, key, self);
        if (!exitHit.valid) return Miss();
        float distanceBetweenHits = exitHit.K - entryHit.K;
        SampleKey::F4 r = key.draw(SLOT_MEDIUM, self);
        float distanceToIntersection = negInvDensity * std::log(r.x);
        if (distanceToIntersection > distanceBetweenHits) return Miss();
        V3 wp = ray.PointAt(distanceToIntersection);           // sic: measured from the ray origin (:61)
        V3 normal = OnUnitSphere(r.y, r.z, r.w);
        return MakeHit(ray, wp, wp, normal, entryHit.K + distanceToIntersection, true, 0, 0, this, distanceToIntersection);
    }
    bool FastTryHit(const Ray& ray, float kMin, float kMax) const override { return Boundary->FastTryHit(ray, kMin, kMax); }  // :77
};

// ---------------------------------------------------------------------------------------------
// Core/Acceleration: BvhNode / BinaryBvhNode / SingleObjectBvhNode / EmptyBvhNode / BvhTree
// ---------------------------------------------------------------------------------------------
struct Stats {
    uint64_t aabbMisses = 0, hittableMisses = 0, hittableIntersections = 0, rayCount = 0, shadowRays = 0;
    uint64_t skyRays = 0, bounceLimit = 0, scatter = 0, absorbed = 0, pixels = 0;
    uint64_t normalsWrong = 0, uvInvalid = 0;       // GraphicsValidator.RecordError counts, summed over objects (RenderJob.cs:597, :606)
    uint64_t* perObject = nullptr;                  // ... and per object, [n_objects][2] = (normals, UVs): GraphicsValidator.Errors[type][object] (GraphicsValidator.cs:35-45); this thread's own table, or null
    uint64_t depth[RT_DEPTH_HIST_SIZE] = {};
    void add(const Stats& o) {
        aabbMisses += o.aabbMisses; hittableMisses += o.hittableMisses; hittableIntersections += o.hittableIntersections;
        rayCount += o.rayCount; shadowRays += o.shadowRays; skyRays += o.skyRays; bounceLimit += o.bounceLimit;
        scatter += o.scatter; absorbed += o.absorbed; pixels += o.pixels; normalsWrong += o.normalsWrong; uvInvalid += o.uvInvalid;
        for (int i = 0; i < RT_DEPTH_HIST_SIZE; i++) depth[i] += o.depth[i];
    }
};

struct Found { uint32_t obj = RT_NO_HIT; HitRecord hit; bool ok() const { return obj != RT_NO_HIT; } };

struct SceneObjectRef { uint32_t index; const Hittable* h; };

struct BvhNode {
    Aabb BoundingBox{};
    virtual ~BvhNode() {}
    virtual Found TryHit(const Ray& ray, float kMin, float kMax, const SampleKey& key, Stats& st) const = 0;
    virtual bool FastTryHit(const Ray& ray, float kMin, float kMax) const = 0;
};
struct EmptyBvhNode : BvhNode {                                                // EmptyBvhNode.cs:7-17
    EmptyBvhNode() { BoundingBox = {{0, 0, 0}, {0, 0, 0}}; }
    Found TryHit(const Ray&, float, float, const SampleKey&, Stats&) const override { return {}; }
    bool FastTryHit(const Ray&, float, float) const override { return false; }
};
struct SingleObjectBvhNode : BvhNode {                                         // SingleObjectBvhNode.cs:4-36
    SceneObjectRef so;
    SingleObjectBvhNode(SceneObjectRef s) : so(s) { BoundingBox = s.h->BoundingVolume; }
    Found TryHit(const Ray& ray, float kMin, float kMax, const SampleKey& key, Stats& st) const override {
        if (!BoundingBox.Hit(ray, kMin, kMax)) { st.aabbMisses++; return {}; }
        HitRecord h = so.h->TryHit(ray, kMin, kMax, key, so.index);
        if (h.valid) { st.hittableIntersections++; Found f; f.obj = so.index; f.hit = h; return f; }
        st.hittableMisses++;
        return {};
    }
    bool FastTryHit(const Ray& ray, float kMin, float kMax) const override { return so.h->FastTryHit(ray, kMin, kMax); }
};
struct BinaryBvhNode : BvhNode {                                               // BinaryBvhNode.cs:7-42
    std::unique_ptr<BvhNode> NodeA, NodeB;
    BinaryBvhNode(std::unique_ptr<BvhNode> a, std::unique_ptr<BvhNode> b) : NodeA(std::move(a)), NodeB(std::move(b)) {
        BoundingBox = Aabb::Encompass(NodeA->BoundingBox, NodeB->BoundingBox);
    }
    Found TryHit(const Ray& ray, float kMin, float kMax, const SampleKey& key, Stats& st) const override {
        if (!BoundingBox.Hit(ray, kMin, kMax)) { st.aabbMisses++; return {}; }
        Found hitA = NodeA->TryHit(ray, kMin, kMax, key, st);
        if (hitA.ok()) {
            Found hitB = NodeB->TryHit(ray, kMin, hitA.hit.K, key, st);
            if (hitB.ok()) {
                // D2: reference returns hitB (K_B <= K_A). On an exact tie the larger object index wins.
                if (hitB.hit.K == hitA.hit.K && hitA.obj > hitB.obj) return hitA;
                return hitB;
            }
            return hitA;
        }
        return NodeB->TryHit(ray, kMin, kMax, key, st);
    }
    bool FastTryHit(const Ray& ray, float kMin, float kMax) const override {   // no AABB test, by design of the reference (:41)
        return NodeA->FastTryHit(ray, kMin, kMax) || NodeB->FastTryHit(ray, kMin, kMax);
    }
};

// BvhTree.cs:37-136.  Array.Sort with the inconsistent comparer of :181-186 is implementation-defined
// in .NET (introsort); a stable merge sort driven by the same comparer is used here.
static float AabbArea(const Aabb& b) { V3 s = b.Max - b.Min; return 2 * ((s.x * s.y) + (s.y * s.z) + (s.z * s.x)); }
static int CompareTo(float a, float b) { if (a < b) return -1; if (a > b) return 1; if (a == b) return 0; if (std::isnan(a)) return std::isnan(b) ? 0 : -1; return 1; }
static int CompareHittables(const Aabb& a, const Aabb& b, int axis) {          // :181-186 (note b.Min + b.Min, sic)
    V3 ca = (a.Min + a.Max) / 2.0f, cb = (b.Min + b.Min) / 2.0f;
    float fa = axis == 0 ? ca.x : axis == 1 ? ca.y : ca.z, fb = axis == 0 ? cb.x : axis == 1 ? cb.y : cb.z;
    return CompareTo(fa, fb);
}
static void MergeSort(std::vector<SceneObjectRef>& v, int axis) {
    if (v.size() < 2) return;
    std::vector<SceneObjectRef> l(v.begin(), v.begin() + v.size() / 2), r(v.begin() + v.size() / 2, v.end());
    MergeSort(l, axis); MergeSort(r, axis);
    size_t i = 0, j = 0, k = 0;
    while (i < l.size() && j < r.size()) v[k++] = (CompareHittables(r[j].h->BoundingVolume, l[i].h->BoundingVolume, axis) < 0) ? r[j++] : l[i++];
    while (i < l.size()) v[k++] = l[i++];
    while (j < r.size()) v[k++] = r[j++];
}
static std::unique_ptr<BvhNode> FromSegment_SAH(const std::vector<SceneObjectRef>& segment, int depth) {
    size_t n = segment.size();
    if (n == 0) return std::make_unique<EmptyBvhNode>();
    if (n == 1) return std::make_unique<SingleObjectBvhNode>(segment[0]);
    if (depth > 4000) {  // guard the C++ stack; the reference would recurse as deep as its list-like trees go
        std::vector<SceneObjectRef> a(segment.begin(), segment.begin() + n / 2), b(segment.begin() + n / 2, segment.end());
        return std::make_unique<BinaryBvhNode>(FromSegment_SAH(a, depth + 1), FromSegment_SAH(b, depth + 1));
    }
    std::vector<Aabb> boxes(n); std::vector<float> leftArea(n), rightArea(n);
    std::vector<SceneObjectRef> objects = segment;
    Aabb mainBox = segment[0].h->BoundingVolume;
    for (size_t i = 1; i < n; i++) mainBox = Aabb::Encompass(segment[i].h->BoundingVolume, mainBox);
    V3 size = mainBox.Max - mainBox.Min;
    float mx = MaxF(MaxF(size.x, size.y), size.z);
    int axis = std::fabs(mx - size.x) < 0.001f ? 0 : std::fabs(mx - size.y) < 0.001f ? 1 : 2;
    MergeSort(objects, axis);
    for (size_t i = 0; i < n; i++) boxes[i] = segment[i].h->BoundingVolume;     // sic: the UNSORTED segment (:84)
    leftArea[0] = AabbArea(boxes[0]);
    Aabb leftBox = boxes[0];
    for (size_t i = 1; i < n - 1; i++) { leftBox = Aabb::Encompass(leftBox, boxes[i]); leftArea[i] = AabbArea(leftBox); }
    rightArea[n - 1] = AabbArea(boxes[n - 1]);
    Aabb rightBox = boxes[n - 1];
    for (size_t i = n - 2; i > 0; i--) { rightBox = Aabb::Encompass(rightBox, boxes[i]); rightArea[i] = AabbArea(rightBox); }
    float minSA = std::numeric_limits<float>::max();
    size_t minSAIndex = 0;
    for (size_t i = 0; i < n - 1; i++) {
        float sah = ((float)i * leftArea[i]) + ((float)(n - i - 1) * rightArea[i + 1]);
        if (sah < minSA) { minSAIndex = i; minSA = sah; }
    }
    std::unique_ptr<BvhNode> left, right;
    if (minSAIndex == 0) left = std::make_unique<SingleObjectBvhNode>(objects[0]);
    else left = FromSegment_SAH(std::vector<SceneObjectRef>(objects.begin(), objects.begin() + minSAIndex + 1), depth + 1);
    if (minSAIndex == n - 2) right = std::make_unique<SingleObjectBvhNode>(objects[minSAIndex + 1]);
    else right = FromSegment_SAH(std::vector<SceneObjectRef>(objects.begin() + minSAIndex + 1, objects.end()), depth + 1);
    return std::make_unique<BinaryBvhNode>(std::move(left), std::move(right));
}

// ---------------------------------------------------------------------------------------------
// Textures, skyboxes
// ---------------------------------------------------------------------------------------------
// PARITY UNPINNED: SharpNoise's gradient table is not in the reference tree.  This is a libnoise-style
// gradient-coherent noise (Fast quality = linear interpolation, persistence .5, lacunarity 3, 5 octaves,
// MarbleTexture.cs:10-16) over a hashed gradient set; the device implements the same function.
static inline uint32_t hash3(int32_t x, int32_t y, int32_t z, int32_t seed) {
    uint32_t h = (uint32_t)x * 1619u + (uint32_t)y * 31337u + (uint32_t)z * 6971u + (uint32_t)seed * 1013u;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
static inline float grad_noise(float fx, float fy, float fz, int32_t ix, int32_t iy, int32_t iz, int32_t seed) {
    uint32_t h = hash3(ix, iy, iz, seed);
    float gx = (float)((int)(h & 1023u) - 512) * (1.0f / 512.0f);
    float gy = (float)((int)((h >> 10) & 1023u) - 512) * (1.0f / 512.0f);
    float gz = (float)((int)((h >> 20) & 1023u) - 512) * (1.0f / 512.0f);
    float dx = fx - (float)ix, dy = fy - (float)iy, dz = fz - (float)iz;
    return ((gx * dx) + (gy * dy) + (gz * dz)) * 1.2f;
}
static inline float lerpf(float a, float b, float t) { return a + (t * (b - a)); }
static float coherent_noise(float x, float y, float z, int32_t seed) {
    float flx = std::floor(x), fly = std::floor(y), flz = std::floor(z);
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
// libnoise's Perlin::GetValue (SharpNoise.Modules.Perlin): octaves of coherent noise, seed + octave each
static float perlin_noise(float x, float y, float z, float frequency, float lacunarity, float persistence, int octaves, int32_t seed) {
    float value = 0.0f, amplitude = 1.0f;
    x = x * frequency; y = y * frequency; z = z * frequency;
    for (int o = 0; o < octaves; o++) {
        value = value + (coherent_noise(x, y, z, seed + o) * amplitude);
        x = x * lacunarity; y = y * lacunarity; z = z * lacunarity;
        amplitude = amplitude * persistence;
    }
    return value;
}
static float marble_noise(float x, float y, float z) { return perlin_noise(x, y, z, 1.0f, 3.0f, 0.5f, 5, 0); }      // MarbleTexture.cs:11-17


// ---------------------------------------------------------------------------------------------
// CullTree — NOT in the reference.  RenderJob.AnyIntersectionFast (Core/RenderJob.cs:575) asks every object's FastTryHit
// (BinaryBvhNode.cs:41 has no box test), which costs O(objects) per shadow ray: 5 M triangle tests per ray on config 4.  To check
// the device's any-hit answers on >= 100 k rays of the full-size scenes the checker needs the same boolean faster.  This is a plain
// median-split box tree over the objects' BoundingVolumes whose box test is CLOSED and whose boxes are GROWN (see MayHit) three
// times as far as the device grows its own, i.e. strictly more conservative than the device's; an object is still decided by
// its own FastTryHit.  tests/ prove it equal to the list scan (every small scene, 100 k rays each, and a sample of rays
// on the full-size scenes) before they use it as the reference for the device.
// ---------------------------------------------------------------------------------------------
struct CullTree {
    struct Node { Aabb box; uint32_t left, right, first, count; };       // count > 0: leaf over order[first .. first+count)
    std::vector<Node> nodes;
    std::vector<SceneObjectRef> order;
    std::vector<SceneObjectRef> always;                                   // objects with a non-finite box (InfinitePlane.cs:28) and media
    static bool Finite(const Aabb& b) {
        return std::isfinite(b.Min.x) && std::isfinite(b.Min.y) && std::isfinite(b.Min.z) && std::isfinite(b.Max.x) && std::isfinite(b.Max.y) && std::isfinite(b.Max.z);
    }
    uint32_t BuildRange(uint32_t lo, uint32_t hi) {
        Node n{}; n.box = order[lo].h->BoundingVolume;
        V3 cmin = (n.box.Min + n.box.Max) / 2.0f, cmax = cmin;
        for (uint32_t i = lo + 1; i < hi; i++) {
            const Aabb& b = order[i].h->BoundingVolume;
            n.box = Aabb::Encompass(n.box, b);
            V3 c = (b.Min + b.Max) / 2.0f; cmin = Min3(cmin, c); cmax = Max3(cmax, c);
        }
        const uint32_t id = (uint32_t)nodes.size();
        nodes.push_back(n);
        if (hi - lo <= 4) { nodes[id].first = lo; nodes[id].count = hi - lo; return id; }
        V3 e = cmax - cmin;
        const int axis = (e.x >= e.y && e.x >= e.z) ? 0 : (e.y >= e.z ? 1 : 2);
        const uint32_t mid = (lo + hi) / 2;
        auto key = [axis](const SceneObjectRef& r) { const Aabb& b = r.h->BoundingVolume; return axis == 0 ? b.Min.x + b.Max.x : axis == 1 ? b.Min.y + b.Max.y : b.Min.z + b.Max.z; };
        std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](const SceneObjectRef& a, const SceneObjectRef& b) { return key(a) < key(b); });
        const uint32_t l = BuildRange(lo, mid), r = BuildRange(mid, hi);
        nodes[id].left = l; nodes[id].right = r;
        return id;
    }
    V3 centre{0, 0, 0}; float radius = 0, maxObjectRadius = 0;            // sphere around every bounded object; largest half-diagonal of one object's box
    void Build(const std::vector<SceneObjectRef>& solids, const std::vector<SceneObjectRef>& media) {
        for (const SceneObjectRef& s : solids) (Finite(s.h->BoundingVolume) ? order : always).push_back(s);
        for (const SceneObjectRef& m : media) always.push_back(m);
        nodes.reserve(order.size());
        if (order.empty()) return;
        BuildRange(0, (uint32_t)order.size());
        const Aabb& all = nodes[0].box;
        centre = (all.Min + all.Max) / 2.0f; radius = Length(all.Max - all.Min) * 0.5001f + 1e-6f;
        for (const SceneObjectRef& s : order) maxObjectRadius = std::max(maxObjectRadius, Length(s.h->BoundingVolume.Max - s.h->BoundingVolume.Min) * 0.5001f);
    }
    // A binary32 intersector can accept a point outside its object's box: the sphere test's discriminant halfB^2 - a c carries an
    // absolute error of up to ~1e-6 D^2 at distance D from the ray origin, so a ray passing the sphere at sqrt(r^2 + 1e-6 D^2) may
    // be reported as a hit, ~1e-3 D outside the box (seen on shadow rays from ground points 2 000 units out in the Demo scene).
    // Every box is therefore grown by pad = 3e-3 * D + 1e-4 (three times the device's), D = the farthest an object the ray can
    // report may be: its range, capped by the far side of the scene, plus twice the largest object.
    static bool MayHit(const Aabb& b, const Ray& ray, V3 invD, float kMin, float kMax, float pad) {
        float lo = kMin, hi = kMax;
        const float mn[3] = {b.Min.x - pad, b.Min.y - pad, b.Min.z - pad}, mx[3] = {b.Max.x + pad, b.Max.y + pad, b.Max.z + pad};
        const float o[3] = {ray.o.x, ray.o.y, ray.o.z}, id[3] = {invD.x, invD.y, invD.z};
        for (int a = 0; a < 3; a++) {
            float t0 = (mn[a] - o[a]) * id[a], t1 = (mx[a] - o[a]) * id[a];
            if (std::isnan(t0) || std::isnan(t1)) continue;               // 0 * inf: this axis cannot decide, so it does not cull
            if (t0 > t1) std::swap(t0, t1);
            if (t0 > lo) lo = t0;
            if (t1 < hi) hi = t1;
        }
        if (std::isnan(lo) || std::isnan(hi)) return true;
        const float slack = 1e-5f * (std::fabs(lo) + std::fabs(hi)) + 1e-6f;
        return !(hi + slack < lo - slack);
    }
    bool Any(const Ray& ray, float kMin, float kMax) const {
        for (const SceneObjectRef& s : always) if (s.h->FastTryHit(ray, kMin, kMax)) return true;
        if (nodes.empty()) return false;
        const V3 invD = v3(1.0f) / ray.d;
        const float farSide = Distance(ray.o, centre) + radius;
        const float reach = ((kMax < farSide) ? kMax : farSide) + (2.0f * maxObjectRadius);
#ifndef ORC_CULL_PAD
#define ORC_CULL_PAD 3e-3f
#endif
        const float pad = (ORC_CULL_PAD * reach) + 1e-4f;
        uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
        while (sp) {
            const Node& n = nodes[stack[--sp]];
            if (!MayHit(n.box, ray, invD, kMin, kMax, pad)) continue;
            if (n.count) { for (uint32_t i = 0; i < n.count; i++) if (order[n.first + i].h->FastTryHit(ray, kMin, kMax)) return true; }
            else if (sp + 2 <= 128) { stack[sp++] = n.left; stack[sp++] = n.right; }
            else {      // a tree deeper than the stack (a median split keeps it near log2 n, so this is not expected): never skip a subtree — ask every object, as the reference does
                for (const SceneObjectRef& o : order) if (o.h->FastTryHit(ray, kMin, kMax)) return true;
                return false;
            }
        }
        return false;
    }
};

struct SceneData;
static Colour TextureColour(const SceneData& sc, uint32_t tex, const HitRecord& hit);

// ---------------------------------------------------------------------------------------------
// The scene + RenderJob state
// ---------------------------------------------------------------------------------------------
struct SceneData {
    rt_camera camera{};
    rt_skybox skybox{};
    std::vector<std::unique_ptr<Hittable>> objects;      // Scene.SceneObjects[i].Hittable
    std::vector<std::unique_ptr<Hittable>> lightShapes;  // DiffuseShapedLight.Shape per light (may be null)
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_light> lights;
    std::unique_ptr<BvhNode> root;                        // over non-medium objects (D3)
    std::vector<SceneObjectRef> media;
    CullTree cull;                                        // checker-only acceleration of AnyIntersectionFast (see CullTree)
    bool culledAny = false;                               // orc_scene_set_culled_any: shadow rays of orc_render go through `cull`
};

static std::unique_ptr<Hittable> MakeHittable(const rt_scene_desc& d, uint32_t type, uint32_t index, std::string& err) {
    auto bad = [&](const char* what) { err = std::string("hittable index out of range: ") + what; return std::unique_ptr<Hittable>(); };
    switch (type) {
        case RT_HIT_SPHERE: if (index >= d.n_spheres) return bad("sphere"); return std::make_unique<Sphere>(d.spheres[index]);
        case RT_HIT_QUAD: if (index >= d.n_quads) return bad("quad"); { const rt_quad& q = d.quads[index]; return std::make_unique<Quad>(q.origin, q.normal, q.inv_col1, q.inv_col2, false); }
        case RT_HIT_TRIANGLE: if (index >= d.n_triangles) return bad("triangle"); { const rt_triangle& q = d.triangles[index]; return std::make_unique<Quad>(q.v0, q.normal, q.inv_col1, q.inv_col2, true); }
        case RT_HIT_INFINITE_PLANE: if (index >= d.n_infinite_planes) return bad("plane"); return std::make_unique<InfinitePlane>(d.infinite_planes[index]);
        case RT_HIT_XY_PLANE: if (index >= d.n_xy_planes) return bad("xy"); return std::make_unique<AxisPlane>(d.xy_planes[index], 0);
        case RT_HIT_XZ_PLANE: if (index >= d.n_xz_planes) return bad("xz"); return std::make_unique<AxisPlane>(d.xz_planes[index], 1);
        case RT_HIT_YZ_PLANE: if (index >= d.n_yz_planes) return bad("yz"); return std::make_unique<AxisPlane>(d.yz_planes[index], 2);
        case RT_HIT_DISK: if (index >= d.n_disks) return bad("disk"); return std::make_unique<Disk>(d.disks[index]);
        case RT_HIT_BOX: if (index >= d.n_boxes) return bad("box"); return std::make_unique<Box>(d.boxes[index]);
        case RT_HIT_CAPSULE: if (index >= d.n_capsules) return bad("capsule"); return std::make_unique<Capsule>(d.capsules[index]);
        case RT_HIT_CYLINDER: if (index >= d.n_cylinders) return bad("cylinder"); return std::make_unique<Cylinder>(d.cylinders[index]);
        case RT_HIT_MEDIUM: {
            if (index >= d.n_media) return bad("medium");
            const rt_medium& m = d.media[index];
            if (m.boundary_type == RT_HIT_MEDIUM) { err = "medium boundary cannot be a medium"; return nullptr; }  // ConstantDensityMedium.cs:25
            auto md = std::make_unique<Medium>();
            md->Boundary = MakeHittable(d, m.boundary_type, m.boundary_index, err);
            if (!md->Boundary) return nullptr;
            md->negInvDensity = m.neg_inv_density;
            return md;
        }
    }
    err = "unknown hittable type";
    return nullptr;
}

// RenderJob.TryFindClosestHit, Core/RenderJob.cs:581-617 (+ D3, D4)
static Found TryFindClosestHit(const SceneData& sc, const Ray& ray, float kMin, float kMax, const SampleKey& key, Stats& st) {
    st.rayCount++;
    if (std::isnan(ray.o.x) || std::isnan(ray.o.y) || std::isnan(ray.o.z) || std::isnan(ray.d.x) || std::isnan(ray.d.y) || std::isnan(ray.d.z)) return {};  // D4
    Found best = sc.root->TryHit(ray, kMin, kMax, key, st);
    for (const SceneObjectRef& m : sc.media) {                                                       // D3
        float km = best.ok() ? best.hit.K : kMax;
        if (!m.h->BoundingVolume.Hit(ray, kMin, km)) { st.aabbMisses++; continue; }
        HitRecord h = m.h->TryHit(ray, kMin, km, key, m.index);
        if (!h.valid) { st.hittableMisses++; continue; }
        st.hittableIntersections++;
        if (best.ok() && h.K == best.hit.K && best.obj > m.index) continue;                           // D2
        best.obj = m.index; best.hit = h;
    }
    if (!best.ok()) return {};
    // the two warn-only rules: CheckVectorNormalized (GraphicsValidator.cs:52-60), CheckUVCoordValid (GraphicsValidator.cs:62-66); RenderJob.cs:588-607 logs, counts, changes nothing
    if (!(!(std::fabs(LengthSquared(best.hit.Normal) - 1.0f) > 0.01f))) { st.normalsWrong++; if (st.perObject) st.perObject[2 * (size_t)best.obj]++; }
    if (!((best.hit.U >= 0.0f) && (best.hit.U <= 1.0f) && (best.hit.V >= 0.0f) && (best.hit.V <= 1.0f))) { st.uvInvalid++; if (st.perObject) st.perObject[2 * (size_t)best.obj + 1]++; }
    // GraphicsValidator.CheckValueRange, Core/Debugging/GraphicsValidator.cs:69 ; RenderJob.cs:609-614
    if (!((best.hit.K >= kMin) && (best.hit.K <= kMax))) return {};
    return best;
}

// RenderJob.AnyIntersectionFast, Core/RenderJob.cs:575 — a DFS over every leaf, no AABB culling.
static bool ListAny(const SceneData& sc, const Ray& ray, float kMin, float kMax) {
    if (sc.root->FastTryHit(ray, kMin, kMax)) return true;
    for (const SceneObjectRef& m : sc.media) if (m.h->FastTryHit(ray, kMin, kMax)) return true;
    return false;
}
static bool AnyIntersectionFast(const SceneData& sc, const Ray& ray, float kMin, float kMax, Stats& st) {
    st.shadowRays++;
    if (sc.culledAny) return sc.cull.Any(ray, kMin, kMax);                                              // checker-only shortcut, proven equal by tests/
    return ListAny(sc, ray, kMin, kMax);
}

static Colour TextureColour(const SceneData& sc, uint32_t tex, const HitRecord& hit) {
    const rt_texture& t = sc.textures[tex];
    if (t.type == RT_TEX_SOLID) return col(t.f);                                                      // SolidColourTexture.cs:14
    if (t.type == RT_TEX_GREYSCALE_PERLIN) {                                                          // GreyscaleNoiseTexture.cs:20 (PARITY UNPINNED, as marble)
        float v = perlin_noise(hit.WorldPoint.x, hit.WorldPoint.y, hit.WorldPoint.z, t.f[0], t.f[1], t.f[2], (int)t.f[3], (int32_t)t.f[4]);
        return Colour{v, v, v};
    }
    // MarbleTexture.cs:38-48
    float Scale = t.f[0], NoiseScale = t.f[1], NoiseStrength = t.f[2], DropoffPower = t.f[3];
    float x = hit.WorldPoint.x / Scale, y = hit.WorldPoint.y / Scale, z = hit.WorldPoint.z / Scale;
    float tt = x + y + z;
    tt += marble_noise(x / NoiseScale, y / NoiseScale, z / NoiseScale) * NoiseStrength;
    float val = std::sin(tt);
    val = (0.5f * val) + 0.5f;
    val = std::pow(val, DropoffPower);
    return ColourLerp(Colour{0, 0, 0}, Colour{1, 1, 0.95f}, val);
}

static Colour SkyColour(const SceneData& sc, const Ray& ray) {
    if (sc.skybox.type == RT_SKY_SINGLE_COLOUR) return col(sc.skybox.colour);                          // SingleColourSkyBox.cs:16
    float t = 0.5f * (ray.d.y + 1);                                                                    // DefaultSkyBox.cs:12-13
    return Colour{(1 - t) + (0.5f * t), (1 - t) + (0.7f * t), (1 - t) + (1.0f * t)};
}

// ---------------------------------------------------------------------------------------------
// Lights: Impl/Lights/SimpleLightBase.cs, PointLight.cs, DiffuseSphereLight.cs, DiffuseShapedLight.cs
// ---------------------------------------------------------------------------------------------
struct ShadowRay { Ray ray; float kMin, kMax; };
static ShadowRay DefaultGetShadowRayForHit(V3 hitPos, V3 lightPos) {                                   // SimpleLightBase.cs:68-77
    Ray r = Ray::FromPoints(hitPos, lightPos);
    const float kMin = 0.01f;
    float kMax = Distance(hitPos, lightPos) - kMin;
    return {r, kMin, kMax};
}
static float Attenuation(const rt_light& l, float x) {                                                 // the factories, SimpleLightBase.cs:83-129
    switch (l.atten_kind) {
        case RT_ATTEN_CONSTANT: return l.atten_p[0];
        case RT_ATTEN_LINEAR: return 1 - x;
        case RT_ATTEN_POWER: return std::pow(1 - x, l.atten_p[0]);
        case RT_ATTEN_POWER_STAY_HIGH: return 1 - std::pow(x, l.atten_p[0]);
        case RT_ATTEN_LOGISTIC: return 1.0f / (1 + std::pow(E_F, l.atten_p[1] * (x - l.atten_p[0])));
        case RT_ATTEN_LOGISTIC_LKX: return l.atten_p[0] / (1.0f + std::pow(E_F, -l.atten_p[1] * (x - l.atten_p[2])));
        default: return std::pow(E_F, l.atten_p[0] * -x);
    }
}
static ShadowRay GetShadowRayForHit(const SceneData& sc, uint32_t li, const HitRecord& hit, const SampleKey& key, uint32_t slotBase, uint32_t sub) {
    const rt_light& l = sc.lights[li];
    V3 Position = v3(l.position);
    if (l.type == RT_LIGHT_POINT) return DefaultGetShadowRayForHit(hit.WorldPoint, Position);          // PointLight.cs:9
    if (l.type == RT_LIGHT_DIFFUSE_SPHERE) {                                                           // DiffuseSphereLight.cs:88-142
        float DiffusionRadius = l.diffusion_radius;
        if (Distance(hit.WorldPoint, Position) <= DiffusionRadius * 1.01f) {
            for (int i = 0; i < 1000; i++) {
                SampleKey::F4 r = key.draw(slotBase + (uint32_t)i, sub);
                V3 randDir = OnUnitSphere(r.x, r.y, r.z);
                if (Dot(randDir, hit.Normal) < 0) randDir = -randDir;
                V3 oc = hit.WorldPoint - Position;
                float a = LengthSquared(randDir);
                float halfB = Dot(oc, randDir);
                float c = LengthSquared(oc) - (DiffusionRadius * DiffusionRadius);
                float discriminant = (halfB * halfB) - (a * c);
                if (discriminant < 0) continue;
                float sqrtD = std::sqrt(discriminant);
                float k = (-halfB - sqrtD) / a;
                if (std::isnan(k)) { k = (-halfB + sqrtD) / a; if (std::isnan(k)) continue; }
                return {Ray(hit.WorldPoint, randDir), 0.001f, k};
            }
            return DefaultGetShadowRayForHit(hit.WorldPoint, Position);                                // :138
        }
        SampleKey::F4 r = key.draw(slotBase, sub);
        V3 randDir = OnUnitSphere(r.x, r.y, r.z);
        if (Dot(randDir, Normalize(hit.WorldPoint - Position)) < 0) randDir = -randDir;
        V3 randPos = Position + (randDir * DiffusionRadius);
        return DefaultGetShadowRayForHit(hit.WorldPoint, randPos);
    }
    // DiffuseShapedLight.cs:69-105
    const Hittable* shape = sc.lightShapes[li].get();
    V3 mn = v3(l.shape_aabb_min), mx = v3(l.shape_aabb_max);
    for (int i = 0; i < 1000; i++) {
        SampleKey::F4 r = key.draw(slotBase + (uint32_t)i, sub);
        V3 p1 = hit.WorldPoint;
        V3 p2 = {lerpf(mn.x, mx.x, r.x), lerpf(mn.y, mx.y, r.y), lerpf(mn.z, mx.z, r.z)};
        Ray ry = Ray::FromPoints(p1, p2);
        HitRecord sh = shape->TryHit(ry, 0.0001f, INF, key, RT_NO_HIT);
        if (sh.valid) return {ry, 0.001f, sh.K - 0.001f};
    }
    return {Ray(hit.WorldPoint, v3(1.0f)), 0.0f, INF};
}
// SimpleLightBase.CalculateLight, SimpleLightBase.cs:48-66
static Colour CalculateLight(const SceneData& sc, uint32_t li, const HitRecord& hit, const SampleKey& key, uint32_t sub, Stats& st, Ray* outRay = nullptr, bool raw = false) {
    const rt_light& l = sc.lights[li];
    ShadowRay s = GetShadowRayForHit(sc, li, hit, key, raw ? SLOT_LIGHT_RAW : SLOT_LIGHT, sub);
    if (outRay) *outRay = s.ray;
    if (AnyIntersectionFast(sc, s.ray, s.kMin, s.kMax, st)) return Black;
    Colour c = col(l.colour);
    if (raw) return c;
    if (s.kMax > l.cutoff_radius) return Black;
    float distScale = Attenuation(l, s.kMax / l.attenuation_radius);
    distScale = MaxF(distScale, 0);
    return c * distScale;
}

struct RenderCtx {
    const SceneData* sc; rt_render_opts opts;
};

// Material.CalculateSimpleColourFromLights, Core/Base Type Definitions/Material.cs:30-45
static Colour CalculateSimpleColourFromLights(const RenderCtx& rc, const HitRecord& hit, const SampleKey& key, Stats& st) {
    Colour sum = Black;
    uint32_t samples = rc.opts.light_sample_count_hint, nl = (uint32_t)rc.sc->lights.size();
    for (uint32_t s = 0; s < samples; s++)
        for (uint32_t i = 0; i < nl; i++) sum = sum + CalculateLight(*rc.sc, i, hit, key, s * nl + i, st);
    sum = sum / (float)samples;
    return sum;
}

// Material.Scatter — never null for the builtin materials
static Ray Scatter(const RenderCtx& rc, const HitRecord& hit, const HitRecord* prev, int prevCount, const SampleKey& key) {
    const rt_material& m = rc.sc->materials[hit.Material];
    SampleKey::F4 r = key.draw(SLOT_SCATTER, 0);
    switch (m.type) {
        case RT_MAT_STANDARD: {                                                                        // StandardMaterial.cs:64-72
            V3 diffuse = OnUnitSphere(r.x, r.y, r.z);
            if (Dot(diffuse, hit.Normal) < 0) diffuse = diffuse * -1.0f;
            V3 reflect = Reflect(hit.IncomingRay.d, hit.Normal);
            return Ray(hit.WorldPoint, Normalize(Lerp(reflect, diffuse, m.f[0])));
        }
        case RT_MAT_PHONG: {                                                                           // PhongMaterial.cs:26-37
            V3 dir = OnUnitSphere(r.x, r.y, r.z);
            if (Dot(dir, hit.Normal) < 0) dir = dir * -1.0f;
            return Ray(hit.WorldPoint, dir);
        }
        case RT_MAT_VOLUMETRIC:                                                                        // ConstantDensityMedium.cs:102
            return Ray(hit.WorldPoint, OnUnitSphere(r.x, r.y, r.z));
        default: {                                                                                     // RefractiveMaterial.cs:37-99
            V3 unitDirection = Normalize(hit.IncomingRay.d);
            float cosTheta = MinF(Dot(-unitDirection, hit.Normal), 1.0f);
            float sinTheta = std::sqrt(1.0f - (cosTheta * cosTheta));
            float RefractiveIndex = m.f[0];
            bool outsideGoingInside = (prevCount == 0) || (prev[prevCount - 1].Material != hit.Material);
            float eta = outsideGoingInside ? 1.0f : RefractiveIndex, etaPrime = outsideGoingInside ? RefractiveIndex : 1.0f;
            float refractionRatio = eta / etaPrime;
            bool cannotRefract = refractionRatio * sinTheta > 1.0f;
            {
                float r0 = (eta - etaPrime) / (eta + etaPrime);
                r0 *= r0;
                float reflectance = r0 + ((1 - r0) * std::pow(1 - cosTheta, 5.0f));
                if (reflectance > r.w) cannotRefract = true;
            }
            V3 outDirection;
            if (cannotRefract) outDirection = Reflect(unitDirection, hit.Normal);
            else if (!(m.flags & RT_MAT_FLAG_ALTERNATE_REFRACTION)) {
                V3 perp = refractionRatio * (unitDirection + (cosTheta * hit.Normal));
                V3 par = -std::sqrt(std::fabs(1.0f - LengthSquared(perp))) * hit.Normal;
                outDirection = perp + par;
            } else {
                float nd = Dot(hit.Normal, unitDirection);
                outDirection = Normalize((std::sqrt((1 - std::pow(refractionRatio, 2.0f)) * (1 - std::pow(nd, 2.0f))) * hit.Normal) + (refractionRatio * (unitDirection - (nd * hit.Normal))));
            }
            return Ray(hit.WorldPoint, outDirection);
        }
    }
}

// Material.CalculateColour
static Colour CalculateColour(const RenderCtx& rc, Colour future, const HitRecord& hit, const HitRecord* prev, int prevCount, const SampleKey& key, Stats& st) {
    const SceneData& sc = *rc.sc;
    const rt_material& m = sc.materials[hit.Material];
    switch (m.type) {
        case RT_MAT_STANDARD: {                                                                        // StandardMaterial.cs:74-80
            Colour colour = CalculateSimpleColourFromLights(rc, hit, key, st) + future;
            colour = colour * TextureColour(sc, m.tex[0], hit);
            colour = colour + TextureColour(sc, m.tex[1], hit);
            return colour;
        }
        case RT_MAT_REFRACTIVE:                                                                        // RefractiveMaterial.cs:101-103
            return (future + CalculateSimpleColourFromLights(rc, hit, key, st)) * TextureColour(sc, m.tex[0], hit);
        case RT_MAT_EMISSIVE_REFRACTIVE: {                                                             // EmissiveRefractiveMaterial.cs:43-64
            Colour colour = future;
            colour = colour + CalculateSimpleColourFromLights(rc, hit, key, st);
            colour = colour * TextureColour(sc, m.tex[0], hit);
            if (m.flags & RT_MAT_FLAG_DIRECT_EMISSION) return colour + TextureColour(sc, m.tex[1], hit);
            if (prevCount == 0) return colour;
            if (prevCount == 1 && prev[0].Material == hit.Material) return colour;
            return colour + TextureColour(sc, m.tex[1], hit);
        }
        case RT_MAT_PHONG: {                                                                           // PhongMaterial.cs:39-85
            Colour Specular = col(m.f), Diffuse = col(m.f + 3), Ambient = col(m.f + 6);
            float Shininess = m.f[9];
            Colour rawDiffuse = Black, rawSpecular = Black;
            uint32_t lightSamples = rc.opts.light_sample_count_hint, nl = (uint32_t)sc.lights.size();
            for (uint32_t a = 0; a < lightSamples; a++)
                for (uint32_t i = 0; i < nl; i++) {
                    Ray dRay, sRay;
                    Colour dCol = CalculateLight(sc, i, hit, key, a * nl + i, st, &dRay, false);
                    Colour sCol = CalculateLight(sc, i, hit, key, a * nl + i, st, &sRay, true);
                    V3 refl = Reflect(sRay.d, -hit.Normal);
                    float spec = std::fabs(Dot(hit.IncomingRay.d, refl));
                    rawSpecular = rawSpecular + (sCol * std::pow(spec, std::pow(2.0f, Shininess)));
                    float diff = std::fabs(Dot(dRay.d, hit.Normal));
                    rawDiffuse = rawDiffuse + (dCol * diff);
                }
            rawDiffuse = rawDiffuse / (float)lightSamples;
            rawSpecular = rawSpecular / (float)lightSamples;
            {
                V3 refl = Reflect(hit.IncomingRay.d, -hit.Normal);
                float spec = std::fabs(Dot(hit.IncomingRay.d, refl));
                rawSpecular = rawSpecular + (future * std::pow(spec, std::pow(2.0f, Shininess)));
                rawDiffuse = rawDiffuse + future;
            }
            return Ambient + (rawDiffuse * Diffuse) + (rawSpecular * Specular);
        }
        default: {                                                                                     // the private VolumetricMaterial.CalculateColour, ConstantDensityMedium.cs:105-129 (class at :87)
            float pw = m.f[3] * hit.ShaderData;
            return Colour{std::pow(m.f[0], pw), std::pow(m.f[1], pw), std::pow(m.f[2], pw)} * future;
        }
    }
}

// Camera.GetRay, Core/Camera.cs:35-42 ; RandUtils.RandomInUnitCircle, Core/RandUtils.cs:53-59
static Ray GetRay(const rt_camera& cam, float u, float v, float thetaU, float rU) {
    float theta = 0 + (((2 * PI_F) - 0) * thetaU);                                                     // RandomFloat(0, 2*PI), RandUtils.cs:21
    float r = std::sqrt(rU);
    float sx = std::sin(theta), cy = std::cos(theta);                                                  // (x, y) = SinCos(theta)
    float randX = (r * sx) * cam.lens_radius, randY = (r * cy) * cam.lens_radius;
    V3 offset = (v3(cam.u) * randX) + (v3(cam.v) * randY);
    V3 origin = v3(cam.look_from) + offset;
    V3 direction = (v3(cam.lower_left_corner) + (u * v3(cam.horizontal)) + (v * v3(cam.vertical))) - origin;
    return Ray(origin, Normalize(direction));
}

// RenderJob.CalculateRayColourLooped, Core/RenderJob.cs:401-491
static Colour CalculateRayColourLooped(const RenderCtx& rc, const Ray& initialRay, SampleKey key, Stats& st, std::vector<HitRecord>& hitStateArray) {
    int maxDepthReached = -1;
    int maxAllowedDepth = (int)rc.opts.max_bounce_depth;
    Ray currentRay = initialRay;
    Colour finalColour;
    hitStateArray.resize((size_t)maxAllowedDepth + 1);
    for (int currentDepth = 0;; currentDepth++) {
        if (currentDepth > maxAllowedDepth) { st.bounceLimit++; finalColour = Black; break; }
        key.depth = (uint32_t)currentDepth;
        Found f = TryFindClosestHit(*rc.sc, currentRay, rc.opts.k_min, rc.opts.k_max, key, st);
        if (f.ok()) {
            maxDepthReached = currentDepth;
            hitStateArray[currentDepth] = f.hit;
            currentRay = Scatter(rc, f.hit, hitStateArray.data(), currentDepth, key);
            st.scatter++;
        } else {
            st.skyRays++;
            finalColour = SkyColour(*rc.sc, currentRay);
            break;
        }
    }
    if (maxDepthReached == -1) return finalColour;
    st.depth[std::min(maxDepthReached, RT_DEPTH_HIST_SIZE - 1)]++;
    for (int currentDepth = maxDepthReached; currentDepth >= 0; currentDepth--) {
        key.depth = (uint32_t)currentDepth;
        finalColour = CalculateColour(rc, finalColour, hitStateArray[currentDepth], hitStateArray.data(), currentDepth, key, st);
    }
    return finalColour;
}

// RenderJob.UniqueColourFromMaterialHash, Core/RenderJob.cs:317-328.  Material.GetHashCode() is an identity hash that differs from
// process to process; (material id + 1) * 2654435761 stands in for it (deviation D5, the same on the device).
static Colour UniqueColourFromMaterialHash(uint32_t material, bool alternate) {
    uint32_t hash = (material + 1u) * 2654435761u;
    int o = alternate ? 1 : 0;
    float rh = (float)((hash >> (8 * (o + 0))) & 0xFFu) / 255.0f;
    float gh = (float)((hash >> (8 * (o + 1))) & 0xFFu) / 255.0f;
    float bh = (float)((hash >> (8 * (o + 2))) & 0xFFu) / 255.0f;
    return {rh, gh, bh};
}

// RenderJob.RenderPixelWithVisualisations, Core/RenderJob.cs:135-310
static Colour RenderPixel(const RenderCtx& rc, uint32_t x, uint32_t y, uint64_t pass, Stats& st, std::vector<HitRecord>& scratch) {
    SampleKey key; key.seed = rc.opts.seed; key.pixel = x + y * rc.opts.width; key.pass = pass; key.depth = 0;
    SampleKey::F4 r = key.draw(SLOT_CAMERA, 0);
    float s = (float)x, t = (float)y;
    const float ssaaRadius = .5f;
    s += PlusMinusOne(r.x) * ssaaRadius;
    t += PlusMinusOne(r.y) * ssaaRadius;
    Ray viewRay = GetRay(rc.sc->camera, s / (float)rc.opts.width, t / (float)rc.opts.height, r.z, r.w);
    const uint32_t vis = rc.opts.debug_visualisation;
    if (vis == RT_VIS_NONE) return CalculateRayColourLooped(rc, viewRay, key, st, scratch);
    if (vis == RT_VIS_BOUNCE_DEPTH) {                                                                   // :251-306
        int maxDepthReached = -1;
        int maxAllowedDepth = (int)rc.opts.max_bounce_depth;
        Ray currentRay = viewRay;
        scratch.resize((size_t)maxAllowedDepth + 1);
        for (int currentDepth = 0;; currentDepth++) {
            if (currentDepth > maxAllowedDepth) { st.bounceLimit++; break; }
            key.depth = (uint32_t)currentDepth;
            Found f = TryFindClosestHit(*rc.sc, currentRay, rc.opts.k_min, rc.opts.k_max, key, st);
            if (f.ok()) {
                maxDepthReached = currentDepth;
                scratch[currentDepth] = f.hit;
                currentRay = Scatter(rc, f.hit, scratch.data(), currentDepth, key);
                st.scatter++;
            } else { st.skyRays++; break; }
        }
        if (maxDepthReached == -1) return {1, 0, 1};                                                    // Colour.Purple
        float val = 1 - std::pow(E_F, -((float)(maxDepthReached * maxDepthReached) * std::sqrt((float)maxAllowedDepth)));
        return ColourLerp({1, 1, 1}, Colour{0, 0, 1} * 0.02f, val);
    }
    Found f = TryFindClosestHit(*rc.sc, viewRay, rc.opts.k_min, rc.opts.k_max, key, st);
    if (!f.ok()) return Black;                                                                          // NoColour
    const HitRecord& hit = f.hit;
    switch (vis) {
        case RT_VIS_NORMALS: { V3 n = (hit.Normal + v3(1.0f)) / 2.0f; return {n.x, n.y, n.z}; }          // :156-162
        case RT_VIS_FACE_DIRECTION: return hit.OutsideFace ? Colour{0, 1, 0} : Colour{1, 0, 0};         // :163-168
        case RT_VIS_DISTANCE_CLOSE: { float z = hit.K - rc.opts.k_min; return col(std::pow(E_F, -(0.01f * z * z))); }
        case RT_VIS_DISTANCE_MID:   { float z = hit.K - rc.opts.k_min; return col(std::pow(E_F, -(0.0001f * z * z))); }
        case RT_VIS_DISTANCE_FAR:   { float z = hit.K - rc.opts.k_min; return col(std::pow(E_F, -(.000001f * z * z))); }
        case RT_VIS_UV_COORDS: return {hit.U, hit.V, 1};                                                // :229-233
        case RT_VIS_PIXEL_COORD_TEXTURE: return UniqueColourFromMaterialHash(hit.Material, std::sin((float)x / 2.0f) * std::sin((float)y / 2.0f) < 0);
        case RT_VIS_WORLD_COORD_TEXTURE: return UniqueColourFromMaterialHash(hit.Material, std::sin(hit.WorldPoint.x * 40.0f) * std::sin(hit.WorldPoint.y * 40.0f) * std::sin(hit.WorldPoint.z * 40.0f) < 0);
        case RT_VIS_LOCAL_COORD_TEXTURE: return UniqueColourFromMaterialHash(hit.Material, std::sin(hit.LocalPoint.x * 40.0f) * std::sin(hit.LocalPoint.y * 40.0f) * std::sin(hit.LocalPoint.z * 40.0f) < 0);
        case RT_VIS_SCATTER_DIRECTION: {                                                                // :221-228 (no builtin material absorbs)
            Ray sc = Scatter(rc, hit, scratch.data(), 0, key);
            V3 n = (sc.d + v3(1.0f)) / 2.0f; return {n.x, n.y, n.z};
        }
        case RT_VIS_ESTIMATED_LIGHT: {                                                                  // :234-250
            Colour sum = Black;
            uint32_t samples = rc.opts.light_sample_count_hint, nl = (uint32_t)rc.sc->lights.size();
            for (uint32_t i = 0; i < samples; i++)
                for (uint32_t li = 0; li < nl; li++) sum = sum + CalculateLight(*rc.sc, li, hit, key, i * nl + li, st);
            return sum / (float)samples;
        }
        default: return Black;
    }
}

} // namespace

// =============================================================================================
// C interface (ctypes)
// =============================================================================================
struct orc_scene { SceneData sc; };

static thread_local std::string g_err;

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

int orc_scene_create(const rt_scene_desc* d, orc_scene** out) {
    if (!d || !out || d->struct_size != sizeof(rt_scene_desc) || d->abi_version != RT_ABI_VERSION) { g_err = "bad scene desc"; return RT_ERR_INVALID_ARG; }
    auto s = std::make_unique<orc_scene>();
    SceneData& sc = s->sc;
    sc.camera = d->camera; sc.skybox = d->skybox;
    sc.materials.assign(d->materials, d->materials + d->n_materials);
    sc.textures.assign(d->textures, d->textures + d->n_textures);
    sc.lights.assign(d->lights, d->lights + d->n_lights);
    std::vector<SceneObjectRef> solids;
    for (uint32_t i = 0; i < d->n_objects; i++) {
        const rt_object& o = d->objects[i];
        auto h = MakeHittable(*d, o.type, o.index, g_err);
        if (!h) return RT_ERR_INVALID_ARG;
        if (o.material >= d->n_materials) { g_err = "material index out of range"; return RT_ERR_INVALID_ARG; }
        h->material = o.material;
        h->BoundingVolume = {v3(o.aabb_min), v3(o.aabb_max)};
        SceneObjectRef ref{i, h.get()};
        if (h->IsMedium()) sc.media.push_back(ref); else solids.push_back(ref);
        sc.objects.push_back(std::move(h));
    }
    for (uint32_t i = 0; i < d->n_lights; i++) {
        std::unique_ptr<Hittable> shape;
        if (d->lights[i].type == RT_LIGHT_DIFFUSE_SHAPED) {
            shape = MakeHittable(*d, d->lights[i].shape_type, d->lights[i].shape_index, g_err);
            if (!shape) return RT_ERR_INVALID_ARG;
        }
        sc.lightShapes.push_back(std::move(shape));
    }
    sc.root = FromSegment_SAH(solids, 0);
    sc.cull.Build(solids, sc.media);
    *out = s.release();
    return RT_OK;
}

int orc_scene_destroy(orc_scene* s) { delete s; return RT_OK; }

int orc_trace_closest(orc_scene* s, const rt_ray* rays, uint64_t n, float kMin, float kMax, uint64_t seed, rt_hit* hits) {
    Stats st;
    for (uint64_t i = 0; i < n; i++) {
        SampleKey key; key.seed = seed; key.pixel = (uint32_t)i; key.pass = 0; key.depth = 0;
        Ray r = Ray::Raw(v3(rays[i].origin), v3(rays[i].direction));
        Found f = TryFindClosestHit(s->sc, r, kMin, kMax, key, st);
        rt_hit& h = hits[i];
        memset(&h, 0, sizeof(h));
        h.object = RT_NO_HIT; h.material = RT_NO_HIT;
        if (f.ok()) {
            h.k = f.hit.K; h.object = f.obj; h.material = f.hit.Material;
            h.normal[0] = f.hit.Normal.x; h.normal[1] = f.hit.Normal.y; h.normal[2] = f.hit.Normal.z;
            h.uv[0] = f.hit.U; h.uv[1] = f.hit.V;
            h.world_point[0] = f.hit.WorldPoint.x; h.world_point[1] = f.hit.WorldPoint.y; h.world_point[2] = f.hit.WorldPoint.z;
            h.outside_face = f.hit.OutsideFace ? 1u : 0u;
        }
    }
    return RT_OK;
}

// AnyIntersectionFast on caller-supplied rays.  culled = 0: the reference's scan of every object; 1: through CullTree.
static int trace_any_impl(orc_scene* s, const rt_ray* rays, const float* kMin, const float* kMax, uint64_t n, uint8_t* occluded, bool culled, uint32_t n_threads) {
    if (n_threads < 1) n_threads = 1;
    std::atomic<uint64_t> next{0};
    auto worker = [&]() {
        Stats st;
        for (;;) {
            const uint64_t b = next.fetch_add(64);
            if (b >= n) break;
            for (uint64_t i = b; i < std::min(n, b + 64); i++) {
                Ray r = Ray::Raw(v3(rays[i].origin), v3(rays[i].direction));
                bool hit;
                if (culled) hit = s->sc.cull.Any(r, kMin[i], kMax[i]);
                else hit = ListAny(s->sc, r, kMin[i], kMax[i]);
                occluded[i] = hit ? 1 : 0;
            }
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < n_threads; t++) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    return RT_OK;
}
int orc_trace_any(orc_scene* s, const rt_ray* rays, const float* kMin, const float* kMax, uint64_t n, uint8_t* occluded) { return trace_any_impl(s, rays, kMin, kMax, n, occluded, false, 1); }
int orc_trace_any_mt(orc_scene* s, const rt_ray* rays, const float* kMin, const float* kMax, uint64_t n, uint8_t* occluded, uint32_t n_threads) { return trace_any_impl(s, rays, kMin, kMax, n, occluded, false, n_threads); }
int orc_trace_any_culled(orc_scene* s, const rt_ray* rays, const float* kMin, const float* kMax, uint64_t n, uint8_t* occluded, uint32_t n_threads) { return trace_any_impl(s, rays, kMin, kMax, n, occluded, true, n_threads); }
// orc_render's shadow rays through CullTree (1) or the reference's scan (0, the default)
int orc_scene_set_culled_any(orc_scene* s, int enable) { s->sc.culledAny = enable != 0; return RT_OK; }

// RenderJob.RenderInternal + UpdateBuffers' accumulation, Core/RenderJob.cs:69-126, 626-647.
// rgb_sum is in image-row order (row 0 = top, the y-flip of :631) and is accumulated into.
// Threads split the pixel index range like Parallel.For (:96); per-thread stats are merged at the end
// (the reference's Interlocked contention is NOT reproduced — this favours the CPU baseline).
static int render_impl(orc_scene* s, const rt_render_opts* opts, uint64_t first_pass, uint32_t n_passes, uint32_t pass_stride,
                       uint32_t n_threads, float* rgb_sum, rt_stats* stats_out, uint64_t* per_object, uint32_t n_objects) {
    if (!s || !opts || !rgb_sum || opts->width < 1 || opts->height < 1 || opts->light_sample_count_hint < 1) { g_err = "bad render args"; return RT_ERR_INVALID_ARG; }
    if (opts->debug_visualisation > RT_VIS_BOUNCE_DEPTH || opts->debug_visualisation == RT_VIS_UNDEFINED_TEST) { g_err = "Wrong enum value for debug visualisation"; return RT_ERR_INVALID_ARG; }   // RenderJob.cs:307-308
    RenderCtx rc{&s->sc, *opts};
    if (n_threads < 1) n_threads = 1;
    if (pass_stride < 1) pass_stride = 1;
    const uint64_t W = opts->width, H = opts->height, N = W * H;
    std::vector<Stats> tstats(n_threads);
    std::vector<std::vector<uint64_t>> tper(per_object ? n_threads : 0);
    for (uint32_t t = 0; t < tper.size(); t++) { tper[t].assign((size_t)n_objects * 2, 0); tstats[t].perObject = tper[t].data(); }
    for (uint32_t p = 0; p < n_passes; p++) {
        uint64_t pass = first_pass + (uint64_t)p * pass_stride;
        std::atomic<uint64_t> next{0};
        auto worker = [&](uint32_t tid) {
            std::vector<HitRecord> scratch;
            const uint64_t chunk = 256;
            for (;;) {
                uint64_t b = next.fetch_add(chunk);
                if (b >= N) break;
                uint64_t e = std::min(N, b + chunk);
                for (uint64_t i = b; i < e; i++) {
                    uint32_t x = (uint32_t)(i % W), y = (uint32_t)(i / W);                       // Decompress2DIndex, Core/MathUtils.cs:38-43
                    Colour c = RenderPixel(rc, x, y, pass, tstats[tid], scratch);
                    uint64_t yy = H - y - 1;                                                    // RenderJob.cs:631
                    float* dst = rgb_sum + 3 * (x + yy * W);
                    dst[0] += c.r; dst[1] += c.g; dst[2] += c.b;                                // :645
                    tstats[tid].pixels++;
                }
            }
        };
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < n_threads; t++) th.emplace_back(worker, t);
        worker(0);
        for (auto& t : th) t.join();
    }
    if (stats_out) {
        Stats tot; for (auto& t : tstats) tot.add(t);
        memset(stats_out, 0, sizeof(*stats_out));
        stats_out->pixels_rendered = tot.pixels; stats_out->passes_rendered = n_passes;
        stats_out->material_scatter_count = tot.scatter; stats_out->material_absorbed_count = tot.absorbed;
        stats_out->aabb_misses = tot.aabbMisses; stats_out->hittable_misses = tot.hittableMisses;
        stats_out->hittable_intersections = tot.hittableIntersections; stats_out->sky_rays = tot.skyRays;
        stats_out->normals_wrong_magnitude = tot.normalsWrong; stats_out->uv_invalid = tot.uvInvalid;
        stats_out->bounce_limit_exceeded = tot.bounceLimit; stats_out->ray_count = tot.rayCount; stats_out->shadow_rays = tot.shadowRays;
        for (int i = 0; i < RT_DEPTH_HIST_SIZE; i++) stats_out->ray_depth_counts[i] = tot.depth[i];
    }
    if (per_object) {
        std::fill(per_object, per_object + (size_t)n_objects * 2, 0ull);
        for (auto& t : tper) for (size_t i = 0; i < t.size(); i++) per_object[i] += t[i];
    }
    return RT_OK;
}
int orc_render(orc_scene* s, const rt_render_opts* opts, uint64_t first_pass, uint32_t n_passes, uint32_t pass_stride,
               uint32_t n_threads, float* rgb_sum, rt_stats* stats_out) {
    return render_impl(s, opts, first_pass, n_passes, pass_stride, n_threads, rgb_sum, stats_out, nullptr, 0);
}
// ... with GraphicsValidator.Errors as the reference files them (GraphicsValidator.cs:35-45 via RenderJob.cs:597,606): per_object[2 * i] =
// closest hits on scene object i whose normal is not a unit vector, [2 * i + 1] = whose UV lies outside [0, 1]^2
int orc_render_validated(orc_scene* s, const rt_render_opts* opts, uint64_t first_pass, uint32_t n_passes, uint32_t pass_stride,
                         uint32_t n_threads, float* rgb_sum, rt_stats* stats_out, uint64_t* per_object, uint32_t n_objects) {
    if (!per_object || !s || n_objects != s->sc.objects.size()) { g_err = "orc_render_validated: per_object must hold two counters for each of the scene's objects"; return RT_ERR_INVALID_ARG; }
    return render_impl(s, opts, first_pass, n_passes, pass_stride, n_threads, rgb_sum, stats_out, per_object, n_objects);
}

// the resolve at the end of UpdateBuffers (Core/RenderJob.cs:626-655), :646-654 ; Colour.Clamp01/Sqrt/(Rgb24), Core/Colour.cs:197-234
int orc_resolve_rgb24(const float* rgb_sum, uint32_t sample_count, uint32_t width, uint32_t height, uint8_t* dst) {
    uint64_t n = (uint64_t)width * height * 3;
    for (uint64_t i = 0; i < n; i++) {
        float c = rgb_sum[i] / (float)(int32_t)sample_count;       // Colour / int -> Colour / (float)int
        c = (c > (double)0.0f) ? c : 0.0f;                         // Max(value, Black): value > other ? value : other
        c = (c < (double)1.0f) ? c : 1.0f;                         // Min(.., White)
        c = std::sqrt(c);
        dst[i] = (uint8_t)(255 * c);                               // (byte)(byte.MaxValue * c): truncation
    }
    return RT_OK;
}

// Camera.Create, Core/Camera.cs:61-83
int orc_camera_create(const float* lookFrom, const float* lookTowards, const float* upVector, float verticalFov, float aspectRatio,
                      float lensRadius, float focusDistance, rt_camera* out) {
    V3 up = Normalize(v3(upVector));
    float theta = verticalFov * (PI_F / 180.0f);
    float h = std::tan(theta / 2.0f);
    float viewportHeight = 2.0f * h;
    float viewportWidth = aspectRatio * viewportHeight;
    V3 lookDir = Normalize(v3(lookFrom) - v3(lookTowards));
    V3 cr = Cross(up, lookDir);
    if (cr.x == 0 && cr.y == 0 && cr.z == 0) { g_err = "Camera cannot point in the same direction as its 'up' vector"; return RT_ERR_INVALID_ARG; }
    V3 u = Normalize(cr);
    V3 v = Cross(lookDir, u);
    V3 horizontal = (viewportWidth * u) * focusDistance;
    V3 vertical = (viewportHeight * v) * focusDistance;
    V3 llc = v3(lookFrom) - (horizontal / 2.0f) - (vertical / 2.0f) - (focusDistance * lookDir);
    auto put = [](float* d, V3 a) { d[0] = a.x; d[1] = a.y; d[2] = a.z; };
    put(out->look_from, v3(lookFrom)); put(out->horizontal, horizontal); put(out->vertical, vertical);
    put(out->lower_left_corner, llc); put(out->u, u); put(out->v, v); out->lens_radius = lensRadius;
    return RT_OK;
}

// Known-answer helper: Sphere.GetSphereUV (Impl/Hittables/Sphere.cs:78-93)
void orc_sphere_uv(const float* p, float* uv) { Sphere::GetSphereUV(v3(p), uv[0], uv[1]); }

// Known-answer helper: one camera ray at the pixel centre (no jitter, no lens), then sky + resolve of a single sample.
void orc_pixel_centre_sky_rgb24(const rt_camera* cam, const rt_skybox* sky, uint32_t x, uint32_t y_from_bottom, uint32_t width, uint32_t height, uint8_t* rgb) {
    SceneData sc; sc.camera = *cam; sc.skybox = *sky;
    rt_camera c = *cam; c.lens_radius = 0;
    Ray r = GetRay(c, (float)x / (float)width, (float)y_from_bottom / (float)height, 0.0f, 0.0f);
    Colour col = SkyColour(sc, r);
    float f[3] = {col.r, col.g, col.b};
    orc_resolve_rgb24(f, 1, 1, 1, rgb);
}

} // extern "C"
