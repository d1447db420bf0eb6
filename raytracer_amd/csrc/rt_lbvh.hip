// rt_lbvh.hip — BVH build on the device for scenes of millions of objects (reference analogue
// BvhTree.FromSegment_SAH, Acceleration/BvhTree.cs:37-136, whose topology is implementation-defined: only the closest hit matters).
//
// The host's binned-SAH builder (rt_bvh.cpp) takes 3.8 s for config 4's 5 M triangles; this one sorts the objects along a 63-bit
// Morton curve of their box centres (rocPRIM radix sort through hipCUB) and links the n-1 inner nodes in one pass (Karras 2012:
// every inner node finds its own range and split by binary search over common key prefixes; equal keys are told apart by their
// position), then fits the boxes bottom-up: one thread per leaf climbs, the second arrival at a node merges and goes on.
// Output = the same 64-B nodes the traversal kernels read (both child boxes in the parent, one object per leaf), root = node 0.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdint>
#include <cstring>
#include <string>

#include "rt_bvh.h"

namespace rtlbvh {
namespace {

constexpr int TB = 256;

__device__ __forceinline__ uint64_t spread21(uint32_t v) {         // 21 bits -> every third bit of 63
    uint64_t x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

struct Frame { float lo[3], scale[3]; };

__global__ void __launch_bounds__(TB) k_morton(const rtbvh::Box* boxes, uint32_t n, Frame f, uint64_t* keys, uint32_t* vals) {
    uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    rtbvh::Box b = boxes[i];
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        float c = 0.5f * b.mn[a] + 0.5f * b.mx[a];
        float t = (c - f.lo[a]) * f.scale[a];
        t = t > 0.0f ? t : 0.0f;                                     // (NaN -> 0)
        q[a] = (uint32_t)fminf(t, 2097151.0f);
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[i] = i;
}

// length of the common prefix of the sorted entries i and j (-1 outside the array); equal keys differ by position
__device__ __forceinline__ int delta(const uint64_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ b));
}

// parent links: (node << 1) | side
__global__ void __launch_bounds__(TB) k_hierarchy(const uint64_t* keys, const uint32_t* vals, const uint32_t* object_ids, int n,
                                                  rtbvh::HostNode* nodes, uint32_t* parent_inner, uint32_t* parent_leaf) {
    int i = blockIdx.x * TB + threadIdx.x;
    if (i >= n - 1) return;
    int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2) if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = delta(keys, n, i, j);
    int s = 0, t = l;
    do { t = (t + 1) / 2; if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t; } while (t > 1);
    int gamma = i + s * d + (d < 0 ? d : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    int32_t childA, childB;
    if (lo == gamma) { childA = ~(int32_t)object_ids[vals[gamma]]; parent_leaf[gamma] = ((uint32_t)i << 1); }
    else { childA = gamma; parent_inner[gamma] = ((uint32_t)i << 1); }
    if (hi == gamma + 1) { childB = ~(int32_t)object_ids[vals[gamma + 1]]; parent_leaf[gamma + 1] = ((uint32_t)i << 1) | 1u; }
    else { childB = gamma + 1; parent_inner[gamma + 1] = ((uint32_t)i << 1) | 1u; }
    nodes[i].childA = childA; nodes[i].childB = childB; nodes[i].pad0 = 0; nodes[i].pad1 = 0;
}

__device__ __forceinline__ float load_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(TB) k_refit(const rtbvh::Box* boxes, const uint32_t* vals, int n, rtbvh::HostNode* nodes, const uint32_t* parent_inner,
                                              const uint32_t* parent_leaf, uint32_t* arrivals, uint32_t* heights /* [2 * (n-1)] */, uint32_t* max_depth) {
    int k = blockIdx.x * TB + threadIdx.x;
    if (k >= n) return;
    rtbvh::Box b = boxes[vals[k]];
    uint32_t h = 0, link = parent_leaf[k];
    for (;;) {
        const uint32_t p = link >> 1, side = link & 1u;
        float* q = nodes[p].q;
        for (int a = 0; a < 3; a++) { q[2 * a + side] = b.mn[a]; q[6 + 2 * a + side] = b.mx[a]; }
        heights[2 * p + side] = h;
        __threadfence();
        if (atomicAdd(&arrivals[p], 1u) == 0u) return;               // the sibling subtree is not finished: its thread will go on
        __threadfence();
        const uint32_t other = side ^ 1u;
        for (int a = 0; a < 3; a++) {
            b.mn[a] = fminf(b.mn[a], load_agent(q + 2 * a + other));
            b.mx[a] = fmaxf(b.mx[a], load_agent(q + 6 + 2 * a + other));
        }
        const uint32_t ho = __hip_atomic_load(heights + 2 * p + other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h = (h > ho ? h : ho) + 1u;
        if (p == 0) { *max_depth = h; return; }
        link = parent_inner[p];
    }
}

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    template <class T> T* as() const { return (T*)p; }
};

} // namespace

#define LBVH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? 2 : 1; } } while (0)

// boxes/object_ids: host arrays of the n >= 2 bounded objects; d_nodes: device buffer of n-1 nodes.  0 = ok, 1 = HIP error, 2 = out of memory.
int build(const rtbvh::Box* boxes, const uint32_t* object_ids, uint32_t n, void* d_nodes, uint32_t* max_depth, std::string& err) {
    if (n < 2 || n > 0x3FFFFFFFu) { err = "rtlbvh::build: object count out of range"; return 1; }
    Frame f;
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < n; i++)
            for (int a = 0; a < 3; a++) { float c = 0.5f * boxes[i].mn[a] + 0.5f * boxes[i].mx[a]; lo[a] = c < lo[a] ? c : lo[a]; hi[a] = c > hi[a] ? c : hi[a]; }
        // one scale for the three axes (cubic cells): a flat scene then spends its key bits on the two axes that matter
        float ext = 0.0f;
        for (int a = 0; a < 3; a++) { float e = hi[a] - lo[a]; if (std::isfinite(e) && e > ext) ext = e; }
        for (int a = 0; a < 3; a++) { f.lo[a] = lo[a]; f.scale[a] = ext > 0.0f ? 2097152.0f / ext : 0.0f; }
    }
    Buf d_boxes, d_ids, keys0, keys1, vals0, vals1, temp, par_in, par_leaf, arrivals, heights, depth;
    LBVH_TRY(d_boxes.alloc((size_t)n * sizeof(rtbvh::Box))); LBVH_TRY(d_ids.alloc((size_t)n * 4));
    LBVH_TRY(keys0.alloc((size_t)n * 8)); LBVH_TRY(keys1.alloc((size_t)n * 8)); LBVH_TRY(vals0.alloc((size_t)n * 4)); LBVH_TRY(vals1.alloc((size_t)n * 4));
    LBVH_TRY(par_in.alloc((size_t)n * 4)); LBVH_TRY(par_leaf.alloc((size_t)n * 4)); LBVH_TRY(arrivals.alloc((size_t)n * 4)); LBVH_TRY(heights.alloc((size_t)n * 8));
    LBVH_TRY(depth.alloc(4));
    LBVH_TRY(hipMemcpy(d_boxes.p, boxes, (size_t)n * sizeof(rtbvh::Box), hipMemcpyHostToDevice));
    LBVH_TRY(hipMemcpy(d_ids.p, object_ids, (size_t)n * 4, hipMemcpyHostToDevice));
    LBVH_TRY(hipMemset(arrivals.p, 0, (size_t)n * 4)); LBVH_TRY(hipMemset(depth.p, 0, 4));
    const uint32_t grid = (n + TB - 1) / TB;
    hipLaunchKernelGGL(k_morton, dim3(grid), dim3(TB), 0, 0, d_boxes.as<rtbvh::Box>(), n, f, keys0.as<uint64_t>(), vals0.as<uint32_t>());
    size_t temp_bytes = 0;
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys0.as<uint64_t>(), keys1.as<uint64_t>(), vals0.as<uint32_t>(), vals1.as<uint32_t>(), (int)n, 0, 63));
    LBVH_TRY(temp.alloc(temp_bytes));
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, keys0.as<uint64_t>(), keys1.as<uint64_t>(), vals0.as<uint32_t>(), vals1.as<uint32_t>(), (int)n, 0, 63));
    hipLaunchKernelGGL(k_hierarchy, dim3(grid), dim3(TB), 0, 0, keys1.as<uint64_t>(), vals1.as<uint32_t>(), d_ids.as<uint32_t>(), (int)n, (rtbvh::HostNode*)d_nodes,
                       par_in.as<uint32_t>(), par_leaf.as<uint32_t>());
    hipLaunchKernelGGL(k_refit, dim3(grid), dim3(TB), 0, 0, d_boxes.as<rtbvh::Box>(), vals1.as<uint32_t>(), (int)n, (rtbvh::HostNode*)d_nodes, par_in.as<uint32_t>(),
                       par_leaf.as<uint32_t>(), arrivals.as<uint32_t>(), heights.as<uint32_t>(), depth.as<uint32_t>());
    LBVH_TRY(hipGetLastError());
    LBVH_TRY(hipDeviceSynchronize());
    LBVH_TRY(hipMemcpy(max_depth, depth.p, 4, hipMemcpyDeviceToHost));
    return 0;
}

} // namespace rtlbvh
