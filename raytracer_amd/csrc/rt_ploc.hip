// rt_ploc.hip — SAH-quality BVH build on the device (reference analogue BvhTree.FromSegment_SAH,
// Acceleration/BvhTree.cs:37-136, whose topology is implementation-defined: only the closest hit matters).
//
// rt_lbvh.hip links a Morton-sorted array in one pass, and its rays pay 15 % more box tests than on the host's binned-SAH tree.
// This builder keeps the Morton order only as a search structure (PLOC, Meister & Bittner 2018): the clusters — at first the
// objects, in Morton order — each look RADIUS places to either side for the neighbour whose union with them has the smallest
// surface area; two clusters that choose each other are merged into one inner node, which takes the lower one's place; the
// array is compacted and the round repeats until one cluster is left.  Bottom-up merging by surface area is what agglomerative
// SAH builders do; the search window makes a round O(n * RADIUS) and the whole build ~35 rounds for 5 M triangles.
//
// Everything is deterministic: pairs are ordered by (area, lower index, upper index), so the smallest pair of a round is always
// mutual (progress) and equal areas never depend on timing; node indices come from a prefix sum, not from an atomic counter.
// Inner nodes are numbered from the end of the array backwards in order of creation, so the root is node 0 and a parent always
// precedes its children.  Output = the 64-B nodes the traversal kernels read (both child boxes in the parent, one object per leaf).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdint>
#include <cstring>
#include <string>

#include "rt_bvh.h"

namespace rtploc {
namespace {

constexpr int TB = 256;
#ifndef RT_PLOC_RADIUS
#define RT_PLOC_RADIUS 24
#endif
constexpr int RADIUS = RT_PLOC_RADIUS;

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

// a cluster: its box, what a parent would call it (>= 0 inner node, < 0 leaf ~object) and the height of its subtree
struct Clusters { rtbvh::Box* box; int32_t* code; uint32_t* height; };

__global__ void __launch_bounds__(TB) k_leaves(const rtbvh::Box* boxes, const uint32_t* vals, const uint32_t* object_ids, uint32_t n, Clusters c) {
    uint32_t k = blockIdx.x * TB + threadIdx.x;
    if (k >= n) return;
    const uint32_t v = vals[k];
    c.box[k] = boxes[v]; c.code[k] = ~(int32_t)object_ids[v]; c.height[k] = 0u;
}

__device__ __forceinline__ float half_area_of_union(const rtbvh::Box& a, const rtbvh::Box& b) {
    const float dx = fmaxf(a.mx[0], b.mx[0]) - fminf(a.mn[0], b.mn[0]);
    const float dy = fmaxf(a.mx[1], b.mx[1]) - fminf(a.mn[1], b.mn[1]);
    const float dz = fmaxf(a.mx[2], b.mx[2]) - fminf(a.mn[2], b.mn[2]);
    return dx * dy + dy * dz + dz * dx;
}

// nearest neighbour of every cluster within RADIUS places, by the pair order (area, lower index, upper index)
__global__ void __launch_bounds__(TB) k_nearest(const rtbvh::Box* box, uint32_t m, uint32_t* nn) {
    __shared__ rtbvh::Box tile[TB + 2 * RADIUS];
    const int first = (int)(blockIdx.x * TB) - RADIUS;
    for (int t = threadIdx.x; t < TB + 2 * RADIUS; t += TB) {
        const int g = first + t;
        if (g >= 0 && g < (int)m) tile[t] = box[g];
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= m) return;
    const rtbvh::Box me = tile[threadIdx.x + RADIUS];
    // start from the array neighbour with an infinite area: whatever the boxes hold (even NaN), every cluster names somebody, and
    // the last two name each other
    uint32_t best = i + 1 < m ? i + 1 : i - 1;
    float best_area = INFINITY;
    const int lo = (int)i - RADIUS < 0 ? 0 : (int)i - RADIUS, hi = (int)i + RADIUS >= (int)m ? (int)m - 1 : (int)i + RADIUS;
    for (int j = lo; j <= hi; j++) {
        if (j == (int)i) continue;
        const float a = half_area_of_union(me, tile[j - first]);
        // (area, min(i,j), max(i,j)) ascending: for a fixed i and j below i the pair is (j, i), above i it is (i, j); among equal
        // areas every j below i comes before every j above i, and smaller j first on either side — the scan order itself
        if (a < best_area) { best_area = a; best = (uint32_t)j; }
    }
    nn[i] = best;
}

// 1 in the low word: the cluster stays in the array (it is not the upper half of a merge); 1 in the high word: it is the lower
// half of a merge and becomes a new inner node
__global__ void __launch_bounds__(TB) k_flags(const uint32_t* nn, uint32_t m, uint64_t* flags) {
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = nn[i];
    const bool mutual = nn[j] == i;
    flags[i] = (mutual && j < i ? 0ull : 1ull) | (mutual && i < j ? 1ull << 32 : 0ull);
}

struct RoundResult { uint32_t clusters, merges; };

__global__ void __launch_bounds__(TB) k_apply(Clusters in, Clusters out, const uint32_t* nn, const uint64_t* flags, const uint64_t* scan, uint32_t m,
                                              rtbvh::HostNode* nodes, uint32_t last_free /* index of the next node to hand out, counting down */, RoundResult* res) {
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= m) return;
    const uint64_t f = flags[i], s = scan[i];
    if (i == m - 1) { res->clusters = (uint32_t)(s & 0xFFFFFFFFull) + (uint32_t)(f & 1ull); res->merges = (uint32_t)(s >> 32) + (uint32_t)(f >> 32); }
    if (!(f & 1ull)) return;
    const uint32_t pos = (uint32_t)(s & 0xFFFFFFFFull);
    rtbvh::Box a = in.box[i];
    int32_t code = in.code[i];
    uint32_t h = in.height[i];
    if (f >> 32) {
        const uint32_t j = nn[i];
        const rtbvh::Box b = in.box[j];
        const uint32_t idx = last_free - (uint32_t)(s >> 32);
        rtbvh::HostNode nd;
        for (int k = 0; k < 3; k++) { nd.q[2 * k] = a.mn[k]; nd.q[2 * k + 1] = b.mn[k]; nd.q[6 + 2 * k] = a.mx[k]; nd.q[6 + 2 * k + 1] = b.mx[k]; }
        nd.childA = code; nd.childB = in.code[j]; nd.pad0 = 0; nd.pad1 = 0;
        nodes[idx] = nd;
        for (int k = 0; k < 3; k++) { a.mn[k] = fminf(a.mn[k], b.mn[k]); a.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
        const uint32_t hj = in.height[j];
        h = (h > hj ? h : hj) + 1u;
        code = (int32_t)idx;
    }
    out.box[pos] = a; out.code[pos] = code; out.height[pos] = h;
}

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    template <class T> T* as() const { return (T*)p; }
};

} // namespace

#define PLOC_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? 2 : 1; } } while (0)

// boxes/object_ids: host arrays of the n >= 2 bounded objects; d_nodes: device buffer of n-1 nodes.  0 = ok, 1 = HIP error, 2 = out of memory.
int build(const rtbvh::Box* boxes, const uint32_t* object_ids, uint32_t n, void* d_nodes, uint32_t* max_depth, uint32_t* rounds, std::string& err) {
    if (n < 2 || n > 0x3FFFFFFFu) { err = "rtploc::build: object count out of range"; return 1; }
    Frame f;
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < n; i++)
            for (int a = 0; a < 3; a++) { float c = 0.5f * boxes[i].mn[a] + 0.5f * boxes[i].mx[a]; lo[a] = c < lo[a] ? c : lo[a]; hi[a] = c > hi[a] ? c : hi[a]; }
        float ext = 0.0f;                                             // one scale for the three axes (cubic cells), as in rt_lbvh.hip
        for (int a = 0; a < 3; a++) { float e = hi[a] - lo[a]; if (std::isfinite(e) && e > ext) ext = e; }
        for (int a = 0; a < 3; a++) { f.lo[a] = lo[a]; f.scale[a] = ext > 0.0f ? 2097152.0f / ext : 0.0f; }
    }
    Buf d_boxes, d_ids, keys0, keys1, vals0, vals1, temp, nn, flags, scan, res, cb[2], cc[2], ch[2];
    PLOC_TRY(d_boxes.alloc((size_t)n * sizeof(rtbvh::Box))); PLOC_TRY(d_ids.alloc((size_t)n * 4));
    PLOC_TRY(keys0.alloc((size_t)n * 8)); PLOC_TRY(keys1.alloc((size_t)n * 8)); PLOC_TRY(vals0.alloc((size_t)n * 4)); PLOC_TRY(vals1.alloc((size_t)n * 4));
    PLOC_TRY(nn.alloc((size_t)n * 4)); PLOC_TRY(flags.alloc((size_t)n * 8)); PLOC_TRY(scan.alloc((size_t)n * 8)); PLOC_TRY(res.alloc(sizeof(RoundResult)));
    for (int k = 0; k < 2; k++) { PLOC_TRY(cb[k].alloc((size_t)n * sizeof(rtbvh::Box))); PLOC_TRY(cc[k].alloc((size_t)n * 4)); PLOC_TRY(ch[k].alloc((size_t)n * 4)); }
    PLOC_TRY(hipMemcpy(d_boxes.p, boxes, (size_t)n * sizeof(rtbvh::Box), hipMemcpyHostToDevice));
    PLOC_TRY(hipMemcpy(d_ids.p, object_ids, (size_t)n * 4, hipMemcpyHostToDevice));
    const uint32_t grid_n = (n + TB - 1) / TB;
    hipLaunchKernelGGL(k_morton, dim3(grid_n), dim3(TB), 0, 0, d_boxes.as<rtbvh::Box>(), n, f, keys0.as<uint64_t>(), vals0.as<uint32_t>());
    size_t sort_bytes = 0, scan_bytes = 0;
    PLOC_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys0.as<uint64_t>(), keys1.as<uint64_t>(), vals0.as<uint32_t>(), vals1.as<uint32_t>(), (int)n, 0, 63));
    PLOC_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, flags.as<uint64_t>(), scan.as<uint64_t>(), (int)n));
    PLOC_TRY(temp.alloc(sort_bytes > scan_bytes ? sort_bytes : scan_bytes));
    size_t tb = sort_bytes;
    PLOC_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, tb, keys0.as<uint64_t>(), keys1.as<uint64_t>(), vals0.as<uint32_t>(), vals1.as<uint32_t>(), (int)n, 0, 63));
    Clusters c[2];
    for (int k = 0; k < 2; k++) { c[k].box = cb[k].as<rtbvh::Box>(); c[k].code = cc[k].as<int32_t>(); c[k].height = ch[k].as<uint32_t>(); }
    hipLaunchKernelGGL(k_leaves, dim3(grid_n), dim3(TB), 0, 0, d_boxes.as<rtbvh::Box>(), vals1.as<uint32_t>(), d_ids.as<uint32_t>(), n, c[0]);
    uint32_t m = n, last_free = n - 2, n_rounds = 0;
    int cur = 0;
    while (m > 1) {
        const uint32_t grid = (m + TB - 1) / TB;
        hipLaunchKernelGGL(k_nearest, dim3(grid), dim3(TB), 0, 0, c[cur].box, m, nn.as<uint32_t>());
        hipLaunchKernelGGL(k_flags, dim3(grid), dim3(TB), 0, 0, nn.as<uint32_t>(), m, flags.as<uint64_t>());
        tb = scan_bytes;
        PLOC_TRY(hipcub::DeviceScan::ExclusiveSum(temp.p, tb, flags.as<uint64_t>(), scan.as<uint64_t>(), (int)m));
        hipLaunchKernelGGL(k_apply, dim3(grid), dim3(TB), 0, 0, c[cur], c[cur ^ 1], nn.as<uint32_t>(), flags.as<uint64_t>(), scan.as<uint64_t>(), m,
                           (rtbvh::HostNode*)d_nodes, last_free, res.as<RoundResult>());
        RoundResult r;
        PLOC_TRY(hipMemcpy(&r, res.p, sizeof(r), hipMemcpyDeviceToHost));     // (synchronises: the next round's grid depends on it)
        if (r.merges == 0 || r.clusters != m - r.merges) { err = "rtploc::build: a round merged nothing"; return 1; }     // cannot happen (the smallest pair is mutual); never loop forever
        m = r.clusters; last_free -= r.merges; cur ^= 1; n_rounds++;
    }
    PLOC_TRY(hipGetLastError());
    uint32_t h = 0;
    PLOC_TRY(hipMemcpy(&h, c[cur].height, 4, hipMemcpyDeviceToHost));
    *max_depth = h;
    if (rounds) *rounds = n_rounds;
    return 0;
}

} // namespace rtploc
