// rt_sah.hip — binned-SAH BVH build on the device (reference analogue BvhTree.FromSegment_SAH,
// Acceleration/BvhTree.cs:37-136, whose topology is implementation-defined: only the closest hit matters).
//
// The same top-down algorithm as the host builder's big-node branch (rt_bvh.cpp: 16 bins per axis over the centroid bounds, the
// cheapest of the 45 bin borders by surface-area cost in double, objects partitioned in stable order, positional halves when no
// axis separates the centroids or the tree gets deeper than 40), run breadth-first: every level is a handful of launches over
// the nodes that are still open.  The objects of a node are a contiguous range of one permutation array, and with one object per
// leaf a subtree over c objects has c-1 inner nodes, so a node's preorder index follows from its parent's (left = parent + 1,
// right = parent + left count): the array comes out in the host builder's order, no renumbering pass.
//   large nodes (> 2048 objects): one workgroup per 2048 positions bins into LDS (ds atomics on order-preserving integer images
//     of the floats) and merges into the node's bins in memory; one wave per node picks the split and turns the workgroups' own
//     bin counts into their write offsets; the same workgroups then scatter their positions into the other permutation array.
//   small nodes: one wave per node does all of that on a copy of the node's range in LDS.
// Counts, minima and maxima do not depend on the order the atomics arrive in: the tree is the same on every run.
// Output = the 64-B nodes the traversal kernels read (both child boxes in the parent, one object per leaf), root = node 0; the
// nodes never exist on the host.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "rt_bvh.h"

namespace rtsah {
namespace {

constexpr int kBins = 16;
constexpr uint32_t SMALL_MAX = 2048;      // nodes of at most this many objects are built by one wave on a copy in LDS
constexpr uint32_t CHUNK = 2048;          // positions per workgroup of a large node
constexpr int TB = 256;
constexpr uint32_t kMaxSahDepth = 40;     // beyond this, halves (as the host builder): the traversal stack stays short
constexpr int MAX_LEVELS = 128;

struct Task {
    uint32_t idx, begin, count, depth;    // node index, its range of the permutation, its level (root = 1)
    float cb[6];                          // bounds of the objects' box centres: min xyz, max xyz
    uint32_t buf, blk_base;               // which permutation array holds the range; first workgroup of a large node
    uint32_t axis, best_bin, nl, pad;     // the split (k_large_split -> k_large_scatter)
};
static_assert(sizeof(Task) == 64, "one line per task");

struct Counters { uint32_t n_large, n_blocks, n_small, pad; };

// per axis and bin: object count, union of the objects' boxes, bounds of their centres (what the children's tasks need)
struct Bins { uint32_t cnt[3][kBins]; uint32_t bmin[3][kBins][3], bmax[3][kBins][3], cmin[3][kBins][3], cmax[3][kBins][3]; };
constexpr int BINS_WORDS = (int)(sizeof(Bins) / 4);
static_assert(BINS_WORDS == 624, "layout assumed by bins_word_kind");
// 0 = a count, 1 = a minimum, 2 = a maximum
__device__ __forceinline__ int bins_word_kind(int j) { return j < 48 ? 0 : j < 192 ? 1 : j < 336 ? 2 : j < 480 ? 1 : 2; }
__device__ __forceinline__ uint32_t bins_word_init(int j) { return bins_word_kind(j) == 1 ? 0xFFFFFFFFu : 0u; }

// float <-> unsigned with the same order (atomic min / max of floats through integer atomics)
__device__ __forceinline__ uint32_t enc(float f) { uint32_t b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ float dec(uint32_t u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

struct Lists { Task* large; Task* small; uint32_t* blk_task; Counters* ctr; };

// how a node's objects are binned: by their centres (SAH) or, when no axis separates them or the node is too deep, by position
struct Plan { bool halves; bool use[3]; float cmn[3], scale[3]; };
__device__ __forceinline__ Plan plan_of(const Task& t) {
    Plan p; bool any = false;
    for (int a = 0; a < 3; a++) {
        const float ext = t.cb[3 + a] - t.cb[a], sc = (float)kBins / ext;
        p.use[a] = ext > 0.0f && isfinite(ext) && isfinite(sc);
        p.scale[a] = p.use[a] ? sc : 0.0f; p.cmn[a] = t.cb[a];
        any |= p.use[a];
    }
    p.halves = !any || t.depth >= kMaxSahDepth;
    if (p.halves) { p.use[0] = true; p.use[1] = false; p.use[2] = false; }
    return p;
}
__device__ __forceinline__ int bin_of(const Plan& p, int a, float c, uint32_t local_pos, uint32_t count) {
    if (p.halves) return local_pos < count / 2 ? 0 : 1;
    int k = (int)((c - p.cmn[a]) * p.scale[a]);
    return k < 0 ? 0 : k > kBins - 1 ? kBins - 1 : k;
}
__device__ __forceinline__ void centre_of(const rtbvh::Box& b, float c[3]) {
    for (int a = 0; a < 3; a++) c[a] = 0.5f * b.mn[a] + 0.5f * b.mx[a];
}
__device__ __forceinline__ void bin_add(Bins& B, int a, int k, const rtbvh::Box& bx, const float c[3]) {      // B in LDS
    atomicAdd(&B.cnt[a][k], 1u);
    for (int d = 0; d < 3; d++) {
        atomicMin(&B.bmin[a][k][d], enc(bx.mn[d])); atomicMax(&B.bmax[a][k][d], enc(bx.mx[d]));
        atomicMin(&B.cmin[a][k][d], enc(c[d])); atomicMax(&B.cmax[a][k][d], enc(c[d]));
    }
}

struct Split { int axis, bin; uint32_t nl; rtbvh::Box lb, rb; float lcb[6], rcb[6]; };

__device__ __forceinline__ double half_area(const float mn[3], const float mx[3]) {
    const double dx = (double)mx[0] - mn[0], dy = (double)mx[1] - mn[1], dz = (double)mx[2] - mn[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0;
    return dx * dy + dy * dz + dz * dx;
}
// union of the boxes of bins [k0, k1] of axis a, and their object count
__device__ __forceinline__ uint32_t bins_union(const Bins& B, int a, int k0, int k1, float mn[3], float mx[3]) {
    uint32_t c = 0;
    for (int d = 0; d < 3; d++) { mn[d] = INFINITY; mx[d] = -INFINITY; }
    for (int k = k0; k <= k1; k++) {
        if (!B.cnt[a][k]) continue;
        c += B.cnt[a][k];
        for (int d = 0; d < 3; d++) { mn[d] = fminf(mn[d], dec(B.bmin[a][k][d])); mx[d] = fmaxf(mx[d], dec(B.bmax[a][k][d])); }
    }
    return c;
}
// One wave.  The 45 bin borders are priced by 45 lanes; lane 0 takes the cheapest (the first one among equals, axis by axis and bin
// by bin, as the host builder's loop does) and works out what the two children are.  `out` and `cost` in LDS.
__device__ void choose_split(const Bins& B, const Plan& p, double* cost, Split* out, int lane) {
    if (lane < 48) cost[lane] = INFINITY;
    if (!p.halves && lane < 45) {
        const int a = lane / 15, b = lane % 15;
        if (p.use[a]) {
            float lmn[3], lmx[3], rmn[3], rmx[3];
            const uint32_t cl = bins_union(B, a, 0, b, lmn, lmx), cr = bins_union(B, a, b + 1, kBins - 1, rmn, rmx);
            if (cl && cr) cost[lane] = half_area(lmn, lmx) * cl + half_area(rmn, rmx) * cr;
        }
    }
    __syncthreads();
    if (lane == 0) {
        int best = 0;
        if (!p.halves) {
            for (int a = 2; a >= 0; a--) if (p.use[a]) best = a * 15;                // (every usable axis has objects in its first and last bin: a valid fallback)
            double bc = INFINITY;
            for (int j = 0; j < 45; j++) if (cost[j] < bc) { bc = cost[j]; best = j; }
        }
        Split s; s.axis = best / 15; s.bin = best % 15;
        s.nl = bins_union(B, s.axis, 0, s.bin, s.lb.mn, s.lb.mx);
        bins_union(B, s.axis, s.bin + 1, kBins - 1, s.rb.mn, s.rb.mx);
        for (int d = 0; d < 3; d++) { s.lcb[d] = INFINITY; s.lcb[3 + d] = -INFINITY; s.rcb[d] = INFINITY; s.rcb[3 + d] = -INFINITY; }
        for (int k = 0; k < kBins; k++) {
            if (!B.cnt[s.axis][k]) continue;
            float* cb = k <= s.bin ? s.lcb : s.rcb;
            for (int d = 0; d < 3; d++) { cb[d] = fminf(cb[d], dec(B.cmin[s.axis][k][d])); cb[3 + d] = fmaxf(cb[3 + d], dec(B.cmax[s.axis][k][d])); }
        }
        *out = s;
    }
    __syncthreads();
}

// the node's record: both children's boxes; the codes of inner children (a leaf's code is written by whoever places its object)
__device__ __forceinline__ void write_node(rtbvh::HostNode* nodes, const Task& t, const Split& s) {
    rtbvh::HostNode& nd = nodes[t.idx];
    for (int a = 0; a < 3; a++) { nd.q[2 * a] = s.lb.mn[a]; nd.q[2 * a + 1] = s.rb.mn[a]; nd.q[6 + 2 * a] = s.lb.mx[a]; nd.q[6 + 2 * a + 1] = s.rb.mx[a]; }
    if (s.nl > 1) nd.childA = (int32_t)(t.idx + 1);
    if (t.count - s.nl > 1) nd.childB = (int32_t)(t.idx + s.nl);
    nd.pad0 = 0; nd.pad1 = 0;
}
// a child of two objects or more becomes a task of the next level (one wave calls this together)
__device__ void emit(uint32_t idx, uint32_t begin, uint32_t count, uint32_t depth, const float cb[6], uint32_t buf, const Lists& next, int lane) {
    if (count < 2) return;
    Task t; t.idx = idx; t.begin = begin; t.count = count; t.depth = depth; t.buf = buf; t.blk_base = 0; t.axis = 0; t.best_bin = 0; t.nl = 0; t.pad = 0;
    for (int d = 0; d < 6; d++) t.cb[d] = cb[d];
    if (count > SMALL_MAX) {
        const uint32_t nblk = (count + CHUNK - 1) / CHUNK;
        uint32_t slot = 0, base = 0;
        if (lane == 0) { slot = atomicAdd(&next.ctr->n_large, 1u); base = atomicAdd(&next.ctr->n_blocks, nblk); }
        slot = __shfl(slot, 0); base = __shfl(base, 0);
        t.blk_base = base;
        if (lane == 0) next.large[slot] = t;
        for (uint32_t k = lane; k < nblk; k += 64) next.blk_task[base + k] = slot;
    } else if (lane == 0) next.small[atomicAdd(&next.ctr->n_small, 1u)] = t;
}

// ---- set-up
__global__ void __launch_bounds__(TB) k_iota_bounds(const rtbvh::Box* boxes, uint32_t n, uint32_t* order, uint32_t* cb_enc /* [6]: min xyz, max xyz */) {
    __shared__ uint32_t s[6];
    if (threadIdx.x < 6) s[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i < n) {
        order[i] = i;
        float c[3]; centre_of(boxes[i], c);
        for (int d = 0; d < 3; d++) { atomicMin(&s[d], enc(c[d])); atomicMax(&s[3 + d], enc(c[d])); }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&cb_enc[threadIdx.x], s[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&cb_enc[threadIdx.x], s[threadIdx.x]);
}
__global__ void __launch_bounds__(64) k_root(uint32_t n, const uint32_t* cb_enc, Lists first) {
    float cb[6];
    for (int d = 0; d < 6; d++) cb[d] = dec(cb_enc[d]);
    emit(0u, 0u, n, 1u, cb, 0u, first, (int)threadIdx.x);
}

// ---- large nodes
__global__ void __launch_bounds__(TB) k_bins_init(uint32_t* gbins, uint32_t n_words) {
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i < n_words) gbins[i] = bins_word_init((int)(i % BINS_WORDS));
}
__global__ void __launch_bounds__(TB) k_large_bin(const Task* tasks, const uint32_t* blk_task, const rtbvh::Box* boxes, const uint32_t* order0, const uint32_t* order1,
                                                  Bins* gbins, uint32_t* blk_hist /* [blocks][48] */) {
    __shared__ Bins B;
    const uint32_t slot = blk_task[blockIdx.x];
    const Task t = tasks[slot];
    const Plan p = plan_of(t);
    const uint32_t* order = t.buf ? order1 : order0;
    for (int j = threadIdx.x; j < BINS_WORDS; j += TB) ((uint32_t*)&B)[j] = bins_word_init(j);
    __syncthreads();
    const uint32_t cbeg = (blockIdx.x - t.blk_base) * CHUNK, cend = min(cbeg + CHUNK, t.count);
    for (uint32_t i = cbeg + threadIdx.x; i < cend; i += TB) {
        const rtbvh::Box bx = boxes[order[t.begin + i]];
        float c[3]; centre_of(bx, c);
        for (int a = 0; a < 3; a++) if (p.use[a]) bin_add(B, a, bin_of(p, a, c[a], i, t.count), bx, c);
    }
    __syncthreads();
    uint32_t* g = (uint32_t*)(gbins + slot);
    for (int j = threadIdx.x; j < BINS_WORDS; j += TB) {
        const uint32_t v = ((const uint32_t*)&B)[j];
        const int kind = bins_word_kind(j);
        if (kind == 0) { blk_hist[(size_t)blockIdx.x * 48 + j] = v; if (v) atomicAdd(&g[j], v); }
        else if (kind == 1) { if (v != 0xFFFFFFFFu) atomicMin(&g[j], v); }
        else if (v) atomicMax(&g[j], v);
    }
}
__global__ void __launch_bounds__(64) k_large_split(Task* tasks, const Bins* gbins, const uint32_t* blk_hist, uint32_t* blk_left, uint32_t* blk_right,
                                                    rtbvh::HostNode* nodes, Lists next, uint32_t* max_depth) {
    __shared__ Bins B; __shared__ double cost[48]; __shared__ Split sp;
    const int lane = (int)threadIdx.x;
    const Task t = tasks[blockIdx.x];
    const Plan p = plan_of(t);
    for (int j = lane; j < BINS_WORDS; j += 64) ((uint32_t*)&B)[j] = ((const uint32_t*)(gbins + blockIdx.x))[j];
    __syncthreads();
    choose_split(B, p, cost, &sp, lane);
    const Split s = sp;
    if (lane == 0) {
        write_node(nodes, t, s);
        tasks[blockIdx.x].axis = (uint32_t)s.axis; tasks[blockIdx.x].best_bin = (uint32_t)s.bin; tasks[blockIdx.x].nl = s.nl;
        atomicMax(max_depth, t.depth);
    }
    // where each workgroup of the node writes its left-going and its right-going objects: a scan of the workgroups' own bin counts
    const uint32_t nblk = (t.count + CHUNK - 1) / CHUNK;
    uint32_t run_left = 0;
    for (uint32_t base = 0; base < nblk; base += 64) {
        const uint32_t b = base + lane;
        uint32_t l = 0;
        if (b < nblk) for (int k = 0; k <= s.bin; k++) l += blk_hist[(size_t)(t.blk_base + b) * 48 + s.axis * kBins + k];
        uint32_t incl = l;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        const uint32_t lefts_before = run_left + incl - l;
        if (b < nblk) { blk_left[t.blk_base + b] = t.begin + lefts_before; blk_right[t.blk_base + b] = t.begin + s.nl + (b * CHUNK - lefts_before); }
        run_left += __shfl(incl, 63);
    }
    emit(t.idx + 1, t.begin, s.nl, t.depth + 1, s.lcb, t.buf ^ 1u, next, lane);
    emit(t.idx + s.nl, t.begin + s.nl, t.count - s.nl, t.depth + 1, s.rcb, t.buf ^ 1u, next, lane);
}
__global__ void __launch_bounds__(TB) k_large_scatter(const Task* tasks, const uint32_t* blk_task, const uint32_t* blk_left, const uint32_t* blk_right,
                                                      const rtbvh::Box* boxes, const uint32_t* obj, uint32_t* order0, uint32_t* order1, rtbvh::HostNode* nodes) {
    constexpr int PER = (int)(CHUNK / TB);
    __shared__ uint32_t wave_total[TB / 64];
    const Task t = tasks[blk_task[blockIdx.x]];
    const Plan p = plan_of(t);
    const uint32_t* in = t.buf ? order1 : order0;
    uint32_t* out = t.buf ? order0 : order1;
    const uint32_t cbeg = (blockIdx.x - t.blk_base) * CHUNK, cend = min(cbeg + CHUNK, t.count);
    uint32_t ids[PER]; uint32_t left_mask = 0, n_left = 0;
    for (int k = 0; k < PER; k++) {                       // thread t owns PER consecutive positions: the order is kept
        const uint32_t i = cbeg + threadIdx.x * PER + k;
        ids[k] = 0;
        if (i < cend) {
            ids[k] = in[t.begin + i];
            const rtbvh::Box bx = boxes[ids[k]];
            const float c = 0.5f * bx.mn[t.axis] + 0.5f * bx.mx[t.axis];
            if (bin_of(p, (int)t.axis, c, i, t.count) <= (int)t.best_bin) { left_mask |= 1u << k; n_left++; }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = n_left;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    uint32_t lefts_before = incl - n_left;
    for (int w = 0; w < wave; w++) lefts_before += wave_total[w];
    const uint32_t lbase = blk_left[blockIdx.x], rbase = blk_right[blockIdx.x];
    for (int k = 0; k < PER; k++) {
        const uint32_t local = threadIdx.x * PER + k;
        if (cbeg + local >= cend) break;
        const bool left = (left_mask >> k) & 1u;
        out[left ? lbase + lefts_before : rbase + (local - lefts_before)] = ids[k];
        if (left) { lefts_before++; if (t.nl == 1) nodes[t.idx].childA = ~(int32_t)obj[ids[k]]; }
        else if (t.count - t.nl == 1) nodes[t.idx].childB = ~(int32_t)obj[ids[k]];
    }
}

// ---- small nodes: one wave each, the node's range and its objects' bins staged in LDS
__global__ void __launch_bounds__(64) k_small(const Task* tasks, const rtbvh::Box* boxes, const uint32_t* obj, uint32_t* order0, uint32_t* order1,
                                              rtbvh::HostNode* nodes, Lists next, uint32_t* max_depth) {
    __shared__ uint32_t ids[SMALL_MAX]; __shared__ uint16_t kb[SMALL_MAX];
    __shared__ Bins B; __shared__ double cost[48]; __shared__ Split sp;
    const int lane = (int)threadIdx.x;
    const Task t = tasks[blockIdx.x];
    const Plan p = plan_of(t);
    uint32_t* order = t.buf ? order1 : order0;
    for (int j = lane; j < BINS_WORDS; j += 64) ((uint32_t*)&B)[j] = bins_word_init(j);
    __syncthreads();
    for (uint32_t i = lane; i < t.count; i += 64) {
        const uint32_t id = order[t.begin + i];
        const rtbvh::Box bx = boxes[id];
        float c[3]; centre_of(bx, c);
        uint32_t packed = 0;
        for (int a = 0; a < 3; a++) if (p.use[a]) { const int k = bin_of(p, a, c[a], i, t.count); bin_add(B, a, k, bx, c); packed |= (uint32_t)k << (4 * a); }
        ids[i] = id; kb[i] = (uint16_t)packed;
    }
    __syncthreads();
    choose_split(B, p, cost, &sp, lane);
    const Split s = sp;
    uint32_t done_l = 0, done_r = 0;
    for (uint32_t base = 0; base < t.count; base += 64) {           // stable partition, 64 positions at a time
        const uint32_t i = base + lane;
        const bool in = i < t.count;
        const bool left = in && (int)((kb[in ? i : 0] >> (4 * s.axis)) & 15u) <= s.bin;
        const unsigned long long ml = __ballot(left), mr = __ballot(in && !left), lt = (1ull << lane) - 1ull;
        if (in) {
            const uint32_t id = ids[i];
            order[left ? t.begin + done_l + (uint32_t)__popcll(ml & lt) : t.begin + s.nl + done_r + (uint32_t)__popcll(mr & lt)] = id;
            if (left) { if (s.nl == 1) nodes[t.idx].childA = ~(int32_t)obj[id]; }
            else if (t.count - s.nl == 1) nodes[t.idx].childB = ~(int32_t)obj[id];
        }
        done_l += (uint32_t)__popcll(ml); done_r += (uint32_t)__popcll(mr);
    }
    if (lane == 0) { write_node(nodes, t, s); atomicMax(max_depth, t.depth); }
    emit(t.idx + 1, t.begin, s.nl, t.depth + 1, s.lcb, t.buf, next, lane);
    emit(t.idx + s.nl, t.begin + s.nl, t.count - s.nl, t.depth + 1, s.rcb, t.buf, next, lane);
}

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    template <class T> T* as() const { return (T*)p; }
};

} // namespace

#define SAH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? 2 : 1; } } while (0)

// boxes/object_ids: host arrays of the n >= 2 bounded objects; d_nodes: device buffer of n-1 nodes.  0 = ok, 1 = HIP error, 2 = out of memory.
int build(const rtbvh::Box* boxes, const uint32_t* object_ids, uint32_t n, void* d_nodes, uint32_t* max_depth, uint32_t* levels, std::string& err) {
    if (n < 2 || n > 0x3FFFFFFFu) { err = "rtsah::build: object count out of range"; return 1; }
    const uint32_t large_cap = n / SMALL_MAX + 2, blocks_cap = n / CHUNK + large_cap + 2, small_cap = n / 2 + 2;
    Buf d_boxes, d_ids, order0, order1, large[2], small[2], blk_task[2], blk_hist, blk_left, blk_right, gbins, ctr, misc;
    SAH_TRY(d_boxes.alloc((size_t)n * sizeof(rtbvh::Box))); SAH_TRY(d_ids.alloc((size_t)n * 4));
    SAH_TRY(order0.alloc((size_t)n * 4)); SAH_TRY(order1.alloc((size_t)n * 4));
    for (int i = 0; i < 2; i++) {
        SAH_TRY(large[i].alloc((size_t)large_cap * sizeof(Task))); SAH_TRY(small[i].alloc((size_t)small_cap * sizeof(Task)));
        SAH_TRY(blk_task[i].alloc((size_t)blocks_cap * 4));
    }
    SAH_TRY(blk_hist.alloc((size_t)blocks_cap * 48 * 4)); SAH_TRY(blk_left.alloc((size_t)blocks_cap * 4)); SAH_TRY(blk_right.alloc((size_t)blocks_cap * 4));
    SAH_TRY(gbins.alloc((size_t)large_cap * sizeof(Bins)));
    SAH_TRY(ctr.alloc((size_t)(MAX_LEVELS + 1) * sizeof(Counters)));
    SAH_TRY(misc.alloc(8 * 4));                              // [0..5] the root's centre bounds (encoded), [6] the deepest level
    SAH_TRY(hipMemcpy(d_boxes.p, boxes, (size_t)n * sizeof(rtbvh::Box), hipMemcpyHostToDevice));
    SAH_TRY(hipMemcpy(d_ids.p, object_ids, (size_t)n * 4, hipMemcpyHostToDevice));
    SAH_TRY(hipMemset(ctr.p, 0, (size_t)(MAX_LEVELS + 1) * sizeof(Counters)));
    {
        const uint32_t init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
        SAH_TRY(hipMemcpy(misc.p, init, sizeof(init), hipMemcpyHostToDevice));
    }
    SAH_TRY(hipMemset(d_nodes, 0, (size_t)(n - 1) * sizeof(rtbvh::HostNode)));
    const rtbvh::Box* bx = d_boxes.as<rtbvh::Box>();
    const uint32_t* ob = d_ids.as<uint32_t>();
    uint32_t *o0 = order0.as<uint32_t>(), *o1 = order1.as<uint32_t>(), *deepest = misc.as<uint32_t>() + 6;
    rtbvh::HostNode* nodes = (rtbvh::HostNode*)d_nodes;
    Counters* counters = ctr.as<Counters>();
    auto lists = [&](int which, int level) { Lists l; l.large = large[which].as<Task>(); l.small = small[which].as<Task>(); l.blk_task = blk_task[which].as<uint32_t>(); l.ctr = counters + level; return l; };

    hipLaunchKernelGGL(k_iota_bounds, dim3((n + TB - 1) / TB), dim3(TB), 0, 0, bx, n, o0, misc.as<uint32_t>());
    hipLaunchKernelGGL(k_root, dim3(1), dim3(64), 0, 0, n, misc.as<uint32_t>(), lists(0, 0));
    int level = 0;
    for (;; level++) {
        if (level >= MAX_LEVELS) { err = "rtsah::build: the tree did not finish in 128 levels"; return 1; }
        Counters c;
        SAH_TRY(hipMemcpy(&c, counters + level, sizeof(c), hipMemcpyDeviceToHost));
        if (!c.n_large && !c.n_small) break;
        if (c.n_large > large_cap || c.n_blocks > blocks_cap || c.n_small > small_cap) { err = "rtsah::build: task list overflow"; return 1; }
        const int cur = level & 1;
        const Lists here = lists(cur, level), next = lists(cur ^ 1, level + 1);
        if (c.n_large) {
            const uint32_t words = c.n_large * BINS_WORDS;
            hipLaunchKernelGGL(k_bins_init, dim3((words + TB - 1) / TB), dim3(TB), 0, 0, gbins.as<uint32_t>(), words);
            hipLaunchKernelGGL(k_large_bin, dim3(c.n_blocks), dim3(TB), 0, 0, here.large, here.blk_task, bx, o0, o1, gbins.as<Bins>(), blk_hist.as<uint32_t>());
            hipLaunchKernelGGL(k_large_split, dim3(c.n_large), dim3(64), 0, 0, here.large, gbins.as<Bins>(), blk_hist.as<uint32_t>(), blk_left.as<uint32_t>(),
                               blk_right.as<uint32_t>(), nodes, next, deepest);
            hipLaunchKernelGGL(k_large_scatter, dim3(c.n_blocks), dim3(TB), 0, 0, here.large, here.blk_task, blk_left.as<uint32_t>(), blk_right.as<uint32_t>(), bx, ob, o0, o1, nodes);
        }
        if (c.n_small) hipLaunchKernelGGL(k_small, dim3(c.n_small), dim3(64), 0, 0, here.small, bx, ob, o0, o1, nodes, next, deepest);
        SAH_TRY(hipGetLastError());
    }
    SAH_TRY(hipDeviceSynchronize());
    SAH_TRY(hipMemcpy(max_depth, deepest, 4, hipMemcpyDeviceToHost));
    if (levels) *levels = (uint32_t)level;
    return 0;
}

} // namespace rtsah
