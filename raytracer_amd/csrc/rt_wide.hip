// rt_wide.hip — builds the 8-wide tree with quantised inner boxes (rt_wide.h) from the binary tree, on the device.
//
// Top-down, one level per round: an open item is a binary subtree that becomes one wide node.  Its children are found by opening,
// again and again, the binary child with the largest surface area until there are eight (or only leaves are left) — the collapse
// of Wald et al. 2008 / Ylitie et al. 2017.  Per level: (1) every item counts its inner children and leaves, (2) an inclusive scan
// turns the counts into the first child node / first leaf record of every item, (3) every item makes its children again (the same
// deterministic procedure), assigns them to slots by octant, quantises their boxes outward in the node's frame, writes the node,
// the leaf records (exact box, object, code, and a copy of the primitive when the tree holds one type) and the next level's items.
// Node numbers and leaf positions follow from the scans alone: the same arrays on every run.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "rt_wide.h"

namespace rtwide {
namespace {

using rtd::Node;
constexpr int TB = 256;
constexpr int MAX_LEVELS = 64;

struct Cand { int ref; uint32_t leaves; float mn[3], mx[3]; };      // a child: >= 0 binary inner node, < 0 leaf (~object); the leaves below it

// Leaves below every binary inner node, bottom-up: a thread starts at each node without inner children and climbs; at a node with
// two inner children the second arrival goes on (the first one's count is visible to it: fence, agent-scope load).
__global__ void __launch_bounds__(TB) k_parents(const Node* nodes, uint32_t n_nodes, int* parent) {
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n_nodes) return;
    const int a = nodes[i].q3.x, b = nodes[i].q3.y;
    if (a >= 0) parent[a] = (int)i;
    if (b >= 0) parent[b] = (int)i;
}
__device__ __forceinline__ uint32_t leaves_of_child(int ref, const uint32_t* leaf_count) {
    return ref == rtd::CHILD_EMPTY ? 0u : ref < 0 ? 1u : __hip_atomic_load(leaf_count + ref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void __launch_bounds__(TB) k_leaf_counts(const Node* nodes, uint32_t n_nodes, const int* parent, uint32_t* arrivals, uint32_t* leaf_count) {
    int i = (int)(blockIdx.x * TB + threadIdx.x);
    if (i >= (int)n_nodes) return;
    if (nodes[i].q3.x >= 0 || nodes[i].q3.y >= 0) return;           // starts only where both children are leaves (or empty)
    for (;;) {
        const int a = nodes[i].q3.x, b = nodes[i].q3.y;
        __hip_atomic_store(leaf_count + i, leaves_of_child(a, leaf_count) + leaves_of_child(b, leaf_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int p = parent[i];
        if (p < 0) return;
        const uint32_t needed = (nodes[p].q3.x >= 0 ? 1u : 0u) + (nodes[p].q3.y >= 0 ? 1u : 0u);
        __threadfence();
        if (atomicAdd(&arrivals[p], 1u) + 1u < needed) return;        // the sibling subtree is not finished: its thread will go on
        __threadfence();
        i = p;
    }
}

__device__ __forceinline__ void child_of(const Node* nodes, const uint32_t* leaf_count, int b, int side, Cand& c) {
    const Node& n = nodes[b];
    const float4 q0 = n.q0, q1 = n.q1, q2 = n.q2;
    c.ref = side ? n.q3.y : n.q3.x;
    c.leaves = c.ref == rtd::CHILD_EMPTY ? 0u : c.ref < 0 ? 1u : leaf_count[c.ref];
    c.mn[0] = side ? q0.y : q0.x; c.mn[1] = side ? q0.w : q0.z; c.mn[2] = side ? q1.y : q1.x;
    c.mx[0] = side ? q1.w : q1.z; c.mx[1] = side ? q2.y : q2.x; c.mx[2] = side ? q2.w : q2.z;
}
__device__ __forceinline__ float half_area(const Cand& c) {
    const float dx = c.mx[0] - c.mn[0], dy = c.mx[1] - c.mn[1], dz = c.mx[2] - c.mn[2];
    return dx * dy + dy * dz + dz * dx;
}
// The (up to) eight children of the wide node made from binary node b.  First the binary children that hold more than eight
// leaves are opened, largest surface area first (the collapse of Wald et al. 2008): big subtrees become inner children.  Then
// subtrees of at most eight leaves: one that fits into the free slots WHOLE is opened (fewest leaves first — it saves a node and a
// level for one slot per leaf), one that does not fit stays closed — it becomes one well-filled node of its own, where opening it
// halfway would leave two nodes of two or three leaves each (5 M triangles: 2.07 M nodes of 3.4 children with area order alone).
__device__ int gather_children(const Node* nodes, const uint32_t* leaf_count, int b, Cand* c) {
    int n = 0;
    child_of(nodes, leaf_count, b, 0, c[n]); if (c[n].ref != rtd::CHILD_EMPTY) n++;
    child_of(nodes, leaf_count, b, 1, c[n]); if (c[n].ref != rtd::CHILD_EMPTY) n++;
    while (n < 8) {
        int best = -1; float ba = -1.0f;
        for (int i = 0; i < n; i++) if (c[i].ref >= 0 && c[i].leaves > 8u) { const float a = half_area(c[i]); if (a > ba) { ba = a; best = i; } }
        if (best < 0) {
            uint32_t fewest = 0xFFFFFFFFu;
            for (int i = 0; i < n; i++) if (c[i].ref >= 0 && (uint32_t)n + c[i].leaves - 1u <= 8u && c[i].leaves < fewest) { fewest = c[i].leaves; best = i; }
        }
        if (best < 0) break;
        const int ref = c[best].ref;
        child_of(nodes, leaf_count, ref, 0, c[best]);
        child_of(nodes, leaf_count, ref, 1, c[n]);
        if (c[n].ref != rtd::CHILD_EMPTY) n++;       // (inner nodes of a tree of two objects or more always have two children)
    }
    return n;
}

__global__ void __launch_bounds__(TB) k_count(const Node* nodes, const uint32_t* leaf_count, const int* items, uint32_t n_items, unsigned long long* counts) {
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n_items) return;
    Cand c[8];
    const int n = gather_children(nodes, leaf_count, items[i], c);
    uint32_t inner = 0;
    for (int k = 0; k < n; k++) inner += c[k].ref >= 0 ? 1u : 0u;
    counts[i] = ((unsigned long long)inner << 32) | (uint32_t)(n - (int)inner);
}

struct Tables {          // what a leaf record copies
    const uint32_t* obj_code; uint32_t code_affine, code_base;
    const float4* spheres; const float4* triangles; const float4* quads;
    uint32_t inline_type; uint32_t leaf_stride;
};

__global__ void __launch_bounds__(TB) k_write(const Node* nodes, const uint32_t* leaf_count, const int* items, uint32_t n_items, const unsigned long long* incl, uint32_t first_node, uint32_t first_child, uint32_t first_leaf,
                                              char* wide_nodes, uint32_t node_stride, char* leaves, Tables tb, int* next_items) {
    const uint32_t i = blockIdx.x * TB + threadIdx.x;
    if (i >= n_items) return;
    Cand c[8];
    const int n = gather_children(nodes, leaf_count, items[i], c);
    uint32_t n_inner = 0;
    for (int k = 0; k < n; k++) n_inner += c[k].ref >= 0 ? 1u : 0u;
    const unsigned long long before = incl[i] - (((unsigned long long)n_inner << 32) | (uint32_t)(n - (int)n_inner));
    const uint32_t child_base = first_child + (uint32_t)(before >> 32), leaf_base = first_leaf + (uint32_t)before;
    // the node's frame
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < n; k++) for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], c[k].mn[a]); hi[a] = fmaxf(hi[a], c[k].mx[a]); }
    // slots by octant: child k to the free slot whose sign vector agrees best with (centre of k - centre of the node); the pair with
    // the largest agreement first, ties to the lower child and slot
    int slot_of[8]; bool slot_used[8];
    for (int k = 0; k < 8; k++) { slot_of[k] = -1; slot_used[k] = false; }
    for (int round = 0; round < n; round++) {
        float best = -INFINITY; int bk = -1, bs = -1;
        for (int k = 0; k < n; k++) {
            if (slot_of[k] >= 0) continue;
            float d[3];
            for (int a = 0; a < 3; a++) d[a] = (0.5f * c[k].mn[a] + 0.5f * c[k].mx[a]) - (0.5f * lo[a] + 0.5f * hi[a]);
            for (int s = 0; s < 8; s++) {
                if (slot_used[s]) continue;
                const float v = ((s & 1) ? d[0] : -d[0]) + ((s & 2) ? d[1] : -d[1]) + ((s & 4) ? d[2] : -d[2]);
                if (v > best) { best = v; bk = k; bs = s; }
            }
        }
        if (bk < 0) {                                   // (every score NaN or -inf: non-finite boxes never reach the tree; any free pairing will do)
            for (int k = 0; k < n && bk < 0; k++) if (slot_of[k] < 0) bk = k;
            for (int s = 0; s < 8 && bs < 0; s++) if (!slot_used[s]) bs = s;
        }
        slot_of[bk] = bs; slot_used[bs] = true;
    }
    int child_in[8];
    for (int s = 0; s < 8; s++) child_in[s] = -1;
    for (int k = 0; k < n; k++) child_in[slot_of[k]] = k;
    // per axis: the smallest power of two with (hi - lo) / 2^e <= 255, then every plane rounded outward — checked in double, where
    // org + q * 2^e is exact: a decoded low plane is never above the child's, a decoded high plane never below
    uint32_t eb[3]; unsigned char qlo[3][8], qhi[3][8];
    for (int a = 0; a < 3; a++) {
        const double ext = (double)hi[a] - (double)lo[a];
        int e = -126;
        if (ext > 0.0) { int k; frexp(ext / 255.0, &k); e = k; }      // 2^k >= ext / 255
        if (e < -126) e = -126;
        for (;;) {
            const double s = ldexp(1.0, e);
            bool ok = true;
            for (int sl = 0; sl < 8 && ok; sl++) {
                const int k = child_in[sl];
                if (k < 0) { qlo[a][sl] = 255; qhi[a][sl] = 0; continue; }
                double ql = floor(((double)c[k].mn[a] - (double)lo[a]) / s), qh = ceil(((double)c[k].mx[a] - (double)lo[a]) / s);
                while (ql > 0.0 && (double)lo[a] + ql * s > (double)c[k].mn[a]) ql -= 1.0;
                while ((double)lo[a] + qh * s < (double)c[k].mx[a]) qh += 1.0;
                if (ql < 0.0) ql = 0.0;
                if (qh > 255.0) { ok = false; break; }
                qlo[a][sl] = (unsigned char)ql; qhi[a][sl] = (unsigned char)qh;
            }
            if (ok || e >= 127) break;
            e++;
        }
        eb[a] = (uint32_t)(e + 127);
    }
    uint32_t imask = 0, lmask = 0;
    for (int s = 0; s < 8; s++) if (child_in[s] >= 0) { if (c[child_in[s]].ref >= 0) imask |= 1u << s; else lmask |= 1u << s; }
    // the node record
    uint32_t* np = (uint32_t*)(wide_nodes + (size_t)(first_node + i) * node_stride);      // (rt_wide.h has the layout; records are 8-B aligned)
    auto pack4 = [](const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
    np[0] = __float_as_uint(lo[0]); np[1] = __float_as_uint(lo[1]); np[2] = __float_as_uint(lo[2]); np[3] = (eb[0] << 23) | imask;
    np[4] = child_base; np[5] = leaf_base; np[6] = (eb[1] << 23) | lmask; np[7] = eb[2] << 23;
    for (int a = 0; a < 3; a++) {
        uint32_t* ap = np + 8 + 6 * a;
        ap[0] = pack4(qlo[a]); ap[1] = pack4(qlo[a] + 4); ap[2] = pack4(qhi[a]); ap[3] = pack4(qhi[a] + 4); ap[4] = ap[0]; ap[5] = ap[1];
    }
    // children in slot order: inner ones become the next level's items, leaves their records
    uint32_t ki = 0, kl = 0;
    for (int s = 0; s < 8; s++) {
        const int k = child_in[s];
        if (k < 0) continue;
        if (c[k].ref >= 0) { next_items[(child_base - first_child) + ki] = c[k].ref; ki++; continue; }
        const uint32_t obj = (uint32_t)~c[k].ref;
        const uint32_t code = tb.code_affine ? tb.code_base + obj : tb.obj_code[obj];
        float4* lp = (float4*)(leaves + (size_t)(leaf_base + kl) * tb.leaf_stride);
        lp[0] = make_float4(c[k].mn[0], c[k].mn[1], c[k].mn[2], __uint_as_float(obj));
        lp[1] = make_float4(c[k].mx[0], c[k].mx[1], c[k].mx[2], __uint_as_float(code));
        const uint32_t idx = code & rtd::OBJ_INDEX_MASK;
        if (tb.inline_type == RT_HIT_SPHERE) lp[2] = tb.spheres[idx];
        else if (tb.inline_type == RT_HIT_TRIANGLE || tb.inline_type == RT_HIT_QUAD) {
            const float4* q = (tb.inline_type == RT_HIT_TRIANGLE ? tb.triangles : tb.quads) + 4 * (size_t)idx;
            lp[2] = q[0]; lp[3] = q[1]; lp[4] = q[2]; lp[5] = q[3];
        }
        kl++;
    }
}

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); rc = e_ == hipErrorOutOfMemory ? 2 : 1; goto done; } } while (0)

} // namespace

// nodes: the binary tree on the device (n_nodes = n_objects_in_tree - 1 >= 1).  Outputs are hipMalloc'ed here and owned by the caller.
// Returns 0, 1 (HIP error) or 2 (out of memory).
int build(const void* d_nodes, uint32_t n_nodes, const rtd::SceneDev& tables, uint32_t inline_type, uint32_t node_stride,
          void** out_nodes, size_t* out_node_bytes, void** out_leaves, size_t* out_leaf_bytes, uint32_t* out_leaf_stride, uint32_t* out_n_wide, uint32_t* out_levels, std::string& err) {
    int rc = 0;
    const uint32_t n_leaves = n_nodes + 1;
    const uint32_t prim_bytes = inline_type == RT_HIT_SPHERE ? 16u : (inline_type == RT_HIT_TRIANGLE || inline_type == RT_HIT_QUAD) ? 64u : 0u;
    const uint32_t leaf_stride = rtd::WIDE_LEAF_BOX_BYTES + prim_bytes;
    void *tmp_nodes = nullptr, *leaves = nullptr, *items[2] = {nullptr, nullptr}, *counts = nullptr, *scan_tmp = nullptr, *final_nodes = nullptr, *parent = nullptr, *arrivals = nullptr, *leaf_count = nullptr;
    const uint32_t node_grid = (n_nodes + TB - 1) / TB;
    size_t scan_bytes = 0;
    uint32_t n_items = 1, first_node = 0, leaves_done = 0, levels = 0;
    Tables tb{tables.obj_code, tables.leaf_code_affine, tables.leaf_code_base, tables.spheres, tables.triangles, tables.quads, inline_type, leaf_stride};
    *out_nodes = nullptr; *out_leaves = nullptr;
    // at most one wide node per binary inner node
    HIP_OK(hipMalloc(&tmp_nodes, (size_t)n_nodes * node_stride));
    HIP_OK(hipMalloc(&leaves, (size_t)n_leaves * leaf_stride));
    HIP_OK(hipMalloc(&items[0], (size_t)n_nodes * sizeof(int)));
    HIP_OK(hipMalloc(&items[1], (size_t)n_nodes * sizeof(int)));
    HIP_OK(hipMalloc(&counts, (size_t)n_nodes * sizeof(unsigned long long)));
    HIP_OK(hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, (unsigned long long*)counts, (unsigned long long*)counts, (int)n_nodes));
    HIP_OK(hipMalloc(&scan_tmp, scan_bytes));
    HIP_OK(hipMemset(items[0], 0, sizeof(int)));                 // level 0: the root
    HIP_OK(hipMalloc(&parent, (size_t)n_nodes * sizeof(int)));
    HIP_OK(hipMalloc(&arrivals, (size_t)n_nodes * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&leaf_count, (size_t)n_nodes * sizeof(uint32_t)));
    HIP_OK(hipMemset(parent, 0xFF, (size_t)n_nodes * sizeof(int)));
    HIP_OK(hipMemset(arrivals, 0, (size_t)n_nodes * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_parents, dim3(node_grid), dim3(TB), 0, 0, (const Node*)d_nodes, n_nodes, (int*)parent);
    hipLaunchKernelGGL(k_leaf_counts, dim3(node_grid), dim3(TB), 0, 0, (const Node*)d_nodes, n_nodes, (const int*)parent, (uint32_t*)arrivals, (uint32_t*)leaf_count);
    HIP_OK(hipGetLastError());
    for (int cur = 0;; cur ^= 1) {
        if (levels >= (uint32_t)MAX_LEVELS) { err = "wide tree deeper than 64 levels"; rc = 1; goto done; }
        const uint32_t grid = (n_items + TB - 1) / TB;
        hipLaunchKernelGGL(k_count, dim3(grid), dim3(TB), 0, 0, (const Node*)d_nodes, (const uint32_t*)leaf_count, (const int*)items[cur], n_items, (unsigned long long*)counts);
        HIP_OK(hipGetLastError());
        size_t sb = scan_bytes;
        HIP_OK(hipcub::DeviceScan::InclusiveSum(scan_tmp, sb, (unsigned long long*)counts, (unsigned long long*)counts, (int)n_items));
        unsigned long long total = 0;
        HIP_OK(hipMemcpy(&total, (unsigned long long*)counts + (n_items - 1), sizeof(total), hipMemcpyDeviceToHost));
        const uint32_t n_inner = (uint32_t)(total >> 32), n_leaf = (uint32_t)total;
        const uint32_t first_child = first_node + n_items;
        if ((size_t)first_child + n_inner > (size_t)n_nodes || (size_t)leaves_done + n_leaf > (size_t)n_leaves) { err = "wide tree: counts exceed the binary tree's"; rc = 1; goto done; }
        hipLaunchKernelGGL(k_write, dim3(grid), dim3(TB), 0, 0, (const Node*)d_nodes, (const uint32_t*)leaf_count, (const int*)items[cur], n_items, (const unsigned long long*)counts, first_node, first_child, leaves_done,
                           (char*)tmp_nodes, node_stride, (char*)leaves, tb, (int*)items[cur ^ 1]);
        HIP_OK(hipGetLastError());
        levels++;
        first_node = first_child; leaves_done += n_leaf; n_items = n_inner;
        if (!n_items) break;
    }
    HIP_OK(hipDeviceSynchronize());
    if (leaves_done != n_leaves) { err = "wide tree: leaf count differs from the binary tree's"; rc = 1; goto done; }
    // the node array at its real size (first_node = number of wide nodes)
    HIP_OK(hipMalloc(&final_nodes, (size_t)first_node * node_stride));
    HIP_OK(hipMemcpy(final_nodes, tmp_nodes, (size_t)first_node * node_stride, hipMemcpyDeviceToDevice));
    *out_nodes = final_nodes; final_nodes = nullptr; *out_node_bytes = (size_t)first_node * node_stride;
    *out_leaves = leaves; leaves = nullptr; *out_leaf_bytes = (size_t)n_leaves * leaf_stride; *out_leaf_stride = leaf_stride;
    *out_n_wide = first_node; *out_levels = levels;
done:
    for (void* p : {tmp_nodes, leaves, items[0], items[1], counts, scan_tmp, final_nodes, parent, arrivals, leaf_count}) if (p) (void)hipFree(p);
    if (rc) (void)hipGetLastError();
    return rc;
}

} // namespace rtwide
