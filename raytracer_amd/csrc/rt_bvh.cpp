// rt_bvh.cpp — binned-SAH BVH2 builder (see rt_bvh.h).
#include "rt_bvh.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace rtbvh {
namespace {

constexpr int kBins = 16;
constexpr uint32_t kMaxSahDepth = 40;   // beyond this, split at the median so the traversal stack stays short

struct Work { uint32_t begin, end; int32_t parent; int side; uint32_t depth; };

inline void grow(Box& b, const Box& o) {
    for (int a = 0; a < 3; a++) { b.mn[a] = std::min(b.mn[a], o.mn[a]); b.mx[a] = std::max(b.mx[a], o.mx[a]); }
}
inline Box empty_box() {
    Box b; for (int a = 0; a < 3; a++) { b.mn[a] = std::numeric_limits<float>::infinity(); b.mx[a] = -std::numeric_limits<float>::infinity(); } return b;
}
inline double half_area(const Box& b) {
    double dx = (double)b.mx[0] - b.mn[0], dy = (double)b.mx[1] - b.mn[1], dz = (double)b.mx[2] - b.mn[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0;
    return dx * dy + dy * dz + dz * dx;
}

} // namespace

BuildResult build(const std::vector<Box>& boxes, const std::vector<uint32_t>& object_ids) {
    BuildResult res;
    const uint32_t n = (uint32_t)boxes.size();
    if (n == 0) return res;

    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::vector<float> cx(n), cy(n), cz(n);
    for (uint32_t i = 0; i < n; i++) {
        cx[i] = 0.5f * boxes[i].mn[0] + 0.5f * boxes[i].mx[0];
        cy[i] = 0.5f * boxes[i].mn[1] + 0.5f * boxes[i].mx[1];
        cz[i] = 0.5f * boxes[i].mn[2] + 0.5f * boxes[i].mx[2];
    }
    const float* cen[3] = {cx.data(), cy.data(), cz.data()};

    res.nodes.reserve(n > 1 ? n - 1 : 1);
    auto set_child = [&](int32_t parent, int side, int32_t code, const Box& b) {
        HostNode& p = res.nodes[parent];
        for (int a = 0; a < 3; a++) { p.q[2 * a + side] = b.mn[a]; p.q[6 + 2 * a + side] = b.mx[a]; }
        (side ? p.childB : p.childA) = code;
    };
    auto new_node = [&]() -> int32_t {
        HostNode nd; std::memset(&nd, 0, sizeof(nd));
        Box e = empty_box();
        for (int s = 0; s < 2; s++) for (int a = 0; a < 3; a++) { nd.q[2 * a + s] = e.mn[a]; nd.q[6 + 2 * a + s] = e.mx[a]; }
        nd.childA = CHILD_EMPTY; nd.childB = CHILD_EMPTY;
        res.nodes.push_back(nd);
        return (int32_t)res.nodes.size() - 1;
    };

    if (n == 1) {      // a single bounded object: a root whose only child is the leaf
        int32_t r = new_node();
        set_child(r, 0, ~(int32_t)object_ids[0], boxes[0]);
        res.max_depth = 1;
        return res;
    }

    std::vector<Work> stack;
    stack.push_back({0, n, -1, 0, 1});
    while (!stack.empty()) {
        Work w = stack.back(); stack.pop_back();
        const uint32_t cnt = w.end - w.begin;
        Box bounds = empty_box();
        for (uint32_t i = w.begin; i < w.end; i++) grow(bounds, boxes[order[i]]);
        if (cnt == 1) {
            set_child(w.parent, w.side, ~(int32_t)object_ids[order[w.begin]], boxes[order[w.begin]]);
            continue;
        }
        int32_t me = new_node();
        if (w.parent >= 0) set_child(w.parent, w.side, me, bounds);
        res.max_depth = std::max(res.max_depth, w.depth);

        // choose the split
        uint32_t mid = w.begin + cnt / 2;
        bool have_split = false;
        if (cnt > 2 && w.depth < kMaxSahDepth) {
            float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = w.begin; i < w.end; i++)
                for (int a = 0; a < 3; a++) { float c = cen[a][order[i]]; cmn[a] = std::min(cmn[a], c); cmx[a] = std::max(cmx[a], c); }
            double best_cost = std::numeric_limits<double>::infinity(); int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; a++) {
                float ext = cmx[a] - cmn[a];
                if (!(ext > 0.0f) || !std::isfinite(ext)) continue;
                Box bb[kBins]; uint32_t bc[kBins];
                for (int b = 0; b < kBins; b++) { bb[b] = empty_box(); bc[b] = 0; }
                const float scale = (float)kBins / ext;
                for (uint32_t i = w.begin; i < w.end; i++) {
                    int b = (int)((cen[a][order[i]] - cmn[a]) * scale); b = std::min(std::max(b, 0), kBins - 1);
                    grow(bb[b], boxes[order[i]]); bc[b]++;
                }
                double right_area[kBins]; uint32_t right_cnt[kBins];
                Box acc = empty_box(); uint32_t c = 0;
                for (int b = kBins - 1; b > 0; b--) { grow(acc, bb[b]); c += bc[b]; right_area[b] = half_area(acc); right_cnt[b] = c; }
                acc = empty_box(); c = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    grow(acc, bb[b]); c += bc[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    double cost = half_area(acc) * c + right_area[b + 1] * right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const int a = best_axis; const float ext = cmx[a] - cmn[a]; const float scale = (float)kBins / ext;
                auto it = std::partition(order.begin() + w.begin, order.begin() + w.end, [&](uint32_t id) {
                    int b = (int)((cen[a][id] - cmn[a]) * scale); b = std::min(std::max(b, 0), kBins - 1);
                    return b <= best_bin;
                });
                mid = (uint32_t)(it - order.begin());
                have_split = mid > w.begin && mid < w.end;
            }
        }
        if (!have_split) {   // median along the widest centroid axis (or plain halves when everything coincides)
            float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = w.begin; i < w.end; i++)
                for (int a = 0; a < 3; a++) { float c = cen[a][order[i]]; cmn[a] = std::min(cmn[a], c); cmx[a] = std::max(cmx[a], c); }
            int a = 0; float e = cmx[0] - cmn[0];
            for (int k = 1; k < 3; k++) if (cmx[k] - cmn[k] > e) { e = cmx[k] - cmn[k]; a = k; }
            mid = w.begin + cnt / 2;
            std::nth_element(order.begin() + w.begin, order.begin() + mid, order.begin() + w.end,
                             [&](uint32_t l, uint32_t r) { return cen[a][l] < cen[a][r] || (cen[a][l] == cen[a][r] && l < r); });
        }
        stack.push_back({mid, w.end, me, 1, w.depth + 1});
        stack.push_back({w.begin, mid, me, 0, w.depth + 1});   // left is processed first -> left subtree is contiguous
    }
    return res;
}

} // namespace rtbvh
