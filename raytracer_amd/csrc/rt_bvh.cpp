// rt_bvh.cpp — SAH BVH2 builder (see rt_bvh.h): 16 bins per axis for big nodes, every split position for nodes of <= 4096 objects.
#include "rt_bvh.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <atomic>
#include <limits>
#include <thread>

namespace rtbvh {
namespace {

constexpr int kBins = 16;
#ifndef RT_BVH_SWEEP_MAX
#define RT_BVH_SWEEP_MAX 4096
#endif
constexpr uint32_t kSweepMax = RT_BVH_SWEEP_MAX;   // nodes of at most this many objects try every split position (full-sweep SAH), not 15 bin borders
constexpr uint32_t kMaxSahDepth = 40;   // beyond this, split at the median so the traversal stack stays short

// `node` = index the subtree's root will get: with one object per leaf a subtree over c objects has exactly c-1 inner nodes, so the
// preorder index of every node is known before its subtree is built — subtrees can be built by different threads into one array
// and the result is the same array whatever the number of threads.
struct Work { uint32_t begin, end; int32_t parent; int side; uint32_t depth; int32_t node; bool has_bounds; Box bounds; };
constexpr uint32_t kParallelMin = 1u << 13;   // objects below which a subtree is not worth a thread
constexpr int kParallelLevels = 6;            // at most 2^6 subtree threads
constexpr uint32_t kTeamMin = 1u << 19;       // objects from which one node's passes are split between threads

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

// fn(t, begin, end) over `team` contiguous chunks of [begin, end), chunk 0 on the calling thread
template <class F> void team_for(unsigned team, uint32_t begin, uint32_t end, F fn) {
    if (team <= 1) { fn(0u, begin, end); return; }
    const uint64_t n = end - begin;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < team; t++) th.emplace_back(fn, t, begin + (uint32_t)(n * t / team), begin + (uint32_t)(n * (t + 1) / team));
    fn(0u, begin, begin + (uint32_t)(n / team));
    for (auto& x : th) x.join();
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

    {   // every node starts with two empty children
        HostNode nd; std::memset(&nd, 0, sizeof(nd));
        Box e = empty_box();
        for (int s = 0; s < 2; s++) for (int a = 0; a < 3; a++) { nd.q[2 * a + s] = e.mn[a]; nd.q[6 + 2 * a + s] = e.mx[a]; }
        nd.childA = CHILD_EMPTY; nd.childB = CHILD_EMPTY;
        res.nodes.assign(n > 1 ? n - 1 : 1, nd);
    }
    auto set_child = [&](int32_t parent, int side, int32_t code, const Box& b) {      // the two sides of a node are distinct floats: no race
        HostNode& p = res.nodes[parent];
        for (int a = 0; a < 3; a++) { p.q[2 * a + side] = b.mn[a]; p.q[6 + 2 * a + side] = b.mx[a]; }
        (side ? p.childB : p.childA) = code;
    };

    if (n == 1) {      // a single bounded object: a root whose only child is the leaf
        set_child(0, 0, ~(int32_t)object_ids[0], boxes[0]);
        res.max_depth = 1;
        return res;
    }

    std::atomic<uint32_t> max_depth{0};
    const unsigned hw = std::thread::hardware_concurrency();
    const int par_levels = hw > 1 ? kParallelLevels : 0;

    // builds the subtree `first` (and everything below it that is not handed to another thread)
    std::function<void(Work, int)> subtree = [&](Work first, int level) {
      std::vector<std::thread> helpers;
      uint32_t deepest = 0;
      std::vector<Work> stack;
      stack.push_back(first);
      while (!stack.empty()) {
        Work w = stack.back(); stack.pop_back();
        const uint32_t cnt = w.end - w.begin;
        if (cnt == 1) {
            set_child(w.parent, w.side, ~(int32_t)object_ids[order[w.begin]], boxes[order[w.begin]]);
            continue;
        }
        // big nodes near the root: their passes over the objects are shared between the threads that have no subtree yet
        const unsigned team = cnt >= kTeamMin && level < par_levels ? std::max(1u, std::min(hw >> level, 16u)) : 1u;
        if (!w.has_bounds) {
            Box part[16];
            team_for(team, w.begin, w.end, [&](unsigned t, uint32_t b, uint32_t e) { Box x = empty_box(); for (uint32_t i = b; i < e; i++) grow(x, boxes[order[i]]); part[t] = x; });
            w.bounds = empty_box();
            for (unsigned t = 0; t < team; t++) grow(w.bounds, part[t]);
        }
        const int32_t me = w.node;
        if (w.parent >= 0) set_child(w.parent, w.side, me, w.bounds);
        deepest = std::max(deepest, w.depth);
        // choose the split (min/max and counts do not depend on the order of the objects: the same tree for any team size)
        uint32_t mid = w.begin + cnt / 2;
        bool have_split = false;
        Box child_bounds[2]; bool have_child_bounds = false;
        if (cnt > 2 && cnt <= kSweepMax && w.depth < kMaxSahDepth) {
            // full sweep: per axis sort by centroid (ties by object index: deterministic), cost of every prefix | suffix split
            std::vector<uint32_t> ids(order.begin() + w.begin, order.begin() + w.end), best_ids;
            std::vector<double> right_area(cnt);
            double best_cost = std::numeric_limits<double>::infinity(); uint32_t best_k = 0; Box best_l = empty_box(), best_r = empty_box();
            for (int a = 0; a < 3; a++) {
                std::sort(ids.begin(), ids.end(), [&](uint32_t l, uint32_t r) { return cen[a][l] < cen[a][r] || (cen[a][l] == cen[a][r] && l < r); });
                Box acc = empty_box();
                for (uint32_t k = cnt - 1; k > 0; k--) { grow(acc, boxes[ids[k]]); right_area[k] = half_area(acc); }
                acc = empty_box();
                bool better = false; uint32_t kk = 0; Box lb = empty_box();
                for (uint32_t k = 1; k < cnt; k++) {          // left = ids[0..k), right = ids[k..cnt)
                    grow(acc, boxes[ids[k - 1]]);
                    const double cost = half_area(acc) * k + right_area[k] * (cnt - k);
                    if (cost < best_cost) { best_cost = cost; kk = k; lb = acc; better = true; }
                }
                if (better) {
                    best_k = kk; best_l = lb; best_ids = ids;
                    best_r = empty_box(); for (uint32_t k = kk; k < cnt; k++) grow(best_r, boxes[ids[k]]);
                }
            }
            if (best_k > 0) {
                std::copy(best_ids.begin(), best_ids.end(), order.begin() + w.begin);
                mid = w.begin + best_k; have_split = true;
                child_bounds[0] = best_l; child_bounds[1] = best_r; have_child_bounds = true;
            }
        } else if (cnt > 2 && w.depth < kMaxSahDepth) {
            float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
            {
                float pmn[16][3], pmx[16][3];
                team_for(team, w.begin, w.end, [&](unsigned t, uint32_t b, uint32_t e) {
                    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                    for (uint32_t i = b; i < e; i++) { const uint32_t id = order[i]; for (int a = 0; a < 3; a++) { float c = cen[a][id]; lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c); } }
                    for (int a = 0; a < 3; a++) { pmn[t][a] = lo[a]; pmx[t][a] = hi[a]; }
                });
                for (unsigned t = 0; t < team; t++) for (int a = 0; a < 3; a++) { cmn[a] = std::min(cmn[a], pmn[t][a]); cmx[a] = std::max(cmx[a], pmx[t][a]); }
            }
            bool use[3]; float scale[3];
            for (int a = 0; a < 3; a++) { float ext = cmx[a] - cmn[a]; use[a] = ext > 0.0f && std::isfinite(ext); scale[a] = use[a] ? (float)kBins / ext : 0.0f; }
            struct Bins { Box bb[3][kBins]; uint32_t bc[3][kBins]; };
            Bins one; std::vector<Bins> many(team > 1 ? team : 0);
            Bins* parts = team > 1 ? many.data() : &one;
            team_for(team, w.begin, w.end, [&](unsigned t, uint32_t b, uint32_t e) {
                Bins& B = parts[t];
                for (int a = 0; a < 3; a++) for (int k = 0; k < kBins; k++) { B.bb[a][k] = empty_box(); B.bc[a][k] = 0; }
                for (uint32_t i = b; i < e; i++) {
                    const uint32_t id = order[i]; const Box& bx = boxes[id];
                    for (int a = 0; a < 3; a++) {
                        if (!use[a]) continue;
                        int k = (int)((cen[a][id] - cmn[a]) * scale[a]); k = std::min(std::max(k, 0), kBins - 1);
                        grow(B.bb[a][k], bx); B.bc[a][k]++;
                    }
                }
            });
            Bins& B = parts[0];
            for (unsigned t = 1; t < team; t++) for (int a = 0; a < 3; a++) for (int k = 0; k < kBins; k++) { grow(B.bb[a][k], parts[t].bb[a][k]); B.bc[a][k] += parts[t].bc[a][k]; }
            double best_cost = std::numeric_limits<double>::infinity(); int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; a++) {
                if (!use[a]) continue;
                double right_area[kBins]; uint32_t right_cnt[kBins];
                Box acc = empty_box(); uint32_t c = 0;
                for (int b = kBins - 1; b > 0; b--) { grow(acc, B.bb[a][b]); c += B.bc[a][b]; right_area[b] = half_area(acc); right_cnt[b] = c; }
                acc = empty_box(); c = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    grow(acc, B.bb[a][b]); c += B.bc[a][b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    double cost = half_area(acc) * c + right_area[b + 1] * right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const int a = best_axis; const float sc = scale[a];
                auto it = std::partition(order.begin() + w.begin, order.begin() + w.end, [&](uint32_t id) {
                    int b = (int)((cen[a][id] - cmn[a]) * sc); b = std::min(std::max(b, 0), kBins - 1);
                    return b <= best_bin;
                });
                mid = (uint32_t)(it - order.begin());
                have_split = mid > w.begin && mid < w.end;
                if (have_split) {      // the children's boxes are unions of bins: one pass over their objects saved
                    child_bounds[0] = empty_box(); child_bounds[1] = empty_box();
                    for (int k = 0; k < kBins; k++) grow(child_bounds[k <= best_bin ? 0 : 1], B.bb[a][k]);
                    have_child_bounds = true;
                }
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
        Work right{mid, w.end, me, 1, w.depth + 1, me + (int32_t)(mid - w.begin), have_child_bounds, child_bounds[1]}, left{w.begin, mid, me, 0, w.depth + 1, me + 1, have_child_bounds, child_bounds[0]};
        if (level < par_levels && right.end - right.begin >= kParallelMin && left.end - left.begin >= kParallelMin) {
            level++;
            helpers.emplace_back(subtree, right, level);
        } else stack.push_back(right);
        stack.push_back(left);                                 // left is processed first -> left subtree is contiguous
      }
      for (auto& t : helpers) t.join();
      uint32_t seen = max_depth.load();
      while (seen < deepest && !max_depth.compare_exchange_weak(seen, deepest)) {}
    };
    subtree(Work{0, n, -1, 0, 1, 0, false, empty_box()}, 0);
    res.max_depth = max_depth.load();
    return res;
}

} // namespace rtbvh
