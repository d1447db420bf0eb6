// rt_lib.hip — C ABI (include/rt_hip.h) of the gfx950 trace-loop library: scene flattening to device tables,
// BVH build, wavefront scheduling of the kernels in rt_kernels.h.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <array>
#include <functional>
#include <memory>
#include <mutex>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_bvh.h"
namespace rtlbvh { int build(const rtbvh::Box* boxes, const uint32_t* object_ids, uint32_t n, void* d_nodes, uint32_t* max_depth, std::string& err); }
namespace rtploc { int build(const rtbvh::Box* boxes, const uint32_t* object_ids, uint32_t n, void* d_nodes, uint32_t* max_depth, uint32_t* rounds, std::string& err); }
namespace rtsah { int build(const rtbvh::Box* boxes, const uint32_t* object_ids, uint32_t n, void* d_nodes, uint32_t* max_depth, uint32_t* levels, std::string& err); }
#define RT_KERNELS_COMMON
#include "rt_kernels.h"
namespace rtwide { int build(const void* d_nodes, uint32_t n_nodes, const rtd::SceneDev& tables, uint32_t inline_type, uint32_t node_stride, void** out_nodes, size_t* out_node_bytes,
                             void** out_leaves, size_t* out_leaf_bytes, uint32_t* out_leaf_stride, uint32_t* out_n_wide, uint32_t* out_levels, std::string& err); }

using namespace rtd;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// owns one device allocation
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { if (p) { (void)hipFree(p); p = nullptr; } bytes = n; if (!n) return hipSuccess; return hipMalloc(&p, n); }
    template <class T> T* as() const { return (T*)p; }
};

template <class T>
int upload(DevBuf& b, const std::vector<T>& v) {
    HIP_TRY(b.alloc(v.size() * sizeof(T)));
    if (!v.empty()) HIP_TRY(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

struct DeviceGuard {      // restore the caller's current device on scope exit
    int prev = -1; bool ok = false;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

inline bool finite3(const float* a) { return std::isfinite(a[0]) && std::isfinite(a[1]) && std::isfinite(a[2]); }

} // namespace

namespace {
// most specific first; the last entry covers everything
const KernelSet* const g_kernel_sets[] = {&ks_spheres, &ks_mesh, &ks_boxes, &ks_all};
constexpr int N_KERNEL_SETS = sizeof(g_kernel_sets) / sizeof(g_kernel_sets[0]);
} // namespace

// every device buffer of a scene (X-macro: allocation, peer copy to a replica, pointer binding all walk the same list)
#define RT_SCENE_BUFFERS(X) \
    X(nodes) X(wide_nodes) X(wide_leaves) X(list_nodes) X(lds_image) X(ldsw_image) X(quad_image) X(obj_code) X(obj_material) X(obj_shade) X(unbounded) X(media) X(spheres) X(quads) X(triangles) X(inf_planes) \
    X(axis0) X(axis1) X(axis2) X(disks) X(boxes) X(capsules) X(cylinders) X(media_rec) X(materials) X(textures) X(lights)

struct rt_scene {
    int device = 0;
    const KernelSet* ks = g_kernel_sets[N_KERNEL_SETS - 1];
    SceneDev dev{};
    uint32_t stack_depth = 1;        // per-lane traversal stack entries
    uint32_t bvh_depth = 0;
    bool use_lds = false;            // LdsTree kernels (small all-sphere scene)
    bool use_ldsw = false;           // ... its any-hit kernels through the W-wide tree in LDS (rt_ldsw.h) instead of the binary one (use_lds stays set: everything else about "tree in LDS" holds)
    bool use_ldsw_closest = false;   // ... and its closest-hit kernels too (RT_HIP_LDS_WIDE=2)
    uint32_t ldsw_count = 0, ldsw_levels = 0;
    bool use_quad = false;           // the any-hit kernels of a tree read through the caches walk its W-wide float form (QuadTree, rt_ldsw.h)
    uint32_t quad_count = 0, quad_levels = 0;
    bool use_wide = false;           // WideTree kernels (rt_wide.h): the 8-wide tree made from the binary one, for trees read through the caches
    uint32_t wide_count = 0, wide_levels = 0;
    bool device_build = false;       // the tree was built by rt_lbvh.hip
    int shadow_rays = -1;            // one lane per shadow ray (k_shadow_rays): -1 = the library's rule, 0/1 = RT_HIP_SHADOW_RAYS
    bool has_phong = false;          // a Phong material somewhere: k_shadow_rays cannot fold its two sums
    bool any_light_cutoff = false;   // a light with a finite CutoffRadius: rays beyond it add an exact zero and are not traced
    int persistent_shadow = -1;      // shadow rays through k_shadow_persistent: -1 = by the number of rays per request, 0/1 = RT_HIP_PERSISTENT_SHADOW
    bool persistent = false;         // bounce rays through k_closest_persistent (scenes with a heavy-tailed cost per ray)
    uint32_t n_objects = 0, n_lights = 0;
    int n_cu = 256;
    std::atomic<int> live_renders{0};            // rt_scene_destroy refuses while a render still uses the scene
#define X(name) DevBuf name;
    RT_SCENE_BUFFERS(X)
#undef X
    // staging of rt_trace_closest / rt_trace_any, kept between calls (grow-only up to TRACE_KEEP_BYTES)
    DevBuf trace_in, trace_out, trace_k;
};

namespace {
// SceneDev's pointers from the scene's own buffers (after upload, and after a replica's peer copies)
void bind_scene_pointers(rt_scene* s) {
    SceneDev& sd = s->dev;
    sd.nodes = s->nodes.as<Node>(); sd.list = s->list_nodes.as<Node>(); sd.lds_image = s->lds_image.as<uint4>(); sd.ldsw_image = s->ldsw_image.as<uint4>(); sd.quad_image = s->quad_image.as<uint4>();
    sd.wide_nodes = s->wide_nodes.as<uint4>(); sd.wide_leaves = s->wide_leaves.as<char>();
    sd.obj_code = s->obj_code.as<uint32_t>(); sd.obj_material = s->obj_material.as<uint32_t>();
    sd.obj_shade = s->obj_shade.bytes ? s->obj_shade.as<float4>() : nullptr;
    sd.unbounded = s->unbounded.as<uint32_t>(); sd.media = s->media.as<uint32_t>();
    sd.spheres = s->spheres.as<float4>(); sd.quads = s->quads.as<float4>(); sd.triangles = s->triangles.as<float4>(); sd.inf_planes = s->inf_planes.as<float4>();
    sd.axis_planes[0] = s->axis0.as<float4>(); sd.axis_planes[1] = s->axis1.as<float4>(); sd.axis_planes[2] = s->axis2.as<float4>();
    sd.disks = s->disks.as<float4>(); sd.boxes = s->boxes.as<float4>(); sd.capsules = s->capsules.as<float4>(); sd.cylinders = s->cylinders.as<float4>();
    sd.media_rec = s->media_rec.as<rt_medium>(); sd.materials = s->materials.as<rt_material>(); sd.textures = s->textures.as<rt_texture>();
    sd.lights = s->lights.as<rt_light>();
}
// more than 64 KB of dynamic LDS needs a per-function opt-in.  The attribute is per function and process-wide, so it is set to the
// cap rt_scene_create admits (80 KB), never to one scene's need: a second scene cannot lower the first one's limit.
constexpr size_t LDS_TREE_CAP = 80 * 1024;
int allow_big_lds(const KernelSet* ks) {
    const void* fns[8] = {(const void*)ks->closest_primary_lds, (const void*)ks->closest_lds, (const void*)ks->trace_closest_lds, (const void*)ks->shadow_lds, (const void*)ks->trace_any_lds, (const void*)ks->shadow_lds_exact,
                          (const void*)ks->shadow_rays_lds, (const void*)ks->shadow_rays_lds_exact};
    for (const void* f : fns) if (f) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_CAP));
    const void* wfns[6] = {(const void*)ks->ldsw.closest, (const void*)ks->ldsw.closest_primary, (const void*)ks->ldsw.shadow, (const void*)ks->ldsw.shadow_exact, (const void*)ks->ldsw.trace_closest, (const void*)ks->ldsw.trace_any};
    for (const void* f : wfns) if (f) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TREE_CAP));
    return RT_OK;
}
} // namespace

// One batch in flight: its own path state, counters and stream.  Batches are independent until they add into the image
// sums, so a few of them run concurrently: the deep bounces of one batch (few paths, latency-bound launches) overlap the
// wide first bounces of the next.  The adds into the sums stay in batch order (event chain), so results do not depend on it.
struct Slot {
    RenderDev rd{};
    hipStream_t stream = nullptr;
    DevBuf f32_pool, counters;
};
constexpr int MAX_SLOTS = 4;
#ifndef RT_TRACE_BLOCKS_PER_CU
#define RT_TRACE_BLOCKS_PER_CU 8
#endif

struct rt_render {
    rt_scene* scene = nullptr;
    int device = 0;                  // the scene's device, cached: every entry point selects it
    rt_render_opts opts{};
    RenderDev rd{};                  // slot 0's view (shared fields)
    bool exact_lights = false;       // RT_RENDER_EXACT_LIGHT_SAMPLING / RT_HIP_EXACT_LIGHTS=1
    bool exact_occlusion = false;    // RT_RENDER_EXACT_OCCLUSION / RT_HIP_EXACT_OCCLUSION=1
    uint32_t passes_per_batch = 1;
    size_t max_paths = 0;
    hipStream_t stream = nullptr; bool own_stream = false;     // the stream results are ordered on (slot 0's)
    int n_slots = 1;
    Slot slots[MAX_SLOTS];
    hipEvent_t ev_acc[MAX_SLOTS] = {};      // recorded after a slot's k_accumulate
    hipEvent_t ev_fork = nullptr;
    uint64_t batches_enqueued = 0;
    float* accum = nullptr; bool own_accum = false;
    uint32_t sample_count = 0;
    uint64_t passes_rendered = 0;
    DevBuf stats, accum_buf, rgb;
    DevBuf validate;                 // RT_RENDER_VALIDATE: u64 [n_objects][2], shared by the batch slots
    std::vector<unsigned long long> stats_host;
    // per-kernel event timing (rt_render_set_profiling)
    bool profiling = false;
    struct Stamp { int cls; hipEvent_t a, b; };
    std::vector<Stamp> stamps;              // recorded, not yet read
    std::vector<hipEvent_t> free_events;
    rt_kernel_profile prof{};
    // n_devices > 1: this handle is a front for one single-device render per shard (shards[0] on the caller's scene, the others on
    // replicas this handle owns); its own slots / streams stay empty.  `combined` and `gather` live on shards[0]'s device.
    std::vector<rt_render*> shards;
    std::vector<rt_scene*> replicas;
    DevBuf combined;
    std::vector<std::unique_ptr<DevBuf>> gather;      // [G - 1] on shards[0]'s device: where shard g's sums arrive
    std::vector<hipEvent_t> ev_gather;                // [G - 1], each created on its shard's device: that shard's copy is done
};

namespace {
// Brackets one launch with two event records when profiling is on.
struct Timed {
    rt_render* r; int cls; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t get(rt_render* r) {
        if (!r->free_events.empty()) { hipEvent_t e = r->free_events.back(); r->free_events.pop_back(); return e; }
        hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
    }
    Timed(rt_render* r_, int cls_, hipStream_t st_ = nullptr) : r(r_), cls(cls_), st(st_ ? st_ : r_->stream) {
        if (!r->profiling) return;
        a = get(r); b = get(r);
        if (a) (void)hipEventRecord(a, st);
    }
    ~Timed() {
        if (!r->profiling) return;
        if (b) (void)hipEventRecord(b, st);
        if (a && b) r->stamps.push_back({cls, a, b});
    }
};
void collect_stamps(rt_render* r) {       // call after the stream has been synchronised
    for (auto& s : r->stamps) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { r->prof.launches[s.cls]++; r->prof.total_ms[s.cls] += ms; }
        r->free_events.push_back(s.a); r->free_events.push_back(s.b);
    }
    r->stamps.clear();
}
} // namespace

// rt_trace_closest for callers that cannot allocate (rt_hip.h): the device copies of the rays and of the hit records live in a
// file-scope device array instead of two hipMalloc'ed buffers.  `scratch` is only a host-side stand-in for the symbol: the address
// the kernels and copies use is resolved once per device with hipGetSymbolAddress (scratch_for_device, called by rt_scene_create with
// that device current) and kept in g_scratch[device]; each GPU has its own copy of the array at its own address.
static __device__ __align__(256) unsigned char scratch[RT_TRACE_NOALLOC_MAX_RAYS * (sizeof(rt_ray) + sizeof(rt_hit))];
namespace {
constexpr int MAX_DEVICES = 64;
void* g_scratch[MAX_DEVICES] = {};
hipStream_t g_scratch_stream[MAX_DEVICES] = {};      // one non-blocking stream per device for rt_trace_closest_noalloc: nothing of it touches the null stream
std::mutex g_scratch_mutex;
// the current device must be `device`
int scratch_for_device(int device) {
    if (device < 0 || device >= MAX_DEVICES) return fail(RT_ERR_INVALID_ARG, "device %d beyond the %d this library keeps scratch for", device, MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch[device] && g_scratch_stream[device]) return RT_OK;
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(RT_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address on device %d", device);
    if (!g_scratch_stream[device]) HIP_TRY(hipStreamCreateWithFlags(&g_scratch_stream[device], hipStreamNonBlocking));      // (a stream is no device allocation; kept until the process ends)
    g_scratch[device] = p;
    return RT_OK;
}
} // namespace


namespace {
// The W-wide tree for LDS (rt_ldsw.h) from the host's binary tree: a node takes over its larger inner children's children until it has
// W of them (largest surface area first), its children sit in the slots of their quadrant / octant, and the inner children of one node
// get consecutive indices (breadth first).  Returns false when the tree does not fit the format (more nodes than a stack entry can name).
struct LdswBuild { std::vector<unsigned char> nodes; uint32_t n_nodes = 0, levels = 0, axis_a = 0, axis_b = 2; };
bool build_ldsw(const std::vector<rtbvh::HostNode>& bin, LdswBuild& out, bool big) {
    const size_t max_nodes = big ? ((size_t)1 << 28) / LDSW_NODE_UNITS : LDSW_MAX_NODES_LDS;
    constexpr int W = LDSW;
    struct Kid { float mn[3], mx[3]; int code; };
    auto kid = [&](int node, int side) { Kid k; const rtbvh::HostNode& h = bin[node]; for (int a = 0; a < 3; a++) { k.mn[a] = h.q[2 * a + side]; k.mx[a] = h.q[6 + 2 * a + side]; } k.code = side ? h.childB : h.childA; return k; };
    auto area = [](const Kid& k) { const double x = (double)k.mx[0] - k.mn[0], y = (double)k.mx[1] - k.mn[1], z = (double)k.mx[2] - k.mn[2]; return x * y + y * z + z * x; };
    {   // W = 4: the quadrants follow the two axes along which the whole tree is largest
        const Kid a = kid(0, 0), b = kid(0, 1);
        double ext[3]; for (int i = 0; i < 3; i++) ext[i] = (double)std::max(a.mx[i], b.mx[i]) - std::min(a.mn[i], b.mn[i]);
        int smallest = 0; for (int i = 1; i < 3; i++) if (ext[i] < ext[smallest]) smallest = i;
        out.axis_a = smallest == 0 ? 1 : 0; out.axis_b = smallest == 2 ? 1 : 2;
    }
    // Which W subtrees a node takes as its children: the choice that minimises the expected number of node visits, sum over the wide nodes
    // of their surface area (Ylitie, Karras, Laine 2017, section 3.1: cost[n][j] = the best cost of the binary subtree n as a forest of at most
    // j + 1 wide subtrees; a leaf costs the same wherever it hangs, so it counts 0).  Greedy "open the largest child" left 263 nodes of
    // 2.8 children for Night Spheres' 484 leaves.
    const size_t nb = bin.size();
    std::vector<double> node_area(nb, 0.0);
    std::vector<std::array<double, W>> cost(nb);
    auto sub_area = [&](int code, const Kid& as_child) { return code >= 0 ? node_area[code] : area(as_child); };
    (void)sub_area;
    std::vector<int> order; order.reserve(nb);      // parents before children; walked backwards for the bottom-up pass
    { std::vector<int> st{0}; while (!st.empty()) { const int n = st.back(); st.pop_back(); order.push_back(n); for (int side = 0; side < 2; side++) { const int c = side ? bin[n].childB : bin[n].childA; if (c >= 0) st.push_back(c); } } }
    auto cost_of = [&](int code, int j) { return code >= 0 ? cost[code][j] : 0.0; };      // j = roots allowed - 1
    for (size_t oi = order.size(); oi-- > 0;) {
        const int n = order[oi];
        const Kid a = kid(n, 0), b = kid(n, 1);
        Kid u = a; if (b.code != rtbvh::CHILD_EMPTY) for (int x = 0; x < 3; x++) { u.mn[x] = std::min(u.mn[x], b.mn[x]); u.mx[x] = std::max(u.mx[x], b.mx[x]); }
        node_area[n] = area(u);
        const int ca = a.code, cb = b.code == rtbvh::CHILD_EMPTY ? -1 : b.code;
        auto distribute = [&](int roots) {      // the subtree as `roots` (>= 2) subtrees: k of them from the left child, the others from the right
            double best = INFINITY;
            for (int k = 1; k < roots; k++) best = std::min(best, cost_of(ca, k - 1) + cost_of(cb, roots - k - 1));
            return best;
        };
        cost[n][0] = node_area[n] + distribute(W);                                     // one wide node here, its W children below
        for (int j = 1; j < W; j++) cost[n][j] = std::min(distribute(j + 1), cost[n][j - 1]);
    }
    std::function<void(int, const Kid&, int, std::vector<Kid>&)> collect = [&](int code, const Kid& self, int roots, std::vector<Kid>& outk) {
        if (code < 0 || roots == 1) { outk.push_back(self); return; }
        if (roots > 1 && cost[code][roots - 1] == cost[code][roots - 2] && roots > 2) { collect(code, self, roots - 1, outk); return; }
        if (roots == 2 && cost[code][1] == cost[code][0]) { outk.push_back(self); return; }
        const Kid a = kid(code, 0), b = kid(code, 1);
        if (b.code == rtbvh::CHILD_EMPTY) { collect(a.code, a, roots, outk); return; }
        int best_k = 1; double best = INFINITY;
        for (int k = 1; k < roots; k++) { const double c = cost_of(a.code, k - 1) + cost_of(b.code, roots - k - 1); if (c < best) { best = c; best_k = k; } }
        collect(a.code, a, best_k, outk); collect(b.code, b, roots - best_k, outk);
    };
    struct Todo { int bin_node; uint32_t level; };
    std::vector<Todo> todo{{0, 1}};           // todo[i] becomes wide node i
    for (size_t i = 0; i < todo.size(); i++) {
        if (todo.size() > max_nodes) return false;
        std::vector<Kid> kids;
        {   // the node's own two children, each opened as far as the cost table says: W subtrees in all
            const Kid a = kid(todo[i].bin_node, 0), b = kid(todo[i].bin_node, 1);
            if (b.code == rtbvh::CHILD_EMPTY) kids.push_back(a);
            else {
                int best_k = 1; double best = INFINITY;
                for (int k = 1; k < W; k++) { const double c = cost_of(a.code, k - 1) + cost_of(b.code, W - k - 1); if (c < best) { best = c; best_k = k; } }
                collect(a.code, a, best_k, kids); collect(b.code, b, W - best_k, kids);
            }
        }
        // slots: kid k in slot s scores sum over the slot's axes of (+-)(its centre - the node's centre); greedily the best (kid, slot) pair first
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (const Kid& k : kids) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], (double)k.mn[a]); hi[a] = std::max(hi[a], (double)k.mx[a]); }
        const int axes[3] = {W == 8 ? 0 : (int)out.axis_a, W == 8 ? 1 : (int)out.axis_b, 2};
        const int n_axes = W == 8 ? 3 : 2;
        int slot_of[W]; bool kid_done[W] = {}, slot_used[W] = {};
        for (size_t placed = 0; placed < kids.size(); placed++) {
            double best = -INFINITY; int bk = -1, bs = -1;
            for (int k = 0; k < (int)kids.size(); k++) if (!kid_done[k])
                for (int sl = 0; sl < W; sl++) if (!slot_used[sl]) {
                    double sc = 0;
                    for (int b = 0; b < n_axes; b++) { const int a = axes[b]; const double c = 0.5 * ((double)kids[k].mn[a] + kids[k].mx[a]) - 0.5 * (lo[a] + hi[a]); sc += ((sl >> b) & 1) ? c : -c; }
                    if (sc > best) { best = sc; bk = k; bs = sl; }
                }
            kid_done[bk] = true; slot_used[bs] = true; slot_of[bk] = bs;
        }
        // the record
        out.nodes.resize((i + 1) * LDSW_NODE_BYTES, 0);
        unsigned char* rec = out.nodes.data() + i * LDSW_NODE_BYTES;
        float* planes = (float*)(rec + LDSW_HEAD_BYTES);
        for (int a = 0; a < 3; a++) for (int sl = 0; sl < W; sl++) { planes[3 * W * a + sl] = 3e38f; planes[3 * W * a + W + sl] = -3e38f; planes[3 * W * a + 2 * W + sl] = 3e38f; }      // an empty slot: no ray meets it
        uint32_t imask = 0, lmask = 0; uint32_t ids[W] = {};
        int by_slot[W]; for (int sl = 0; sl < W; sl++) by_slot[sl] = -1;
        for (int k = 0; k < (int)kids.size(); k++) by_slot[slot_of[k]] = k;
        const uint32_t child_base = (uint32_t)todo.size() * LDSW_NODE_UNITS;      // (16-B units)
        for (int sl = 0; sl < W; sl++) {
            if (by_slot[sl] < 0) continue;
            const Kid& k = kids[by_slot[sl]];
            for (int a = 0; a < 3; a++) { planes[3 * W * a + sl] = k.mn[a]; planes[3 * W * a + W + sl] = k.mx[a]; planes[3 * W * a + 2 * W + sl] = k.mn[a]; }
            if (k.code >= 0) { imask |= 1u << sl; todo.push_back({k.code, todo[i].level + 1}); }          // (slots ascending: the k-th inner child is node child_base + k)
            else { lmask |= 1u << sl; ids[sl] = (uint32_t)~k.code; }
        }
        if (!big && child_base + W * LDSW_NODE_UNITS > 65536u) return false;
        const uint32_t head[2] = {child_base, imask | (lmask << 8)};
        memcpy(rec, head, 8); memcpy(rec + 16, ids, sizeof(ids));
        out.levels = std::max(out.levels, todo[i].level);
    }
    if (todo.size() > max_nodes) return false;
    out.n_nodes = (uint32_t)todo.size();
    {   // every leaf of the binary tree hangs in exactly one slot, every node but the root is exactly one node's child, and every child's
        // box lies inside what the node's slot says: a tree that fails this is not used
        std::vector<int> seen;
        size_t bin_leaves = 0, wide_leaves = 0, inner_kids = 0; bool ok = true;
        auto note = [&](int code) { if (code < 0 && code != rtbvh::CHILD_EMPTY) { bin_leaves++; const size_t o = (size_t)~code; if (seen.size() <= o) seen.resize(o + 1, 0); } };
        for (const rtbvh::HostNode& h : bin) { note(h.childA); note(h.childB); }
        std::vector<int> parents(out.n_nodes, 0);
        for (uint32_t i = 0; i < out.n_nodes; i++) {
            const unsigned char* rec = out.nodes.data() + (size_t)i * LDSW_NODE_BYTES;
            uint32_t head[2]; memcpy(head, rec, 8);
            const uint32_t imask = head[1] & 0xFFu, lmask = (head[1] >> 8) & 0xFFu;
            ok = ok && !(imask & lmask);
            uint32_t rank = 0;
            for (int sl = 0; sl < W; sl++) {
                if ((lmask >> sl) & 1u) { uint32_t id; memcpy(&id, rec + 16 + 4 * sl, 4); if (id < seen.size()) seen[id]++; else ok = false; wide_leaves++; }
                if ((imask >> sl) & 1u) { const uint32_t child = (head[0] / LDSW_NODE_UNITS) + rank++; if (child < out.n_nodes) parents[child]++; else ok = false; inner_kids++; }
            }
        }
        // an inner child's record must be the subtree its slot's box was made for: the union of the child's own slots is that box
        for (uint32_t i = 0; i < out.n_nodes; i++) {
            const unsigned char* rec = out.nodes.data() + (size_t)i * LDSW_NODE_BYTES;
            uint32_t head[2]; memcpy(head, rec, 8);
            const float* pl = (const float*)(rec + LDSW_HEAD_BYTES);
            uint32_t rank = 0;
            for (int sl = 0; sl < W; sl++) if ((head[1] >> sl) & 1u) {
                const uint32_t child = (head[0] / LDSW_NODE_UNITS) + rank++;
                if (child >= out.n_nodes) { ok = false; continue; }
                const unsigned char* crec = out.nodes.data() + (size_t)child * LDSW_NODE_BYTES;
                uint32_t chead[2]; memcpy(chead, crec, 8);
                const float* cpl = (const float*)(crec + LDSW_HEAD_BYTES);
                const uint32_t used = (chead[1] | (chead[1] >> 8)) & 0xFFu;
                for (int a = 0; a < 3; a++) {
                    float lo = INFINITY, hi = -INFINITY;
                    for (int cs = 0; cs < W; cs++) if ((used >> cs) & 1u) { lo = std::min(lo, cpl[3 * W * a + cs]); hi = std::max(hi, cpl[3 * W * a + W + cs]); }
                    ok = ok && lo == pl[3 * W * a + sl] && hi == pl[3 * W * a + W + sl] && pl[3 * W * a + 2 * W + sl] == pl[3 * W * a + sl];
                }
            }
        }
        for (int c : seen) ok = ok && c <= 1;
        size_t seen_once = 0; for (int c : seen) seen_once += c == 1;
        for (uint32_t i = 1; i < out.n_nodes; i++) ok = ok && parents[i] == 1;
        if (getenv("RT_HIP_TIMING")) fprintf(stderr, "rt_hip: %d-wide LDS tree: %zu leaves of %zu (%zu distinct), %zu inner children for %u nodes, %s\n", W, wide_leaves, bin_leaves, seen_once, inner_kids, out.n_nodes, ok ? "consistent" : "INCONSISTENT");
        if (!ok || seen_once != bin_leaves || wide_leaves != bin_leaves || inner_kids + 1 != out.n_nodes) return false;
    }
    return true;
}
} // namespace

extern "C" {

int rt_version(void) { return RT_ABI_VERSION; }
const char* rt_last_error(void) { return g_err.c_str(); }

int rt_device_count(int* n) {
    if (!n) return fail(RT_ERR_INVALID_ARG, "rt_device_count: null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(RT_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n = c;
    return RT_OK;
}

// -------------------------------------------------------------------------------------------------
// Scene
// -------------------------------------------------------------------------------------------------
static int check_prim(const rt_scene_desc* d, uint32_t type, uint32_t index, bool allow_medium) {
    uint32_t n = 0;
    switch (type) {
        case RT_HIT_SPHERE: n = d->n_spheres; break;
        case RT_HIT_QUAD: n = d->n_quads; break;
        case RT_HIT_TRIANGLE: n = d->n_triangles; break;
        case RT_HIT_INFINITE_PLANE: n = d->n_infinite_planes; break;
        case RT_HIT_XY_PLANE: n = d->n_xy_planes; break;
        case RT_HIT_XZ_PLANE: n = d->n_xz_planes; break;
        case RT_HIT_YZ_PLANE: n = d->n_yz_planes; break;
        case RT_HIT_DISK: n = d->n_disks; break;
        case RT_HIT_BOX: n = d->n_boxes; break;
        case RT_HIT_CAPSULE: n = d->n_capsules; break;
        case RT_HIT_CYLINDER: n = d->n_cylinders; break;
        case RT_HIT_MEDIUM: if (!allow_medium) return fail(RT_ERR_INVALID_ARG, "a ConstantDensityMedium cannot be used here"); n = d->n_media; break;
        default: return fail(RT_ERR_UNSUPPORTED, "unknown hittable type %u (user-defined Hittables are rejected, not emulated)", type);
    }
    if (index >= n) return fail(RT_ERR_INVALID_ARG, "hittable index %u out of range for type %u (table has %u)", index, type, n);
    if (index > OBJ_INDEX_MASK) return fail(RT_ERR_TOO_LARGE, "more than 2^28 primitives of one type");
    return RT_OK;
}

int rt_scene_create(const rt_scene_desc* d, int device, rt_scene** out) {
    if (!d || !out) return fail(RT_ERR_INVALID_ARG, "rt_scene_create: null argument");
    *out = nullptr;
    if (d->struct_size != sizeof(rt_scene_desc) || d->abi_version != RT_ABI_VERSION)
        return fail(RT_ERR_INVALID_ARG, "rt_scene_desc size/version mismatch (got %u/%u, want %zu/%d)", d->struct_size, d->abi_version, sizeof(rt_scene_desc), RT_ABI_VERSION);
#define NEED(ptr, n) if ((n) && !(ptr)) return fail(RT_ERR_INVALID_ARG, "table %s is null but its count is %u", #ptr, (unsigned)(n))
    NEED(d->objects, d->n_objects); NEED(d->spheres, d->n_spheres); NEED(d->quads, d->n_quads); NEED(d->triangles, d->n_triangles);
    NEED(d->infinite_planes, d->n_infinite_planes); NEED(d->xy_planes, d->n_xy_planes); NEED(d->xz_planes, d->n_xz_planes); NEED(d->yz_planes, d->n_yz_planes);
    NEED(d->disks, d->n_disks); NEED(d->boxes, d->n_boxes); NEED(d->capsules, d->n_capsules); NEED(d->cylinders, d->n_cylinders); NEED(d->media, d->n_media);
    NEED(d->materials, d->n_materials); NEED(d->textures, d->n_textures); NEED(d->lights, d->n_lights);
    if (d->n_materials >= 0xFFFFu) return fail(RT_ERR_TOO_LARGE, "%u materials: the path state keeps material ids in 16 bits (at most 65534)", d->n_materials);
#undef NEED
    if (d->skybox.type > RT_SKY_SINGLE_COLOUR) return fail(RT_ERR_UNSUPPORTED, "unknown skybox type %u (user-defined SkyBoxes are rejected)", d->skybox.type);
    for (uint32_t i = 0; i < d->n_textures; i++)
    {
        if (d->textures[i].type > RT_TEX_GREYSCALE_PERLIN) return fail(RT_ERR_UNSUPPORTED, "texture %u: unknown type %u (CustomFunction textures and noise modules other than Perlin are rejected)", i, d->textures[i].type);
        if (d->textures[i].type == RT_TEX_GREYSCALE_PERLIN) {
            const float oc = d->textures[i].f[3], sd = d->textures[i].f[4];
            if (!(oc >= 1.0f && oc <= 30.0f) || oc != std::floor(oc)) return fail(RT_ERR_INVALID_ARG, "texture %u: Perlin OctaveCount %g outside 1..30", i, (double)oc);     // SharpNoise's Perlin.MaxOctaves
            if (!(std::fabs(sd) <= 16777216.0f) || sd != std::floor(sd)) return fail(RT_ERR_INVALID_ARG, "texture %u: Perlin Seed %g is not an integer a float holds", i, (double)sd);
        }
    }
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rt_material& m = d->materials[i];
        if (m.type > RT_MAT_VOLUMETRIC) return fail(RT_ERR_UNSUPPORTED, "material %u: unknown type %u (user-defined Materials are rejected)", i, m.type);
        uint32_t ntex = m.type == RT_MAT_STANDARD || m.type == RT_MAT_EMISSIVE_REFRACTIVE ? 2 : m.type == RT_MAT_REFRACTIVE ? 1 : 0;
        for (uint32_t t = 0; t < ntex; t++) if (m.tex[t] >= d->n_textures) return fail(RT_ERR_INVALID_ARG, "material %u: texture index %u out of range", i, m.tex[t]);
    }
    for (uint32_t i = 0; i < d->n_lights; i++) {
        const rt_light& l = d->lights[i];
        if (l.type > RT_LIGHT_DIFFUSE_SHAPED) return fail(RT_ERR_UNSUPPORTED, "light %u: unknown type %u (user-defined Lights are rejected)", i, l.type);
        if (l.atten_kind > RT_ATTEN_EXP_DECAY) return fail(RT_ERR_UNSUPPORTED, "light %u: unknown attenuation kind %u", i, l.atten_kind);
        if (l.type == RT_LIGHT_DIFFUSE_SHAPED) { int rc = check_prim(d, l.shape_type, l.shape_index, false); if (rc) return rc; }
    }
    for (uint32_t i = 0; i < d->n_media; i++) { int rc = check_prim(d, d->media[i].boundary_type, d->media[i].boundary_index, false); if (rc) return rc; }

    std::vector<uint32_t> obj_code(d->n_objects), obj_mat(d->n_objects), unbounded, media;
    std::vector<rtbvh::Box> boxes; std::vector<uint32_t> box_obj;
    for (uint32_t i = 0; i < d->n_objects; i++) {
        const rt_object& o = d->objects[i];
        int rc = check_prim(d, o.type, o.index, true); if (rc) return rc;
        if (o.material >= d->n_materials) return fail(RT_ERR_INVALID_ARG, "object %u: material index %u out of range", i, o.material);
        obj_code[i] = (o.type << OBJ_TYPE_SHIFT) | o.index; obj_mat[i] = o.material;
        if (o.type == RT_HIT_MEDIUM) { media.push_back(i); continue; }
        bool nan = false;
        for (int a = 0; a < 3; a++) nan |= std::isnan(o.aabb_min[a]) || std::isnan(o.aabb_max[a]);
        if (nan) return fail(RT_ERR_INVALID_ARG, "object %u: NaN in its bounding volume", i);
        if (!finite3(o.aabb_min) || !finite3(o.aabb_max)) { unbounded.push_back(i); continue; }
        rtbvh::Box b; for (int a = 0; a < 3; a++) { b.mn[a] = o.aabb_min[a]; b.mx[a] = o.aabb_max[a]; }
        boxes.push_back(b); box_obj.push_back(i);
    }
    // Builder: binned SAH.  On the host (rt_bvh.cpp: every split position for nodes of <= 4096 objects, subtrees on all host
    // threads) for scenes of fewer than 256 k objects — they may want the tree in LDS or as a list, which are made from the host's
    // nodes — and on the device (rt_sah.hip: the same 16-bin algorithm breadth-first, 5 M triangles in tens of milliseconds, the
    // nodes never cross PCIe) from there on; rays do the same number of box tests in either tree.  RT_HIP_BVH=host / =sah force
    // one or the other; =ploc (rt_ploc.hip: clustering by surface area in Morton windows) and =device (rt_lbvh.hip: Morton order
    // + Karras links + refit, the fastest build) are the two other device builders, both with 10-20 % more box tests per ray.
    constexpr size_t kDeviceBuildMin = 1u << 18;
    bool ploc = false, lbvh = false, sah = boxes.size() >= kDeviceBuildMin;
    if (const char* e = getenv("RT_HIP_BVH")) { ploc = !strcmp(e, "ploc"); lbvh = !strcmp(e, "device"); sah = !strcmp(e, "sah") || (sah && !ploc && !lbvh && strcmp(e, "host")); }
    const bool device_build = (ploc || lbvh || sah) && boxes.size() >= 2;
    rtbvh::BuildResult bvh;
    if (!device_build) {
        const auto t0 = std::chrono::steady_clock::now();
        bvh = rtbvh::build(boxes, box_obj);
        if (getenv("RT_HIP_TIMING")) fprintf(stderr, "rt_hip: host BVH build of %zu objects: %.1f ms, depth %u\n", boxes.size(),
                                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), bvh.max_depth);
    }

    // what the scene contains -> the most specific kernel variant that covers it (RT_HIP_VARIANT=generic forces the general one; test knob)
    uint32_t m_prims = 0, m_mats = 0, m_texs = 0, m_lights = 0;
    for (uint32_t i = 0; i < d->n_objects; i++) m_prims |= 1u << d->objects[i].type;
    for (uint32_t i = 0; i < d->n_media; i++) m_prims |= 1u << d->media[i].boundary_type;
    for (uint32_t i = 0; i < d->n_lights; i++) { m_lights |= 1u << d->lights[i].type; if (d->lights[i].type == RT_LIGHT_DIFFUSE_SHAPED) m_prims |= 1u << d->lights[i].shape_type; }
    for (uint32_t i = 0; i < d->n_materials; i++) m_mats |= 1u << d->materials[i].type;
    for (uint32_t i = 0; i < d->n_textures; i++) m_texs |= 1u << d->textures[i].type;
    const KernelSet* ks = g_kernel_sets[N_KERNEL_SETS - 1];
    const char* force = getenv("RT_HIP_VARIANT");
    if (!(force && !strcmp(force, "generic")))
        for (int i = 0; i < N_KERNEL_SETS; i++) {
            const KernelSet& k = *g_kernel_sets[i];
            if (!(m_prims & ~k.prims) && !(m_mats & ~k.mats) && !(m_texs & ~k.texs) && !(m_lights & ~k.lights)) { ks = &k; break; }
        }

    // float4-aligned records
    auto pack = [](const float* src, size_t n_rec, size_t floats_in, size_t quads_out) {
        std::vector<float4> v(n_rec * quads_out, make_float4(0, 0, 0, 0));
        for (size_t r = 0; r < n_rec; r++) memcpy((float*)&v[r * quads_out], src + r * floats_in, floats_in * sizeof(float));
        return v;
    };
    std::vector<float4> spheres = pack((const float*)d->spheres, d->n_spheres, 4, 1);
    std::vector<float4> quads(d->n_quads * 4), tris(d->n_triangles * 4), planes(d->n_infinite_planes * 2), disks(d->n_disks * 4), caps(d->n_capsules * 5), cyls(d->n_cylinders * 2);
    for (uint32_t i = 0; i < d->n_quads; i++) {
        const rt_quad& q = d->quads[i];
        quads[4 * i + 0] = make_float4(q.origin[0], q.origin[1], q.origin[2], 0); quads[4 * i + 1] = make_float4(q.normal[0], q.normal[1], q.normal[2], 0);
        quads[4 * i + 2] = make_float4(q.inv_col1[0], q.inv_col1[1], q.inv_col1[2], q.inv_col1[3]); quads[4 * i + 3] = make_float4(q.inv_col2[0], q.inv_col2[1], q.inv_col2[2], q.inv_col2[3]);
    }
    for (uint32_t i = 0; i < d->n_triangles; i++) {
        const rt_triangle& q = d->triangles[i];
        tris[4 * i + 0] = make_float4(q.v0[0], q.v0[1], q.v0[2], 0); tris[4 * i + 1] = make_float4(q.normal[0], q.normal[1], q.normal[2], 0);
        tris[4 * i + 2] = make_float4(q.inv_col1[0], q.inv_col1[1], q.inv_col1[2], q.inv_col1[3]); tris[4 * i + 3] = make_float4(q.inv_col2[0], q.inv_col2[1], q.inv_col2[2], q.inv_col2[3]);
    }
    for (uint32_t i = 0; i < d->n_infinite_planes; i++) {
        const rt_infinite_plane& p = d->infinite_planes[i];
        planes[2 * i + 0] = make_float4(p.point[0], p.point[1], p.point[2], p.neg_point_dot_normal); planes[2 * i + 1] = make_float4(p.normal[0], p.normal[1], p.normal[2], 0);
    }
    std::vector<float4> axis[3];
    const rt_axis_plane* ap[3] = {d->xy_planes, d->xz_planes, d->yz_planes}; const uint32_t an[3] = {d->n_xy_planes, d->n_xz_planes, d->n_yz_planes};
    for (int a = 0; a < 3; a++) {
        axis[a].resize(an[a] * 2);
        for (uint32_t i = 0; i < an[a]; i++) { axis[a][2 * i] = make_float4(ap[a][i].a_lo, ap[a][i].a_hi, ap[a][i].b_lo, ap[a][i].b_hi); axis[a][2 * i + 1] = make_float4(ap[a][i].c, 0, 0, 0); }
    }
    for (uint32_t i = 0; i < d->n_disks; i++) {
        const rt_disk& k = d->disks[i];
        disks[4 * i + 0] = make_float4(k.centre[0], k.centre[1], k.centre[2], k.radius); disks[4 * i + 1] = make_float4(k.normal[0], k.normal[1], k.normal[2], k.neg_centre_dot_norm);
        disks[4 * i + 2] = make_float4(k.inv_col1[0], k.inv_col1[1], k.inv_col1[2], k.inv_col1[3]); disks[4 * i + 3] = make_float4(k.inv_col2[0], k.inv_col2[1], k.inv_col2[2], k.inv_col2[3]);
    }
    std::vector<float4> bxs = pack((const float*)d->boxes, d->n_boxes, 21, 6);
    for (uint32_t i = 0; i < d->n_capsules; i++) {
        const rt_capsule& c = d->capsules[i];
        caps[5 * i + 0] = make_float4(c.p1[0], c.p1[1], c.p1[2], c.radius); caps[5 * i + 1] = make_float4(c.p2[0], c.p2[1], c.p2[2], 0);
        caps[5 * i + 2] = make_float4(c.uv_matrix[0], c.uv_matrix[1], c.uv_matrix[2], c.uv_matrix[3]); caps[5 * i + 3] = make_float4(c.uv_matrix[4], c.uv_matrix[5], c.uv_matrix[6], c.uv_matrix[7]);
        caps[5 * i + 4] = make_float4(c.uv_matrix[8], 0, 0, 0);
    }
    for (uint32_t i = 0; i < d->n_cylinders; i++) {
        const rt_cylinder& c = d->cylinders[i];
        cyls[2 * i + 0] = make_float4(c.p1[0], c.p1[1], c.p1[2], c.radius); cyls[2 * i + 1] = make_float4(c.p2[0], c.p2[1], c.p2[2], 0);
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(RT_ERR_NO_DEVICE, "no HIP device is visible");
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return fail(RT_ERR_NO_DEVICE, "hipGetDevice failed"); }
    if (device >= ndev) return fail(RT_ERR_INVALID_ARG, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard guard(device);
    if (!guard.ok) return fail(RT_ERR_HIP, "cannot select device %d", device);
    (void)scratch_for_device(device);      // rt_trace_closest_noalloc's array on this device (once per device).  Best effort: only that entry point needs it, and it resolves the address itself if this failed

    auto s = std::make_unique<rt_scene>();
    s->device = device; s->ks = ks;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    s->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int rc;
    std::vector<Node> nodes(bvh.nodes.size());
    static_assert(sizeof(Node) == sizeof(rtbvh::HostNode), "node layouts must agree");
    if (!nodes.empty()) memcpy(nodes.data(), bvh.nodes.data(), nodes.size() * sizeof(Node));
#define UP(buf, vec) if ((rc = upload(s->buf, vec))) return rc
    size_t n_nodes = nodes.size();
    if (device_build) {
        n_nodes = boxes.size() - 1;
        HIP_TRY(s->nodes.alloc(n_nodes * sizeof(Node)));
        std::string err;
        uint32_t rounds = 0;
        const auto t0 = std::chrono::steady_clock::now();
        int brc = sah  ? rtsah::build(boxes.data(), box_obj.data(), (uint32_t)boxes.size(), s->nodes.p, &bvh.max_depth, &rounds, err)
                : ploc ? rtploc::build(boxes.data(), box_obj.data(), (uint32_t)boxes.size(), s->nodes.p, &bvh.max_depth, &rounds, err)
                       : rtlbvh::build(boxes.data(), box_obj.data(), (uint32_t)boxes.size(), s->nodes.p, &bvh.max_depth, err);
        if (brc) return fail(brc == 2 ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, "device BVH build: %s", err.c_str());
        if (getenv("RT_HIP_TIMING")) fprintf(stderr, "rt_hip: device BVH build (%s) of %zu objects: %.1f ms, %u levels / rounds, depth %u\n", sah ? "sah" : ploc ? "ploc" : "lbvh", boxes.size(),
                                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), rounds, bvh.max_depth);
    } else UP(nodes, nodes);
    s->device_build = device_build;
    {   // leaf_code: do the tree's objects all satisfy obj_code[obj] == base + obj ?
        bool affine = !box_obj.empty();
        const uint32_t base = affine ? obj_code[box_obj[0]] - box_obj[0] : 0u;
        for (uint32_t obj : box_obj) if (obj_code[obj] - obj != base) { affine = false; break; }
        s->dev.leaf_code_affine = affine ? 1u : 0u; s->dev.leaf_code_base = base;
        bool all = affine;
        for (uint32_t i = 0; all && i < d->n_objects; i++) all = obj_code[i] - i == base;
        s->dev.all_code_affine = all ? 1u : 0u;
        bool one = d->n_objects > 0;
        for (uint32_t i = 1; one && i < d->n_objects; i++) one = obj_mat[i] == obj_mat[0];
        s->dev.uniform_material = one ? obj_mat[0] : RT_NO_HIT;
    }
    // a handful of objects: a list of all of them, two per record, instead of the tree (rt_device.h: list_closest)
    {
        const char* e = getenv("RT_HIP_LIST");
        const size_t list_max = e ? (size_t)std::max(0, atoi(e)) : 16;
        if (!boxes.empty() && boxes.size() <= list_max) {
            // Order: objects inside the scene before the ones that bound it.  Any-hit stops at the first occluder, and what occludes
            // is what stands in the room, not its walls (the Cornell box lists its five walls first); closest hit asks everybody
            // anyway and decides ties by object index, so the order changes no result.  "Bounds it" = how many faces of the object's
            // box lie on the faces of the box around everything (stable sort by that count).
            std::vector<size_t> ord(boxes.size());
            {
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (const rtbvh::Box& b : boxes) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.mn[a]); hi[a] = std::max(hi[a], b.mx[a]); }
                std::vector<int> on_hull(boxes.size(), 0);
                for (size_t k = 0; k < boxes.size(); k++)
                    for (int a = 0; a < 3; a++) {
                        const float tol = 2e-3f * std::max(hi[a] - lo[a], 1e-30f);
                        on_hull[k] += (boxes[k].mn[a] - lo[a] <= tol) + (hi[a] - boxes[k].mx[a] <= tol);
                    }
                for (size_t k = 0; k < ord.size(); k++) ord[k] = k;
                const char* o = getenv("RT_HIP_LIST_ORDER");
                if (!(o && !strcmp(o, "scene"))) std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return on_hull[x] < on_hull[y]; });
            }
            std::vector<Node> ln((boxes.size() + 1) / 2);
            for (size_t i = 0; i < ln.size(); i++) {
                rtbvh::HostNode h; std::memset(&h, 0, sizeof(h));
                for (int side = 0; side < 2; side++) {
                    const bool have = 2 * i + side < boxes.size();
                    const size_t k = have ? ord[2 * i + side] : 0;
                    for (int a = 0; a < 3; a++) { h.q[2 * a + side] = have ? boxes[k].mn[a] : INFINITY; h.q[6 + 2 * a + side] = have ? boxes[k].mx[a] : -INFINITY; }
                    (side ? h.childB : h.childA) = have ? ~(int32_t)box_obj[k] : rtbvh::CHILD_EMPTY;
                }
                memcpy(&ln[i], &h, sizeof(Node));
            }
            UP(list_nodes, ln);
            s->dev.n_list_pairs = (uint32_t)ln.size();
        }
    }
    UP(obj_code, obj_code); UP(obj_material, obj_mat); UP(unbounded, unbounded); UP(media, media);
    {   // Shading records per object (rt_kernels.h: k_shade): when every texture is a solid colour and every material one of the three
        // that read (tex[0], tex[1], f[0], flags) only, what a hit needs from its material is 32 B that the host can resolve once.
        // Not for big scenes (the table would be 32 B per object of cache misses where one material record serves millions of hits).
        bool ok = d->n_objects > 0 && d->n_objects <= (1u << 16) && !getenv("RT_HIP_NO_OBJ_SHADE");
        for (uint32_t i = 0; ok && i < d->n_textures; i++) ok = d->textures[i].type == RT_TEX_SOLID;
        for (uint32_t i = 0; ok && i < d->n_materials; i++) {
            const rt_material& m = d->materials[i];
            ok = (m.type == RT_MAT_STANDARD || m.type == RT_MAT_REFRACTIVE || m.type == RT_MAT_EMISSIVE_REFRACTIVE) && m.tex[0] < d->n_textures &&
                 (m.type == RT_MAT_REFRACTIVE || m.tex[1] < d->n_textures) && m.flags < (1u << 24);
        }
        if (ok) {
            std::vector<float4> sh(4 * (size_t)d->n_objects, make_float4(0, 0, 0, 0));
            auto bits = [](uint32_t u) { float w; memcpy(&w, &u, 4); return w; };
            for (uint32_t i = 0; i < d->n_objects; i++) {
                const rt_material& m = d->materials[obj_mat[i]];
                const rt_texture& t0 = d->textures[m.tex[0]];
                float e[3] = {0, 0, 0};
                if (m.type != RT_MAT_REFRACTIVE) { const rt_texture& t1 = d->textures[m.tex[1]]; e[0] = t1.f[0]; e[1] = t1.f[1]; e[2] = t1.f[2]; }
                sh[4 * i] = make_float4(t0.f[0], t0.f[1], t0.f[2], m.f[0]); sh[4 * i + 1] = make_float4(e[0], e[1], e[2], bits(m.type | (m.flags << 8)));
                // what finalize_hit reads of the primitive (no UV wanted), where that is 16 B: a sphere's (centre, radius), a normal, an axis plane's offset
                const rt_object& o = d->objects[i];
                uint32_t have_geo = 0;
                if (o.type == RT_HIT_SPHERE) { sh[4 * i + 2] = spheres[o.index]; have_geo = 1; }
                else if (o.type == RT_HIT_INFINITE_PLANE) { sh[4 * i + 2] = planes[2 * o.index + 1]; have_geo = 1; }
                else if (o.type == RT_HIT_QUAD) { sh[4 * i + 2] = quads[4 * o.index + 1]; have_geo = 1; }
                else if (o.type == RT_HIT_TRIANGLE) { sh[4 * i + 2] = tris[4 * o.index + 1]; have_geo = 1; }
                else if (o.type == RT_HIT_DISK) { sh[4 * i + 2] = disks[4 * o.index + 1]; have_geo = 1; }
                else if (o.type >= RT_HIT_XY_PLANE && o.type <= RT_HIT_YZ_PLANE) { sh[4 * i + 2] = axis[o.type - RT_HIT_XY_PLANE][2 * o.index + 1]; have_geo = 1; }
                sh[4 * i + 3] = make_float4(bits(obj_code[i]), bits(obj_mat[i]), bits(have_geo), 0.0f);
            }
            UP(obj_shade, sh);
        }
    }
    UP(spheres, spheres); UP(quads, quads); UP(triangles, tris); UP(inf_planes, planes); UP(axis0, axis[0]); UP(axis1, axis[1]); UP(axis2, axis[2]);
    UP(disks, disks); UP(boxes, bxs); UP(capsules, caps); UP(cylinders, cyls);
    std::vector<rt_medium> mrec(d->media, d->media + d->n_media);
    std::vector<rt_material> mats(d->materials, d->materials + d->n_materials);
    std::vector<rt_texture> texs(d->textures, d->textures + d->n_textures);
    std::vector<rt_light> lts(d->lights, d->lights + d->n_lights);
    UP(media_rec, mrec); UP(materials, mats); UP(textures, texs); UP(lights, lts);
#undef UP
    SceneDev& sd = s->dev;
    bind_scene_pointers(s.get());
    sd.n_nodes = (uint32_t)n_nodes; sd.any_near_first = n_nodes >= 1024 ? 1u : 0u;
    sd.n_objects = d->n_objects; sd.n_unbounded = (uint32_t)unbounded.size(); sd.n_media = (uint32_t)media.size(); sd.n_lights = d->n_lights;
    sd.camera = d->camera; sd.skybox = d->skybox;
    {   // any_pad (rt_device.h): a sphere around every bounded object and the largest object's half-diagonal, in double on the host
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, max_half = 0.0;
        for (const rtbvh::Box& b : boxes) {
            double dd = 0.0;
            for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], (double)b.mn[a]); hi[a] = std::max(hi[a], (double)b.mx[a]); const double e = 0.5 * ((double)b.mx[a] - b.mn[a]); dd += e * e; }
            max_half = std::max(max_half, std::sqrt(dd));
        }
        double r2 = 0.0;
        for (int a = 0; a < 3; a++) { const double c = boxes.empty() ? 0.0 : 0.5 * (lo[a] + hi[a]), e = boxes.empty() ? 0.0 : 0.5 * (hi[a] - lo[a]); sd.bounds_centre[a] = (float)c; r2 += e * e; }
        sd.bounds_radius = (float)(std::sqrt(r2) * 1.0001 + 1e-6); sd.max_object_radius = (float)(max_half * 1.0001);
        // the smallest sphere / capsule / cylinder among the bounded objects (media boundaries and light shapes are not in the tree)
        float rmin = FLT_MAX;
        for (uint32_t obj : box_obj) {
            const rt_object& o = d->objects[obj];
            if (o.type == RT_HIT_SPHERE) rmin = std::min(rmin, std::fabs(d->spheres[o.index].radius));
            else if (o.type == RT_HIT_CAPSULE) rmin = std::min(rmin, std::fabs(d->capsules[o.index].radius));
            else if (o.type == RT_HIT_CYLINDER) rmin = std::min(rmin, std::fabs(d->cylinders[o.index].radius));
        }
        sd.min_round_radius = rmin;
        sd.any_pad_scale = 1.0f; sd.any_again = RT_ANY_GROWTH_CAP;
        if (const char* e = getenv("RT_HIP_ANY_AGAIN")) sd.any_again = (float)atof(e);
        if (const char* e = getenv("RT_HIP_ANY_PAD_SCALE")) sd.any_pad_scale = (float)atof(e);
    }
    s->stack_depth = std::max<uint32_t>(bvh.max_depth + 1, 2); s->bvh_depth = bvh.max_depth;
    if ((size_t)s->stack_depth * BLOCK * sizeof(int) > 64 * 1024) return fail(RT_ERR_TOO_LARGE, "BVH depth %u needs more LDS than one workgroup may use", bvh.max_depth);
    s->n_objects = d->n_objects; s->n_lights = d->n_lights;
    // LDS image: every BVH leaf a sphere, and two workgroups' image + 16-bit stacks fit one CU's 160 KB
    {
        bool all_spheres = ks->closest_lds != nullptr && !nodes.empty();
        for (uint32_t obj : box_obj) all_spheres = all_spheres && d->objects[obj].type == RT_HIT_SPHERE;
        const size_t nn = nodes.size(), no = d->n_objects;
        const size_t image_bytes = nn * LDS_NODE_BYTES + no * 16;
        const size_t need = image_bytes + (size_t)s->stack_depth * LDS_BLOCK_CLOSEST * sizeof(unsigned short);
        const char* no_lds = getenv("RT_HIP_NO_LDS_TREE");
        if (all_spheres && boxes.size() >= 2 && !s->dev.n_list_pairs && nn * (LDS_NODE_BYTES / 16) < 32768 && no < 32768 && need <= LDS_TREE_CAP && !(no_lds && no_lds[0] == '1') && !(force && !strcmp(force, "generic"))) {
            // node record (rt_device.h LdsTree): per axis (min A, min B, max A, max B, min A, min B), then (child A, child B)
            static_assert(LDS_NODE_BYTES == 80 && LDS_NODE_BYTES % 16 == 0, "LDS node layout");
            std::vector<float4> img(image_bytes / 16, make_float4(0, 0, 0, 0));
            for (size_t i = 0; i < nn; i++) {
                const rtbvh::HostNode& h = bvh.nodes[i];              // q = min.x(A,B) min.y(A,B) min.z(A,B) max.x(A,B) max.y(A,B) max.z(A,B)
                float* rec = (float*)img.data() + i * (LDS_NODE_BYTES / 4);
                for (int a = 0; a < 3; a++) {
                    float* ax = rec + 6 * a;
                    ax[0] = h.q[2 * a]; ax[1] = h.q[2 * a + 1]; ax[2] = h.q[6 + 2 * a]; ax[3] = h.q[6 + 2 * a + 1]; ax[4] = ax[0]; ax[5] = ax[1];
                }
                // an inner child is stored as its record's offset in 16-B units (index * 5): the lane's address is a shift, not a 32-bit multiply
                const int32_t cA = h.childA >= 0 ? h.childA * (int32_t)(LDS_NODE_BYTES / 16) : h.childA, cB = h.childB >= 0 ? h.childB * (int32_t)(LDS_NODE_BYTES / 16) : h.childB;
                memcpy(rec + 18, &cA, 4); memcpy(rec + 19, &cB, 4);
            }
            float4* sp = img.data() + nn * (LDS_NODE_BYTES / 16);
            for (size_t i = 0; i < no; i++) if (d->objects[i].type == RT_HIT_SPHERE) sp[i] = spheres[d->objects[i].index];
            if ((rc = upload(s->lds_image, img))) return rc;
            sd.lds_image = s->lds_image.as<uint4>(); sd.lds_image_bytes = (uint32_t)image_bytes;
            s->use_lds = true;
            if ((rc = allow_big_lds(ks))) return rc;
            // ... and the same tree W-wide (rt_ldsw.h): the kernels that walk it are the default when both forms fit (RT_HIP_LDS_WIDE=0: the
            // binary form).  The kernels it has no form of (ray lanes, persistent lanes: both measured slower with the tree in LDS) keep
            // the binary image, so forcing one of those leaves this out.
            const char* ew = getenv("RT_HIP_LDS_WIDE");
            const bool want_w = ew ? (ew[0] == '1' || ew[0] == '2') : RT_LDSW_DEFAULT != 0;
            LdswBuild wb;
            if (want_w && ks->ldsw.closest && !getenv("RT_HIP_SHADOW_RAYS") && !getenv("RT_HIP_PERSISTENT") && !getenv("RT_HIP_PERSISTENT_SHADOW") && build_ldsw(bvh.nodes, wb, false)) {
                const size_t node_bytes = wb.nodes.size(), any_bytes = node_bytes + no * 16, all_bytes = (any_bytes + no * LDSW_BOX_BYTES + 15) & ~(size_t)15;
                const size_t need_c = all_bytes + (size_t)(wb.levels + 1) * LDSW_BLOCK_CLOSEST * sizeof(LdswEntry), need_a = any_bytes + (size_t)(wb.levels + 1) * LDS_BLOCK_SHADOW * sizeof(LdswEntry);
                if (getenv("RT_HIP_TIMING")) fprintf(stderr, "rt_hip: %d-wide LDS tree: %u nodes, %u levels, %zu B for closest-hit kernels, %zu B for any-hit kernels\n", LDSW, wb.n_nodes, wb.levels, need_c, need_a);
                if (std::max(need_c, need_a) <= LDS_TREE_CAP) {
                    std::vector<unsigned char> wimg(all_bytes, 0);
                    memcpy(wimg.data(), wb.nodes.data(), node_bytes);
                    float4* wsp = (float4*)(wimg.data() + node_bytes);
                    float* wbx = (float*)(wimg.data() + any_bytes);
                    for (size_t i = 0; i < no; i++) {
                        if (d->objects[i].type == RT_HIT_SPHERE) wsp[i] = spheres[d->objects[i].index];
                        for (int a = 0; a < 3; a++) { wbx[6 * i + a] = d->objects[i].aabb_min[a]; wbx[6 * i + 3 + a] = d->objects[i].aabb_max[a]; }
                    }
                    if ((rc = upload(s->ldsw_image, wimg))) return rc;
                    sd.ldsw_image = s->ldsw_image.as<uint4>(); sd.quad_image = s->quad_image.as<uint4>(); sd.ldsw_node_bytes = (uint32_t)node_bytes; sd.ldsw_any_bytes = (uint32_t)any_bytes; sd.ldsw_all_bytes = (uint32_t)all_bytes;
                    sd.ldsw_axis_a = wb.axis_a; sd.ldsw_axis_b = wb.axis_b;
                    s->use_ldsw = true; s->use_ldsw_closest = ew && ew[0] == '2'; s->ldsw_count = wb.n_nodes; s->ldsw_levels = wb.levels;
                }
            }
        }
    }
    // The 8-wide tree with quantised inner boxes (rt_wide.h, rt_wide.hip) for trees that are read through the caches: a third of the
    // dependent node fetches per ray.  From 1024 inner nodes; RT_HIP_WIDE=1 / =0 forces it for any tree of two objects or more / never.
    {
        const char* e = getenv("RT_HIP_WIDE");
        const bool want = e ? e[0] == '1' : n_nodes >= 1024;
        sd.wide_inline = RT_NO_HIT;
        if (want && !s->use_lds && !sd.n_list_pairs && boxes.size() >= 2 && n_nodes == boxes.size() - 1 && n_nodes < WIDE_MAX_NODES) {      // (more nodes than a visit's 24-bit multiply addresses: the binary tree serves)
            uint32_t one_type = d->objects[box_obj[0]].type;
            for (uint32_t obj : box_obj) if (d->objects[obj].type != one_type) { one_type = RT_NO_HIT; break; }
            const uint32_t inline_type = (one_type == RT_HIT_SPHERE || one_type == RT_HIT_TRIANGLE || one_type == RT_HIT_QUAD) ? one_type : (uint32_t)RT_NO_HIT;
            uint32_t stride = WIDE_NODE_BYTES;
            if (const char* st = getenv("RT_HIP_WIDE_STRIDE")) stride = std::max<uint32_t>(WIDE_NODE_BYTES, ((uint32_t)atoi(st) + 7u) & ~7u);
            void *wn = nullptr, *wl = nullptr; size_t wnb = 0, wlb = 0; uint32_t lstride = 0;
            std::string err;
            const auto t0 = std::chrono::steady_clock::now();
            const int brc = rtwide::build(s->nodes.p, (uint32_t)n_nodes, sd, inline_type, stride, &wn, &wnb, &wl, &wlb, &lstride, &s->wide_count, &s->wide_levels, err);
            if (brc) return fail(brc == 2 ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, "wide tree build: %s", err.c_str());
            if (getenv("RT_HIP_TIMING")) fprintf(stderr, "rt_hip: wide tree of %u nodes (%zu MB) + %zu leaf records of %u B from %zu binary nodes: %.1f ms, %u levels\n", s->wide_count, wnb >> 20, boxes.size(), lstride, n_nodes,
                                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), s->wide_levels);
            s->wide_nodes.p = wn; s->wide_nodes.bytes = wnb; s->wide_leaves.p = wl; s->wide_leaves.bytes = wlb;
            sd.wide_nodes = (const uint4*)wn; sd.wide_stride = stride; sd.wide_leaves = (const char*)wl; sd.wide_leaf_stride = lstride; sd.wide_inline = inline_type;
            s->use_wide = true;
            s->stack_depth = std::max<uint32_t>(s->wide_levels + 1, 2);
            if ((size_t)s->stack_depth * BLOCK * sizeof(uint2) > 64 * 1024) return fail(RT_ERR_TOO_LARGE, "wide tree of %u levels needs more LDS than one workgroup may use", s->wide_levels);
        }
    }
    // ... and, on request (RT_HIP_QUAD=1), when every leaf is a sphere and the host built the binary tree: the W-wide float form of the same
    // tree for the ANY-HIT kernels (QuadTree, rt_ldsw.h: one FMA per plane on full-precision boxes, no byte conversions, no near / far
    // selects, no mask permutations — under half the vector instructions of a quantised 8-wide visit).  Measured on 100 k spheres + 64
    // lights: 41 instead of 57 box tests per shadow ray, but 12.5 dependent 176-B fetches instead of 7.4 of 80 B: k_shadow_persistent
    // 425 -> 445 ms per batch.  Not the default.
    {
        const char* e = getenv("RT_HIP_QUAD");
        if (s->use_wide && !device_build && sd.wide_inline == RT_HIT_SPHERE && ks->quad.shadow && e && e[0] == '1') {
            LdswBuild wb;
            const auto t0 = std::chrono::steady_clock::now();
            if (build_ldsw(bvh.nodes, wb, true)) {
                const size_t no = d->n_objects, node_bytes = wb.nodes.size(), any_bytes = node_bytes + no * 16, all_bytes = (any_bytes + no * LDSW_BOX_BYTES + 15) & ~(size_t)15;
                std::vector<unsigned char> img(all_bytes, 0);
                memcpy(img.data(), wb.nodes.data(), node_bytes);
                float4* wsp = (float4*)(img.data() + node_bytes);
                float* wbx = (float*)(img.data() + any_bytes);
                for (size_t i = 0; i < no; i++) {
                    if (d->objects[i].type == RT_HIT_SPHERE) wsp[i] = spheres[d->objects[i].index];
                    for (int a = 0; a < 3; a++) { wbx[6 * i + a] = d->objects[i].aabb_min[a]; wbx[6 * i + 3 + a] = d->objects[i].aabb_max[a]; }
                }
                if ((rc = upload(s->quad_image, img))) return rc;
                sd.quad_image = s->quad_image.as<uint4>(); sd.quad_node_bytes = (uint32_t)node_bytes;
                sd.ldsw_axis_a = wb.axis_a; sd.ldsw_axis_b = wb.axis_b;
                s->use_quad = true; s->quad_count = wb.n_nodes; s->quad_levels = wb.levels;
                if ((size_t)(wb.levels + 1) * BLOCK * sizeof(uint2) > 64 * 1024) return fail(RT_ERR_TOO_LARGE, "quad tree of %u levels needs more LDS than one workgroup may use", wb.levels);
                if (getenv("RT_HIP_TIMING")) fprintf(stderr, "rt_hip: %d-wide float tree for any-hit: %u nodes (%zu MB), %u levels: %.1f ms\n", LDSW, wb.n_nodes, node_bytes >> 20, wb.levels,
                                                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
        }
    }
    // Bounce rays in a tree far larger than the L2s have a heavy-tailed cost (measured: 5 M triangles 790 -> 1014 Msamples/s with
    // persistent lanes, 100 k spheres no change, the 10-object Cornell box -11 %).  RT_HIP_PERSISTENT=0/1 overrides for A/B runs.
    { const char* e = getenv("RT_HIP_PERSISTENT_SHADOW"); s->persistent_shadow = e ? (e[0] == '1') : -1; }
    { const char* e = getenv("RT_HIP_SHADOW_RAYS"); s->shadow_rays = e ? (e[0] == '1') : -1; }
    s->has_phong = (m_mats >> RT_MAT_PHONG) & 1u;
    for (uint32_t i = 0; i < d->n_lights; i++) s->any_light_cutoff = s->any_light_cutoff || std::isfinite(d->lights[i].cutoff_radius);
    { const char* e = getenv("RT_HIP_PERSISTENT"); s->persistent = e ? e[0] == '1' : (!s->use_lds && n_nodes * sizeof(Node) > (64u << 20)); }
    *out = s.release();
    return RT_OK;
}

int rt_scene_get_info(rt_scene* s, rt_scene_info* out) {
    if (!s || !out) return fail(RT_ERR_INVALID_ARG, "rt_scene_get_info: null argument");
    memset(out, 0, sizeof(*out));
    out->n_objects = s->n_objects; out->n_unbounded = s->dev.n_unbounded; out->n_nodes = s->dev.n_nodes; out->bvh_depth = s->bvh_depth;
    out->tree_in_lds = s->use_lds ? 1u : 0u; out->lds_image_bytes = s->dev.lds_image_bytes;
    out->built_on_device = s->device_build ? 1u : 0u; out->object_list = s->dev.n_list_pairs ? 1u : 0u;
    out->wide_nodes = s->use_wide ? s->wide_count : s->use_ldsw ? s->ldsw_count : 0u; out->wide_levels = s->use_wide ? s->wide_levels : s->use_ldsw ? s->ldsw_levels : 0u; out->device = s->device;
    strncpy(out->variant, s->ks->name, sizeof(out->variant) - 1);
    return RT_OK;
}

int rt_scene_destroy(rt_scene* s) {
    if (!s) return RT_OK;
    // every rt_render_* call reads its scene: destroying the scene first would leave the renders with dangling device pointers
    if (s->live_renders.load() > 0) return fail(RT_ERR_INVALID_ARG, "rt_scene_destroy: %d render(s) still use this scene; destroy them first", s->live_renders.load());
    DeviceGuard guard(s->device);
    delete s;
    return RT_OK;
}

// A copy of `src` on `device` for rt_render_create's n_devices > 1: every table and the BVH travel by peer copy (xGMI), nothing is
// rebuilt — the replica's tree is the source's bit for bit, so every shard traces the same arithmetic.
static int clone_scene(const rt_scene* src, int device, rt_scene** out) {
    *out = nullptr;
    DeviceGuard guard(device);
    if (!guard.ok) return fail(RT_ERR_HIP, "cannot select device %d", device);
    (void)scratch_for_device(device);      // (best effort, as in rt_scene_create)
    auto s = std::make_unique<rt_scene>();
    s->device = device; s->ks = src->ks; s->dev = src->dev; s->stack_depth = src->stack_depth; s->bvh_depth = src->bvh_depth;
    s->use_quad = src->use_quad; s->quad_count = src->quad_count; s->quad_levels = src->quad_levels; s->use_lds = src->use_lds; s->use_ldsw = src->use_ldsw; s->use_ldsw_closest = src->use_ldsw_closest; s->ldsw_count = src->ldsw_count; s->ldsw_levels = src->ldsw_levels; s->use_wide = src->use_wide; s->wide_count = src->wide_count; s->wide_levels = src->wide_levels; s->device_build = src->device_build; s->persistent_shadow = src->persistent_shadow; s->shadow_rays = src->shadow_rays; s->has_phong = src->has_phong; s->any_light_cutoff = src->any_light_cutoff; s->persistent = src->persistent;
    s->n_objects = src->n_objects; s->n_lights = src->n_lights;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    s->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
#define X(name) if (src->name.bytes) { HIP_TRY(s->name.alloc(src->name.bytes)); HIP_TRY(hipMemcpyPeer(s->name.p, device, src->name.p, src->device, src->name.bytes)); }
    RT_SCENE_BUFFERS(X)
#undef X
    HIP_TRY(hipDeviceSynchronize());
    bind_scene_pointers(s.get());
    if (s->use_lds) if (int rc = allow_big_lds(s->ks)) return rc;
    *out = s.release();
    return RT_OK;
}

// -------------------------------------------------------------------------------------------------
// Launch helpers
// -------------------------------------------------------------------------------------------------
static inline uint32_t grid_for(const rt_scene* s, uint64_t work, uint32_t blocks_per_cu) {
    uint64_t need = (work + BLOCK - 1) / BLOCK;
    uint64_t cap = (uint64_t)s->n_cu * blocks_per_cu;
    return (uint32_t)std::max<uint64_t>(1, std::min(need, cap));
}
static inline size_t stack_bytes(const rt_scene* s) { return (size_t)s->stack_depth * BLOCK * (s->use_wide ? sizeof(uint2) : sizeof(int)); }
static inline size_t any_stack_bytes(const rt_scene* s) { return s->use_quad ? (size_t)(s->quad_levels + 1) * BLOCK * sizeof(uint2) : stack_bytes(s); }      // the any-hit kernels' stack columns
// dynamic LDS of a kernel that holds the tree: the image and the lanes' stack columns.  closest: the W-wide form also carries the objects' exact boxes
static inline size_t lds_binary_bytes(const rt_scene* s, int block) { return s->dev.lds_image_bytes + (size_t)s->stack_depth * block * sizeof(unsigned short); }
static inline size_t lds_tree_bytes(const rt_scene* s, int block, bool closest) {
    if (closest ? s->use_ldsw_closest : s->use_ldsw) return (closest ? s->dev.ldsw_all_bytes : s->dev.ldsw_any_bytes) + (size_t)(s->ldsw_levels + 1) * block * sizeof(LdswEntry);
    return lds_binary_bytes(s, block);
}
static inline uint32_t lds_grid_for(const rt_scene* s, uint64_t work, int block, size_t lds_bytes) {
    uint64_t need = (work + block - 1) / block;
    uint64_t per_cu = std::max<uint64_t>(1, (160 * 1024) / lds_bytes);
    per_cu = std::min<uint64_t>(per_cu, 2048 / block);
    return (uint32_t)std::max<uint64_t>(1, std::min(need, (uint64_t)s->n_cu * per_cu));
}
static inline int closest_block(const rt_scene* s) { return s->use_ldsw_closest ? LDSW_BLOCK_CLOSEST : LDS_BLOCK_CLOSEST; }
static inline uint32_t lds_grid(const rt_scene* s, uint64_t work, int block, bool closest) { return lds_grid_for(s, work, block, lds_tree_bytes(s, block, closest)); }

// -------------------------------------------------------------------------------------------------
// Parity hooks
// -------------------------------------------------------------------------------------------------
// Staging for the two hooks lives in the scene and is reused by later calls (the hooks are the public TryFindClosestHit /
// AnyIntersectionFast replacements, RenderJob.cs:575-617: a caller that asks often must not pay two hipMallocs per call).  A request
// beyond TRACE_KEEP_BYTES still works; its buffers are released again afterwards.
constexpr size_t TRACE_KEEP_BYTES = 256u << 20;
static int trace_staging(DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return RT_OK;
    HIP_TRY(b.alloc(bytes));
    return RT_OK;
}
static void trace_release_large(rt_scene* s) {
    for (DevBuf* b : {&s->trace_in, &s->trace_out, &s->trace_k}) if (b->bytes > TRACE_KEEP_BYTES) (void)b->alloc(0);
}

int rt_trace_closest(rt_scene* s, const rt_ray* rays, uint64_t n, float k_min, float k_max, uint64_t seed, rt_hit* hits) {
    if (!s || (n && (!rays || !hits))) return fail(RT_ERR_INVALID_ARG, "rt_trace_closest: null argument");
    if (!n) return RT_OK;
    if (n > (1ull << 31)) return fail(RT_ERR_TOO_LARGE, "rt_trace_closest: at most 2^31 rays per call");
    DeviceGuard guard(s->device);
    if (int rc = trace_staging(s->trace_in, n * sizeof(rt_ray))) return rc;
    if (int rc = trace_staging(s->trace_out, n * sizeof(rt_hit))) return rc;
    rt_ray* dr = s->trace_in.as<rt_ray>(); rt_hit* dh = s->trace_out.as<rt_hit>();
    HIP_TRY(hipMemcpy(dr, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice));
    if (s->use_lds) hipLaunchKernelGGL(s->use_ldsw_closest ? s->ks->ldsw.trace_closest : s->ks->trace_closest_lds, dim3(lds_grid(s, n, closest_block(s), true)), dim3(closest_block(s)), lds_tree_bytes(s, closest_block(s), true), 0, s->dev, dr, n, k_min, k_max, seed, dh);
    else hipLaunchKernelGGL(s->use_wide ? s->ks->wide.trace_closest : s->ks->trace_closest, dim3(grid_for(s, n, 8)), dim3(BLOCK), stack_bytes(s), 0, s->dev, dr, n, k_min, k_max, seed, dh);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(hits, dh, n * sizeof(rt_hit), hipMemcpyDeviceToHost));           // (null stream: ordered after the kernel, blocks until the copy is done)
    trace_release_large(s);
    return RT_OK;
}

int rt_trace_any(rt_scene* s, const rt_ray* rays, const float* k_min, const float* k_max, uint64_t n, uint8_t* occluded) {
    if (!s || (n && (!rays || !k_min || !k_max || !occluded))) return fail(RT_ERR_INVALID_ARG, "rt_trace_any: null argument");
    if (!n) return RT_OK;
    if (n > (1ull << 31)) return fail(RT_ERR_TOO_LARGE, "rt_trace_any: at most 2^31 rays per call");
    DeviceGuard guard(s->device);
    if (int rc = trace_staging(s->trace_in, n * sizeof(rt_ray))) return rc;
    if (int rc = trace_staging(s->trace_k, 2 * n * sizeof(float))) return rc;
    if (int rc = trace_staging(s->trace_out, n)) return rc;
    rt_ray* dr = s->trace_in.as<rt_ray>(); float* d0 = s->trace_k.as<float>(); float* d1 = d0 + n; uint8_t* dout = s->trace_out.as<uint8_t>();
    HIP_TRY(hipMemcpy(dr, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d0, k_min, n * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d1, k_max, n * sizeof(float), hipMemcpyHostToDevice));
    if (s->use_lds) hipLaunchKernelGGL(s->use_ldsw ? s->ks->ldsw.trace_any : s->ks->trace_any_lds, dim3(lds_grid(s, n, LDS_BLOCK_SHADOW, false)), dim3(LDS_BLOCK_SHADOW), lds_tree_bytes(s, LDS_BLOCK_SHADOW, false), 0, s->dev, dr, d0, d1, n, dout);
    else hipLaunchKernelGGL(s->use_quad ? s->ks->quad.trace_any : s->use_wide ? s->ks->wide.trace_any : s->ks->trace_any, dim3(grid_for(s, n, 8)), dim3(BLOCK), any_stack_bytes(s), 0, s->dev, dr, d0, d1, n, dout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(occluded, dout, n, hipMemcpyDeviceToHost));
    trace_release_large(s);
    return RT_OK;
}

int rt_trace_closest_noalloc(rt_scene* s, const rt_ray* rays, uint64_t n, float k_min, float k_max, uint64_t seed, rt_hit* hits) {
    if (!s || (n && (!rays || !hits))) return fail(RT_ERR_INVALID_ARG, "rt_trace_closest_noalloc: null argument");
    if (n > RT_TRACE_NOALLOC_MAX_RAYS) return fail(RT_ERR_TOO_LARGE, "rt_trace_closest_noalloc: %llu rays, at most %u per call (use rt_trace_closest, which allocates)", (unsigned long long)n, (unsigned)RT_TRACE_NOALLOC_MAX_RAYS);
    if (!n) return RT_OK;
    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(RT_ERR_HIP, "cannot select device %d", s->device);
    if (int rc = scratch_for_device(s->device)) return rc;      // normally resolved by rt_scene_create already; hipGetSymbolAddress allocates nothing
    void* base = nullptr; hipStream_t st = nullptr;
    { std::lock_guard<std::mutex> lock(g_scratch_mutex); base = g_scratch[s->device]; st = g_scratch_stream[s->device]; }
    // [rays: MAX x rt_ray][hits: MAX x rt_hit]; every hit record of 0..n-1 is written by the kernel, nothing has to start at zero.
    // Copies and launch go to the device's own non-blocking stream: the call neither waits for nor orders itself against the null stream.
    rt_ray* d_rays = (rt_ray*)base;
    rt_hit* d_hits = (rt_hit*)((unsigned char*)base + (size_t)RT_TRACE_NOALLOC_MAX_RAYS * sizeof(rt_ray));
    HIP_TRY(hipMemcpyAsync(d_rays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice, st));
    if (s->use_lds) hipLaunchKernelGGL(s->use_ldsw_closest ? s->ks->ldsw.trace_closest : s->ks->trace_closest_lds, dim3(lds_grid(s, n, closest_block(s), true)), dim3(closest_block(s)), lds_tree_bytes(s, closest_block(s), true), st, s->dev, d_rays, n, k_min, k_max, seed, d_hits);
    else hipLaunchKernelGGL(s->use_wide ? s->ks->wide.trace_closest : s->ks->trace_closest, dim3(grid_for(s, n, 8)), dim3(BLOCK), stack_bytes(s), st, s->dev, d_rays, n, k_min, k_max, seed, d_hits);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(hits, d_hits, n * sizeof(rt_hit), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

// -------------------------------------------------------------------------------------------------
// Render job
// -------------------------------------------------------------------------------------------------
// everything a render handle owns (also the clean-up of a half-built one: rt_render_create fails through it)
static void release_render(rt_render* r) {
    for (size_t g = 0; g < r->ev_gather.size(); g++) if (r->ev_gather[g]) { DeviceGuard dg(r->shards[g + 1]->device); (void)hipEventDestroy(r->ev_gather[g]); }
    r->ev_gather.clear();
    for (rt_render* sh : r->shards) if (sh) { DeviceGuard g(sh->device); release_render(sh); }
    r->shards.clear();
    for (rt_scene* sc : r->replicas) if (sc) { DeviceGuard g(sc->device); delete sc; }
    r->replicas.clear();
    for (int si = 0; si < MAX_SLOTS; si++) if (r->slots[si].stream) (void)hipStreamSynchronize(r->slots[si].stream);
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    collect_stamps(r);
    for (hipEvent_t e : r->free_events) (void)hipEventDestroy(e);
    for (int si = 0; si < MAX_SLOTS; si++) {
        if (r->ev_acc[si]) (void)hipEventDestroy(r->ev_acc[si]);
        if (si > 0 && r->slots[si].stream) (void)hipStreamDestroy(r->slots[si].stream);
    }
    if (r->ev_fork) (void)hipEventDestroy(r->ev_fork);
    if (r->own_stream && r->stream) (void)hipStreamDestroy(r->stream);
    if (r->scene) r->scene->live_renders.fetch_sub(1);
    delete r;
}
struct RenderDeleter { void operator()(rt_render* r) const { release_render(r); } };

static int check_render_opts(const rt_render_opts* o) {
    if (o->struct_size != sizeof(rt_render_opts)) return fail(RT_ERR_INVALID_ARG, "rt_render_opts size mismatch (got %u, want %zu)", o->struct_size, sizeof(rt_render_opts));
    if (o->width < 1 || o->height < 1) return fail(RT_ERR_INVALID_ARG, "Image cannot have a width/height of less than 1 pixel");          // RenderOptions.cs:52,63
    if (o->k_min < 0 || std::isnan(o->k_min)) return fail(RT_ERR_INVALID_ARG, "KMin cannot have a value of less than 0");                   // :74
    if (o->k_max < o->k_min || std::isnan(o->k_max)) return fail(RT_ERR_INVALID_ARG, "KMax cannot have a value of less than 0 or less than KMin"); // :86
    if (o->light_sample_count_hint < 1) return fail(RT_ERR_INVALID_ARG, "Must have at least 1 sample per light");                         // :160
    if (o->debug_visualisation > RT_VIS_BOUNCE_DEPTH || o->debug_visualisation == RT_VIS_UNDEFINED_TEST)
        return fail(RT_ERR_INVALID_ARG, "Wrong enum value for debug visualisation (%u)", o->debug_visualisation);                           // RenderJob.cs:307-308
    if (o->max_bounce_depth > 4096) return fail(RT_ERR_TOO_LARGE, "max_bounce_depth above 4096");
    if ((uint64_t)o->width * o->height > (1ull << 30)) return fail(RT_ERR_TOO_LARGE, "more than 2^30 pixels");
    if (o->flags & ~(RT_RENDER_EXACT_LIGHT_SAMPLING | RT_RENDER_EXACT_OCCLUSION | RT_RENDER_VALIDATE)) return fail(RT_ERR_INVALID_ARG, "unknown bits in rt_render_opts.flags (0x%x)", o->flags);
    if (o->n_devices > RT_MAX_DEVICES) return fail(RT_ERR_TOO_LARGE, "n_devices %u above RT_MAX_DEVICES (%d)", o->n_devices, RT_MAX_DEVICES);
    return RT_OK;
}

// one GPU: the scene's
static int create_single(rt_scene* s, const rt_render_opts* o, rt_render** out) {
    if (o->device >= 0 && o->device != s->device) return fail(RT_ERR_INVALID_ARG, "render device %d differs from the scene's device %d", o->device, s->device);
    const uint64_t n_pixels = (uint64_t)o->width * o->height;

    DeviceGuard guard(s->device);
    if (!guard.ok) return fail(RT_ERR_HIP, "cannot select device %d", s->device);
    std::unique_ptr<rt_render, RenderDeleter> r(new rt_render());
    r->scene = s; r->device = s->device; r->opts = *o;
    s->live_renders.fetch_add(1);
    { const char* e = getenv("RT_HIP_EXACT_LIGHTS"); r->exact_lights = (o->flags & RT_RENDER_EXACT_LIGHT_SAMPLING) || (e && e[0] == '1'); }
    { const char* e = getenv("RT_HIP_EXACT_OCCLUSION"); r->exact_occlusion = (o->flags & RT_RENDER_EXACT_OCCLUSION) || (e && e[0] == '1'); }
    // paths per batch: the alive paths fall to ~1 % by bounce 10, and a launch needs >> 0.5 M rays to fill 256 CUs, so batches are
    // big (128 M paths = 26 GB of state per slot): sized for 288 GB of HBM, not for a cache — but never for more passes than the
    // caller will ask for in one call (max_passes_per_call), and never for more than the device has free right now.
    const uint64_t target_paths = 128ull << 20;
    uint32_t ppb = o->passes_per_batch;
    if (!ppb) {
        ppb = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(64, target_paths / n_pixels));
        if (o->max_passes_per_call) ppb = std::min(ppb, o->max_passes_per_call);
        // A call shorter than two full batches is split in two, so that both batch slots work on it (one batch per slot leaves no
        // other batch to overlap with).  Measured, Night Spheres 1080p, one job end to end (scripts/strong_batching.py): 64 passes as
        // 1 x 64 3298, as 2 x 32 3431 Msamples/s; 32 passes 3084 -> 3163; 16 passes 2796 -> 2894 (steady state, 256 passes: 3590).
        if (o->max_passes_per_call >= 16 && o->max_passes_per_call < 2 * ppb) ppb = (o->max_passes_per_call + 1) / 2;
        // ... and a batch's passes are numbered in groups of the largest power of two that divides their number (enqueue_batch), so a
        // multiple of 8 where there is room for one (groups of 8 give most of what groups of 16 do), of 4 or 2 below
        if (ppb >= 8) ppb &= ~7u; else if (ppb >= 4) ppb = 4; else if (ppb >= 2) ppb = 2;
    }
    if (n_pixels * ppb > (1ull << 30)) return fail(RT_ERR_TOO_LARGE, "passes_per_batch * pixels exceeds 2^30 paths");
    // concurrent batches: default 2 (RT_HIP_SLOTS overrides, test/tuning knob)
    int n_slots = 2;
    if (const char* e = getenv("RT_HIP_SLOTS")) n_slots = std::max(1, std::min(MAX_SLOTS, atoi(e)));
    // path state: float4 arrays (2 x 3 state, 1 out, 3 (+1) shadow), then ids (2 x uint2), hit (float2) and hit_aux (float).
    // passes_per_batch is an upper bound: when the device cannot hold it (another job, a smaller card) the batch is halved until it fits.
    const bool has_phong = (s->ks->mats >> RT_MAT_PHONG) & 1u, has_medium = (s->ks->prims >> RT_HIT_MEDIUM) & 1u;
    const size_t n_f4 = 6 + 1 + 3 + (has_phong ? 1 : 0);
    const size_t bytes_per_path = n_f4 * sizeof(float4) + 2 * sizeof(uint2) + sizeof(float2) + (has_medium ? sizeof(float) : 0);
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b)
            while (ppb > 1 && (double)n_pixels * ppb * bytes_per_path * n_slots > 0.92 * (double)free_b) ppb = std::max<uint32_t>(1, ppb / 2);
    }
    for (;;) {
        hipError_t e = hipSuccess;
        for (int si = 0; si < n_slots && e == hipSuccess; si++) e = r->slots[si].f32_pool.alloc((size_t)(n_pixels * ppb) * bytes_per_path);
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        for (int si = 0; si < n_slots; si++) (void)r->slots[si].f32_pool.alloc(0);
        if (e != hipErrorOutOfMemory) return fail(RT_ERR_HIP, "hipMalloc of the path state failed: %s", hipGetErrorString(e));     // only a shortage of memory is a reason to shrink
        if (ppb == 1) return fail(RT_ERR_OUT_OF_MEMORY, "not enough device memory for the path state of one pass (%zu bytes per slot)", (size_t)n_pixels * bytes_per_path);
        ppb = std::max<uint32_t>(1, ppb / 2);
    }
    r->passes_per_batch = ppb;
    const size_t P = (size_t)(n_pixels * ppb);
    r->max_paths = P;

    if (o->stream) r->stream = (hipStream_t)o->stream;
    else { HIP_TRY(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking)); r->own_stream = true; }

    r->n_slots = n_slots;
    if (o->flags & RT_RENDER_VALIDATE) {
        if (s->n_objects > (1u << 16)) return fail(RT_ERR_TOO_LARGE, "RT_RENDER_VALIDATE keeps two counters per object: at most 65536 objects (the scene has %u)", s->n_objects);
        HIP_TRY(r->validate.alloc(std::max<size_t>(1, s->n_objects) * 2 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(r->validate.p, 0, r->validate.bytes, r->stream));
    }
    HIP_TRY(r->stats.alloc(ST_COUNT * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(r->stats.p, 0, r->stats.bytes, r->stream));
    HIP_TRY(r->rgb.alloc(n_pixels * 3));
    if (o->accum_device) r->accum = o->accum_device;
    else {
        HIP_TRY(r->accum_buf.alloc(n_pixels * 3 * sizeof(float)));
        HIP_TRY(hipMemsetAsync(r->accum_buf.p, 0, r->accum_buf.bytes, r->stream));
        r->accum = r->accum_buf.as<float>(); r->own_accum = true;
    }
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming));

    const uint32_t n_cnt = o->max_bounce_depth + 2;
    for (int si = 0; si < n_slots; si++) {
        Slot& sl = r->slots[si];
        if (si == 0) sl.stream = r->stream;
        else if (const char* e = getenv("RT_HIP_SLOT_PRIORITY")) {      // A/B knob: the other batch slots' streams at the highest (1) or lowest (-1) priority
            int lo = 0, hi = 0; HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIP_TRY(hipStreamCreateWithPriority(&sl.stream, hipStreamNonBlocking, atoi(e) > 0 ? hi : atoi(e) < 0 ? lo : 0));
        }
        else HIP_TRY(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&r->ev_acc[si], hipEventDisableTiming));
        HIP_TRY(sl.counters.alloc(2 * n_cnt * sizeof(uint32_t)));
        RenderDev& rd = sl.rd;
        float4* f = sl.f32_pool.as<float4>();
        auto nf = [&]() { float4* p = f; f += P; return p; };
        for (int h = 0; h < 2; h++) for (int i = 0; i < 3; i++) rd.st[h][i] = nf();
        rd.out = nf();
        for (int i = 0; i < 3; i++) rd.sh[i] = nf();
        rd.sh[3] = has_phong ? nf() : nullptr;
        rd.ids[0] = (uint2*)f; rd.ids[1] = rd.ids[0] + P;
        rd.hit = (float2*)(rd.ids[1] + P);
        rd.hit_aux = has_medium ? (float*)(rd.hit + P) : nullptr;
        rd.qcount = sl.counters.as<uint32_t>(); rd.sqcount = rd.qcount + n_cnt;
        rd.stats = r->stats.as<unsigned long long>();
        rd.width = o->width; rd.height = o->height; rd.n_pixels = (uint32_t)n_pixels;
        {   // slots of a pass tile by tile (rt_kernels.h: slot_pixel): 32 x 32 pixels where the width is a multiple of 32, else 8 x 8 where it is one
            // of 8, rows that fill no tile in row order; RT_HIP_TILES=0 / 8 / 32: row order / 8 x 8 at most / as said (A/B, tests)
            const char* e = getenv("RT_HIP_TILES");
            const uint32_t want = e ? (uint32_t)atoi(e) : 32u;
            uint32_t sh = 0;
            for (uint32_t t : {5u, 3u}) if (!sh && want >= (1u << t) && (o->width % (1u << t)) == 0 && o->width >= (2u << t) && o->height >= (1u << t)) sh = t;
            const uint32_t tpr = sh ? o->width >> sh : 1u, bands = sh ? o->height >> sh : 0u;
            const bool tiles = sh && (uint64_t)tpr * bands * tpr < (1ull << 32);
            rd.tiled_slots = tiles ? (bands << sh) * o->width : 0u; rd.tiles_per_row = tiles ? tpr : 1u; rd.tile_shift = tiles ? sh : 0u;
            rd.tile_magic = tiles ? (uint32_t)(((1ull << 32) + tpr - 1) / tpr) : 0u;
        }
        rd.seed = o->seed; rd.k_min = o->k_min; rd.k_max = o->k_max;
        rd.max_depth = o->max_bounce_depth; rd.light_samples = o->light_sample_count_hint;
        rd.compact_requests = (!s->has_phong && o->debug_visualisation == RT_VIS_NONE && !getenv("RT_HIP_FULL_REQUESTS")) ? 1u : 0u;
        rd.misses_end_in_closest = (s->use_lds && o->debug_visualisation == RT_VIS_NONE) ? 1u : 0u;
        rd.any_cap = r->exact_occlusion ? INFINITY : RT_ANY_GROWTH_CAP;
        rd.skip_mute_rays = ((uint64_t)rd.light_samples * s->n_lights > 2 || s->any_light_cutoff) ? 1u : 0u;
        rd.vis = o->debug_visualisation;
        rd.validate = r->validate.as<unsigned long long>();
    }
    r->rd = r->slots[0].rd;
    r->stats_host.assign(ST_COUNT, 0);
    *out = r.release();
    return RT_OK;
}

int rt_render_create(rt_scene* s, const rt_render_opts* o, rt_render** out) {
    if (!s || !o || !out) return fail(RT_ERR_INVALID_ARG, "rt_render_create: null argument");
    *out = nullptr;
    if (int rc = check_render_opts(o)) return rc;
    if (o->n_devices <= 1) {
        if (o->n_devices == 1 && o->device_ids[0] != s->device) return fail(RT_ERR_INVALID_ARG, "device_ids[0] = %d differs from the scene's device %d", o->device_ids[0], s->device);
        return create_single(s, o, out);
    }
    // ---- several GPUs behind one handle: shard 0 on the caller's scene, shard g on a peer copy of it on device_ids[g]
    if (o->stream || o->accum_device) return fail(RT_ERR_INVALID_ARG, "stream / accum_device belong to one device: they must be NULL when n_devices > 1");
    if (o->device_ids[0] != s->device) return fail(RT_ERR_INVALID_ARG, "device_ids[0] = %d must be the scene's device %d (the shards are combined there)", o->device_ids[0], s->device);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(RT_ERR_NO_DEVICE, "no HIP device is visible");
    for (uint32_t g = 0; g < o->n_devices; g++)
        if (o->device_ids[g] < 0 || o->device_ids[g] >= ndev) return fail(RT_ERR_INVALID_ARG, "device_ids[%u] = %d out of range (%d visible)", g, o->device_ids[g], ndev);
    std::unique_ptr<rt_render, RenderDeleter> r(new rt_render());
    r->scene = s; r->device = s->device; r->opts = *o;
    s->live_renders.fetch_add(1);
    rt_render_opts so = *o;
    so.n_devices = 0; so.device = -1;
    // a call's passes are split G ways, so a shard never sees more than ceil(max / G) of them
    if (so.max_passes_per_call) so.max_passes_per_call = (so.max_passes_per_call + o->n_devices - 1) / o->n_devices;
    for (uint32_t g = 0; g < o->n_devices; g++) {
        rt_scene* sc = s;
        if (g > 0) {
            if (int rc = clone_scene(s, o->device_ids[g], &sc)) return rc;
            r->replicas.push_back(sc);
        }
        rt_render* sh = nullptr;
        if (int rc = create_single(sc, &so, &sh)) return rc;
        r->shards.push_back(sh);
    }
    for (uint32_t g = 1; g < o->n_devices; g++) {         // (the event a shard's copy is recorded with lives on that shard's device)
        DeviceGuard dg(r->shards[g]->device);
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        r->ev_gather.push_back(e);
    }
    DeviceGuard guard(s->device);
    const size_t bytes = (size_t)o->width * o->height * 3 * sizeof(float);
    HIP_TRY(r->combined.alloc(bytes));
    for (uint32_t g = 1; g < o->n_devices; g++) { r->gather.emplace_back(new DevBuf()); HIP_TRY(r->gather.back()->alloc(bytes)); }
    r->rd = r->shards[0]->rd;
    *out = r.release();
    return RT_OK;
}

int rt_render_destroy(rt_render* r) {
    if (!r) return RT_OK;
    DeviceGuard guard(r->device);
    release_render(r);
    return RT_OK;
}

// One wavefront batch of `np` passes starting at pass `first` (stride `stride`) on slot `si`; enqueues only.
static int enqueue_batch(rt_render* r, int si, uint64_t first, uint32_t np, uint32_t stride) {
    rt_scene* s = r->scene;
    Slot& sl = r->slots[si];
    RenderDev rd = sl.rd;
    rd.n_paths = rd.n_pixels * np; rd.first_pass = first; rd.pass_stride = stride;
    {   // the batch's passes in groups of the largest power of two that divides their number, at most 16 (rt_kernels.h: path_pass_pixel); RT_HIP_PASS_GROUP=n: at most 2^n, n <= 6.
        // Measured: 5 M triangles (16 passes per batch) 1068 / 1110 / 1115 / 1142 Msamples/s in short runs with groups of 1 / 4 / 8 / 16; Night Spheres
        // (64 per batch) 3794 / 3804 / 3820 / 3802, and 3682 with groups of 64, where k_accumulate's four adding threads per workgroup cost more than the walks gain
        const char* e = getenv("RT_HIP_PASS_GROUP"); const int cap = e ? std::min(6, std::max(0, atoi(e))) : 4;
        uint32_t sh = 0;
        while ((int)sh < cap && !((np >> sh) & 1u)) sh++;
        rd.pass_shift = sh; rd.group_paths = rd.n_pixels << sh;
    }
    hipStream_t st = sl.stream;
    const KernelSet* ks = s->ks;
    const uint32_t n_cnt = r->opts.max_bounce_depth + 2;
    // debug visualisations look at the first hit only (RenderJob.cs:154), except BounceDepth which runs the bounce loop
    const uint32_t vis = r->opts.debug_visualisation;
    const bool first_hit_only = vis != RT_VIS_NONE && vis != RT_VIS_BOUNCE_DEPTH;
    const uint32_t max_depth = first_hit_only ? 0u : r->opts.max_bounce_depth;
    if (first_hit_only) rd.max_depth = 0;
    const bool validate = (r->opts.flags & RT_RENDER_VALIDATE) != 0;
    auto shade_kernel = vis != RT_VIS_NONE ? (validate ? k_shade_vis_validate : k_shade_vis) : validate ? k_shade_validate : ks->shade;
    const bool trace_shadows = s->n_lights && (vis == RT_VIS_NONE || vis == RT_VIS_ESTIMATED_LIGHT);
    HIP_TRY(hipMemsetAsync(rd.qcount, 0, 2 * n_cnt * sizeof(uint32_t), st));
    static const int trace_blocks_per_cu = [] { const char* e = getenv("RT_HIP_TRACE_BLOCKS_PER_CU"); return e ? std::max(1, atoi(e)) : RT_TRACE_BLOCKS_PER_CU; }();      // (A/B knob)
    const uint32_t g_light = grid_for(s, rd.n_paths, 8), g_trace = grid_for(s, rd.n_paths, (uint32_t)trace_blocks_per_cu);
    const size_t lds = stack_bytes(s);
    // A request with many rays (64 lights: 64 traversals, each ending at its own time) is where a lane should not wait for its
    // wave: 100 k spheres + 64 lights 766 -> 728 ms per batch; with 2-4 rays per request the refill rounds cost more than they save.
    const bool shadow_persistent = s->persistent_shadow >= 0 ? s->persistent_shadow != 0 : (uint64_t)rd.light_samples * s->n_lights >= 32;
    const bool ex = r->exact_lights;
    const uint64_t per_request = (uint64_t)rd.light_samples * s->n_lights;
    const bool ray_lanes_ok = !s->has_phong && per_request >= 1 && per_request <= (uint64_t)(s->use_lds ? LDS_BLOCK_RAYS : BLOCK) && (s->use_lds ? ks->shadow_rays_lds != nullptr : ks->shadow_rays != nullptr);
    // Measured per shadow kernel and batch (RT_HIP_SHADOW_RAYS=0/1): Cornell (list, 2 rays per request) 53.4 -> 49.6 ms, the 5 M-triangle
    // mesh 47.7 -> 42.8, Demo 2.20 -> 2.12; Night Spheres (tree in LDS) 67.3 -> 75.9 — two rays per request do not pay for loading the
    // request twice and the two barriers — and 100 k spheres + 64 lights 570 (persistent request lanes) -> 768: a wave that IS one
    // request waits for the longest of its 64 rays.  Hence: trees outside LDS, requests of fewer than 32 rays.
    const bool ray_lanes = ray_lanes_ok && (s->shadow_rays >= 0 ? s->shadow_rays != 0 : (s->persistent_shadow < 0 && !s->use_lds && per_request < 32));
    // depth 0 without stored state (rt_kernels.h: primary_ray) unless a debug visualisation's kernel reads it (RT_HIP_RAYGEN=1: the stored form, for A/B runs)
    {   // the object records in LDS when four workgroups' copies fit a CU beside everything else (<= 32 KB: 512 objects)
        const size_t rec_bytes = s->obj_shade.bytes;
        const char* e = getenv("RT_HIP_OBJ_SHADE_LDS");
        rd.obj_shade_lds_bytes = (vis == RT_VIS_NONE && rec_bytes && rec_bytes <= 32 * 1024 && !(e && e[0] == '0')) ? (uint32_t)rec_bytes : 0u;
    }
    rd.implicit_primary = (vis == RT_VIS_NONE && !getenv("RT_HIP_RAYGEN")) ? (s->use_lds && RT_PRIMARY_STORE ? 2u : 1u) : 0u;
    if (!rd.implicit_primary) { Timed t(r, RT_K_RAYGEN, st); hipLaunchKernelGGL(k_raygen, dim3(g_light), dim3(BLOCK), 0, st, s->dev, rd); }
    for (uint32_t depth = 0; depth <= max_depth; depth++) {
        { Timed t(r, RT_K_CLOSEST, st);
          const bool prim = rd.implicit_primary && depth == 0;
          if (s->use_ldsw_closest) hipLaunchKernelGGL(prim ? ks->ldsw.closest_primary : ks->ldsw.closest, dim3(lds_grid(s, rd.n_paths, LDSW_BLOCK_CLOSEST, true)), dim3(LDSW_BLOCK_CLOSEST), lds_tree_bytes(s, LDSW_BLOCK_CLOSEST, true), st, s->dev, rd, depth);
          else if (s->use_lds) hipLaunchKernelGGL(prim ? ks->closest_primary_lds : ks->closest_lds, dim3(lds_grid(s, rd.n_paths, LDS_BLOCK_CLOSEST, true)), dim3(LDS_BLOCK_CLOSEST), lds_tree_bytes(s, LDS_BLOCK_CLOSEST, true), st, s->dev, rd, depth);
          else if (s->use_wide) hipLaunchKernelGGL(prim ? ks->wide.closest_primary : s->persistent && depth > 0 ? ks->wide.closest_persistent : ks->wide.closest, dim3(g_trace), dim3(BLOCK), lds, st, s->dev, rd, depth);
          else hipLaunchKernelGGL(prim ? ks->closest_primary : s->persistent && depth > 0 ? ks->closest_persistent : ks->closest, dim3(g_trace), dim3(BLOCK), lds, st, s->dev, rd, depth); }
        { Timed t(r, RT_K_SHADE, st); hipLaunchKernelGGL(shade_kernel, dim3(g_light), dim3(BLOCK), rd.obj_shade_lds_bytes, st, s->dev, rd, depth); }
        if (trace_shadows) { Timed t(r, RT_K_SHADOW, st);
          if (ray_lanes && s->use_lds) hipLaunchKernelGGL(ex ? ks->shadow_rays_lds_exact : ks->shadow_rays_lds, dim3(lds_grid_for(s, (uint64_t)rd.n_paths * per_request, LDS_BLOCK_RAYS, lds_binary_bytes(s, LDS_BLOCK_RAYS) + 3 * LDS_BLOCK_RAYS * sizeof(float))), dim3(LDS_BLOCK_RAYS), lds_binary_bytes(s, LDS_BLOCK_RAYS), st, s->dev, rd, depth);
          else if (ray_lanes) hipLaunchKernelGGL(s->use_quad ? (ex ? ks->quad.shadow_rays_exact : ks->quad.shadow_rays) : s->use_wide ? (ex ? ks->wide.shadow_rays_exact : ks->wide.shadow_rays) : (ex ? ks->shadow_rays_exact : ks->shadow_rays), dim3(grid_for(s, (uint64_t)rd.n_paths * per_request, 8)), dim3(BLOCK), any_stack_bytes(s), st, s->dev, rd, depth);
          else if (s->use_ldsw) hipLaunchKernelGGL(ex ? ks->ldsw.shadow_exact : ks->ldsw.shadow, dim3(lds_grid(s, rd.n_paths, LDS_BLOCK_SHADOW, false)), dim3(LDS_BLOCK_SHADOW), lds_tree_bytes(s, LDS_BLOCK_SHADOW, false), st, s->dev, rd, depth);
          else if (s->use_lds) hipLaunchKernelGGL(ex ? ks->shadow_lds_exact : ks->shadow_lds, dim3(lds_grid(s, rd.n_paths, LDS_BLOCK_SHADOW, false)), dim3(LDS_BLOCK_SHADOW), lds_tree_bytes(s, LDS_BLOCK_SHADOW, false), st, s->dev, rd, depth);
          else if (s->use_quad) hipLaunchKernelGGL(shadow_persistent ? (ex ? ks->quad.shadow_persistent_exact : ks->quad.shadow_persistent) : (ex ? ks->quad.shadow_exact : ks->quad.shadow), dim3(g_trace), dim3(BLOCK), any_stack_bytes(s), st, s->dev, rd, depth);
          else if (s->use_wide) hipLaunchKernelGGL(shadow_persistent ? (ex ? ks->wide.shadow_persistent_exact : ks->wide.shadow_persistent) : (ex ? ks->wide.shadow_exact : ks->wide.shadow), dim3(g_trace), dim3(BLOCK), any_stack_bytes(s), st, s->dev, rd, depth);
          else hipLaunchKernelGGL(shadow_persistent ? (ex ? ks->shadow_persistent_exact : ks->shadow_persistent) : (ex ? ks->shadow_exact : ks->shadow), dim3(g_trace), dim3(BLOCK), lds, st, s->dev, rd, depth); }
    }
    // the adds into the image sums happen in batch order whatever slot a batch ran on
    if (r->batches_enqueued > 0 && r->n_slots > 1) HIP_TRY(hipStreamWaitEvent(st, r->ev_acc[(si + r->n_slots - 1) % r->n_slots], 0));
    { Timed t(r, RT_K_ACCUMULATE, st); hipLaunchKernelGGL(k_accumulate, dim3(grid_for(s, rd.n_pixels, 8)), dim3(BLOCK), 0, st, rd, r->accum, np); }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(r->ev_acc[si], st));
    r->batches_enqueued++;
    return RT_OK;
}

static int render_passes_single(rt_render* r, uint64_t first_pass, uint32_t n_passes, uint32_t pass_stride) {
    DeviceGuard guard(r->device);
    if (!guard.ok) return fail(RT_ERR_HIP, "cannot select device %d", r->device);
    // per-kernel event timing wants kernels that do not overlap: one slot while profiling
    const int n_slots = r->profiling ? 1 : r->n_slots;
    // the other slots start after whatever the caller already queued on the main stream
    if (n_slots > 1) {
        HIP_TRY(hipEventRecord(r->ev_fork, r->stream));
        for (int si = 1; si < n_slots; si++) HIP_TRY(hipStreamWaitEvent(r->slots[si].stream, r->ev_fork, 0));
    }
    uint32_t done = 0;
    int last_slot = 0;
    while (done < n_passes) {
        uint32_t np = std::min(r->passes_per_batch, n_passes - done);
        // slot = batch number mod slots, so that the accumulate chain (slot i waits for slot i-1) follows the batch order
        int si = n_slots > 1 ? (int)(r->batches_enqueued % (uint64_t)r->n_slots) : 0;
        if (si >= n_slots) si = 0;
        int rc = enqueue_batch(r, si, first_pass + (uint64_t)done * pass_stride, np, pass_stride);
        if (rc) return rc;
        last_slot = si;
        done += np;
    }
    // join: the main stream continues after the last batch's accumulate (which itself follows all earlier ones)
    if (last_slot != 0) HIP_TRY(hipStreamWaitEvent(r->stream, r->ev_acc[last_slot], 0));
    HIP_TRY(hipStreamSynchronize(r->stream));
    for (int si = 1; si < r->n_slots; si++) HIP_TRY(hipStreamSynchronize(r->slots[si].stream));
    collect_stamps(r);
    r->sample_count += n_passes; r->passes_rendered += n_passes;
    return RT_OK;
}

int rt_render_passes(rt_render* r, uint64_t first_pass, uint32_t n_passes, uint32_t pass_stride) {
    if (!r) return fail(RT_ERR_INVALID_ARG, "rt_render_passes: null handle");
    if (pass_stride < 1) pass_stride = 1;
    if (r->shards.empty()) return render_passes_single(r, first_pass, n_passes, pass_stride);
    // Pass i of this call (global pass first + i * stride) goes to shard i mod G: shard g renders first + g * stride, then every
    // G-th from there.  One host thread per device drives its shard's streams; the call returns when every shard has finished.
    const uint32_t G = (uint32_t)r->shards.size();
    std::vector<int> rcs(G, RT_OK); std::vector<std::string> errs(G);
    std::vector<std::thread> th;
    for (uint32_t g = 0; g < G; g++) {
        const uint32_t n_g = n_passes > g ? (n_passes - g + G - 1) / G : 0;
        if (!n_g) continue;
        th.emplace_back([=, &rcs, &errs] {
            rcs[g] = render_passes_single(r->shards[g], first_pass + (uint64_t)g * pass_stride, n_g, pass_stride * G);
            if (rcs[g]) errs[g] = g_err;                                   // (g_err is per thread)
        });
    }
    for (auto& t : th) t.join();
    for (uint32_t g = 0; g < G; g++) if (rcs[g]) return fail(rcs[g], "device %d: %s", r->shards[g]->device, errs[g].c_str());
    r->sample_count += n_passes; r->passes_rendered += n_passes;
    return RT_OK;
}

// n_devices > 1: the shards' colour sums, added in shard order on shards[0]'s device -> r->combined: the one exchange step of the
// path (width * height * 3 floats per shard: 24.9 MB at 1080p, 99.5 MB at 4K).  Every shard pushes its sums over its own xGMI link
// into its own gather buffer, on its own stream — the G - 1 copies run side by side, not as a chain through one stream — and one
// kernel on shards[0]'s stream, which waits for the G - 1 copy events, adds them: ((s0 + s1) + s2) + ..., the order of the shard
// numbers whatever order the copies arrived in, so the image does not depend on timing.  (Not ncclReduce: its summation order is
// the collective's to choose, and the image would then differ from run to run in the last bit; for a C# host it would also be a
// second native dependency.  One process per GPU — raytracer_amd.RenderJob(rank, world_size), bench.py — does reduce with RCCL.)
static int combine_shards(rt_render* r) {
    rt_render* r0 = r->shards[0];
    const size_t nv = (size_t)r->rd.n_pixels * 3, bytes = nv * sizeof(float);
    ShardSums src{}; src.n = (uint32_t)r->shards.size(); src.p[0] = r0->accum;
    for (size_t g = 1; g < r->shards.size(); g++) {
        rt_render* sh = r->shards[g];
        DeviceGuard dg(sh->device);
        HIP_TRY(hipMemcpyPeerAsync(r->gather[g - 1]->p, r0->device, sh->accum, sh->device, bytes, sh->stream));
        HIP_TRY(hipEventRecord(r->ev_gather[g - 1], sh->stream));
        src.p[g] = r->gather[g - 1]->as<float>();
    }
    DeviceGuard guard(r0->device);
    for (size_t g = 1; g < r->shards.size(); g++) HIP_TRY(hipStreamWaitEvent(r0->stream, r->ev_gather[g - 1], 0));
    hipLaunchKernelGGL(k_add_shards, dim3(grid_for(r0->scene, nv, 8)), dim3(BLOCK), 0, r0->stream, r->combined.as<float>(), src, nv);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_render_resolve_device(rt_render* r, const float* accum_device, uint32_t sample_count, uint8_t* dst_host) {
    if (!r || !accum_device || !dst_host) return fail(RT_ERR_INVALID_ARG, "rt_render_resolve_device: null argument");
    if (sample_count < 1) return fail(RT_ERR_INVALID_ARG, "rt_render_resolve_device: no samples to resolve");
    if (!r->shards.empty()) r = r->shards[0];
    DeviceGuard guard(r->device);
    const size_t nv = (size_t)r->rd.n_pixels * 3;
    { Timed t(r, RT_K_RESOLVE); hipLaunchKernelGGL(k_resolve, dim3(grid_for(r->scene, nv, 8)), dim3(BLOCK), 0, r->stream, accum_device, sample_count, nv, r->rgb.as<uint8_t>()); }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dst_host, r->rgb.p, nv, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    collect_stamps(r);
    return RT_OK;
}

int rt_render_read_rgb24(rt_render* r, uint8_t* dst) {
    if (!r || !dst) return fail(RT_ERR_INVALID_ARG, "rt_render_read_rgb24: null argument");
    // before the first pass the reference's Image<Rgb24> is what the RenderJob constructor left (RenderJob.cs:40-47): black — no sample has been resolved into it yet
    if (r->sample_count == 0) { memset(dst, 0, (size_t)r->rd.n_pixels * 3); return RT_OK; }
    if (r->shards.empty()) return rt_render_resolve_device(r, r->accum, r->sample_count, dst);
    DeviceGuard guard(r->device);
    if (int rc = combine_shards(r)) return rc;
    return rt_render_resolve_device(r, r->combined.as<float>(), r->sample_count, dst);      // (same stream as the adds)
}

int rt_render_read_accum(rt_render* r, float* rgb_sum, uint32_t* sample_count) {
    if (!r || !rgb_sum) return fail(RT_ERR_INVALID_ARG, "rt_render_read_accum: null argument");
    DeviceGuard guard(r->device);
    const float* src = r->accum; hipStream_t st = r->stream;
    if (!r->shards.empty()) { if (int rc = combine_shards(r)) return rc; src = r->combined.as<float>(); st = r->shards[0]->stream; }
    HIP_TRY(hipMemcpyAsync(rgb_sum, src, (size_t)r->rd.n_pixels * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (sample_count) *sample_count = r->sample_count;
    return RT_OK;
}

static int stats_single(rt_render* r, rt_stats* out) {
    DeviceGuard guard(r->device);
    HIP_TRY(hipMemcpyAsync(r->stats_host.data(), r->stats.p, ST_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    const auto& h = r->stats_host;
    memset(out, 0, sizeof(*out));
    out->pixels_rendered = h[ST_SAMPLES]; out->passes_rendered = r->passes_rendered;
    out->material_scatter_count = h[ST_SCATTER]; out->material_absorbed_count = h[ST_ABSORBED];
    out->aabb_misses = h[ST_BOX_MISS]; out->hittable_intersections = h[ST_PRIM_HIT]; out->hittable_misses = h[ST_PRIMS] - h[ST_PRIM_HIT];
    out->sky_rays = h[ST_SKY]; out->bounce_limit_exceeded = h[ST_LIMIT]; out->ray_count = h[ST_RAYS]; out->shadow_rays = h[ST_SHADOW_RAYS];
    out->node_visits = h[ST_NODES]; out->prim_tests = h[ST_PRIMS]; out->shadow_node_visits = h[ST_SH_NODES]; out->shadow_prim_tests = h[ST_SH_PRIMS];
    out->normals_wrong_magnitude = h[ST_NORMAL_WARN]; out->uv_invalid = h[ST_UV_WARN];
    for (int i = 0; i < RT_DEPTH_HIST_SIZE; i++) out->ray_depth_counts[i] = h[ST_DEPTH0 + i];
    return RT_OK;
}

int rt_render_stats(rt_render* r, rt_stats* out) {
    if (!r || !out) return fail(RT_ERR_INVALID_ARG, "rt_render_stats: null argument");
    if (r->shards.empty()) return stats_single(r, out);
    memset(out, 0, sizeof(*out));
    static_assert(sizeof(rt_stats) % sizeof(uint64_t) == 0, "rt_stats is an array of 64-bit counters");
    for (rt_render* sh : r->shards) {                      // every field is a count: the job's statistics are the shards' sums
        rt_stats one;
        if (int rc = stats_single(sh, &one)) return rc;
        for (size_t i = 0; i < sizeof(rt_stats) / sizeof(uint64_t); i++) ((uint64_t*)out)[i] += ((const uint64_t*)&one)[i];
    }
    return RT_OK;
}

int rt_render_validator_counts(rt_render* r, uint64_t* normals_wrong, uint64_t* uv_invalid, uint32_t n_objects) {
    if (!r) return fail(RT_ERR_INVALID_ARG, "rt_render_validator_counts: null handle");
    if (!(r->opts.flags & RT_RENDER_VALIDATE)) return fail(RT_ERR_INVALID_ARG, "rt_render_validator_counts: the render was created without RT_RENDER_VALIDATE");
    if (n_objects != r->scene->n_objects) return fail(RT_ERR_INVALID_ARG, "rt_render_validator_counts: n_objects %u, the scene has %u", n_objects, r->scene->n_objects);
    if (normals_wrong) std::fill(normals_wrong, normals_wrong + n_objects, 0ull);
    if (uv_invalid) std::fill(uv_invalid, uv_invalid + n_objects, 0ull);
    std::vector<rt_render*> parts = r->shards.empty() ? std::vector<rt_render*>{r} : r->shards;
    std::vector<unsigned long long> host((size_t)n_objects * 2);
    for (rt_render* sh : parts) {
        if (!n_objects) break;
        DeviceGuard guard(sh->device);
        HIP_TRY(hipMemcpyAsync(host.data(), sh->validate.p, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, sh->stream));
        HIP_TRY(hipStreamSynchronize(sh->stream));
        for (uint32_t i = 0; i < n_objects; i++) { if (normals_wrong) normals_wrong[i] += host[2 * i]; if (uv_invalid) uv_invalid[i] += host[2 * i + 1]; }
    }
    return RT_OK;
}

int rt_render_set_profiling(rt_render* r, int enable) {
    if (!r) return fail(RT_ERR_INVALID_ARG, "rt_render_set_profiling: null handle");
    if (!r->shards.empty()) { for (rt_render* sh : r->shards) if (int rc = rt_render_set_profiling(sh, enable)) return rc; return RT_OK; }
    DeviceGuard guard(r->device);
    HIP_TRY(hipStreamSynchronize(r->stream));
    collect_stamps(r);
    r->profiling = enable != 0;
    memset(&r->prof, 0, sizeof(r->prof));
    return RT_OK;
}

int rt_render_kernel_profile(rt_render* r, rt_kernel_profile* out) {
    if (!r || !out) return fail(RT_ERR_INVALID_ARG, "rt_render_kernel_profile: null argument");
    if (!r->shards.empty()) {                              // launches and milliseconds summed over the shards (they run side by side)
        memset(out, 0, sizeof(*out));
        for (rt_render* sh : r->shards) {
            rt_kernel_profile one;
            if (int rc = rt_render_kernel_profile(sh, &one)) return rc;
            for (int k = 0; k < RT_K_COUNT; k++) { out->launches[k] += one.launches[k]; out->total_ms[k] += one.total_ms[k]; }
        }
        return RT_OK;
    }
    DeviceGuard guard(r->device);
    HIP_TRY(hipStreamSynchronize(r->stream));
    collect_stamps(r);
    *out = r->prof;
    return RT_OK;
}

} // extern "C"
