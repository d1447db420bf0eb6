// rtbench — a host with no Python, no PyTorch and no oracle in it: loads a .rtscene file (the rt_scene_desc tables as
// raytracer_amd/flatten.py writes them), drives include/rt_hip.h exactly as csharp/HipRenderBackend.cs does
// (scene_create -> render_create -> render_passes in batches -> read_rgb24 / stats) and prints one JSON line.
// It is the C-ABI driver SURVEY.md §8b asks for; tests/test_rtbench.py checks its picture against the ctypes host's, byte for byte.
//
//   rtbench scene.rtscene [--width W] [--height H] [--passes P] [--depth D] [--light-samples S] [--seed N]
//                         [--passes-per-call N] [--vis V] [--ppm out.ppm]
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/rt_hip.h"

namespace {

struct Table { uint32_t count = 0, elem = 0; std::vector<unsigned char> bytes; };

bool read_exact(FILE* f, void* dst, size_t n) { return n == 0 || fread(dst, 1, n, f) == n; }

// File layout (little endian): "RTSCENE1", u32 abi_version, rt_camera, rt_skybox, then 16 tables in rt_scene_desc order,
// each { u32 count, u32 element size, count * size bytes }.
bool load(const char* path, rt_scene_desc& d, std::vector<Table>& tables, std::string& err) {
    FILE* f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    char magic[8]; uint32_t abi = 0;
    bool ok = read_exact(f, magic, 8) && memcmp(magic, "RTSCENE1", 8) == 0 && read_exact(f, &abi, 4);
    memset(&d, 0, sizeof(d));
    d.struct_size = (uint32_t)sizeof(d); d.abi_version = abi;
    ok = ok && read_exact(f, &d.camera, sizeof(d.camera)) && read_exact(f, &d.skybox, sizeof(d.skybox));
    static const uint32_t sizes[16] = {sizeof(rt_object), sizeof(rt_sphere), sizeof(rt_quad), sizeof(rt_triangle), sizeof(rt_infinite_plane), sizeof(rt_axis_plane),
                                       sizeof(rt_axis_plane), sizeof(rt_axis_plane), sizeof(rt_disk), sizeof(rt_box), sizeof(rt_capsule), sizeof(rt_cylinder),
                                       sizeof(rt_medium), sizeof(rt_material), sizeof(rt_texture), sizeof(rt_light)};
    tables.assign(16, Table());
    for (int i = 0; ok && i < 16; i++) {
        Table& t = tables[i];
        ok = read_exact(f, &t.count, 4) && read_exact(f, &t.elem, 4);
        if (ok && t.elem != sizes[i]) { err = "table " + std::to_string(i) + ": element size " + std::to_string(t.elem) + " != " + std::to_string(sizes[i]); fclose(f); return false; }
        if (ok) { t.bytes.resize((size_t)t.count * t.elem); ok = read_exact(f, t.bytes.data(), t.bytes.size()); }
    }
    fclose(f);
    if (!ok) { if (err.empty()) err = "not a .rtscene file, or truncated"; return false; }
#define TBL(i, n_field, p_field, T) d.n_field = tables[i].count; d.p_field = (const T*)tables[i].bytes.data()
    TBL(0, n_objects, objects, rt_object); TBL(1, n_spheres, spheres, rt_sphere); TBL(2, n_quads, quads, rt_quad); TBL(3, n_triangles, triangles, rt_triangle);
    TBL(4, n_infinite_planes, infinite_planes, rt_infinite_plane); TBL(5, n_xy_planes, xy_planes, rt_axis_plane); TBL(6, n_xz_planes, xz_planes, rt_axis_plane);
    TBL(7, n_yz_planes, yz_planes, rt_axis_plane); TBL(8, n_disks, disks, rt_disk); TBL(9, n_boxes, boxes, rt_box); TBL(10, n_capsules, capsules, rt_capsule);
    TBL(11, n_cylinders, cylinders, rt_cylinder); TBL(12, n_media, media, rt_medium); TBL(13, n_materials, materials, rt_material);
    TBL(14, n_textures, textures, rt_texture); TBL(15, n_lights, lights, rt_light);
#undef TBL
    return true;
}

int die(const char* what) { fprintf(stderr, "rtbench: %s: %s\n", what, rt_last_error()); return 1; }
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: rtbench scene.rtscene [--width W] [--height H] [--passes P] [--depth D] [--light-samples S] [--seed N] [--passes-per-call N] [--vis V] [--ppm out.ppm]\n"); return 2; }
    uint32_t W = 1920, H = 1080, passes = 64, depth = 10, ls = 2, per_call = 0, vis = 0; uint64_t seed = 0x5EED; const char* ppm = nullptr;
    for (int i = 2; i < argc; i++) {
        auto val = [&](const char* name) -> const char* { if (strcmp(argv[i], name) != 0) return nullptr; if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", name); exit(2); } return argv[++i]; };
        const char* v;
        if ((v = val("--width"))) W = (uint32_t)strtoul(v, nullptr, 0);
        else if ((v = val("--height"))) H = (uint32_t)strtoul(v, nullptr, 0);
        else if ((v = val("--passes"))) passes = (uint32_t)strtoul(v, nullptr, 0);
        else if ((v = val("--depth"))) depth = (uint32_t)strtoul(v, nullptr, 0);
        else if ((v = val("--light-samples"))) ls = (uint32_t)strtoul(v, nullptr, 0);
        else if ((v = val("--seed"))) seed = strtoull(v, nullptr, 0);
        else if ((v = val("--passes-per-call"))) per_call = (uint32_t)strtoul(v, nullptr, 0);
        else if ((v = val("--vis"))) vis = (uint32_t)strtoul(v, nullptr, 0);
        else if ((v = val("--ppm"))) ppm = v;
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    rt_scene_desc desc; std::vector<Table> tables; std::string err;
    if (!load(argv[1], desc, tables, err)) { fprintf(stderr, "rtbench: %s\n", err.c_str()); return 2; }
    if (desc.abi_version != (uint32_t)rt_version()) { fprintf(stderr, "rtbench: file written for ABI %u, library is %d\n", desc.abi_version, rt_version()); return 2; }

    double t0 = now();
    rt_scene* scene = nullptr;
    if (rt_scene_create(&desc, -1, &scene) != RT_OK) return die("rt_scene_create");
    double t_scene = now() - t0;
    rt_scene_info info; memset(&info, 0, sizeof(info));
    if (rt_scene_get_info(scene, &info) != RT_OK) return die("rt_scene_get_info");

    rt_render_opts o; memset(&o, 0, sizeof(o));
    o.struct_size = (uint32_t)sizeof(o); o.width = W; o.height = H; o.k_min = 0.001f; o.k_max = __builtin_inff();
    o.max_bounce_depth = depth; o.light_sample_count_hint = ls; o.debug_visualisation = vis; o.seed = seed; o.device = -1;
    rt_render* job = nullptr;
    if (rt_render_create(scene, &o, &job) != RT_OK) return die("rt_render_create");

    t0 = now();
    for (uint32_t done = 0; done < passes;) {                       // the pass loop of RenderJob.RenderInternal, RenderJob.cs:92-122
        uint32_t n = per_call ? (per_call < passes - done ? per_call : passes - done) : passes - done;
        if (rt_render_passes(job, done, n, 1) != RT_OK) return die("rt_render_passes");
        done += n;
    }
    std::vector<uint8_t> rgb((size_t)W * H * 3);
    if (rt_render_read_rgb24(job, rgb.data()) != RT_OK) return die("rt_render_read_rgb24");
    double dt = now() - t0;
    rt_stats st; memset(&st, 0, sizeof(st));
    if (rt_render_stats(job, &st) != RT_OK) return die("rt_render_stats");

    uint64_t fnv = 1469598103934665603ull;                           // FNV-1a of the picture: two hosts can compare without a file
    for (uint8_t b : rgb) { fnv ^= b; fnv *= 1099511628211ull; }
    if (ppm) {
        FILE* f = fopen(ppm, "wb");
        if (!f) { fprintf(stderr, "rtbench: cannot write %s\n", ppm); return 1; }
        fprintf(f, "P6\n%u %u\n255\n", W, H); fwrite(rgb.data(), 1, rgb.size(), f); fclose(f);
    }
    printf("{\"scene\": \"%s\", \"objects\": %u, \"bvh_nodes\": %u, \"bvh_depth\": %u, \"variant\": \"%s\", \"tree_in_lds\": %s, \"object_list\": %s, \"width\": %u, \"height\": %u, "
           "\"passes\": %u, \"seconds\": %.6f, \"msamples_per_s\": %.3f, \"closest_rays\": %" PRIu64 ", \"shadow_rays\": %" PRIu64 ", \"pixels_rendered\": %" PRIu64
           ", \"scene_create_s\": %.4f, \"rgb24_fnv1a\": \"%016" PRIx64 "\"}\n",
           argv[1], info.n_objects, info.n_nodes, info.bvh_depth, info.variant, info.tree_in_lds ? "true" : "false", info.object_list ? "true" : "false", W, H, passes, dt,
           (double)W * H * passes / dt / 1e6, (uint64_t)st.ray_count, (uint64_t)st.shadow_rays, (uint64_t)st.pixels_rendered, t_scene, fnv);
    rt_render_destroy(job);
    rt_scene_destroy(scene);
    return 0;
}
