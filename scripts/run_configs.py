#!/usr/bin/env python3
"""Throughput and roofline of every BASELINE.json config on one GPU (full scene size and resolution, bounded spp): one JSON line per
config that a reader can recompute.

    python scripts/run_configs.py [c1 c2 c3 c4 c5] [--short] [--traffic profiles/r02_traffic_configs.json]

Per config: Msamples/s and Mrays/s over `passes` passes (two batch slots in flight), then ONE batch with per-kernel HIP-event timing
(rt_render_set_profiling, kernels one at a time) from which the roofline of both traversal kernels is derived:

    flops = 22 * V + F_prim * T (+ 60 * S for the shadow kernel: light sampling per shadow ray)        SURVEY §8d
    bytes = 32 * V + B_prim * T
    V = box tests, T = primitive tests, S = shadow rays: the device counters of that batch (rt_render_stats deltas)
    F_prim / B_prim = the scene's census-weighted mean over its tree primitives (sphere 28/16, triangle or quad 42/48, axis plane 14/24,
                      box 70/112, capsule 60/32, disk 42/48, cylinder 60/32), plane 12/24 for the unbounded InfinitePlane tests
    frac_valu = flops / time / 157.3 TFLOP/s (FP32 vector peak; the kernels are built -ffp-contract=off, so half of it is their ceiling)
    frac_hbm_model = bytes / time / 8 TB/s — the §8d byte model counts every node and primitive fetch, cached or not (8-wide trees:
                     80 B read per node visit (of a 104-B record) spread over its child slots in use, 32 B per leaf box, see wide_model)
    frac_hbm_pmc   = ((FETCH_SIZE + WRITE_SIZE) * 1024 per launch + half the coalesced state reads) / time / 8 TB/s when --traffic names a
                     PMC file for this config (scripts/pmc_passes.sh with BENCH_CMD="python3 scripts/run_configs.py cN --short", then
                     scripts/pmc_traffic.py, which explains the calibration: gathers of 64-B node records are counted in full, coalesced
                     16-B-per-lane streams at half — so 16 B per bounce ray / 32 B per shadow request are added back here).  The counters
                     sit on the L2's memory side: Infinity Cache hits are included, it is an upper bound on HBM bytes.
`roofline.frac` is frac_valu for scenes whose tree fits the caches (c1, c2, c3, c5: <= 6.4 MB of nodes); for c4 (88 MB of wide nodes + 480 MB
of leaf records) it is the larger of frac_valu and frac_hbm_pmc (falling back to the model): whichever limit the kernel is nearer to."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs, capi
from raytracer_amd.core import RenderOptions

VALU_PEAK_TFLOPS, HBM_PEAK_GBS = 157.3, 8000.0
# (label, scene, W, H, passes, depth, light samples, passes per batch)
CONFIGS = {
    "c1": ("Demo 480x270 depth 8", bs.Demo, 480, 270, 64, 8, 2, 64),
    "c2": ("Night Spheres 1920x1080 depth 10", bs.Testing, 1920, 1080, 128, 10, 2, 64),
    "c3": ("Cornell (I=1.0, direct) 1920x1080 depth 32", lambda: bs.CornellBox(1.0, True), 1920, 1080, 64, 32, 2, 32),
    "c4": ("Triangle mesh 4 999 122 tris 3840x2160 depth 8", bs.TriangleMesh, 3840, 2160, 32, 8, 2, 16),
    "c5": ("100 k spheres + 64 lights 3840x2160 depth 10", bs.ManySpheresManyLights, 3840, 2160, 4, 10, 1, 2),
}
F_PRIM = {capi.HIT_SPHERE: (28, 16), capi.HIT_QUAD: (42, 48), capi.HIT_TRIANGLE: (42, 48), capi.HIT_INFINITE_PLANE: (12, 24), capi.HIT_XY_PLANE: (14, 24),
          capi.HIT_XZ_PLANE: (14, 24), capi.HIT_YZ_PLANE: (14, 24), capi.HIT_DISK: (42, 48), capi.HIT_BOX: (70, 112), capi.HIT_CAPSULE: (60, 32),
          capi.HIT_CYLINDER: (60, 32), capi.HIT_MEDIUM: (56, 32)}
F_NODE, B_NODE = 22, 32
STAT_KEYS = ("RayCount", "ShadowRays", "NodeVisits", "PrimTests", "ShadowNodeVisits", "ShadowPrimTests", "PixelsRendered", "AabbMisses", "HittableIntersections")


def census(flat):
    """Object types of the scene: (count of unbounded InfinitePlanes, mean flops / bytes of a primitive test inside the tree or list)."""
    import ctypes as C
    n = flat.n_objects
    raw = np.frombuffer((C.c_char * (n * 36)).from_address(C.addressof(flat.desc.objects.contents)), dtype=np.uint32).reshape(n, 9) if n else np.zeros((0, 9), np.uint32)
    types = raw[:, 0]
    unbounded = int((types == capi.HIT_INFINITE_PLANE).sum())
    inside = types[types != capi.HIT_INFINITE_PLANE]
    if len(inside) == 0:
        return unbounded, 0.0, 0.0
    f = np.array([F_PRIM[int(t)][0] for t in np.unique(inside)]); b = np.array([F_PRIM[int(t)][1] for t in np.unique(inside)])
    w = np.array([(inside == t).sum() for t in np.unique(inside)], dtype=np.float64); w /= w.sum()
    return unbounded, float((f * w).sum()), float((b * w).sum())


def snap(job):
    s = job.UpdateStats()
    return {k: getattr(s, k) for k in STAT_KEYS}


def wide_model(info, b_prim):
    """Bytes per box test in the 8-wide tree (rt_wide.h): a node visit reads 80 B of one 104-B record (header, and per axis the 16 B its
    direction picks) and tests every child slot in use, so an inner box test costs 80 B x nodes / child slots in use (the tree's mean; a binary tree of n inner nodes has n + 1 leaves, and every wide node
    but the root is somebody's child); a leaf's exact box is the 32 B in front of its primitive record."""
    if not info.get("wide_nodes"):
        return None
    n_wide, n_leaves = info["wide_nodes"], info["n_nodes"] + 1
    return {"bytes_per_inner_box_test": 80.0 * n_wide / max(n_wide - 1 + n_leaves, 1), "bytes_per_leaf_box_test": 32.0}


def roofline(which, d, ms, launches, n_unbounded, f_prim, b_prim, wide=None):
    rays, V, T = (d["RayCount"], d["NodeVisits"], d["PrimTests"]) if which == "closest" else (d["ShadowRays"], d["ShadowNodeVisits"], d["ShadowPrimTests"])
    t_plane = min(rays * n_unbounded, T)                     # every ray asks every unbounded plane first (any-hit rays stop at the first hit)
    t_tree = T - t_plane
    flops = F_NODE * V + f_prim * t_tree + F_PRIM[capi.HIT_INFINITE_PLANE][0] * t_plane + (60 * rays if which == "shadow" else 0)
    if wide:     # closest hit tests a leaf's own box (counted in V; a failed one is an AabbMiss, a passed one a primitive test); any-hit has no leaf box
        v_leaf = min(V, t_tree + d["AabbMisses"]) if which == "closest" else 0
        byts = wide["bytes_per_inner_box_test"] * (V - v_leaf) + wide["bytes_per_leaf_box_test"] * v_leaf + b_prim * t_tree + F_PRIM[capi.HIT_INFINITE_PLANE][1] * t_plane
    else:
        byts = B_NODE * V + b_prim * t_tree + F_PRIM[capi.HIT_INFINITE_PLANE][1] * t_plane
    sec = ms / 1e3
    return {"kernel": "k_" + which, "rays": rays, "box_tests_V": V, "prim_tests_T": T, "launches": launches, "total_ms": ms, "avg_launch_ms": ms / max(launches, 1),
            "algorithmic_flops": flops, "algorithmic_bytes": byts, "tflops": flops / sec / 1e12 if sec else 0.0, "frac_valu": flops / sec / 1e12 / VALU_PEAK_TFLOPS if sec else 0.0,
            "model_gbs": byts / sec / 1e9 if sec else 0.0, "frac_hbm_model": byts / sec / 1e9 / HBM_PEAK_GBS if sec else 0.0}


def run_config(c, short=False, traffic=None, exact=False):
    """One config -> the dict that main() prints (bench.py's `configs` leg calls this too)."""
    traffic = traffic or {}
    label, mk, W, H, passes, depth, ls, ppb = CONFIGS[c]
    if short:
        passes = ppb
    sc = mk()
    t0 = time.perf_counter(); dev = rt.DeviceScene(sc); t_scene = time.perf_counter() - t0
    n_unb, f_prim, b_prim = census(dev.flat)
    opt = RenderOptions(RenderWidth=W, RenderHeight=H, Passes=passes, MaxBounceDepth=depth, LightSampleCountHint=ls, PassesPerBatch=ppb,
                        ExactLightSampling=exact, ExactOcclusion=exact)
    job = rt.RenderJob(sc, opt, device_scene=dev)
    job.RenderPasses(10_000, ppb, 1)                      # warm-up batch
    s0 = snap(job)
    t0 = time.perf_counter(); job.RenderPasses(0, passes, 1); job.UpdateImage(); dt = time.perf_counter() - t0
    s1 = snap(job)
    reps = 1
    if dt < 0.25:      # a job of milliseconds (config 1: 8.3 M samples) measures the clocks ramping up, not the kernels: the same job again until half a second has been timed
        reps = int(0.5 / max(dt, 1e-4)) + 1
        s0 = snap(job)
        t0 = time.perf_counter()
        for i in range(reps): job.RenderPasses((i + 1) * passes, passes, 1)
        job.UpdateImage(); dt = (time.perf_counter() - t0) / reps
        s1 = snap(job)
    d = {k: (s1[k] - s0[k]) // reps for k in s0}
    rays, sh = d["RayCount"], d["ShadowRays"]
    info = dev.Info()
    out = {"config": c, "workload": label, "passes": passes, "seconds": dt, "jobs_timed": reps, "msamples_per_s": W * H * passes / dt / 1e6,
           "mode": "exact light sampling + exact occlusion" if exact else "default",
           "mrays_per_s": {"closest": rays / dt / 1e6, "shadow": sh / dt / 1e6},
           "box_tests_per_closest_ray": d["NodeVisits"] / max(rays, 1), "prim_tests_per_closest_ray": d["PrimTests"] / max(rays, 1),
           "box_tests_per_shadow_ray": d["ShadowNodeVisits"] / max(sh, 1), "prim_tests_per_shadow_ray": d["ShadowPrimTests"] / max(sh, 1),
           "aabb_misses_per_closest_ray": d["AabbMisses"] / max(rays, 1), "scene_create_s": t_scene, "scene_info": info,
           "prim_cost_model": {"unbounded_planes": n_unb, "flops_per_tree_prim_test": f_prim, "bytes_per_tree_prim_test": b_prim}}
    # one batch with per-kernel event timing, and the counters of exactly that batch
    job.SetProfiling(True); p0 = snap(job); job.RenderPasses(20_000, ppb, 1); prof = job.KernelProfile(); p1 = snap(job); job.SetProfiling(False)
    pd = {k: p1[k] - p0[k] for k in p0}
    out["kernel_ms_one_batch"] = {k: round(v["total_ms"], 3) for k, v in prof.items()}
    wide = wide_model(info, b_prim)
    rl = {w: roofline(w, pd, prof[w]["total_ms"], prof[w]["launches"], n_unb, f_prim, b_prim, wide) for w in ("closest", "shadow")}
    dominant = max(rl, key=lambda w: rl[w]["total_ms"])
    r = dict(rl[dominant])
    tr = traffic.get(c, {}).get("kernels", {})
    tnames = [k for k in tr if k.startswith("k_" + dominant)]
    # a tree far larger than the caches (the node array alone beyond 64 MB, binary or wide): its fetches are what can bound the kernel
    tree_bytes = info["wide_nodes"] * 104 if info["wide_nodes"] else info["n_nodes"] * 64
    hbm_bound = tree_bytes > (64 << 20)
    if tnames:
        # the half of the coalesced path-state reads that FETCH_SIZE leaves out (k_closest: 2 float4 per ray; k_shadow: 3 float4 per request + its 16-B target)
        n_light_rays = max(ls * max(len(sc.Lights), 1), 1)
        left_out = 0.5 * (32 * r["rays"] if dominant == "closest" else 64 * r["rays"] / n_light_rays) / max(r["launches"], 1)
        tot_bytes = sum(tr[k]["hbm_bytes_per_launch"] * tr[k]["launches_profiled"] for k in tnames); tot_launches = sum(tr[k]["launches_profiled"] for k in tnames)
        per_launch = tot_bytes / max(tot_launches, 1) + left_out
        r["traffic"] = per_launch
        r["traffic_note"] = "L2 memory-side bytes per launch (Infinity Cache hits included): FETCH_SIZE + WRITE_SIZE, + the uncounted half of the coalesced state reads"
        r["pmc_gbs"] = per_launch * r["launches"] / (r["total_ms"] / 1e3) / 1e9
        r["frac_hbm_pmc"] = r["pmc_gbs"] / HBM_PEAK_GBS
    else:
        r["traffic"] = None
    frac_hbm = r.get("frac_hbm_pmc", r["frac_hbm_model"])
    if hbm_bound and frac_hbm >= r["frac_valu"]:
        r.update(bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s", achieved=r.get("pmc_gbs", r["model_gbs"]), frac=frac_hbm,
                 frac_uses="PMC FETCH_SIZE/WRITE_SIZE traffic" if tnames else "the byte model (counts cached node fetches too: an upper bound on HBM traffic)")
    else:
        r.update(bound="valu", peak=VALU_PEAK_TFLOPS, unit="TFLOP/s", achieved=r["tflops"], frac=r["frac_valu"], frac_uses="algorithmic flops / FP32 vector peak")
    out["roofline"] = r
    out["roofline_other"] = rl["shadow" if dominant == "closest" else "closest"]
    job.Dispose(); dev.Dispose()
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    short = "--short" in sys.argv
    traffic = {}
    if "--traffic" in sys.argv:
        traffic = json.load(open(sys.argv[sys.argv.index("--traffic") + 1]))
        args = [a for a in args if not a.endswith(".json")]
    for c in (args or list(CONFIGS)):
        print(json.dumps(run_config(c, short, traffic, exact="--exact" in sys.argv)), flush=True)


if __name__ == "__main__":
    main()
