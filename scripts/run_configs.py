#!/usr/bin/env python3
"""Throughput of every BASELINE.json config on one GPU (full scene size and resolution, bounded spp): one JSON line per config.
usage: python scripts/run_configs.py [c2 c3 c4 c5]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs
from raytracer_amd.core import RenderOptions

CONFIGS = {
    "c1": ("Demo 480x270 depth 8", bs.Demo, 480, 270, 64, 8, 2, 64),
    "c2": ("Night Spheres 1920x1080 depth 10", bs.Testing, 1920, 1080, 128, 10, 2, 64),
    "c3": ("Cornell (I=1.0, direct) 1920x1080 depth 32", lambda: bs.CornellBox(1.0, True), 1920, 1080, 64, 32, 2, 32),
    "c4": ("Triangle mesh 4 999 122 tris 3840x2160 depth 8", bs.TriangleMesh, 3840, 2160, 32, 8, 2, 16),
    "c5": ("100 k spheres + 64 lights 3840x2160 depth 10", bs.ManySpheresManyLights, 3840, 2160, 4, 10, 1, 2),
}

def main():
    which = sys.argv[1:] or ["c1", "c2", "c3", "c4", "c5"]
    for c in which:
        label, mk, W, H, passes, depth, ls, ppb = CONFIGS[c]
        sc = mk()
        t0 = time.perf_counter(); dev = rt.DeviceScene(sc); t_scene = time.perf_counter() - t0
        opt = RenderOptions(RenderWidth=W, RenderHeight=H, Passes=passes, MaxBounceDepth=depth, LightSampleCountHint=ls, PassesPerBatch=ppb)
        job = rt.RenderJob(sc, opt, device_scene=dev)
        job.RenderPasses(10_000, ppb, 1)                      # warm-up batch
        s0 = job.UpdateStats(); r0, sh0, nv0, pt0, snv0, spt0 = s0.RayCount, s0.ShadowRays, s0.NodeVisits, s0.PrimTests, s0.ShadowNodeVisits, s0.ShadowPrimTests
        t0 = time.perf_counter(); job.RenderPasses(0, passes, 1); job.UpdateImage(); dt = time.perf_counter() - t0
        s1 = job.UpdateStats()
        rays, sh = s1.RayCount - r0, s1.ShadowRays - sh0
        out = {"config": c, "workload": label, "passes": passes, "seconds": dt, "msamples_per_s": W * H * passes / dt / 1e6,
               "mrays_per_s": {"closest": rays / dt / 1e6, "shadow": sh / dt / 1e6},
               "box_tests_per_closest_ray": (s1.NodeVisits - nv0) / max(rays, 1), "prim_tests_per_closest_ray": (s1.PrimTests - pt0) / max(rays, 1),
               "box_tests_per_shadow_ray": (s1.ShadowNodeVisits - snv0) / max(sh, 1), "prim_tests_per_shadow_ray": (s1.ShadowPrimTests - spt0) / max(sh, 1),
               "scene_create_s": t_scene, "scene_info": dev.Info()}
        job.SetProfiling(True); job.RenderPasses(20_000, ppb, 1); prof = job.KernelProfile(); job.SetProfiling(False)
        out["kernel_ms_one_batch"] = {k: round(v["total_ms"], 3) for k, v in prof.items()}
        print(json.dumps(out), flush=True)
        job.Dispose(); dev.Dispose()

if __name__ == "__main__":
    main()
