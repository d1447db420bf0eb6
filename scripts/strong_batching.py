#!/usr/bin/env python3
"""What a short job loses to the two-slot steady state, and which batch size a short call should use.

    python scripts/strong_batching.py            # config 2 (Night Spheres 1080p), jobs of 256, 128, 64, 32 and 16 passes on ONE GPU

A job of P passes is what one of N GPUs sees of a 256-pass job (P = 256 / N).  For every P and every batch size that divides it:
Msamples/s of the job end to end (passes + RGB24 read), best of 3, the sums starting at zero each time (a new handle)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from raytracer_amd.core import RenderOptions

W, H = 1920, 1080
sc = rt.BuiltinScenes.Testing()
dev = rt.DeviceScene(sc)
for P in (256, 128, 64, 32, 16):
    for ppb in (0, 64, 32, 16, 8, 4):
        if ppb > P:
            continue
        best = 1e9
        for rep in range(4):
            opt = RenderOptions(RenderWidth=W, RenderHeight=H, Passes=P, MaxBounceDepth=10, LightSampleCountHint=2, PassesPerBatch=ppb)
            job = rt.RenderJob(sc, opt, device_scene=dev)
            t0 = time.perf_counter(); job.RenderPasses(rep * 1000, P, 1); job.UpdateImage(); dt = time.perf_counter() - t0
            job.Dispose()
            if rep:
                best = min(best, dt)
        print(json.dumps({"passes": P, "passes_per_batch": ppb or "library", "ms": round(best * 1e3, 2), "msamples_per_s": round(W * H * P / best / 1e6, 1)}), flush=True)
