#!/usr/bin/env python3
"""Image parity device vs oracle per scene, in both light-sampling modes: fraction of pixels within 1e-3, the worst outliers.
usage: python scripts/parity_report.py [scene ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as rt
from raytracer_amd.core import RenderOptions
from raytracer_amd.flatten import flatten_scene
from oracle import binding as ob
import helpers

CASES = {"Demo": (96, 54, 8, 8, 2), "Testing": (96, 54, 8, 10, 2), "CornellBox": (64, 64, 8, 32, 2), "Zoo": (64, 36, 8, 6, 1), "MiniMesh": (64, 36, 4, 8, 2),
         "MiniManyLights": (64, 36, 4, 10, 1), "Empty": (16, 9, 2, 3, 1), "RgbSpheres": (64, 36, 4, 10, 2), "Noise": (96, 54, 6, 8, 2)}

def main():
    for name in (sys.argv[1:] or CASES):
        w, h, passes, depth, ls = CASES[name]
        sc = helpers.small_scenes()[name](); flat = flatten_scene(sc); dev, orc = rt.DeviceScene(flat), ob.OracleScene(flat)
        ref = None
        for exact in (False, True):
            opt = RenderOptions(RenderWidth=w, RenderHeight=h, Passes=passes, MaxBounceDepth=depth, LightSampleCountHint=ls, Seed=1234, ExactLightSampling=exact, ExactOcclusion=exact)
            job = rt.RenderJob(sc, opt, device_scene=dev); job.Wait(); got, _ = job.ReadAccum(); st = job.RenderStats; job.Dispose()
            if ref is None:
                ref, rst = orc.render(opt, threads=8)
            fin = np.isfinite(ref).all(axis=2) & np.isfinite(got).all(axis=2)
            err = np.abs(got - ref); scale = np.maximum(np.abs(ref), 1e-3)
            rel = (err / scale).max(axis=2); rel[~fin] = 0
            close = ((err <= 1e-3 * scale + 1e-5).all(axis=2) | ~fin)
            worst = np.argsort(rel.ravel())[::-1][:5]
            print(f"{name:16s} exact={int(exact)} close={close.mean():.5f} nonfinite={int((~fin).sum())} max_abs_err={err[fin].max():.4g} "
                  f"worst(rel,abs)={[(round(float(rel.ravel()[i]),4), round(float(err.reshape(-1,3)[i].max()),4)) for i in worst]} "
                  f"rays {st.RayCount}/{rst['ray_count']} shadow {st.ShadowRays}/{rst['shadow_rays']}", flush=True)
        dev.Dispose()

if __name__ == "__main__":
    main()
