"""Primitive tests per shadow ray on the Cornell box against the scale of the any-hit box pad (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs
from raytracer_amd.core import RenderOptions
sc = bs.CornellBox(1.0, True)
for scale in ("0", "0.001", "0.01", "0.1", "1"):
    os.environ["RT_HIP_ANY_PAD_SCALE"] = scale
    dev = rt.DeviceScene(sc)
    opt = RenderOptions(RenderWidth=960, RenderHeight=540, Passes=4, MaxBounceDepth=32, LightSampleCountHint=2, PassesPerBatch=4)
    job = rt.RenderJob(sc, opt, device_scene=dev)
    t0 = time.perf_counter(); job.RenderPasses(0, 4, 1); dt = time.perf_counter() - t0
    s = job.UpdateStats()
    print(f"pad scale {scale}: {dt*1e3:.1f} ms  shadow rays {s.ShadowRays}  boxes/ray {s.ShadowNodeVisits/max(s.ShadowRays,1):.2f}  prims/ray {s.ShadowPrimTests/max(s.ShadowRays,1):.3f}", flush=True)
    job.Dispose(); dev.Dispose()
