"""A/B of the any-hit knobs on the 100 k-sphere scene (GPU box): second-look threshold x persistent lanes x exact occlusion."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs
from raytracer_amd.core import RenderOptions
sc = bs.ManySpheresManyLights()
W, H = (int(a) for a in (sys.argv[1:3] or (1920, 1080)))
for again, exact in (("0.1", False), ("0.25", False), ("1.0", False), ("0.1", True)):
    for pers in ("0", "1"):
        os.environ["RT_HIP_ANY_AGAIN"] = again; os.environ["RT_HIP_PERSISTENT_SHADOW"] = pers
        dev = rt.DeviceScene(sc)
        opt = RenderOptions(RenderWidth=W, RenderHeight=H, Passes=1, MaxBounceDepth=10, LightSampleCountHint=1, PassesPerBatch=1, ExactOcclusion=exact)
        job = rt.RenderJob(sc, opt, device_scene=dev)
        t0 = time.perf_counter(); job.RenderPasses(0, 1, 1); dt = time.perf_counter() - t0
        s = job.UpdateStats()
        print(f"again {again} exact {exact} persistent {pers}: {dt*1e3:.1f} ms  shadow rays {s.ShadowRays}  boxes/ray {s.ShadowNodeVisits/max(s.ShadowRays,1):.1f}  prims/ray {s.ShadowPrimTests/max(s.ShadowRays,1):.2f}", flush=True)
        job.Dispose(); dev.Dispose()
