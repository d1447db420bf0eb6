"""one-off: lane use of the step / leaf rounds of closest_wide (RT_WIDE_PROF builds) on config 4's primary rays and bounce rays (non-persistent kernels)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs
from raytracer_amd.core import RenderOptions
sc = bs.TriangleMesh()
for depth, pers in ((0, "0"), (8, "0"), (8, "1")):
    os.environ["RT_HIP_PERSISTENT"] = pers; dev = rt.DeviceScene(sc)
    opt = RenderOptions(RenderWidth=3840, RenderHeight=2160, Passes=4, MaxBounceDepth=depth, LightSampleCountHint=2, PassesPerBatch=4)
    job = rt.RenderJob(sc, opt, device_scene=dev); job.RenderPasses(0, 4, 1); s = job.UpdateStats()
    print(os.environ.get("RT_HIP_LIB", "")[-12:], "depth", depth, "persistent", pers, "rays", s.RayCount, "slots/ray", s.AabbMisses / s.RayCount, "used/ray", s.HittableIntersections / s.RayCount, "use", s.HittableIntersections / max(s.AabbMisses, 1),
          "box/ray", s.NodeVisits / s.RayCount, "prims/ray", s.PrimTests / s.RayCount, flush=True)
    job.Dispose(); dev.Dispose()
