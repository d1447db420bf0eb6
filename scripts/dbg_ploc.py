"""Device builders against the host SAH tree on configs 4 and 5 (GPU box): build time, tree depth, box tests per ray, Msamples/s."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs
from raytracer_amd.core import RenderOptions
from raytracer_amd.flatten import flatten_scene
CASES = {"c4": (bs.TriangleMesh, 3840, 2160, 16, 8, 2), "c5": (bs.ManySpheresManyLights, 3840, 2160, 2, 10, 1), "c2": (bs.Testing, 1920, 1080, 64, 10, 2)}
for c in (sys.argv[1:] or ["c5", "c4"]):
    mk, W, H, ppb, depth, ls = CASES[c]
    sc = mk()
    t0 = time.perf_counter(); flat = flatten_scene(sc); print(f"{c}: flattening (host, Python) {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
    for builder in (os.environ.get("BUILDERS", "host sah ploc device").split()):
        os.environ["RT_HIP_BVH"] = builder
        os.environ["RT_HIP_NO_LDS_TREE"] = "1" if c == "c2" else "0"
        t0 = time.perf_counter(); dev = rt.DeviceScene(flat); t_scene = time.perf_counter() - t0
        opt = RenderOptions(RenderWidth=W, RenderHeight=H, Passes=2 * ppb, MaxBounceDepth=depth, LightSampleCountHint=ls, PassesPerBatch=ppb)
        job = rt.RenderJob(sc, opt, device_scene=dev)
        job.RenderPasses(10_000, ppb, 1)
        s0 = job.UpdateStats(); a = (s0.RayCount, s0.NodeVisits, s0.ShadowRays, s0.ShadowNodeVisits)
        t0 = time.perf_counter(); job.RenderPasses(0, 2 * ppb, 1); job.UpdateImage(); dt = time.perf_counter() - t0
        s1 = job.UpdateStats(); b = (s1.RayCount, s1.NodeVisits, s1.ShadowRays, s1.ShadowNodeVisits)
        info = dev.Info()
        print(f"{c} {builder:6s}: scene create {t_scene*1e3:8.1f} ms  depth {info['bvh_depth']:3d}  {W*H*2*ppb/dt/1e6:8.1f} Msamples/s  "
              f"boxes/closest ray {(b[1]-a[1])/max(b[0]-a[0],1):6.1f}  boxes/shadow ray {(b[3]-a[3])/max(b[2]-a[2],1):6.1f}", flush=True)
        job.Dispose(); dev.Dispose()
