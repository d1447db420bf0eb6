import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs
from raytracer_amd.flatten import flatten_scene
from oracle import binding as ob
import helpers
sc = bs.ManySpheresManyLights(); flat = flatten_scene(sc)
rs = np.random.default_rng(55)
o1, d1 = helpers.camera_rays(sc, 64, 64, 100_000, rs); o2, d2 = helpers.secondary_rays(flat, 100_000, rs)
o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
i = 53376
oo, dd = o[i:i + 1], d[i:i + 1]
print("ray", oo, dd)
orc = ob.OracleScene(flat)
r = orc.trace_closest(oo, dd, 0.001, np.inf, seed=3); print("oracle", r["object"], r["k"])
bulk = sc.BulkObjects[0]
for wide in ("1", "0"):
    for b in ("host", "ploc", "sah", "device"):
        os.environ["RT_HIP_WIDE"] = wide; os.environ["RT_HIP_BVH"] = b
        dev = rt.DeviceScene(flat); h = dev.TraceClosest(oo, dd, 0.001, np.inf, seed=3); print("wide", wide, b, h["object"], h["k"]); dev.Dispose()
