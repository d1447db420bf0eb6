import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as rt
from raytracer_amd.flatten import flatten_scene
from test_gpu_builders import _field
os.environ["RT_HIP_NO_LDS_TREE"] = "1"; os.environ["RT_HIP_LIST"] = "0"; os.environ["RT_HIP_BVH"] = "host"
n = 3000
flat = flatten_scene(_field(np.tile(np.array([[1.0, 2.0, 3.0]]), (n, 1)), np.full(n, 0.5)))
os.environ["RT_HIP_WIDE"] = "1"; wide = rt.DeviceScene(flat)
os.environ["RT_HIP_WIDE"] = "0"; binary = rt.DeviceScene(flat)
print(wide.Info())
c = np.array([1.0, 2.0, 3.0], np.float32)
o = np.stack([c + np.array([0, 0, 3.0], np.float32), c - np.array([2.0, 0, 0], np.float32), c + np.array([0.1, 0.05, 3.0], np.float32)]).astype(np.float32)
d = np.array([[0, 0, -1.0], [1.0, 0, 0], [0, 0, -1.0]], np.float32)
a, b = wide.TraceClosest(o, d, 0.001, np.inf), binary.TraceClosest(o, d, 0.001, np.inf)
print("wide", a["object"], a["k"]); print("bin ", b["object"], b["k"])
