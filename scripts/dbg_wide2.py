import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as rt
from raytracer_amd.flatten import flatten_scene
from oracle import binding as ob
from test_gpu_builders import _cases, _field
case = [c for c in _cases() if c[0] == "geometric_spread"][0]
name, centres, radii = case[:3]
flat = flatten_scene(_field(centres, radii)); n = len(radii)
os.environ.update(RT_HIP_NO_LDS_TREE="1", RT_HIP_LIST="0", RT_HIP_BVH="host")
os.environ["RT_HIP_WIDE"] = "1"; wide = rt.DeviceScene(flat)
os.environ["RT_HIP_WIDE"] = "0"; binary = rt.DeviceScene(flat)
orc = ob.OracleScene(flat)
rs = np.random.default_rng(11)
c32, r32 = np.asarray(centres, np.float32).reshape(-1, 3), np.asarray(radii, np.float32).reshape(-1)
m = 30_000
pick = rs.integers(0, n, m)
u = rs.normal(0, 1, (m, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
span = float(np.abs(c32).max()) + 5.0
tgt = c32[pick] + (rs.normal(0, 0.5, (m, 3)) * r32[pick][:, None]).astype(np.float32)
o = (tgt + u * rs.uniform(1.5, 8.0, (m, 1)) * r32[pick][:, None]).astype(np.float32)
d = tgt - o; d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-20); d = d.astype(np.float32)
off = r32[pick[:600], None] * np.array([0.13, 0.07, 0.0], np.float32)
o[:300] = c32[pick[:300]] + off[:300] + r32[pick[:300], None] * np.array([0.0, 0.0, 6.0], np.float32); d[:300] = np.array([0.0, 0.0, -1.0], np.float32)
o[300:600] = c32[pick[300:600]] + off[300:, ::-1] - r32[pick[300:600], None] * np.array([4.0, 0.0, 0.0], np.float32); d[300:600] = np.array([1.0, 0.0, 0.0], np.float32)
kmin = np.full(m, 0.01, np.float32); kmax = rs.uniform(0.5, 2.5 * span, m).astype(np.float32)
a, b = wide.TraceAny(o, d, kmin, kmax), binary.TraceAny(o, d, kmin, kmax)
t = orc.trace_any(o, d, kmin, kmax).astype(bool)
bad = np.nonzero(a != b)[0]
print("differ", len(bad), bad[:10], "wide!=oracle", (a.astype(bool) != t).sum(), "binary!=oracle", (b.astype(bool) != t).sum())
for i in bad[:6]:
    print(i, "o", o[i], "d", d[i], "kmax", kmax[i], "wide", a[i], "bin", b[i], "oracle", t[i], "target r", r32[pick[i]])
