"""Would ordering a tile's bounce rays by direction class pay?  Depth-1-like rays of Night Spheres (origins = primary hits in image order,
directions = random hemisphere), traced through rt_trace_closest in queue order and re-ordered inside 1024-ray tiles by class."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs
sc = bs.Testing(); dev = rt.DeviceScene(sc)
W, H = 1920, 1080
cam = sc.Camera
ys, xs = np.mgrid[0:H, 0:W]
u = ((xs + 0.5) / W).astype(np.float32).ravel(); v = ((ys + 0.5) / H).astype(np.float32).ravel()
llc, hor, ver, lf = (np.asarray(getattr(cam, n), np.float32) for n in ("LowerLeftCorner", "Horizontal", "Vertical", "LookFrom"))
d = llc[None] + u[:, None] * hor[None] + v[:, None] * ver[None] - lf[None]; d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.repeat(lf[None], len(d), 0)
h = dev.TraceClosest(o, d.astype(np.float32), 0.001, np.inf, seed=1)
hit = h["object"] != 0xFFFFFFFF
wp, n = h["world_point"][hit], h["normal"][hit]
rs = np.random.default_rng(3)
r = rs.uniform(-1, 1, (len(wp), 3)).astype(np.float32); r /= np.linalg.norm(r, axis=1, keepdims=True)
r[(r * n).sum(1) < 0] *= -1
o2, d2 = np.ascontiguousarray(wp), np.ascontiguousarray(r)
print(len(o2), "bounce rays")
def run(o, d, label):
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); dev.TraceClosest(o, d, 0.001, np.inf, seed=1); best = min(best, time.perf_counter() - t0)
    print(f"{label}: {best*1e3:.1f} ms (copies included)")
run(o2, d2, "queue order")
for ncls, key in ((2, lambda d: (np.abs(d[:, 1]) < 0.5)), (4, lambda d: np.digitize(d[:, 1], [-0.5, 0.0, 0.5])), (16, lambda d: np.digitize(d[:, 1], np.linspace(-1, 1, 17)[1:-1]))):
    T = 1024
    idx = np.arange(len(o2)); cls = key(d2).astype(np.int64)
    order = np.lexsort((idx, cls, idx // T))
    run(np.ascontiguousarray(o2[order]), np.ascontiguousarray(d2[order]), f"{ncls} classes of dir.y inside {T}-ray tiles")
order = np.argsort(d2[:, 1], kind="stable")
run(np.ascontiguousarray(o2[order]), np.ascontiguousarray(d2[order]), "globally sorted by dir.y")
run(o2[:1], d2[:1], "one ray (fixed cost)")
