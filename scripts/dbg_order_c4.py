"""Would ordering the bounce rays of the 5 M-triangle mesh by direction class pay?  Depth-1-like rays (origins = primary hits, each pixel's hit
four times over in Z order as the renderer's queue has them, directions = random hemisphere) through rt_trace_closest in queue order and
re-ordered inside windows of T rays by the octant of their direction.  Kernel times come from the rocprofv3 trace of this script (the
k_trace_closest dispatches in order: 3 per line printed); wall times here include the copies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs
sc = bs.TriangleMesh(); dev = rt.DeviceScene(sc)
W, H = 960, 544
cam = sc.Camera
def morton(n):
    i = np.arange(n * n); x = np.zeros_like(i); y = np.zeros_like(i)
    for b in range(5): x |= ((i >> (2 * b)) & 1) << b; y |= ((i >> (2 * b + 1)) & 1) << b
    return x, y
mx, my = morton(32)
tx, ty = np.meshgrid(np.arange(W // 32), np.arange(H // 32))
xs = (tx.ravel()[:, None] * 32 + mx[None]).ravel(); ys = (ty.ravel()[:, None] * 32 + my[None]).ravel()
xs = np.repeat(xs, 4); ys = np.repeat(ys, 4)                        # four samples of a pixel side by side
u = ((xs + 0.5) / W).astype(np.float32); v = ((ys + 0.5) / H).astype(np.float32)
llc, hor, ver, lf = (np.asarray(getattr(cam, n), np.float32) for n in ("LowerLeftCorner", "Horizontal", "Vertical", "LookFrom"))
d = llc[None] + u[:, None] * hor[None] + v[:, None] * ver[None] - lf[None]; d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.repeat(lf[None], len(d), 0)
h = dev.TraceClosest(o, d.astype(np.float32), 0.001, np.inf, seed=1)
hit = h["object"] != 0xFFFFFFFF
wp, n = h["world_point"][hit], h["normal"][hit]
rs = np.random.default_rng(3)
r = rs.normal(size=(len(wp), 3)).astype(np.float32); r /= np.linalg.norm(r, axis=1, keepdims=True)
r[(r * n).sum(1) < 0] *= -1
o2, d2 = np.ascontiguousarray(wp + 1e-3 * n), np.ascontiguousarray(r)
print(len(o2), "bounce rays", flush=True)
def run(o, d, label):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); dev.TraceClosest(o, d, 0.001, np.inf, seed=1); best = min(best, time.perf_counter() - t0)
    print(f"{label}: {best*1e3:.1f} ms (copies included)", flush=True)
run(o2, d2, "queue order")
octant = (d2[:, 0] < 0).astype(np.int64) | ((d2[:, 1] < 0).astype(np.int64) << 1) | ((d2[:, 2] < 0).astype(np.int64) << 2)
idx = np.arange(len(o2))
for T in (256, 1024, 4096, 65536):
    order = np.lexsort((idx, octant, idx // T))
    run(np.ascontiguousarray(o2[order]), np.ascontiguousarray(d2[order]), f"by octant inside {T}-ray windows")
order = np.lexsort((idx, octant))
run(np.ascontiguousarray(o2[order]), np.ascontiguousarray(d2[order]), "globally by octant")
