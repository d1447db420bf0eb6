"""Parity soak (GPU box): many more rays than the test suite asks, several seeds, on every small scene — closest hits and any-hit
answers of the HIP library against the oracle.  Prints one line per scene and seed; exits non-zero on the first mismatch beyond
the near-tie rule.  usage: python scripts/soak_parity.py [rays per set, default 1000000] [seeds, default 3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracer_amd as rt
from raytracer_amd.flatten import flatten_scene
from oracle import binding as ob
import helpers
import test_gpu_parity as tp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bad = 0
for name, mk in helpers.small_scenes().items():
    sc = mk(); flat = flatten_scene(sc); dev, orc = rt.DeviceScene(flat), ob.OracleScene(flat)
    for seed in range(seeds):
        t0 = time.perf_counter()
        o, d = tp.ray_set(sc, flat, n // 2, 1000 + seed)
        g, r = dev.TraceClosest(o, d, 0.001, np.inf, seed=seed), orc.trace_closest(o, d, 0.001, np.inf, seed=seed)
        try:
            excl = helpers.assert_closest_parity(g, r)
            msg = f"closest ok ({len(excl)} near ties)"
        except AssertionError as e:
            msg = f"CLOSEST MISMATCH: {e}"; bad += 1
        rs = np.random.default_rng(seed)
        kmin = np.full(len(o), 0.01, np.float32)
        kmax = np.where(rs.random(len(o)) < 0.5, rs.uniform(0.05, 30.0, len(o)), rs.uniform(30.0, 3000.0, len(o))).astype(np.float32)
        ga, ra = dev.TraceAny(o, d, kmin, kmax), orc.trace_any(o, d, kmin, kmax)
        diff = int((ga != ra).sum())
        if diff: bad += 1
        print(f"{name:16s} seed {seed}: {len(o)} rays, {msg}; any-hit {'ok' if not diff else f'{diff} DIFFER'} ({(ra != 0).mean():.2f} occluded)  {time.perf_counter() - t0:.1f} s", flush=True)
    dev.Dispose()
sys.exit(1 if bad else 0)
