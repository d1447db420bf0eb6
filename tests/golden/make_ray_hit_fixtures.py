"""Golden ray -> hit vectors for the fixed-ray-set parity check (BASELINE north_star check 1).

The reference cannot run here (C#/.NET) and holds no ray-level fixtures, so these vectors come from oracle/ — the CPU
restatement pinned by the reference's sky golden and comment KATs (tests/test_oracle_kat.py).  They freeze its answers: a later
change to the oracle or to the HIP kernels that moves a hit id shows up against this file on both sides.

    python tests/golden/make_ray_hit_fixtures.py        # rewrites tests/golden/ray_hits.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = ["Demo", "CornellBox", "Testing", "Zoo", "MiniMesh"]
N = 2000


def main():
    import helpers
    from raytracer_amd.flatten import flatten_scene
    from oracle import binding as ob
    out = {}
    for si, name in enumerate(SCENES):
        sc = helpers.small_scenes()[name]()
        flat = flatten_scene(sc)
        orc = ob.OracleScene(flat)
        rs = np.random.default_rng(1000 + si)
        o1, d1 = helpers.camera_rays(sc, 64, 64, N, rs)
        o2, d2 = helpers.secondary_rays(flat, N, rs)
        o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
        h = orc.trace_closest(o, d, 0.001, np.inf, seed=5)
        kmax = rs.uniform(0.1, 30.0, len(o)).astype(np.float32)
        occ = orc.trace_any(o, d, np.full(len(o), 0.01, np.float32), kmax)
        out[name + "_o"], out[name + "_d"] = o, d
        out[name + "_k"] = np.where(h["object"] != 0xFFFFFFFF, h["k"], 0).astype(np.float32)
        out[name + "_object"], out[name + "_material"], out[name + "_outside"] = h["object"], h["material"], h["outside_face"].astype(np.uint8)
        out[name + "_normal"] = np.where((h["object"] != 0xFFFFFFFF)[:, None], h["normal"], 0).astype(np.float32)
        out[name + "_any_kmax"], out[name + "_any"] = kmax, occ.astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "ray_hits.npz"), **out)
    print("wrote", os.path.join(HERE, "ray_hits.npz"))


if __name__ == "__main__":
    main()
