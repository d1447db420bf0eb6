"""Fixed ray sets with committed answers (tests/golden/ray_hits.npz, made by tests/golden/make_ray_hit_fixtures.py).
CPU: the oracle still gives exactly these answers.  GPU: the HIP library gives them through the C ABI."""
import os

import numpy as np
import pytest

import helpers
from raytracer_amd.flatten import flatten_scene

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ray_hits.npz"))
SCENES = sorted({k.rsplit("_", 1)[0] for k in GOLD.files if k.endswith("_o")})
NO_HIT = 0xFFFFFFFF


def _check(name, hits, occ):
    g = GOLD
    assert np.array_equal(hits["object"], g[name + "_object"])
    assert np.array_equal(hits["material"], g[name + "_material"])
    hit = g[name + "_object"] != NO_HIT
    assert hit.sum() > 500
    gk, hk = g[name + "_k"][hit], hits["k"][hit]
    assert (np.abs(hk - gk) <= 1e-5 * np.abs(gk)).all()                           # the bar; observed bit-equal except where logf/acosf feed K (media)
    assert (hk == gk).mean() > 0.95
    assert np.array_equal(hits["outside_face"][hit].astype(np.uint8), g[name + "_outside"][hit])
    assert np.allclose(hits["normal"][hit], g[name + "_normal"][hit], rtol=0, atol=2e-6)
    assert np.array_equal(occ.astype(np.uint8), g[name + "_any"])


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_the_golden_ray_hits(name):
    from oracle import binding as ob
    flat = flatten_scene(helpers.small_scenes()[name]())
    orc = ob.OracleScene(flat)
    o, d = GOLD[name + "_o"], GOLD[name + "_d"]
    _check(name, orc.trace_closest(o, d, 0.001, np.inf, seed=5), orc.trace_any(o, d, np.full(len(o), 0.01, np.float32), GOLD[name + "_any_kmax"]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_hip_library_reproduces_the_golden_ray_hits(name):
    import raytracer_amd as rt
    dev = rt.DeviceScene(flatten_scene(helpers.small_scenes()[name]()))
    o, d = GOLD[name + "_o"], GOLD[name + "_d"]
    _check(name, dev.TraceClosest(o, d, 0.001, np.inf, seed=5), dev.TraceAny(o, d, np.full(len(o), 0.01, np.float32), GOLD[name + "_any_kmax"]))
