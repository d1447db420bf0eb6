"""The device's SAH builder (rt_sah.hip, the default from 256 k objects) on object sets chosen to reach its corners: the smallest
trees, node sizes on both sides of its one-wave / many-workgroup border (2048 objects), objects whose box centres coincide (no
axis to bin along: positional halves), centres on one line, and a geometric spread that peels one object per level until the
depth limit turns splits into halves.  Another tree never changes a closest hit or an any-hit answer: the host builder's tree
is the reference (BvhTree.FromSegment_SAH, Acceleration/BvhTree.cs:37-136, only promises the hits)."""
import math

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.core import Camera, Colour, Scene
from raytracer_amd.flatten import flatten_scene
from raytracer_amd.impl import PointLight, SingleColourSkyBox, SphereArray, StandardMaterial
import helpers

NO_HIT = 0xFFFFFFFF

pytestmark = pytest.mark.gpu


def _field(centres, radii):
    centres = np.asarray(centres, np.float32).reshape(-1, 3); radii = np.asarray(radii, np.float32).reshape(-1)
    mats = [StandardMaterial(Colour(0.7), Colour(0.0), 0.5), StandardMaterial(Colour(0.2, 0.4, 0.8), Colour(0.0), 0.1)]
    spheres = SphereArray(centres, radii, mats, np.arange(len(radii)) % 2)
    cam = Camera.Create((0, 2, 30), (0, 0, 0), (0, 1, 0), 40, 16.0 / 9.0, 0.0, 10.0)
    return Scene(f"{len(radii)} spheres", cam, [], [PointLight(Position=(0, 50, 0), Colour=Colour(1.0), AttenuationRadius=100.0)], SingleColourSkyBox(Colour(0.1)),
                 BulkObjects=[spheres])


def _cases():
    rs = np.random.default_rng(20260)
    def cube(n, side=20.0):
        return rs.uniform(-side, side, (n, 3)), rs.uniform(0.05, 0.6, n)
    yield "two", np.array([[-1.0, 0, 0], [1.5, 0, 0]]), np.array([0.5, 0.7])
    yield "three", np.array([[-1.0, 0, 0], [1.5, 0, 0], [0.2, 2, 0]]), np.array([0.5, 0.7, 0.3])
    yield "five_in_a_row", np.stack([np.arange(5.0), np.zeros(5), np.zeros(5)], axis=1), np.full(5, 0.3)
    for n in (2048, 2049, 4096, 4097, 10_000):
        c, r = cube(n)
        yield f"cube_{n}", c, r
    yield "same_centre_3000", np.tile(np.array([[1.0, 2.0, 3.0]]), (3000, 1)), np.full(3000, 0.5)                   # no axis separates the centres
    yield "same_centre_two_sizes", np.tile(np.array([[0.0, 0.0, 0.0]]), (2500, 1)), np.where(np.arange(2500) % 2, 0.5, 1.0)
    # (rays start within 5 units of their targets here: from the far end of the line a 0.1-unit sphere's K is float noise, and which of
    # two noisy hits is the closest then depends on a tree's visiting order — helpers.assert_closest_parity's near-tie rule)
    yield "on_a_line_2500", np.stack([np.zeros(2500), np.zeros(2500), np.linspace(-300, 300, 2500)], axis=1), np.full(2500, 0.1), 5.0
    # 14 centres along each axis in a geometric progression of ratio 17 (> 16 bins: every bin border cuts off only the farthest one),
    # so 42 levels peel one object each — past level 40 splits turn into halves — among 4000 ordinary ones.  Rays start a few
    # radii from their targets (seen from the cluster, an object 1e17 away is float noise).
    far = 17.0 ** np.arange(1, 15)
    pts = np.concatenate([np.stack([far * (a == 0), far * (a == 1), far * (a == 2)], axis=1) for a in range(3)])
    c, r = cube(4000, 1.0)
    yield "geometric_spread", np.concatenate([pts, c]), np.concatenate([0.3 * np.tile(far, 3), r * 0.1]), "relative"
    grid = np.stack(np.meshgrid(np.arange(20.0), np.arange(20.0), np.arange(10.0), indexing="ij"), axis=-1).reshape(-1, 3)      # many equal coordinates per axis
    yield "lattice_4000", grid, np.full(len(grid), 0.25)


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_device_sah_tree_gives_the_host_trees_hits(case, monkeypatch):
    name, centres, radii = case[:3]
    reach = case[3] if len(case) > 3 else None
    sc = _field(centres, radii); flat = flatten_scene(sc)
    n = len(radii)
    monkeypatch.setenv("RT_HIP_BVH", "host"); monkeypatch.setenv("RT_HIP_NO_LDS_TREE", "1"); monkeypatch.setenv("RT_HIP_LIST", "0")
    host = rt.DeviceScene(flat)
    monkeypatch.setenv("RT_HIP_BVH", "sah")
    dev = rt.DeviceScene(flat)
    again = rt.DeviceScene(flat)
    hi, di = host.Info(), dev.Info()
    assert di["built_on_device"] and not hi["built_on_device"]
    assert di["n_nodes"] == hi["n_nodes"] == n - 1
    assert 1 <= di["bvh_depth"] <= 41 + math.ceil(math.log2(n)), di                      # 40 levels of SAH, then halves
    if name == "geometric_spread": assert di["bvh_depth"] > 40, di                       # (the case is there to get past the limit)
    rs = np.random.default_rng(7)
    c32 = np.asarray(centres, np.float32)
    # rays from a shell around the objects towards randomly chosen objects (jittered), and rays between objects
    r32 = np.asarray(radii, np.float32)
    pick = rs.integers(0, n, 30_000)
    tgt = c32[pick] + rs.normal(0, 1, (30_000, 3)).astype(np.float32) * r32[rs.integers(0, n, 30_000)][:, None]
    span = float(np.abs(c32).max()) + 5.0
    u = rs.normal(0, 1, (30_000, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    if reach == "relative":
        tgt = c32[pick] + (rs.normal(0, 0.5, (30_000, 3)) * r32[pick][:, None]).astype(np.float32)
        o = (tgt + u * rs.uniform(1.5, 8.0, (30_000, 1)) * r32[pick][:, None]).astype(np.float32)
        reach = None
    elif reach is not None:
        o = (tgt + u * rs.uniform(0.3, 1.0, (30_000, 1)) * reach).astype(np.float32)
    else:
        o = (rs.normal(0, 1, (30_000, 3)) * span).astype(np.float32)
    d = tgt - o; d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-20); d = d.astype(np.float32)
    a, b = dev.TraceClosest(o, d, 0.001, np.inf, seed=3), host.TraceClosest(o, d, 0.001, np.inf, seed=3)
    helpers.assert_closest_parity(a, b)
    assert (b["object"] != NO_HIT).mean() > 0.2, name                             # the rays do meet the objects
    c = again.TraceClosest(o, d, 0.001, np.inf, seed=3)
    assert np.array_equal(a["object"], c["object"]) and np.array_equal(a["k"], c["k"])    # the same tree on every build: the same tie decisions
    kmin = np.full(len(o), 0.01, np.float32); kmax = rs.uniform(0.5, 2.5 * (reach or span), len(o)).astype(np.float32)
    assert np.array_equal(dev.TraceAny(o, d, kmin, kmax), host.TraceAny(o, d, kmin, kmax))
    dev.Dispose(); host.Dispose(); again.Dispose()
