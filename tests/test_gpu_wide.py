"""The 8-wide tree with quantised inner boxes (rt_wide.h / rt_wide.hip; trees that are read through the caches) against the binary
tree it is made from.  Inner boxes only have to be conservative and a leaf's own box is still tested in the reference's arithmetic
(SingleObjectBvhNode.cs:10-16), so another tree shape changes no closest hit and no any-hit answer: same ids, same K bits, same
booleans, same pictures — on the builders' corner-case object sets, on small scenes of every leaf kind (spheres and triangles,
whose records are copied into the leaf table, and mixed types, which are not), and through every kernel that walks the tree."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.core import RenderOptions
from raytracer_amd.flatten import flatten_scene
from oracle import binding as ob
import helpers
from test_gpu_builders import _cases, _field

NO_HIT = 0xFFFFFFFF

pytestmark = pytest.mark.gpu


def _pair(flat, monkeypatch, **env):
    """(wide, binary) device scenes of the same flattened scene, trees outside LDS and never a list"""
    monkeypatch.setenv("RT_HIP_NO_LDS_TREE", "1"); monkeypatch.setenv("RT_HIP_LIST", "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("RT_HIP_WIDE", "1"); wide = rt.DeviceScene(flat)
    monkeypatch.setenv("RT_HIP_WIDE", "0"); binary = rt.DeviceScene(flat)
    assert wide.Info()["wide_nodes"] >= 1 and binary.Info()["wide_nodes"] == 0
    return wide, binary


def _same_hits(a, b, flat, rays):
    both = helpers.assert_closest_parity(a, b, flat=flat, rays=rays)
    same = np.ones(len(a), bool); same[both] = False
    assert np.array_equal(a["k"][same], b["k"][same]) and np.array_equal(a["normal"][same], b["normal"][same])      # the same intersector on the same object: the same bits
    return both


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_wide_tree_gives_the_binary_trees_hits_on_the_builders_corner_cases(case, monkeypatch):
    name, centres, radii = case[:3]
    reach = case[3] if len(case) > 3 else None
    flat = flatten_scene(_field(centres, radii))
    n = len(radii)
    wide, binary = _pair(flat, monkeypatch, RT_HIP_BVH="host")
    wi = wide.Info()
    assert 1 <= wi["wide_nodes"] <= max(1, n - 1) and 1 <= wi["wide_levels"] <= wi["bvh_depth"] + 1, wi
    rs = np.random.default_rng(11)
    c32, r32 = np.asarray(centres, np.float32).reshape(-1, 3), np.asarray(radii, np.float32).reshape(-1)
    m = 30_000
    pick = rs.integers(0, n, m)
    u = rs.normal(0, 1, (m, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    span = float(np.abs(c32).max()) + 5.0
    if reach == "relative":
        tgt = c32[pick] + (rs.normal(0, 0.5, (m, 3)) * r32[pick][:, None]).astype(np.float32)
        o = (tgt + u * rs.uniform(1.5, 8.0, (m, 1)) * r32[pick][:, None]).astype(np.float32); reach = None
    else:
        tgt = c32[pick] + rs.normal(0, 1, (m, 3)).astype(np.float32) * r32[rs.integers(0, n, m)][:, None]
        o = (tgt + u * rs.uniform(0.3, 1.0, (m, 1)) * reach).astype(np.float32) if reach is not None else (rs.normal(0, 1, (m, 3)) * span).astype(np.float32)
    d = tgt - o; d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-20); d = d.astype(np.float32)
    # axis-parallel rays too: a zero direction component is where 1/d is infinite.  (Off the centre line: a ray through a sphere's
    # centre along an axis meets the sphere exactly on its box's face, K == the box's entry distance, and AxisAlignedBoundingBox.Hit's
    # `kMax <= kMin` then culls every later object of equal K: the first one visited wins, a near-tie by visiting order.)
    off = r32[pick[:600], None] * np.array([0.13, 0.07, 0.0], np.float32)
    o[:300] = c32[pick[:300]] + off[:300] + r32[pick[:300], None] * np.array([0.0, 0.0, 6.0], np.float32); d[:300] = np.array([0.0, 0.0, -1.0], np.float32)
    o[300:600] = c32[pick[300:600]] + off[300:, ::-1] - r32[pick[300:600], None] * np.array([4.0, 0.0, 0.0], np.float32); d[300:600] = np.array([1.0, 0.0, 0.0], np.float32)
    a, b = wide.TraceClosest(o, d, 0.001, np.inf, seed=3), binary.TraceClosest(o, d, 0.001, np.inf, seed=3)
    _same_hits(a, b, flat, (o, d))
    assert (b["object"] != NO_HIT).mean() > 0.2, name
    kmin = np.full(m, 0.01, np.float32); kmax = rs.uniform(0.5, 2.5 * (reach or span), m).astype(np.float32)
    assert np.array_equal(wide.TraceAny(o, d, kmin, kmax), binary.TraceAny(o, d, kmin, kmax))
    wide.Dispose(); binary.Dispose()


SCENES = ["MiniMesh", "MiniManyLights", "RtInAWeekendCover1", "Testing", "Zoo", "Demo", "CornellBox"]


@pytest.mark.parametrize("name", SCENES)
def test_wide_tree_on_small_scenes_rays_and_pictures(name, monkeypatch):
    sc = helpers.small_scenes()[name](); flat = flatten_scene(sc)
    wide, binary = _pair(flat, monkeypatch)
    orc = ob.OracleScene(flat)
    rs = np.random.default_rng(5)
    o1, d1 = helpers.camera_rays(sc, 64, 64, 40_000, rs); o2, d2 = helpers.secondary_rays(flat, 60_000, rs)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    a, b = wide.TraceClosest(o, d, 0.001, np.inf, seed=1), binary.TraceClosest(o, d, 0.001, np.inf, seed=1)
    _same_hits(a, b, flat, (o, d))
    helpers.assert_closest_parity(a, orc.trace_closest(o, d, 0.001, np.inf, seed=1), flat=flat, rays=(o, d))
    so, sd, kmin, kmax = helpers.shadow_ray_set(sc, flat, orc, 60_000, 9)
    wa = wide.TraceAny(so, sd, kmin, kmax)
    assert np.array_equal(wa, binary.TraceAny(so, sd, kmin, kmax))
    assert np.array_equal(wa.astype(bool), orc.trace_any(so, sd, kmin, kmax).astype(bool))          # = the reference's every-object scan
    # pictures: every mode of the light kernels, same seeds — the same rays find the same hits, so the sums are the same bits
    for kw in (dict(), dict(ExactLightSampling=True, ExactOcclusion=True)):
        opt = RenderOptions(RenderWidth=96, RenderHeight=54, Passes=3, MaxBounceDepth=6, Seed=4, **kw)
        imgs, stats = [], []
        for dev in (wide, binary):
            job = rt.RenderJob(sc, opt, device_scene=dev); job.Wait()
            imgs.append(job.ReadAccum()[0]); st = job.RenderStats; stats.append((st.RayCount, st.ShadowRays, st.PixelsRendered)); job.Dispose()
        same = (imgs[0] == imgs[1]) | (np.isnan(imgs[0]) & np.isnan(imgs[1]))
        # exact modes: the same bits (a near-tie may pick the other of two coinciding surfaces).  Default occlusion caps the pad of the
        # any-hit boxes: which of the reference's far-away noise hits are still found depends on the boxes, and quantised
        # boxes are a little larger — Demo's ground plane, lit from 2 000 units away, is where that shows
        assert same.all(axis=2).mean() >= (0.9995 if kw else 0.99), (name, kw, same.all(axis=2).mean())
        assert stats[0] == stats[1], (name, kw, stats)
    wide.Dispose(); binary.Dispose()


@pytest.mark.parametrize("env", [dict(RT_HIP_PERSISTENT="1", RT_HIP_PERSISTENT_SHADOW="1"), dict(RT_HIP_SHADOW_RAYS="1"), dict(RT_HIP_SHADOW_RAYS="0", RT_HIP_PERSISTENT_SHADOW="0"),
                                 dict(RT_HIP_WIDE_STRIDE="128"), dict(RT_HIP_VARIANT="generic"),
                                 dict(RT_HIP_QUAD="1"), dict(RT_HIP_QUAD="1", RT_HIP_PERSISTENT_SHADOW="1"), dict(RT_HIP_QUAD="1", RT_HIP_SHADOW_RAYS="0", RT_HIP_PERSISTENT_SHADOW="0")], ids=lambda e: "+".join(f"{k[7:]}={v}" for k, v in e.items()))
def test_every_kernel_that_walks_the_wide_tree(env, monkeypatch):
    """(RT_HIP_QUAD=1: the any-hit kernels walk the 4-wide float form of the tree instead, rt_ldsw.h QuadTree — an opt-in, measured slower on config 5)"""
    sc = helpers.small_scenes()["MiniManyLights"](); flat = flatten_scene(sc)
    wide, binary = _pair(flat, monkeypatch, **env)
    if "RT_HIP_QUAD" in env:
        rs = np.random.default_rng(2); orc = ob.OracleScene(flat)
        so, sd, kmin, kmax = helpers.shadow_ray_set(sc, flat, orc, 60_000, 4)
        assert np.array_equal(wide.TraceAny(so, sd, kmin, kmax).astype(bool), orc.trace_any(so, sd, kmin, kmax).astype(bool))
    opt = RenderOptions(RenderWidth=128, RenderHeight=72, Passes=2, MaxBounceDepth=5, LightSampleCountHint=1, Seed=8, ExactOcclusion=True)
    out = []
    for dev in (wide, binary):
        job = rt.RenderJob(sc, opt, device_scene=dev); job.Wait()
        st = job.RenderStats; out.append((job.ReadAccum()[0], st.RayCount, st.ShadowRays)); job.Dispose()
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1:] == out[1][1:]
    wide.Dispose(); binary.Dispose()


@pytest.mark.parametrize("name", ["Testing", "RtInAWeekendCover1"])
def test_lds_wide_tree_against_the_binary_lds_tree(name, monkeypatch):
    """Small all-sphere scenes keep their tree in LDS in two forms (rt_ldsw.h): the binary one and a 4-wide one whose inner boxes are only
    conservative.  RT_HIP_LDS_WIDE=0 walks the binary form everywhere, =1 (the default) gives the any-hit kernels the wide form, =2 the
    closest-hit kernels too: same hits (K bits included), same any-hit answers, same pictures bit for bit, same ray counts."""
    sc = helpers.small_scenes()[name](); flat = flatten_scene(sc)
    devs = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("RT_HIP_LDS_WIDE", mode); devs[mode] = rt.DeviceScene(flat)
    monkeypatch.delenv("RT_HIP_LDS_WIDE")
    assert all(d.Info()["tree_in_lds"] for d in devs.values())
    assert devs["0"].Info()["wide_nodes"] == 0 and devs["1"].Info()["wide_nodes"] >= 1 and devs["2"].Info()["wide_nodes"] == devs["1"].Info()["wide_nodes"]
    assert rt.DeviceScene(flat).Info()["wide_nodes"] == devs["1"].Info()["wide_nodes"]              # the default is mode 1
    orc = ob.OracleScene(flat)
    rs = np.random.default_rng(6)
    o1, d1 = helpers.camera_rays(sc, 64, 64, 50_000, rs); o2, d2 = helpers.secondary_rays(flat, 50_000, rs)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    o[:200] = o[200:400] * np.array([1, 0, 1], np.float32) + np.array([0, 9, 0], np.float32); d[:200] = np.array([0, -1, 0], np.float32)      # axis-parallel rays: 1/d infinite on two axes
    base = devs["0"].TraceClosest(o, d, 0.001, np.inf, seed=1)
    for mode in ("1", "2"):
        _same_hits(devs[mode].TraceClosest(o, d, 0.001, np.inf, seed=1), base, flat, (o, d))
    so, sd, kmin, kmax = helpers.shadow_ray_set(sc, flat, orc, 100_000, 3)
    ref = orc.trace_any(so, sd, kmin, kmax).astype(bool)                                            # = the reference's every-object scan
    for mode, dev in devs.items():
        assert np.array_equal(dev.TraceAny(so, sd, kmin, kmax).astype(bool), ref), mode
    for kw in (dict(), dict(ExactLightSampling=True, ExactOcclusion=True)):
        opt = RenderOptions(RenderWidth=160, RenderHeight=90, Passes=3, MaxBounceDepth=8, Seed=4, **kw)
        out = {}
        for mode, dev in devs.items():
            job = rt.RenderJob(sc, opt, device_scene=dev); job.Wait()
            st = job.RenderStats; out[mode] = (job.ReadAccum()[0], st.RayCount, st.ShadowRays, st.PixelsRendered); job.Dispose()
        for mode in ("1", "2"):
            same = (out[mode][0] == out["0"][0]) | (np.isnan(out[mode][0]) & np.isnan(out["0"][0]))
            # exact occlusion: every answer is the scan's, so not a bit differs; default occlusion caps the pad, and which far-away noise hits
            # are still found depends on the boxes (Night Spheres: none within reach, so still the same bits)
            assert same.all(axis=2).mean() >= (1.0 if kw else 0.9995), (name, mode, kw, same.all(axis=2).mean())
            assert out[mode][1:] == out["0"][1:]
    for dev in devs.values():
        dev.Dispose()
