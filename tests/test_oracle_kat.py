"""Pins oracle/oracle.cpp (the CPU restatement) against every known answer the reference holds for the path
(SURVEY §8c): the sky rows of the committed Demo render, the GetSphereUV table, the capsule basis comment,
the GraphicsValidator invariants, plus analytic cases."""
import os

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs, impl, numerics as nm
from raytracer_amd.core import Camera, Colour, RenderOptions, Scene, SceneObject
from raytracer_amd.flatten import flatten_scene
from oracle import binding as ob

import helpers


def test_sky_golden_of_demo_render(golden_dir):
    """Renders/Demo Scene/image.png (1920x1080, 1000 passes): the sky region is Camera.Create + GetRay + y-flip +
    DefaultSkyBox + Clamp01/Sqrt/(byte) with nothing random but the sub-pixel jitter, whose mean is the pixel centre.
    Bit-exact at the centre for all but pixels sitting on a quantisation boundary (then off by 1 LSB)."""
    g = np.load(os.path.join(golden_dir, "demo_sky_pixels.npz"))
    W, H = int(g["width"]), int(g["height"])
    flat = flatten_scene(bs.Demo())
    got = np.stack([ob.pixel_centre_sky_rgb24(flat.desc.camera, flat.desc.skybox, int(x), H - 1 - int(y), W, H) for x, y in zip(g["x"], g["y_top"])])
    diff = np.abs(got.astype(int) - g["rgb"].astype(int))
    assert diff.max() <= 1
    assert (diff.max(axis=1) == 0).mean() > 0.995


def test_python_camera_matches_oracle_camera_bit_for_bit():
    for args in [((0, 2.87, 7), (0, 3, 0), (0, 1, 0), 70, 16 / 9, 0.0, 1.0), ((13, 2, 3), (0, 0, 0), (0, 1, 0), 20, 16 / 9, 0.0, 10.0),
                 ((278, 278, -800), (278, 278, 0), (0, 1, 0), 40, 1.0, 0.0, 1.0), ((0, 0, 5), (0, 0, 0), (0, 3, 0), 5, 16 / 9, .00002, 5.0)]:
        py = Camera.Create(*args)
        oc = ob.camera_create(*args)
        for name, field in (("LookFrom", "look_from"), ("Horizontal", "horizontal"), ("Vertical", "vertical"), ("LowerLeftCorner", "lower_left_corner"), ("U", "u"), ("V", "v")):
            assert np.array_equal(getattr(py, name), np.array(list(getattr(oc, field)), dtype=np.float32)), name


def test_camera_rejects_up_parallel_to_view():
    with pytest.raises(ArithmeticError):
        Camera.Create((0, 5, 0), (0, 0, 0), (0, 1, 0), 40, 1.0, 0.0, 1.0)


def test_sphere_uv_table():
    """Impl/Hittables/Sphere.cs:83-85"""
    table = {(1, 0, 0): (0.50, 0.50), (-1, 0, 0): (0.00, 0.50), (0, 1, 0): (0.50, 1.00), (0, -1, 0): (0.50, 0.00), (0, 0, 1): (0.25, 0.50), (0, 0, -1): (0.75, 0.50)}
    for p, (u, v) in table.items():
        got = ob.sphere_uv(p)
        if p == (-1, 0, 0):     # atan2(-0, -1) = -pi -> u = 0, as the table says (+pi would give 1)
            assert got[0] in (0.0, 1.0)
        else:
            assert abs(got[0] - u) < 1e-6, p
        assert abs(got[1] - v) < 1e-6, p


def test_capsule_basis_comment():
    """Impl/Hittables/Capsule.cs:126-130: basis vectors and extent of the Demo scene's capsule."""
    c = impl.Capsule((-2, .7, -3), (0, 1.5, -1), .7)
    U, V, Wv = c.uvMatrix[:, 0], c.uvMatrix[:, 1], c.uvMatrix[:, 2]
    assert np.allclose(U, [0.37139067, -0.9284767, 0], atol=1e-7)
    assert np.allclose(V, [-0.6317484, -0.25269935, 0.73282814], atol=1e-7)
    assert np.allclose(Wv, [0.68041384, 0.27216554, 0.68041384], atol=1e-7)
    assert abs(float(nm.distance(c.P1, c.P2)) - 2.9393876) < 1e-6
    # the oracle's hit UVs on that capsule stay inside the rawUv ranges the comment lists, i.e. in [0, 1]
    sc = Scene("c", Camera.Create((0, 1, 5), (-1, 1, -2), (0, 1, 0), 40, 1.0, 0, 1), [SceneObject("Capsule", c)], [], impl.DefaultSkyBox())
    o = ob.OracleScene(sc)
    rs = np.random.default_rng(1)
    org, d = helpers.camera_rays(sc, 64, 64, 4000, rs)
    h = o.trace_closest(org, d)
    hit = h["object"] == 0
    assert hit.sum() > 200
    assert (h["uv"][hit] >= -1e-5).all() and (h["uv"][hit] <= 1 + 1e-5).all()


@pytest.mark.parametrize("name", ["Demo", "Testing", "CornellBox", "Zoo", "MiniMesh", "MiniManyLights"])
def test_graphics_validator_invariants(name):
    """Core/Debugging/GraphicsValidator.cs:57-69 on the oracle's hits: |N|^2 ~ 1 (+-0.01), UV in [0,1]^2, K in [kMin,kMax].
    Known reference behaviour that violates them is excluded: InfinitePlane keeps a non-unit normal (InfinitePlane.cs:24),
    Cylinder cap normals are ba/baba (Cylinder.cs:93) and the volumetric hit's normal/uv are synthetic."""
    sc = helpers.small_scenes()[name]()
    flat = flatten_scene(sc)
    o = ob.OracleScene(flat)
    rs = np.random.default_rng(7)
    org, d = helpers.camera_rays(sc, 64, 64, 6000, rs)
    o2, d2 = helpers.secondary_rays(flat, 6000, rs)
    org, d = np.concatenate([org, o2]), np.concatenate([d, d2])
    kmin = 0.001
    h = o.trace_closest(org, d, kmin, np.inf)
    hit = h["object"] != 0xFFFFFFFF
    assert hit.sum() > 500
    assert (h["k"][hit] >= kmin).all()
    n2 = (h["normal"][hit] ** 2).sum(axis=1)
    names = [so.Name for so in sc.SceneObjects]
    objs = h["object"][hit]
    skip = np.zeros(len(objs), bool)
    for i, so in enumerate(sc.SceneObjects):
        if isinstance(so.Hittable, (impl.InfinitePlane, impl.Cylinder)):
            skip |= objs == i
    assert (np.abs(n2[~skip] - 1) <= 0.01).all()
    uv = h["uv"][hit]
    assert (uv >= -1e-5).all() and (uv <= 1 + 1e-5).all()
    # the hit point lies on the ray at K (media report WorldPoint from the ray origin at `dist`, ConstantDensityMedium.cs:61)
    med = np.zeros(len(objs), bool)
    for i, so in enumerate(sc.SceneObjects):
        if isinstance(so.Hittable, impl.ConstantDensityMedium):
            med |= objs == i
    wp = org[hit] + h["k"][hit][:, None] * d[hit]
    assert np.allclose(wp[~med], h["world_point"][hit][~med], rtol=1e-4, atol=1e-3)


def test_analytic_sphere_hits():
    sc = bs.SphereScene()              # sphere r=0.1 at the origin
    o = ob.OracleScene(sc)
    org = np.array([[0, 0, 2], [0, 0, 0], [0, 0, 2], [0.2, 0, 2]], np.float32)
    d = np.array([[0, 0, -1], [0, 1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    h = o.trace_closest(org, d)
    assert h["object"][0] == 0 and abs(h["k"][0] - 1.9) < 1e-6 and h["outside_face"][0] == 1 and np.allclose(h["normal"][0], [0, 0, 1])
    assert h["object"][1] == 0 and abs(h["k"][1] - 0.1) < 1e-6 and h["outside_face"][1] == 0 and np.allclose(h["normal"][1], [0, -1, 0])   # from inside: far root, flipped normal
    assert h["object"][2] == 0xFFFFFFFF and h["object"][3] == 0xFFFFFFFF
    # K range: kMax below the near root -> falls to the far root -> also out of range -> miss (Sphere.cs:48-56)
    assert o.trace_closest(org[:1], d[:1], 0.001, 1.0)["object"][0] == 0xFFFFFFFF
    assert o.trace_any(org[:1], d[:1], 0.001, 5.0)[0] and not o.trace_any(org[:1], d[:1], 0.001, 1.0)[0]


def test_nan_and_degenerate_rays_miss():
    o = ob.OracleScene(bs.Demo())
    org = np.array([[np.nan, 0, 0], [0, 1, 0], [0, 1, 0]], np.float32)
    d = np.array([[0, 0, -1], [np.nan, 0, 0], [0, 0, 0]], np.float32)
    h = o.trace_closest(org, d)
    assert h["object"][0] == 0xFFFFFFFF and h["object"][1] == 0xFFFFFFFF


def test_resolve_rgb24_truncates_after_sqrt():
    """RenderJob.cs:646-654 / Colour.cs:197-234: clamp, sqrt, (byte)(255*c) truncation; NaN sums clamp like Max/Min with ?:"""
    s = np.array([[[0.25, 4.0, -1.0], [1.0, 0.5, np.nan]]], np.float32) * 4
    out = ob.resolve_rgb24(s, 4)
    assert out[0, 0].tolist() == [127, 255, 0]
    assert out[0, 1, 0] == 255 and out[0, 1, 1] == int(255 * np.sqrt(np.float32(0.5)))
    assert out[0, 1, 2] == 0      # NaN > 0 is false -> Black's 0


def test_oracle_render_is_deterministic_and_partition_independent():
    sc = bs.Demo()
    o = ob.OracleScene(sc)
    opt = RenderOptions(RenderWidth=48, RenderHeight=27, Passes=4, MaxBounceDepth=6)
    a, st = o.render(opt, threads=4)
    b, _ = o.render(opt, threads=1)
    assert np.array_equal(a, b)
    # pass sharding: ranks {0,2} + {1,3} summed pass-by-pass equals the single run only up to float association
    e, _ = o.render(opt, first_pass=0, n_passes=2, stride=2)
    f, _ = o.render(opt, first_pass=1, n_passes=2, stride=2)
    assert np.allclose(e + f, a, rtol=1e-5, atol=1e-6)
    assert st["pixels_rendered"] == 48 * 27 * 4 and st["ray_count"] > st["pixels_rendered"]
    assert st["sky_rays"] + st["bounce_limit_exceeded"] == st["pixels_rendered"]      # no material absorbs (Scatter never returns null)


def test_validator_errors_are_filed_per_object():
    """GraphicsValidator.Errors[type][object] (GraphicsValidator.cs:18, 35-45; RenderJob.cs:597, 606): the Zoo scene's ground plane was built
    with the normal (0, 2, 0), which InfinitePlane.cs:20-25 keeps as given, so every hit on it fails CheckVectorNormalized; the only other
    object that ever does is the cylinder, whose cap normal is the axis divided by its squared length (Cylinder.cs:88)."""
    import helpers
    from raytracer_amd.core import RenderOptions
    from raytracer_amd.flatten import flatten_scene
    sc = helpers.zoo_scene(); orc = ob.OracleScene(flatten_scene(sc))
    opt = RenderOptions(RenderWidth=64, RenderHeight=36, Passes=2, MaxBounceDepth=5, Seed=3)
    _, st, nw, uv = orc.render_validated(opt, threads=4)
    ground = [i for i, o in enumerate(sc.SceneObjects) if o.Name == "ground"][0]
    cyl = [i for i, o in enumerate(sc.SceneObjects) if o.Name == "cyl"][0]
    assert nw[ground] > 0 and set(np.nonzero(nw)[0]) <= {ground, cyl} and nw.sum() == st["normals_wrong_magnitude"]
    assert uv.sum() == st["uv_invalid"]
    _, st2 = orc.render(opt, threads=4)
    assert (st2["normals_wrong_magnitude"], st2["uv_invalid"], st2["ray_count"]) == (st["normals_wrong_magnitude"], st["uv_invalid"], st["ray_count"])
