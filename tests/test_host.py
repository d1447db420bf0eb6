"""Host-side mirror of the reference's Scene API: flattening rules, rejection of user-defined types, option validation,
pass sharding.  No GPU."""
import ctypes as C
import math

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import builtin_scenes as bs, capi, impl, numerics as nm
from raytracer_amd.core import (ArgumentOutOfRangeError, Camera, Colour, Hittable, Light, Material, NotSupportedError, RenderOptions, Scene,
                                SceneObject, SkyBox, Texture)
from raytracer_amd.flatten import flatten_scene

import helpers


def _objects(flat):
    n = flat.n_objects
    raw = np.frombuffer((C.c_char * (n * 36)).from_address(C.addressof(flat.desc.objects.contents)), dtype=np.uint32).reshape(n, 9)
    return raw


def test_material_ids_follow_object_identity_not_value():
    """RefractiveMaterial.cs:51-52 / EmissiveRefractiveMaterial.cs:53-63 compare materials by reference."""
    shared = impl.RefractiveMaterial(1.5, Colour.White)
    twin_a, twin_b = impl.RefractiveMaterial(1.5, Colour.White), impl.RefractiveMaterial(1.5, Colour.White)
    objs = [SceneObject("a", impl.Sphere((0, 0, 0), 1, Material=shared)), SceneObject("b", impl.Sphere((3, 0, 0), 1, Material=shared)),
            SceneObject("c", impl.Sphere((6, 0, 0), 1, Material=twin_a)), SceneObject("d", impl.Sphere((9, 0, 0), 1, Material=twin_b))]
    flat = flatten_scene(Scene("s", Camera.Create((0, 0, 9), (0, 0, 0), (0, 1, 0), 40, 1, 0, 1), objs, [], impl.DefaultSkyBox()))
    mats = _objects(flat)[:, 2]
    assert mats[0] == mats[1] and mats[2] != mats[3] and mats[0] != mats[2]
    assert flat.n_materials == 3


def test_cornell_box_shares_the_grey_wall_material():
    flat = flatten_scene(bs.CornellBox())
    mats = _objects(flat)[:, 2]
    assert mats[2] == mats[3] == mats[4]          # Back, Top, Bottom use greyWallMaterial (BuiltinScenes.cs:206,212-214)
    assert len(set(mats.tolist())) == 8


def test_default_material_is_a_new_instance_per_hittable():
    """BuiltinMaterials.DefaultDiffuseMaterial is a property that news up a material per get (BuiltinMaterials.cs:10)."""
    a, b = impl.Sphere((0, 0, 0), 1), impl.Sphere((1, 0, 0), 1)
    assert a.Material is not b.Material


class MyHittable(Hittable):
    BoundingVolume = rt.AxisAlignedBoundingBox((0, 0, 0), (1, 1, 1))
    Material = impl.DefaultDiffuseMaterial()


class MyMaterial(Material):
    pass


class MySky(SkyBox):
    pass


class MyLight(Light):
    pass


@pytest.mark.parametrize("make", [
    lambda: Scene("s", bs.Demo().Camera, [SceneObject("x", MyHittable())], [], impl.DefaultSkyBox()),
    lambda: Scene("s", bs.Demo().Camera, [SceneObject("x", impl.Sphere((0, 0, 0), 1, Material=MyMaterial()))], [], impl.DefaultSkyBox()),
    lambda: Scene("s", bs.Demo().Camera, [SceneObject("x", impl.Sphere((0, 0, 0), 1, Material=impl.StandardMaterial(impl.CustomFunctionTexture(lambda h: Colour.Red), 1.0)))], [], impl.DefaultSkyBox()),
    lambda: Scene("s", bs.Demo().Camera, [SceneObject("x", impl.Sphere((0, 0, 0), 1, Material=impl.StandardMaterial(impl.GreyscaleNoiseTexture(object()), 1.0)))], [], impl.DefaultSkyBox()),     # a Module that is not Perlin
    lambda: Scene("s", bs.Demo().Camera, [], [], MySky()),
    lambda: Scene("s", bs.Demo().Camera, [], [MyLight()], impl.DefaultSkyBox()),
    lambda: Scene("s", bs.Demo().Camera, [], [impl.PointLight(DistanceAttenuationFunc=lambda l, x: 1.0)], impl.DefaultSkyBox()),
])
def test_user_defined_types_are_rejected_not_emulated(make):
    with pytest.raises(NotSupportedError):
        flatten_scene(make())


def test_greyscale_noise_over_perlin_is_flattened():
    from raytracer_amd import capi
    sc = Scene("s", bs.Demo().Camera, [SceneObject("x", impl.Sphere((0, 0, 0), 1, Material=impl.StandardMaterial(impl.GreyscaleNoiseTexture(impl.Perlin(Frequency=2, OctaveCount=3, Seed=7)), 1.0)))],
               [], impl.DefaultSkyBox())
    flat = flatten_scene(sc)
    t = [flat.desc.textures[i] for i in range(flat.desc.n_textures)]
    noise = [x for x in t if x.type == capi.TEX_GREYSCALE_PERLIN]
    assert len(noise) == 1 and list(noise[0].f[:5]) == [2.0, 2.0, 0.5, 3.0, 7.0]          # Frequency, Lacunarity (default 2), Persistence (.5), OctaveCount, Seed


def test_render_options_validation_follows_the_reference_setters():
    o = RenderOptions()
    assert (o.RenderWidth, o.RenderHeight, o.Passes, o.MaxBounceDepth, o.LightSampleCountHint) == (1920, 1080, 100, 10, 2)     # RenderOptions.cs:11-19
    assert abs(o.KMin - 0.001) < 1e-9 and o.KMax == math.inf
    for kw in (dict(RenderWidth=0), dict(RenderHeight=0), dict(KMin=-1), dict(KMax=-1), dict(KMin=2, KMax=1), dict(LightSampleCountHint=0), dict(ConcurrencyLevel=0)):
        with pytest.raises(ArgumentOutOfRangeError):
            RenderOptions(**kw)


def test_shard_passes_partitions_every_pass_exactly_once():
    for total in (0, 1, 7, 8, 256, 1000):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                first, n, stride = rt.shard_passes(total, r, world)
                seen += [first + i * stride for i in range(n)]
            assert sorted(seen) == list(range(total))
    with pytest.raises(ValueError):
        rt.shard_passes(8, 2, 2)


def test_seeded_scenes_are_pure_functions_of_the_seed():
    a, b, c = flatten_scene(bs.Testing(1)), flatten_scene(bs.Testing(1)), flatten_scene(bs.Testing(2))
    raw = lambda f: bytes((C.c_char * (f.desc.n_spheres * 16)).from_address(C.addressof(f.desc.spheres.contents)))
    assert raw(a) == raw(b) and raw(a) != raw(c)
    t = bs.Testing()
    assert 440 <= len(t.SceneObjects) <= 487 and len(t.Lights) == 1       # 22 x 22 grid minus the cells near (4, .2, 0), + B, C, Ground
    names = [o.Name for o in t.SceneObjects[-3:]]
    assert names == ["Sphere B", "Sphere C", "Ground"]


def test_box_constructor_quirks():
    """Box.cs:84-93 (c2 = Max(c1', c2) keeps c2, so swapped corners collapse) and :45-59 (AABB from the +-1 corners)."""
    b = impl.Box((-3, 0.75, 0.5), (-2, 1.25, 1.5))
    assert np.allclose(b.BoundingVolume.Min, [-3.5, 0.5, 0.0]) and np.allclose(b.BoundingVolume.Max, [-1.5, 1.5, 2.0])
    with pytest.raises(ArithmeticError):
        impl.Box((1, 1, 1), (0, 0, 0))              # size collapses to zero -> singular matrix


def test_quad_matrix_maps_corners_to_unit_square():
    q = impl.Quad((2, 0, 1), (0, 1, -1), (1, 0, 0))
    for p, uv in (((0, 0, 0), (0, 0)), ((0, 1, -1), (1, 0)), ((1, 0, 0), (0, 1)), ((1, 1, -1), (1, 1))):
        lp = np.array(p, np.float32)
        u = lp @ q._col1[:3] + q._col1[3]; v = lp @ q._col2[:3] + q._col2[3]
        assert abs(u - uv[0]) < 1e-6 and abs(v - uv[1]) < 1e-6
    with pytest.raises(ArithmeticError):
        impl.Quad((0, 0, 0), (1, 0, 0), (2, 0, 0))


def test_bulk_arrays_flatten_like_individual_objects():
    v0 = np.array([[0, 0, 0], [1, 0, 0]], np.float32); v1 = v0 + [1, 0, 0]; v2 = v0 + [0, 1, 0]
    mat = impl.StandardMaterial(Colour(0.7), 1.0)
    bulk = flatten_scene(Scene("b", bs.Demo().Camera, [], [], impl.DefaultSkyBox(), BulkObjects=[impl.TriangleArray(v0, v1, v2, Material=mat)]))
    single = flatten_scene(Scene("s", bs.Demo().Camera, [SceneObject(str(i), impl.Triangle(v0[i], v1[i], v2[i], Material=mat)) for i in range(2)], [], impl.DefaultSkyBox()))
    ta = np.frombuffer((C.c_char * 160).from_address(C.addressof(bulk.desc.triangles.contents)), np.float32)
    tb = np.frombuffer((C.c_char * 160).from_address(C.addressof(single.desc.triangles.contents)), np.float32)
    assert np.allclose(ta, tb, atol=1e-7)
    assert np.array_equal(_objects(bulk)[:, :3], _objects(single)[:, :3])
    assert np.allclose(_objects(bulk)[:, 3:].view(np.float32), _objects(single)[:, 3:].view(np.float32), atol=1e-6)


def test_full_size_synthetic_scenes_have_the_configured_counts():
    mesh = bs.TriangleMesh(grid=200)
    assert len(mesh.BulkObjects[0]) == 2 * 200 * 200
    assert 2 * 1581 * 1581 == 4_999_122                   # BASELINE config 4: ~5 M triangles
    many = bs.ManySpheresManyLights(n_spheres=5000, n_lights=64)
    assert len(many.BulkObjects[0]) == 5000 and len(many.Lights) == 64
