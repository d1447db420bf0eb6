"""BuiltinScenes (Ararem.RayTracer.Impl/Builtin/BuiltinScenes.cs) plus the two synthetic benchmark scenes of
BASELINE.json.  The reference's Testing and RtInAWeekendCover1 draw from the unseeded Random.Shared at
property-get time (BuiltinScenes.cs:66-99, 244-285), i.e. a different scene every call; here the same recipes
run on a seeded generator so a scene is a pure function of its seed.
"""
from __future__ import annotations

import math
from typing import List

import numpy as np

from . import numerics as nm
from .core import Camera, Colour, Scene, SceneObject
from .impl import (Box, Capsule, ConstantDensityMedium, DefaultSkyBox, DiffuseSphereLight, Disk, DistanceAttenuation,
                   EmissiveRefractiveMaterial, InfinitePlane, MarbleTexture, PointLight, Quad, RefractiveMaterial, SingleColourSkyBox,
                   SolidColourTexture, Sphere, SphereArray, StandardMaterial, TriangleArray, XYPlane, XZPlane, YZPlane)

f32 = np.float32
White, Black, Red, Green, Blue = Colour.White, Colour.Black, Colour.Red, Colour.Green, Colour.Blue
UnitY, Zero = nm.vec3(0, 1, 0), nm.vec3(0, 0, 0)
DEFAULT_SEED = 0x5EED


class SeededRandom:
    """Stand-in for Random.Shared in scene recipes: splitmix64, NextSingle = top 24 bits * 2^-24 (in [0,1))."""

    def __init__(self, seed: int):
        self.state = seed & 0xFFFFFFFFFFFFFFFF

    def next_u64(self) -> int:
        self.state = (self.state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def RandomFloat01(self) -> np.float32:                             # RandUtils.cs:17
        return f32((self.next_u64() >> 40) * (1.0 / 16777216.0))

    def RandomFloat(self, lo, hi) -> np.float32:                       # RandUtils.cs:21
        lo, hi = f32(lo), f32(hi)
        return f32(lo + f32(f32(hi - lo) * self.RandomFloat01()))

    def RandomColour(self, lo: Colour, hi: Colour) -> Colour:          # RandUtils.cs:72-73, MathUtils.Lerp a + t(b - a)
        return Colour(*(f32(lo.v[i] + f32(self.RandomFloat01() * f32(hi.v[i] - lo.v[i]))) for i in range(3)))


def _sphere_grid(rng: SeededRandom, choose):
    """The 22 x 22 jittered grid both cover scenes share (BuiltinScenes.cs:62-101 / 240-286)."""
    objects: List[SceneObject] = []
    for a in range(-11, 11):
        for b in range(-11, 11):
            chooseMat = rng.RandomFloat01()
            center = nm.vec3(f32(a) + f32(f32(0.9) * rng.RandomFloat01()), 0.2, f32(b) + f32(f32(0.9) * rng.RandomFloat01()))
            if float(nm.length(center - nm.vec3(4.0, 0.2, 0.0))) > 0.9:
                mat = choose(float(chooseMat), center)
                objects.append(SceneObject(f"Sphere ({a},{b})", Sphere(center, 0.2, Material=mat)))
    return objects


def Testing(seed: int = DEFAULT_SEED) -> Scene:
    """BuiltinScenes.Testing, BuiltinScenes.cs:60-109 — the in-tree scene closest to the "Night Spheres" render
    (Renders/Night Spheres/info.md: 1 light, 484 grid cells, black sky)."""
    rng = SeededRandom(seed)

    def choose(chooseMat, center):
        if chooseMat < 0.7:
            albedo = rng.RandomColour(Black, White)
            return StandardMaterial(albedo, rng.RandomFloat01())
        if chooseMat < 0.9:
            albedo = rng.RandomColour(Black, White * f32(.7))
            return StandardMaterial(albedo, albedo, 1.0)
        tint = rng.RandomColour(Black, White)
        return RefractiveMaterial(rng.RandomFloat(1.0, 5.0), SolidColourTexture(tint))

    objects = _sphere_grid(rng, choose)
    lights = [DiffuseSphereLight(Position=(0, 3, 0), Colour=White, AttenuationRadius=8, DiffusionRadius=1.0)]
    objects.append(SceneObject("Sphere B", Sphere((-4, 1, 0), 1, Material=StandardMaterial(Colour(.4, .2, .1), 1.0))))
    objects.append(SceneObject("Sphere C", Sphere((4, 1, 0), 1, Material=StandardMaterial(Colour(.7, .6, .5), 0.0))))
    objects.append(SceneObject("Ground", InfinitePlane(Zero, UnitY, Material=StandardMaterial(Colour(0.5), Black, .3))))
    return Scene("Testing", Camera.Create((13, 2, 3), Zero, UnitY, 20, 16.0 / 9.0, 0.0, 10.0), objects, lights, SingleColourSkyBox(Black))


NightSpheres = Testing


def Demo() -> Scene:
    """BuiltinScenes.Demo, BuiltinScenes.cs:114-180"""
    objects: List[SceneObject] = []
    lights = []
    camera = Camera.Create((0, 2.87, 7), UnitY * f32(3), UnitY, 70, 16.0 / 9.0, 0.0, 1.0)
    objects.append(SceneObject("Ground", InfinitePlane((0, -0.001, 0), UnitY, Material=StandardMaterial(MarbleTexture(), SolidColourTexture(Black), .5))))
    low, high = nm.vec3(-7, 0, -2), nm.vec3(-5, 2.8, -.5)
    objects.append(SceneObject("XY", XYPlane(low[0], high[0], low[1], high[1], low[2], Material=StandardMaterial(Colour(1, .5, .5), .5))))
    objects.append(SceneObject("YZ", YZPlane(low[1], high[1], low[2], high[2], low[0], Material=StandardMaterial(Colour(.5, 1, .5), .5))))
    objects.append(SceneObject("XZ", XZPlane(low[0], high[0], low[2], high[2], low[1], Material=StandardMaterial(Colour(.5, .5, 1), Black, .5))))
    objects.append(SceneObject("Planes Sphere Light", Sphere(((low + high) / f32(2)) - UnitY, .5, Material=StandardMaterial(Black, White * f32(0.8), 0.0))))
    objects.append(SceneObject("Lonely Sphere", Sphere((-1, 3, -2), 1.0, Material=StandardMaterial(Colour(f32(165) / f32(255), f32(42) / f32(255), f32(42) / f32(255)), 0.0))))
    objects.append(SceneObject("Capsule", Capsule((-2, .7, -3), (0, 1.5, -1), .7,
                                                  Material=RefractiveMaterial(RefractiveMaterial.GlassIndex, SolidColourTexture(Colour(0.27058825, 0.77254903, 1.0))))))
    lights.append(PointLight(Position=(-1, 5, -2), Colour=Red * f32(.25), AttenuationRadius=math.inf, CutoffRadius=math.inf,
                             DistanceAttenuationFunc=DistanceAttenuation.Constant(1.0)))
    objects.append(SceneObject("Infinite Light Visualiser", Sphere((-1, 5.1, -2), .05, Material=StandardMaterial(Black, Red, 0.0))))
    lights.append(PointLight(Position=(-5, 1, -7), Colour=Green * f32(10), AttenuationRadius=1.5))
    objects.append(SceneObject("Sized Light Visualiser", Sphere((-5, 1.1, -7), .05, Material=StandardMaterial(Black, Green, 0.0))))
    objects.append(SceneObject("Sized Light Blocker", Sphere((-5, .6, -7), .2, Material=StandardMaterial(Black, 0.0))))
    lights.append(DiffuseSphereLight(Position=(3, 1, -7), DiffusionRadius=.3, Colour=Blue, AttenuationRadius=2.0))
    objects.append(SceneObject("Diffuse Light Visualiser", Sphere((3, 1.1, -7), .1, Material=StandardMaterial(Black, Blue, 0.0))))
    objects.append(SceneObject("Diffuse Light Blocker", Sphere((3, .6, -7), .3, Material=StandardMaterial(Black, 0.0))))
    objects.append(SceneObject("Smoke Box", ConstantDensityMedium(Box((-4, 0, 0), (-1, 1, 2)), 2.0, Black)))
    objects.append(SceneObject("Hex Box", Box((-3, 0.75, 0.5), (-2, 1.25, 1.5), Material=StandardMaterial(Colour.Orange * f32(.5), 1.0))))
    objects.append(SceneObject("Disk", Disk((5, .5, 1), nm.normalize(nm.vec3(-1, 1, -1)), .7, Material=StandardMaterial(Colour.Purple * f32(.6), .3))))
    objects.append(SceneObject("Quad", Quad((2, 0, 1), (0, 1, -1), (1, 0, 0), Material=StandardMaterial(Colour.Yellow * f32(.6), .1))))
    return Scene("Demo", camera, objects, lights, DefaultSkyBox())


def SphereScene() -> Scene:
    """BuiltinScenes.Sphere, BuiltinScenes.cs:183-188"""
    return Scene("A lonely Sphere", Camera.Create((0, 0, 2), (0.1, 0, 0), UnitY, 20, 16.0 / 9.0, 0, 1.0),
                 [SceneObject("Sphere", Sphere(Zero, .1, Material=StandardMaterial(Red, 1.0)))], [], DefaultSkyBox())


def RgbSpheres() -> Scene:
    """BuiltinScenes.RgbSpheres, BuiltinScenes.cs:191-199"""
    return Scene("RGB Spheres", Camera.Create((0, 0, 5), Zero, UnitY, 5, 16.0 / 9.0, .00002, 5.0), [
        SceneObject("Sphere 1", Sphere(nm.vec3(0.1), .1, Material=StandardMaterial(Colour.Lerp(Red, White, 0.5), 1.0))),
        SceneObject("Sphere 2", Sphere(nm.vec3(0.0), .1, Material=StandardMaterial(Colour.Lerp(Green, White, 0.5), 1.0))),
        SceneObject("Sphere 3", Sphere(nm.vec3(-0.1), .1, Material=StandardMaterial(Colour.Lerp(Blue, White, 0.5), 1.0))),
    ], [], DefaultSkyBox())


def CornellBox(emission_intensity: float = 0.1, direct_emission: bool = True, aspect: float = 1.0) -> Scene:
    """BuiltinScenes.CornellBox, BuiltinScenes.cs:202-229.  The "Refractive Material Indirect Emission Comparison"
    renders vary the Small Box Sphere's emission intensity {0.1, 1.0} and DirectEmission flag
    (Renders/Refractive Material Indirect Emission Comparison/info.md:8-9); the defaults are the in-tree values."""
    grey = StandardMaterial(Colour(0.73, 0.73, 0.73), 1.0)
    rad = nm.PI / f32(180.0)

    def box(sy, yaw_deg, t):
        m = nm.matmul4(nm.matmul4(nm.create_scale(165, sy, 165), nm.create_from_yaw_pitch_roll(f32(yaw_deg) * rad, f32(0) * rad, f32(0) * rad)),
                       nm.create_translation(*t))
        return Box(m, Material=StandardMaterial(Colour(0.73, 0.73, 0.73), 1.0))

    objects = [
        SceneObject("Left", YZPlane(0, 555, 0, 555, 0, Material=StandardMaterial(Colour(0.5, 0.1, 0.1), 1.0))),
        SceneObject("Right", YZPlane(0, 555, 0, 555, 555, Material=StandardMaterial(Colour(0.1, 0.5, 0.1), 1.0))),
        SceneObject("Back", XYPlane(0, 555, 0, 555, 555, Material=grey)),
        SceneObject("Top", XZPlane(0, 555, 0, 555, 555, Material=grey)),
        SceneObject("Bottom", XZPlane(0, 555, 0, 555, 0, Material=grey)),
        SceneObject("Light", XZPlane(213, 343, 227, 332, 554.9, Material=StandardMaterial(White, White, 1.0))),
        SceneObject("Small Box", box(165, -18, (212.5, 82.5, 147.5))),
        SceneObject("Tall Box", box(330, 15, (347.5, 165.0, 377.5))),
        SceneObject("Small Box Sphere", Sphere((212.5, 265.0, 147.5), 100, Material=EmissiveRefractiveMaterial(
            RefractiveMaterial.GlassIndex, SolidColourTexture(White), SolidColourTexture(Blue * f32(emission_intensity)), direct_emission))),
        SceneObject("Tall Box Sphere", Sphere((347.5, 430.0, 377.5), 100, Material=RefractiveMaterial(RefractiveMaterial.GlassIndex, SolidColourTexture(White)))),
    ]
    lights = [DiffuseSphereLight(Position=(f32(213 + 343) / f32(2), 554 - 50, f32(227 + 332) / f32(2)), DiffusionRadius=40, Colour=White * f32(0.5), AttenuationRadius=150)]
    return Scene("Cornell Box", Camera.Create((278, 278, -800), (278, 278, 0), UnitY, 40.0, aspect, 0.0, 1.0), objects, lights, SingleColourSkyBox(Black))


def RtInAWeekendCover1(seed: int = DEFAULT_SEED) -> Scene:
    """BuiltinScenes.RtInAWeekendCover1, BuiltinScenes.cs:233-295"""
    rng = SeededRandom(seed)
    lights = []

    def choose(chooseMat, center):
        if chooseMat < 0.3:
            return StandardMaterial(rng.RandomColour(Black, White), 1.0)
        if chooseMat < 0.5:
            return StandardMaterial(White, rng.RandomColour(Black, White), 1.0)
        if chooseMat < 0.65:
            albedo = rng.RandomColour(Black, White)
            fuzz = rng.RandomFloat(0.0, 0.5)
            return StandardMaterial(albedo, f32(1) - fuzz)
        if chooseMat < 0.655:
            colour = rng.RandomColour(Colour.HalfGrey, White)
            lights.append(DiffuseSphereLight(Position=center, Colour=colour, AttenuationRadius=1.0, DiffusionRadius=.4))
            return StandardMaterial(White, 0.0)
        tint = rng.RandomColour(Black, White)
        return RefractiveMaterial(rng.RandomFloat(1.0, 5.0), SolidColourTexture(tint))

    objects = _sphere_grid(rng, choose)
    objects.append(SceneObject("Sphere A", Sphere((0, 1, 0), 1, Material=RefractiveMaterial(1.5, SolidColourTexture(White)))))
    objects.append(SceneObject("Sphere B", Sphere((-4, 1, 0), 1, Material=StandardMaterial(Colour(.4, .2, .1), 1.0))))
    objects.append(SceneObject("Sphere C", Sphere((4, 1, 0), 1, Material=StandardMaterial(Colour(.7, .6, .5), 0.0))))
    objects.append(SceneObject("Ground", InfinitePlane(Zero, UnitY, Material=StandardMaterial(Colour(0.5), Black, 1.0))))
    return Scene("RayTracing Chapter 1", Camera.Create((13, 2, 3), Zero, UnitY, 20, 16.0 / 9.0, 0.0, 10.0), objects, lights, DefaultSkyBox())


# ---------------------------------------------------------------------------------------------------------
# Synthetic scenes of BASELINE.json configs 4 and 5 (no reference counterpart)
# ---------------------------------------------------------------------------------------------------------
def TriangleMesh(grid: int = 1581, seed: int = DEFAULT_SEED) -> Scene:
    """A displaced grid x grid height field, two triangles per cell (1581 -> 4 999 122 triangles), Standard diffuse
    albedo 0.7, DefaultSkyBox, one PointLight."""
    rs = np.random.Generator(np.random.Philox(seed))
    n = grid + 1
    xs = np.linspace(-10.0, 10.0, n, dtype=np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    Y = (0.6 * np.sin(X * 0.9) * np.cos(Z * 0.7) + 0.25 * np.sin(X * 3.1 + Z * 2.3)).astype(np.float32)
    Y += (rs.random((n, n), dtype=np.float32) - 0.5) * np.float32(4.0 / grid)
    P = np.stack([X, Y, Z], axis=-1).astype(np.float32)
    p00, p10, p01, p11 = P[:-1, :-1].reshape(-1, 3), P[1:, :-1].reshape(-1, 3), P[:-1, 1:].reshape(-1, 3), P[1:, 1:].reshape(-1, 3)
    v0 = np.concatenate([p00, p11]); v1 = np.concatenate([p01, p10]); v2 = np.concatenate([p10, p01])    # normals face +y
    mesh = TriangleArray(v0, v1, v2, Material=StandardMaterial(Colour(0.7), 1.0))
    cam = Camera.Create((0, 7, 16), (0, 0, 0), UnitY, 40, 16.0 / 9.0, 0.0, 1.0)
    lights = [PointLight(Position=(4, 12, 6), Colour=White, AttenuationRadius=40.0)]
    return Scene(f"Triangle mesh {2 * grid * grid}", cam, [], lights, DefaultSkyBox(), BulkObjects=[mesh])


def ManySpheresManyLights(n_spheres: int = 100_000, n_lights: int = 64, seed: int = DEFAULT_SEED) -> Scene:
    """n_spheres spheres r in U(0.05, 0.25) in a 100 x 100 slab over a ground plane, materials drawn as in Testing
    (70 % Standard, 20 % emissive Standard, 10 % Refractive), n_lights PointLights on a grid (AttenuationRadius 8)."""
    rs = np.random.Generator(np.random.Philox(seed))
    r = rs.uniform(0.05, 0.25, n_spheres).astype(np.float32)
    c = np.stack([rs.uniform(-50, 50, n_spheres), r + rs.uniform(0, 1.5, n_spheres), rs.uniform(-50, 50, n_spheres)], axis=1).astype(np.float32)
    n_mats = 256
    rng = SeededRandom(seed)
    mats = []
    for _ in range(n_mats):
        ch = float(rng.RandomFloat01())
        if ch < 0.7:
            mats.append(StandardMaterial(rng.RandomColour(Black, White), rng.RandomFloat01()))
        elif ch < 0.9:
            a = rng.RandomColour(Black, White * f32(.7)); mats.append(StandardMaterial(a, a, 1.0))
        else:
            mats.append(RefractiveMaterial(rng.RandomFloat(1.0, 5.0), SolidColourTexture(rng.RandomColour(Black, White))))
    spheres = SphereArray(c, r, mats, rs.integers(0, n_mats, n_spheres))
    side = max(1, int(round(math.sqrt(n_lights))))
    lights = []
    for i in range(n_lights):
        gx, gz = i % side, i // side
        lights.append(PointLight(Position=(-40 + 80 * (gx + 0.5) / side, 4.0, -40 + 80 * (gz + 0.5) / side), Colour=White, AttenuationRadius=8.0))
    ground = SceneObject("Ground", InfinitePlane(Zero, UnitY, Material=StandardMaterial(Colour(0.5), Black, .3)))
    cam = Camera.Create((0, 18, 70), (0, 0, 0), UnitY, 40, 16.0 / 9.0, 0.0, 10.0)
    return Scene(f"{n_spheres} spheres, {n_lights} lights", cam, [ground], lights, SingleColourSkyBox(Black), BulkObjects=[spheres])


def GetAll():
    """BuiltinScenes.GetAll, BuiltinScenes.cs:298-301"""
    return [Testing(), Demo(), SphereScene(), RgbSpheres(), CornellBox(), RtInAWeekendCover1()]
