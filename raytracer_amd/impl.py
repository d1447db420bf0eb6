"""Host-side mirror of Ararem.RayTracer.Impl: the builtin Hittables, Materials, Textures, Lights and SkyBoxes.

Each class keeps the reference's constructor signature and caches exactly the fields the reference's
constructor caches (Impl/ = Ararem.RayTracer.Impl/), computed in binary32 in the same operator order; its
`flatten(builder)` copies those fields into the C ABI tables of include/rt_hip.h.  Nothing here intersects
rays or shades: that is the device's job.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np

from . import capi
from . import numerics as nm
from .core import AxisAlignedBoundingBox, Colour, Hittable, Light, Material, NotSupportedError, SkyBox, Texture

f32 = np.float32
_UNIT_X, _UNIT_Y, _UNIT_Z = nm.vec3(1, 0, 0), nm.vec3(0, 1, 0), nm.vec3(0, 0, 1)
_White = Colour.White


def _rows(*parts) -> np.ndarray:
    return np.concatenate([np.atleast_1d(np.asarray(p, dtype=np.float32)) for p in parts]).astype(np.float32)


# =========================================================================================================
# Textures (Impl/Textures)
# =========================================================================================================
class SolidColourTexture(Texture):
    """SolidColourTexture.cs:8-18"""

    def __init__(self, colour: Colour):
        self.Colour = colour

    def flatten(self, b):
        return b.add_texture(capi.TEX_SOLID, self.Colour.v)


class MarbleTexture(Texture):
    """MarbleTexture.cs:27-33 (defaults .15, 6, 3, 1/6)"""

    def __init__(self, scale=.15, noiseScale=6.0, noiseStrength=3.0, dropoffPower=None):
        self.Scale, self.NoiseScale, self.NoiseStrength = f32(scale), f32(noiseScale), f32(noiseStrength)
        self.DropoffPower = f32(1.0) / f32(6.0) if dropoffPower is None else f32(dropoffPower)

    def flatten(self, b):
        return b.add_texture(capi.TEX_MARBLE, [self.Scale, self.NoiseScale, self.NoiseStrength, self.DropoffPower])


class Perlin:
    """SharpNoise.Modules.Perlin with its defaults (Frequency 1, Lacunarity 2, Persistence .5, OctaveCount 6, Seed 0).  Quality is
    not modelled: the stand-in gradient noise interpolates linearly (NoiseQuality.Fast), as MarbleTexture's does."""

    def __init__(self, Frequency=1.0, Lacunarity=2.0, Persistence=0.5, OctaveCount=6, Seed=0):
        self.Frequency, self.Lacunarity, self.Persistence = f32(Frequency), f32(Lacunarity), f32(Persistence)
        self.OctaveCount, self.Seed = int(OctaveCount), int(Seed)


class GreyscaleNoiseTexture(Texture):
    """GreyscaleNoiseTexture.cs:9-20.  A Perlin module runs on the device; any other SharpNoise Module is opaque and rejected."""

    def __init__(self, module=None):
        self.Module = module

    def flatten(self, b):
        if not isinstance(self.Module, Perlin):
            raise NotSupportedError(f"GreyscaleNoiseTexture over {type(self.Module).__name__}: only a Perlin module can run on the device")
        m = self.Module
        return b.add_texture(capi.TEX_GREYSCALE_PERLIN, [m.Frequency, m.Lacunarity, m.Persistence, f32(m.OctaveCount), f32(m.Seed)])


class CustomFunctionTexture(Texture):
    """CustomFunctionTexture.cs:10-19 — wraps a delegate: opaque, rejected."""

    def __init__(self, func):
        self.Func = func


# =========================================================================================================
# Materials (Impl/Materials)
# =========================================================================================================
def _as_texture(x) -> Texture:
    return SolidColourTexture(x) if isinstance(x, Colour) else x


class StandardMaterial(Material):
    """StandardMaterial.cs:16-61.  StandardMaterial(albedo, diffusion) or StandardMaterial(albedo, emission, diffusion);
    albedo/emission may be Colours or Textures."""

    def __init__(self, albedo, emission_or_diffusion, diffusion=None):
        if diffusion is None:
            emission, diffusion = Colour.Black, emission_or_diffusion
        else:
            emission = emission_or_diffusion
        self.Albedo, self.Emission, self.Diffusion = _as_texture(albedo), _as_texture(emission), f32(diffusion)

    def flatten(self, b):
        return b.add_material(capi.MAT_STANDARD, tex=(b.texture(self.Albedo), b.texture(self.Emission)), f=[self.Diffusion])


class RefractiveMaterial(Material):
    """RefractiveMaterial.cs:20-34"""
    AirIndex, GlassIndex, DiamondIndex = 1.0, 1.5, 2.4

    def __init__(self, refractiveIndex, tint, alternateRefractionMode=False):
        self.RefractiveIndex, self.Tint, self.AlternateRefractionMode = f32(refractiveIndex), _as_texture(tint), bool(alternateRefractionMode)

    def _flags(self):
        return capi.MAT_FLAG_ALTERNATE_REFRACTION if self.AlternateRefractionMode else 0

    def flatten(self, b):
        return b.add_material(capi.MAT_REFRACTIVE, tex=(b.texture(self.Tint), 0), flags=self._flags(), f=[self.RefractiveIndex])


class EmissiveRefractiveMaterial(RefractiveMaterial):
    """EmissiveRefractiveMaterial.cs:30-40"""

    def __init__(self, refractiveIndex, tint, emission, directEmission=False, alternateRefractionMode=False):
        super().__init__(refractiveIndex, tint, alternateRefractionMode)
        self.Emission, self.DirectEmission = _as_texture(emission), bool(directEmission)

    def flatten(self, b):
        flags = self._flags() | (capi.MAT_FLAG_DIRECT_EMISSION if self.DirectEmission else 0)
        return b.add_material(capi.MAT_EMISSIVE_REFRACTIVE, tex=(b.texture(self.Tint), b.texture(self.Emission)), flags=flags, f=[self.RefractiveIndex])


class PhongMaterial(Material):
    """PhongMaterial.cs:11-23 (init-only properties with defaults)"""

    def __init__(self, SpecularColour=None, DiffuseColour=None, AmbientColour=None, Shininess=0.0):
        self.SpecularColour = Colour.White if SpecularColour is None else SpecularColour
        self.DiffuseColour = Colour.HalfGrey if DiffuseColour is None else DiffuseColour
        self.AmbientColour = Colour.White * f32(0.001) if AmbientColour is None else AmbientColour
        self.Shininess = f32(Shininess)

    def flatten(self, b):
        return b.add_material(capi.MAT_PHONG, f=_rows(self.SpecularColour.v, self.DiffuseColour.v, self.AmbientColour.v, self.Shininess))


class VolumetricMaterial(Material):
    """The private material of ConstantDensityMedium (ConstantDensityMedium.cs:87-100)."""

    def __init__(self, albedo: Colour, density):
        self.Albedo, self.Density = albedo, f32(density)

    def flatten(self, b):
        return b.add_material(capi.MAT_VOLUMETRIC, f=_rows(self.Albedo.v, self.Density))


def DefaultDiffuseMaterial() -> Material:
    """BuiltinMaterials.DefaultDiffuseMaterial (Impl/Builtin/BuiltinMaterials.cs:10): a NEW instance per get."""
    return StandardMaterial(Colour.HalfGrey, 1.0)


# =========================================================================================================
# Hittables (Impl/Hittables)
# =========================================================================================================
class SingleMaterialHittable(Hittable):
    """SingleMaterialHittable.cs:8-22"""

    def __init__(self, Material: Optional[Material] = None):
        self.Material = DefaultDiffuseMaterial() if Material is None else Material


class Sphere(SingleMaterialHittable):
    """Sphere.cs:17-22"""

    def __init__(self, centre, radius, Material=None):
        super().__init__(Material)
        self.Centre, self.Radius = nm.vec3(centre), f32(radius)
        self.BoundingVolume = AxisAlignedBoundingBox(self.Centre - self.Radius, self.Centre + self.Radius)

    def flatten(self, b):
        return capi.HIT_SPHERE, b.add_prim(capi.HIT_SPHERE, _rows(self.Centre, self.Radius))


def _plane_basis_columns(u, v, normal):
    """Quad.cs:52-54 / Disk.cs:38-40: invert the (u, v, normal) -> world matrix, return columns 1 and 2."""
    m = np.zeros((4, 4), dtype=np.float32)
    m[0, :3], m[1, :3], m[2, :3], m[3, 3] = u, v, normal, 1.0
    inv = nm.invert4(m)
    return inv[:, 0].copy(), inv[:, 1].copy()


class Quad(SingleMaterialHittable):
    """Quad.cs:37-55"""

    def __init__(self, origin, u, v, Material=None):
        super().__init__(Material)
        self.Origin, self.U, self.V = nm.vec3(origin), nm.vec3(u), nm.vec3(v)
        self.Normal = nm.normalize(nm.cross(self.U, self.V))
        self.BoundingVolume = AxisAlignedBoundingBox.Encompass(self.Origin, self.Origin + self.U, self.Origin + self.V,
                                                              (self.Origin + self.U) + self.V).WithPadding(.0001)
        try:
            self._col1, self._col2 = _plane_basis_columns(self.U, self.V, self.Normal)
        except ArithmeticError:
            raise ArithmeticError("Could not invert Quad to world transformation matrix (UV coords were probably parallel)")

    @staticmethod
    def CreateFromPoints(o, oPlusU, oPlusV, Material=None):     # Quad.cs:79
        o = nm.vec3(o)
        return Quad(o, nm.vec3(oPlusU) - o, nm.vec3(oPlusV) - o, Material)

    def flatten(self, b):
        return capi.HIT_QUAD, b.add_prim(capi.HIT_QUAD, _rows(self.Origin, self.Normal, self._col1, self._col2))


class Triangle(SingleMaterialHittable):
    """NEW primitive (the reference has no mesh hittable, SURVEY §8a row aT), built exactly like Quad with
    u = v1 - v0, v = v2 - v0: same plane test, barycentric inside test, normal never flipped, AABB padded 1e-4."""

    def __init__(self, v0, v1, v2, Material=None):
        super().__init__(Material)
        self.V0 = nm.vec3(v0)
        self.E1, self.E2 = nm.vec3(v1) - self.V0, nm.vec3(v2) - self.V0
        self.Normal = nm.normalize(nm.cross(self.E1, self.E2))
        self.BoundingVolume = AxisAlignedBoundingBox.Encompass(self.V0, self.V0 + self.E1, self.V0 + self.E2).WithPadding(.0001)
        try:
            self._col1, self._col2 = _plane_basis_columns(self.E1, self.E2, self.Normal)
        except ArithmeticError:
            raise ArithmeticError("Could not invert Triangle to world transformation matrix (degenerate triangle)")

    def flatten(self, b):
        return capi.HIT_TRIANGLE, b.add_prim(capi.HIT_TRIANGLE, _rows(self.V0, self.E1, self.E2, self.Normal, self._col1, self._col2))


class InfinitePlane(SingleMaterialHittable):
    """InfinitePlane.cs:20-28 — the normal is stored as given (not normalised); BoundingVolume is infinite."""

    def __init__(self, point, normal, Material=None):
        super().__init__(Material)
        self.Point, self.Normal = nm.vec3(point), nm.vec3(normal)
        self.negPointDotNormal = -nm.dot(self.Point, self.Normal)
        self.BoundingVolume = AxisAlignedBoundingBox.Infinite

    def flatten(self, b):
        return capi.HIT_INFINITE_PLANE, b.add_prim(capi.HIT_INFINITE_PLANE, _rows(self.Point, self.Normal, self.negPointDotNormal))


class XYPlane(SingleMaterialHittable):
    """XYPlane.cs:19-29"""

    def __init__(self, xLow, xHigh, yLow, yHigh, z, aabbPadding=0.001, Material=None):
        super().__init__(Material)
        self.XLow, self.XHigh, self.YLow, self.YHigh, self.Z, self.AABBPadding = (f32(x) for x in (xLow, xHigh, yLow, yHigh, z, aabbPadding))
        self.BoundingVolume = AxisAlignedBoundingBox(nm.vec3(self.XLow, self.YLow, self.Z - self.AABBPadding), nm.vec3(self.XHigh, self.YHigh, self.Z + self.AABBPadding))

    def flatten(self, b):
        return capi.HIT_XY_PLANE, b.add_prim(capi.HIT_XY_PLANE, _rows(self.XLow, self.XHigh, self.YLow, self.YHigh, self.Z))


class XZPlane(SingleMaterialHittable):
    """XZPlane.cs:19-29"""

    def __init__(self, xLow, xHigh, zLow, zHigh, y, aabbPadding=0.001, Material=None):
        super().__init__(Material)
        self.XLow, self.XHigh, self.ZLow, self.ZHigh, self.Y, self.AABBPadding = (f32(x) for x in (xLow, xHigh, zLow, zHigh, y, aabbPadding))
        self.BoundingVolume = AxisAlignedBoundingBox(nm.vec3(self.XLow, self.Y - self.AABBPadding, self.ZLow), nm.vec3(self.XHigh, self.Y + self.AABBPadding, self.ZHigh))

    def flatten(self, b):
        return capi.HIT_XZ_PLANE, b.add_prim(capi.HIT_XZ_PLANE, _rows(self.XLow, self.XHigh, self.ZLow, self.ZHigh, self.Y))


class YZPlane(SingleMaterialHittable):
    """YZPlane.cs:17-26"""

    def __init__(self, yLow, yHigh, zLow, zHigh, x, aabbPadding=0.001, Material=None):
        super().__init__(Material)
        self.YLow, self.YHigh, self.ZLow, self.ZHigh, self.X, self.AABBPadding = (f32(v) for v in (yLow, yHigh, zLow, zHigh, x, aabbPadding))
        self.BoundingVolume = AxisAlignedBoundingBox(nm.vec3(self.X - self.AABBPadding, self.YLow, self.ZLow), nm.vec3(self.X + self.AABBPadding, self.YHigh, self.ZHigh))

    def flatten(self, b):
        return capi.HIT_YZ_PLANE, b.add_prim(capi.HIT_YZ_PLANE, _rows(self.YLow, self.YHigh, self.ZLow, self.ZHigh, self.X))


class Disk(SingleMaterialHittable):
    """Disk.cs:23-41"""

    def __init__(self, centre, normal, radius, Material=None):
        super().__init__(Material)
        self.Normal = nm.normalize(nm.vec3(normal))
        self.Centre, self.Radius = nm.vec3(centre), f32(radius)
        self.BoundingVolume = AxisAlignedBoundingBox(self.Centre - self.Radius, self.Centre + self.Radius)
        self.negCentreDotNorm = -nm.dot(self.Centre, self.Normal)
        randCrossDir = _UNIT_Y if abs(nm.dot(self.Normal, _UNIT_X)) < f32(0.01) else _UNIT_X
        u = nm.normalize(nm.cross(self.Normal, randCrossDir))
        v = nm.normalize(nm.cross(self.Normal, u))
        try:
            self._col1, self._col2 = _plane_basis_columns(u, v, self.Normal)
        except ArithmeticError:
            raise ArithmeticError("Could not invert Disk to world transformation matrix (UV coords were probably parallel)")

    def flatten(self, b):
        return capi.HIT_DISK, b.add_prim(capi.HIT_DISK, _rows(self.Centre, self.Normal, self.Radius, self.negCentreDotNorm, self._col1, self._col2))


class Box(SingleMaterialHittable):
    """Box.cs:35-93.  Box(boxToWorldTransform) or Box(corner1, corner2)."""

    def __init__(self, a, b=None, Material=None):
        super().__init__(Material)
        m = np.asarray(a, dtype=np.float32) if b is None else Box._TransformMatrixFromCorners(nm.vec3(a), nm.vec3(b))
        if m.shape != (4, 4):
            raise ValueError("Box expects a 4x4 matrix or two corners")
        self.BoxToWorldTransform = m
        try:
            self.WorldToBoxTransform = nm.invert4(m)
        except ArithmeticError:
            raise ArithmeticError("Could not invert Box=>World transform matrix")
        corners = [nm.transform_point(nm.vec3(x, y, z), m) for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)]     # :45-57 (sic: +-1, the solid is +-0.5)
        self.BoundingVolume = AxisAlignedBoundingBox.Encompass(*corners)

    @staticmethod
    def _TransformMatrixFromCorners(c1, c2):                   # :84-93 (sic: c2 = Max(c1', c2))
        c1 = np.where(c1 < c2, c1, c2).astype(np.float32)
        c2 = np.where(c1 > c2, c1, c2).astype(np.float32)
        size = c2 - c1
        centre = (c2 + c1) / f32(2.0)
        return nm.matmul4(nm.create_scale(*size), nm.create_translation(*centre))

    def flatten(self, b):
        w = self.WorldToBoxTransform
        return capi.HIT_BOX, b.add_prim(capi.HIT_BOX, _rows(w[:, :3].reshape(-1), self.BoxToWorldTransform[:3, :3].reshape(-1)))


class Capsule(SingleMaterialHittable):
    """Capsule.cs:35-55"""

    def __init__(self, p1, p2, radius, Material=None):
        super().__init__(Material)
        self.P1, self.P2, self.Radius = nm.vec3(p1), nm.vec3(p2), f32(radius)
        mn, mx = np.minimum(self.P1, self.P2), np.maximum(self.P1, self.P2)
        self.BoundingVolume = AxisAlignedBoundingBox(mn - self.Radius, mx + self.Radius)
        w = nm.normalize(self.P2 - self.P1)
        u = nm.normalize(nm.cross(w, _UNIT_Z))
        v = nm.normalize(nm.cross(u, w))
        self.uvMatrix = np.array([[u[0], v[0], w[0]], [u[1], v[1], w[1]], [u[2], v[2], w[2]]], dtype=np.float32)   # :48

    def flatten(self, b):
        return capi.HIT_CAPSULE, b.add_prim(capi.HIT_CAPSULE, _rows(self.P1, self.P2, self.Radius, self.uvMatrix.reshape(-1)))


class Cylinder(SingleMaterialHittable):
    """Cylinder.cs:28-40"""

    def __init__(self, p1, p2, radius, Material=None):
        super().__init__(Material)
        self.P1, self.P2, self.Radius = nm.vec3(p1), nm.vec3(p2), f32(radius)
        mn, mx = np.minimum(self.P1, self.P2), np.maximum(self.P1, self.P2)
        self.BoundingVolume = AxisAlignedBoundingBox(mn - self.Radius, mx + self.Radius)

    def flatten(self, b):
        return capi.HIT_CYLINDER, b.add_prim(capi.HIT_CYLINDER, _rows(self.P1, self.P2, self.Radius))


class ConstantDensityMedium(Hittable):
    """ConstantDensityMedium.cs:23-30,80"""

    def __init__(self, boundary: Hittable, density, colour: Colour):
        if isinstance(boundary, ConstantDensityMedium):
            raise ValueError("Cannot create a constant density volume using another volume")
        self.Boundary, self.Density = boundary, f32(density)
        self.negInvDensity = f32(-1.0) / self.Density
        self.Material = VolumetricMaterial(colour, density)

    @property
    def BoundingVolume(self):
        return self.Boundary.BoundingVolume

    def flatten(self, b):
        btype, bindex = self.Boundary.flatten(b)
        return capi.HIT_MEDIUM, b.add_medium(btype, bindex, self.Density, self.negInvDensity)


# ---- bulk extensions (no reference counterpart): many primitives of one kind as numpy arrays ----------------
class SphereArray:
    """N spheres; materials[material_index[i]] is sphere i's Material.  Equivalent to N SceneObjects of Sphere."""

    def __init__(self, centres, radii, materials: Sequence[Material], material_index):
        self.centres = np.ascontiguousarray(centres, dtype=np.float32).reshape(-1, 3)
        self.radii = np.ascontiguousarray(radii, dtype=np.float32).reshape(-1)
        self.materials = list(materials)
        self.material_index = np.ascontiguousarray(material_index, dtype=np.uint32).reshape(-1)
        if not (len(self.centres) == len(self.radii) == len(self.material_index)):
            raise ValueError("SphereArray: array lengths differ")
        if len(self.material_index) and int(self.material_index.max()) >= len(self.materials):
            raise ValueError("SphereArray: material index out of range")

    def __len__(self):
        return len(self.radii)

    def flatten_bulk(self, b):
        mats = np.array([b.material(m) for m in self.materials], dtype=np.uint32)
        r = self.radii[:, None]
        rows = np.concatenate([self.centres, r], axis=1)
        b.add_bulk(capi.HIT_SPHERE, rows, mats[self.material_index] if len(mats) else self.material_index, self.centres - r, self.centres + r)


class TriangleArray:
    """N triangles (v0, v1, v2) sharing one Material.  Equivalent to N SceneObjects of Triangle."""

    def __init__(self, v0, v1, v2, Material: Optional[Material] = None):
        self.v0 = np.ascontiguousarray(v0, dtype=np.float32).reshape(-1, 3)
        self.v1 = np.ascontiguousarray(v1, dtype=np.float32).reshape(-1, 3)
        self.v2 = np.ascontiguousarray(v2, dtype=np.float32).reshape(-1, 3)
        self.Material = DefaultDiffuseMaterial() if Material is None else Material

    def __len__(self):
        return len(self.v0)

    def flatten_bulk(self, b):
        n = len(self.v0)
        e1, e2 = self.v1 - self.v0, self.v2 - self.v0
        cr = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2], e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], axis=1)
        ln = np.sqrt((cr[:, 0] * cr[:, 0] + cr[:, 1] * cr[:, 1]) + cr[:, 2] * cr[:, 2], dtype=np.float32)
        nrm = (cr / ln[:, None]).astype(np.float32)
        m = np.zeros((n, 4, 4), dtype=np.float32)
        m[:, 0, :3], m[:, 1, :3], m[:, 2, :3], m[:, 3, 3] = e1, e2, nrm, 1.0
        inv = nm.invert4(m)
        rows = np.concatenate([self.v0, e1, e2, nrm, inv[:, :, 0], inv[:, :, 1]], axis=1).astype(np.float32)
        pts = np.stack([self.v0, self.v0 + e1, self.v0 + e2])
        pad = f32(.0001)
        mat = np.full(n, b.material(self.Material), dtype=np.uint32)
        b.add_bulk(capi.HIT_TRIANGLE, rows, mat, pts.min(axis=0) - pad, pts.max(axis=0) + pad)


# =========================================================================================================
# Lights (Impl/Lights)
# =========================================================================================================
class DistanceAttenuation:
    """SimpleLightBase.DistanceAttenuationDelegate restricted to the factory families of SimpleLightBase.cs:83-129
    plus a constant (the Demo scene's `(_, _) => 1f`, BuiltinScenes.cs:146).  An arbitrary callable cannot be
    flattened and is rejected."""

    def __init__(self, kind, p=(0.0, 0.0, 0.0)):
        self.kind, self.p = kind, tuple(float(x) for x in p) + (0.0,) * (3 - len(p))

    @staticmethod
    def Constant(value=1.0): return DistanceAttenuation(capi.ATTEN_CONSTANT, (value,))
    @staticmethod
    def Linear(): return DistanceAttenuation(capi.ATTEN_LINEAR)
    @staticmethod
    def Power(power, stayHighInitially=False): return DistanceAttenuation(capi.ATTEN_POWER_STAY_HIGH if stayHighInitially else capi.ATTEN_POWER, (power,))
    @staticmethod
    def LogisticsCurve(midpoint=.5, steepness=16.0): return DistanceAttenuation(capi.ATTEN_LOGISTIC, (midpoint, steepness))
    @staticmethod
    def LogisticsCurveLKX(l, k, x0): return DistanceAttenuation(capi.ATTEN_LOGISTIC_LKX, (l, k, x0))
    @staticmethod
    def ExponentialDecay(a): return DistanceAttenuation(capi.ATTEN_EXP_DECAY, (a,))


class SimpleLightBase(Light):
    """SimpleLightBase.cs:15-35,132 (init-only properties and their defaults)"""
    _type = None

    def __init__(self, Position=(0, 0, 0), Colour=None, AttenuationRadius=1.0, CutoffRadius=math.inf, DistanceAttenuationFunc=None):
        self.Position = nm.vec3(Position)
        self.Colour = _White if Colour is None else Colour
        self.AttenuationRadius, self.CutoffRadius = f32(AttenuationRadius), f32(CutoffRadius)
        self.DistanceAttenuationFunc = DistanceAttenuation.ExponentialDecay(5) if DistanceAttenuationFunc is None else DistanceAttenuationFunc

    def _row(self, b, diffusion_radius=0.0, shape=None):
        a = self.DistanceAttenuationFunc
        if not isinstance(a, DistanceAttenuation):
            raise NotSupportedError("DistanceAttenuationFunc is an arbitrary delegate; only the SimpleLightBase factory families can run on the device")
        shape_type = shape_index = 0
        mn = mx = np.zeros(3, np.float32)
        if shape is not None:
            shape_type, shape_index = shape.flatten(b)
            mn, mx = shape.BoundingVolume.Min, shape.BoundingVolume.Max
        return b.add_light(self._type, self.Position, self.Colour.v, self.AttenuationRadius, self.CutoffRadius, diffusion_radius, a.kind, a.p,
                           shape_type, shape_index, mn, mx)


class PointLight(SimpleLightBase):
    """PointLight.cs:6-10"""
    _type = capi.LIGHT_POINT

    def flatten(self, b):
        return self._row(b)


class DiffuseSphereLight(SimpleLightBase):
    """DiffuseSphereLight.cs:10-13"""
    _type = capi.LIGHT_DIFFUSE_SPHERE

    def __init__(self, DiffusionRadius=0.0, **kw):
        super().__init__(**kw)
        self.DiffusionRadius = f32(DiffusionRadius)

    def flatten(self, b):
        return self._row(b, self.DiffusionRadius)


class DiffuseShapedLight(SimpleLightBase):
    """DiffuseShapedLight.cs:9-15"""
    _type = capi.LIGHT_DIFFUSE_SHAPED

    def __init__(self, shape: Hittable, **kw):
        super().__init__(**kw)
        if isinstance(shape, ConstantDensityMedium):
            raise NotSupportedError("DiffuseShapedLight over a ConstantDensityMedium is not supported on the device")
        self.Shape = shape

    def flatten(self, b):
        return self._row(b, 0.0, self.Shape)


# =========================================================================================================
# SkyBoxes (Impl/Skyboxes)
# =========================================================================================================
class DefaultSkyBox(SkyBox):
    """DefaultSkyBox.cs:8-16"""

    def flatten(self, b):
        return b.set_skybox(capi.SKY_DEFAULT, (0, 0, 0))


class SingleColourSkyBox(SkyBox):
    """SingleColourSkyBox.cs:8-21"""

    def __init__(self, colour: Colour):
        self.Colour = colour

    def flatten(self, b):
        return b.set_skybox(capi.SKY_SINGLE_COLOUR, self.Colour.v)
