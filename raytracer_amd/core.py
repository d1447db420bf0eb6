"""Host-side mirror of Ararem.RayTracer.Core's public data types (the plugin API the GPU path sits behind).

Names, argument meaning and error behaviour follow the reference (paths relative to the reference root,
Core/ = Ararem.RayTracer.Core/):

  Colour            Core/Colour.cs
  AxisAlignedBoundingBox  Core/Acceleration/AxisAlignedBoundingBox.cs:20,101-132
  Camera / Camera.Create  Core/Camera.cs:26-83
  Hittable / Material / Texture / Light / SkyBox   Core/Base Type Definitions/*.cs
  SceneObject / Scene     Core/SceneObject.cs:6-8, Core/Scene.cs:9
  RenderOptions           Core/RenderOptions.cs:9-166
  RenderStats             Core/RenderStats.cs:12-93

The abstract types carry no arithmetic here: the trace loop runs on the device.  What a subclass must provide
instead of TryHit/Scatter/... is `flatten(builder)` (the IDeviceFlattenable contract of INTEGRATION.md);
a subclass that does not is a user-defined type and is rejected with NotSupportedError, never emulated.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import numerics as nm

f32 = np.float32


class NotSupportedError(TypeError):
    """System.NotSupportedException: a scene-graph type the device kernels have no table for."""


class ArgumentOutOfRangeError(ValueError):
    """LibArarem ArgumentOutOfRangeException<T> as thrown by RenderOptions' setters."""


# ---------------------------------------------------------------------------------------------------------
# Colour (Core/Colour.cs).  A float32 RGB triple; operators are lane-wise, scalars are splatted (:162,:172).
# ---------------------------------------------------------------------------------------------------------
class Colour:
    __slots__ = ("v",)

    def __init__(self, r, g=None, b=None):
        if g is None:
            a = np.asarray(r, dtype=np.float32)
            self.v = np.full(3, a, dtype=np.float32) if a.ndim == 0 else a.astype(np.float32).copy()
        else:
            self.v = np.array([r, g, b], dtype=np.float32)

    R = property(lambda s: s.v[0])
    G = property(lambda s: s.v[1])
    B = property(lambda s: s.v[2])

    @staticmethod
    def _v(o):
        return o.v if isinstance(o, Colour) else np.full(3, o, dtype=np.float32)

    def __add__(self, o): return Colour(self.v + Colour._v(o))
    def __sub__(self, o): return Colour(self.v - Colour._v(o))
    def __mul__(self, o): return Colour(self.v * Colour._v(o))
    __rmul__ = __mul__
    def __truediv__(self, o): return Colour(self.v / Colour._v(o))
    def __eq__(self, o): return isinstance(o, Colour) and bool(np.array_equal(self.v, o.v))
    def __hash__(self): return hash(self.v.tobytes())
    def __repr__(self): return f"Colour({self.v[0]!r}, {self.v[1]!r}, {self.v[2]!r})"

    @staticmethod
    def Lerp(a: "Colour", b: "Colour", amount) -> "Colour":        # Colour.cs:214
        amount = f32(amount)
        return (a * (f32(1.0) - amount)) + (b * amount)


Colour.White = Colour(1, 1, 1)
Colour.Black = Colour(0, 0, 0)
Colour.Red = Colour(1, 0, 0)
Colour.Green = Colour(0, 1, 0)
Colour.Blue = Colour(0, 0, 1)
Colour.HalfGrey = Colour(0.5)
Colour.Purple = Colour(1, 0, 1)
Colour.Yellow = Colour(1, 1, 0)
Colour.Orange = Colour(1, 0.5, 0)
Colour.Aqua = Colour(0, 1, 1)


# ---------------------------------------------------------------------------------------------------------
# AxisAlignedBoundingBox
# ---------------------------------------------------------------------------------------------------------
@dataclass
class AxisAlignedBoundingBox:
    Min: np.ndarray
    Max: np.ndarray

    def __post_init__(self):
        self.Min = nm.vec3(self.Min)
        self.Max = nm.vec3(self.Max)

    @staticmethod
    def Encompass(*points) -> "AxisAlignedBoundingBox":          # AxisAlignedBoundingBox.cs:101-111
        mn = np.full(3, np.inf, dtype=np.float32)
        mx = np.full(3, -np.inf, dtype=np.float32)
        for p in points:
            p = nm.vec3(p)
            mn = np.where(mn < p, mn, p)       # Vector3.Min lane-wise
            mx = np.where(mx > p, mx, p)
        return AxisAlignedBoundingBox(mn, mx)

    def WithPadding(self, padding) -> "AxisAlignedBoundingBox":  # :116
        p = f32(padding)
        return AxisAlignedBoundingBox(self.Min - p, self.Max + p)


AxisAlignedBoundingBox.Infinite = AxisAlignedBoundingBox(np.full(3, -np.inf, np.float32), np.full(3, np.inf, np.float32))  # :20


# ---------------------------------------------------------------------------------------------------------
# Camera
# ---------------------------------------------------------------------------------------------------------
@dataclass
class Camera:
    LookFrom: np.ndarray
    LookTowards: np.ndarray
    LookDirection: np.ndarray
    UpVector: np.ndarray
    VerticalFov: float
    AspectRatio: float
    FocusDistance: float
    LensRadius: float
    Horizontal: np.ndarray
    Vertical: np.ndarray
    LowerLeftCorner: np.ndarray
    U: np.ndarray
    V: np.ndarray

    @staticmethod
    def Create(lookFrom, lookTowards, upVector, verticalFov, aspectRatio, lensRadius, focusDistance) -> "Camera":
        """Camera.Create, Core/Camera.cs:61-83, one binary32 rounding per operator."""
        lookFrom, lookTowards = nm.vec3(lookFrom), nm.vec3(lookTowards)
        upVector = nm.normalize(nm.vec3(upVector))
        verticalFov, aspectRatio, lensRadius, focusDistance = f32(verticalFov), f32(aspectRatio), f32(lensRadius), f32(focusDistance)
        theta = f32(verticalFov * f32(nm.PI / f32(180.0)))
        h = f32(np.tan(f32(theta / f32(2.0)), dtype=np.float32))
        viewportHeight = f32(f32(2.0) * h)
        viewportWidth = f32(aspectRatio * viewportHeight)
        lookDir = nm.normalize(lookFrom - lookTowards)
        cr = nm.cross(upVector, lookDir)
        if not np.any(cr != 0):
            raise ArithmeticError("Camera cannot point in the same direction as its 'up' vector (Cross(CameraUpVector, LookDirection) must != Zero)")
        u = nm.normalize(cr)
        v = nm.cross(lookDir, u)
        horizontal = ((viewportWidth * u).astype(np.float32) * focusDistance).astype(np.float32)
        vertical = ((viewportHeight * v).astype(np.float32) * focusDistance).astype(np.float32)
        llc = (((lookFrom - (horizontal / f32(2.0))) - (vertical / f32(2.0))) - (focusDistance * lookDir)).astype(np.float32)
        return Camera(lookFrom, lookTowards, lookDir, upVector, float(verticalFov), float(aspectRatio), float(focusDistance), float(lensRadius),
                      horizontal, vertical, llc, u, v)


# ---------------------------------------------------------------------------------------------------------
# The five abstract plugin types.  `flatten` is what a device-capable subclass implements.
# ---------------------------------------------------------------------------------------------------------
class _Flattenable:
    def flatten(self, builder):
        raise NotSupportedError(
            f"{type(self).__module__}.{type(self).__name__} is a user-defined {self._kind} with no device table; "
            "the HIP backend rejects it (there is no CPU fallback)")


class Hittable(_Flattenable):
    _kind = "Hittable"
    BoundingVolume: AxisAlignedBoundingBox


class Material(_Flattenable):
    _kind = "Material"


class Texture(_Flattenable):
    _kind = "Texture"


class Light(_Flattenable):
    _kind = "Light"


class SkyBox(_Flattenable):
    _kind = "SkyBox"


@dataclass
class SceneObject:                 # Core/SceneObject.cs:6-8
    Name: str
    Hittable: Hittable


@dataclass
class Scene:                       # Core/Scene.cs:9
    Name: str
    Camera: Camera
    SceneObjects: Sequence[SceneObject]
    Lights: Sequence[Light]
    SkyBox: SkyBox
    # Extension (no reference counterpart): bulk primitive arrays appended after SceneObjects, so that a
    # 5 M-triangle mesh does not need 5 M Python objects.  See impl.SphereArray / impl.TriangleArray.
    BulkObjects: Sequence[object] = field(default_factory=list)


# ---------------------------------------------------------------------------------------------------------
# RenderOptions (Core/RenderOptions.cs).  Validation and messages follow the property setters.
# ---------------------------------------------------------------------------------------------------------
class GraphicsDebugVisualisation:
    """Ararem.RayTracer.Core/Debugging/GraphicsDebugVisualisation.cs:4-61 (same values as rt_debug_vis)."""
    (None_, Normals, FaceDirection, DistanceFromCamera_Close, DistanceFromCamera_Mid, DistanceFromCamera_Far, UVCoords, PixelCoordDebugTexture,
     ScatterDirection, LocalCoordDebugTexture, WorldCoordDebugTexture, EstimatedLight, UndefinedTestVisualisation, BounceDepth) = range(14)


class RenderOptions:
    def __init__(self, RenderWidth=1920, RenderHeight=1080, KMin=0.001, KMax=math.inf, ConcurrencyLevel=None, Passes=100,
                 InfinitePasses=False, MaxBounceDepth=10, DebugVisualisation=0, LightSampleCountHint=2,
                 Seed=0x5EED, PassesPerBatch=0, ExactLightSampling=False, ExactOcclusion=False, Validate=False):
        if RenderWidth < 1:
            raise ArgumentOutOfRangeError("Image cannot have a width of less than 1 pixel")           # :52
        if RenderHeight < 1:
            raise ArgumentOutOfRangeError("Image cannot have a height of less than 1 pixel")          # :63
        if KMin < 0:
            raise ArgumentOutOfRangeError("KMin cannot have a value of less than 0")                  # :74
        if KMax < 0 or KMax < KMin:
            raise ArgumentOutOfRangeError("KMax cannot have a value of less than 0 or less than KMin")  # :86
        if ConcurrencyLevel is not None and ConcurrencyLevel < 1:
            raise ArgumentOutOfRangeError("Cannot have a concurrency level of less than 1")           # :100
        if LightSampleCountHint < 1:
            raise ArgumentOutOfRangeError("Must have at least 1 sample per light")                    # :160
        if Passes < 0 or MaxBounceDepth < 0:
            raise ArgumentOutOfRangeError("Passes / MaxBounceDepth are unsigned")
        self.RenderWidth, self.RenderHeight = int(RenderWidth), int(RenderHeight)
        self.KMin, self.KMax = float(KMin), float(KMax)
        self.ConcurrencyLevel = ConcurrencyLevel
        self.Passes, self.InfinitePasses = int(Passes), bool(InfinitePasses)
        self.MaxBounceDepth = int(MaxBounceDepth)
        self.DebugVisualisation = int(DebugVisualisation)
        self.LightSampleCountHint = int(LightSampleCountHint)
        # not in the reference: key of the counter RNG that replaces the unseeded Random.Shared (Core/RandUtils.cs:17)
        self.Seed = int(Seed)
        self.PassesPerBatch = int(PassesPerBatch)
        # not in the reference: light sampling in correctly rounded arithmetic (rt_hip.h RT_RENDER_EXACT_LIGHT_SAMPLING) instead of 1-ulp hardware forms
        self.ExactLightSampling = bool(ExactLightSampling)
        # not in the reference: shadow rays that answer as the scan of all objects does even at the horizon (rt_hip.h RT_RENDER_EXACT_OCCLUSION)
        self.ExactOcclusion = bool(ExactOcclusion)
        # not in the reference (which always validates, RenderJob.cs:588-607): also check every hit's UV and count both warn-only rules per object (rt_hip.h RT_RENDER_VALIDATE)
        self.Validate = bool(Validate)


@dataclass
class RenderStats:                 # Core/RenderStats.cs:12-93 (+ the counters the reference does not keep)
    TotalTruePixels: int = 0
    TotalPasses: int = 0
    PixelsRendered: int = 0
    PassesRendered: int = 0
    MaterialScatterCount: int = 0
    MaterialAbsorbedCount: int = 0
    AabbMisses: int = 0
    HittableMisses: int = 0
    HittableIntersections: int = 0
    SkyRays: int = 0
    BounceLimitExceeded: int = 0
    RayCount: int = 0
    ShadowRays: int = 0
    NodeVisits: int = 0
    PrimTests: int = 0
    ShadowNodeVisits: int = 0
    ShadowPrimTests: int = 0
    NormalsWrongMagnitude: int = 0   # GraphicsValidator.Errors[NormalsWrongMagnitude] summed over objects (RenderJob.cs:588-598)
    UVInvalid: int = 0               # ...[UVInvalid] (:600-607); counted in debug-visualisation renders
    RayDepthCounts: List[int] = field(default_factory=lambda: [0] * 64)
