"""raytracer_amd — MI355X (gfx950) backend for the per-sample trace loop of Ararem.RayTracer's RenderJob.

Layout: csrc/ (HIP kernels + the C ABI of include/rt_hip.h, built into librt_hip.so), capi (ctypes binding),
core / impl / builtin_scenes (host-side mirror of the reference's Scene API), flatten (DeviceSceneBuilder),
render_job (RenderJob + pass sharding).  There is no CPU trace path in this package.
"""
from .core import (ArgumentOutOfRangeError, AxisAlignedBoundingBox, Camera, Colour, GraphicsDebugVisualisation, Hittable, Light, Material, NotSupportedError,
                   RenderOptions, RenderStats, Scene, SceneObject, SkyBox, Texture)
from .render_job import DeviceScene, OperationCanceledError, RenderJob, shard_passes
from . import builtin_scenes as BuiltinScenes
from . import impl

__all__ = ["GraphicsDebugVisualisation", "ArgumentOutOfRangeError", "AxisAlignedBoundingBox", "Camera", "Colour", "Hittable", "Light", "Material", "NotSupportedError",
           "RenderOptions", "RenderStats", "Scene", "SceneObject", "SkyBox", "Texture", "DeviceScene", "OperationCanceledError", "RenderJob",
           "shard_passes", "BuiltinScenes", "impl"]
