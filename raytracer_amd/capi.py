"""ctypes binding of include/rt_hip.h.  The structures below must stay field-for-field identical to the header
(tests/test_abi.py checks sizes and exported symbols)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")     # RT_HIP_LIB: A/B a differently built library

RT_ABI_VERSION = 5
RT_MAX_DEVICES = 16
RENDER_EXACT_LIGHT_SAMPLING = 1
RENDER_EXACT_OCCLUSION = 2
RENDER_VALIDATE = 4
RT_NO_HIT = 0xFFFFFFFF
RT_DEPTH_HIST_SIZE = 64

# rt_hittable_type
(HIT_SPHERE, HIT_QUAD, HIT_TRIANGLE, HIT_INFINITE_PLANE, HIT_XY_PLANE, HIT_XZ_PLANE, HIT_YZ_PLANE, HIT_DISK, HIT_BOX, HIT_CAPSULE,
 HIT_CYLINDER, HIT_MEDIUM) = range(12)
TEX_SOLID, TEX_MARBLE, TEX_GREYSCALE_PERLIN = 0, 1, 2
MAT_STANDARD, MAT_REFRACTIVE, MAT_EMISSIVE_REFRACTIVE, MAT_PHONG, MAT_VOLUMETRIC = range(5)
MAT_FLAG_ALTERNATE_REFRACTION, MAT_FLAG_DIRECT_EMISSION = 1, 2
LIGHT_POINT, LIGHT_DIFFUSE_SPHERE, LIGHT_DIFFUSE_SHAPED = range(3)
(ATTEN_CONSTANT, ATTEN_LINEAR, ATTEN_POWER, ATTEN_POWER_STAY_HIGH, ATTEN_LOGISTIC, ATTEN_LOGISTIC_LKX, ATTEN_EXP_DECAY) = range(7)
SKY_DEFAULT, SKY_SINGLE_COLOUR = 0, 1

F3 = C.c_float * 3
F4 = C.c_float * 4


class Sphere(C.Structure):
    _fields_ = [("centre", F3), ("radius", C.c_float)]


class Quad(C.Structure):
    _fields_ = [("origin", F3), ("normal", F3), ("inv_col1", F4), ("inv_col2", F4)]


class Triangle(C.Structure):
    _fields_ = [("v0", F3), ("e1", F3), ("e2", F3), ("normal", F3), ("inv_col1", F4), ("inv_col2", F4)]


class InfinitePlane(C.Structure):
    _fields_ = [("point", F3), ("normal", F3), ("neg_point_dot_normal", C.c_float)]


class AxisPlane(C.Structure):
    _fields_ = [("a_lo", C.c_float), ("a_hi", C.c_float), ("b_lo", C.c_float), ("b_hi", C.c_float), ("c", C.c_float)]


class Disk(C.Structure):
    _fields_ = [("centre", F3), ("normal", F3), ("radius", C.c_float), ("neg_centre_dot_norm", C.c_float), ("inv_col1", F4), ("inv_col2", F4)]


class Box(C.Structure):
    _fields_ = [("world_to_box", C.c_float * 12), ("box_to_world_rows", C.c_float * 9)]


class Capsule(C.Structure):
    _fields_ = [("p1", F3), ("p2", F3), ("radius", C.c_float), ("uv_matrix", C.c_float * 9)]


class Cylinder(C.Structure):
    _fields_ = [("p1", F3), ("p2", F3), ("radius", C.c_float)]


class Medium(C.Structure):
    _fields_ = [("boundary_type", C.c_uint32), ("boundary_index", C.c_uint32), ("density", C.c_float), ("neg_inv_density", C.c_float)]


class Object(C.Structure):
    _fields_ = [("type", C.c_uint32), ("index", C.c_uint32), ("material", C.c_uint32), ("aabb_min", F3), ("aabb_max", F3)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("f", C.c_float * 7)]


class Material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("tex", C.c_uint32 * 2), ("flags", C.c_uint32), ("f", C.c_float * 10)]


class Light(C.Structure):
    _fields_ = [("type", C.c_uint32), ("position", F3), ("colour", F3), ("attenuation_radius", C.c_float), ("cutoff_radius", C.c_float),
                ("diffusion_radius", C.c_float), ("atten_kind", C.c_uint32), ("atten_p", F3), ("shape_type", C.c_uint32),
                ("shape_index", C.c_uint32), ("shape_aabb_min", F3), ("shape_aabb_max", F3)]


class SkyBox(C.Structure):
    _fields_ = [("type", C.c_uint32), ("colour", F3)]


class Camera(C.Structure):
    _fields_ = [("look_from", F3), ("horizontal", F3), ("vertical", F3), ("lower_left_corner", F3), ("u", F3), ("v", F3), ("lens_radius", C.c_float)]


def _tbl(name, typ):
    return [("n_" + name, C.c_uint32), (name, C.POINTER(typ))]


class SceneDesc(C.Structure):
    _fields_ = ([("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("camera", Camera), ("skybox", SkyBox)]
                + _tbl("objects", Object) + _tbl("spheres", Sphere) + _tbl("quads", Quad) + _tbl("triangles", Triangle)
                + _tbl("infinite_planes", InfinitePlane) + _tbl("xy_planes", AxisPlane) + _tbl("xz_planes", AxisPlane) + _tbl("yz_planes", AxisPlane)
                + _tbl("disks", Disk) + _tbl("boxes", Box) + _tbl("capsules", Capsule) + _tbl("cylinders", Cylinder) + _tbl("media", Medium)
                + _tbl("materials", Material) + _tbl("textures", Texture) + _tbl("lights", Light))


class RenderOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("k_min", C.c_float), ("k_max", C.c_float),
                ("max_bounce_depth", C.c_uint32), ("light_sample_count_hint", C.c_uint32), ("debug_visualisation", C.c_uint32),
                ("seed", C.c_uint64), ("passes_per_batch", C.c_uint32), ("device", C.c_int32), ("stream", C.c_void_p),
                ("accum_device", C.c_void_p), ("flags", C.c_uint32), ("max_passes_per_call", C.c_uint32), ("n_devices", C.c_uint32),
                ("device_ids", C.c_int32 * RT_MAX_DEVICES)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("pixels_rendered", "passes_rendered", "material_scatter_count", "material_absorbed_count", "aabb_misses",
                                          "hittable_misses", "hittable_intersections", "sky_rays", "bounce_limit_exceeded", "ray_count",
                                          "shadow_rays", "node_visits", "prim_tests", "shadow_node_visits", "shadow_prim_tests",
                                          "normals_wrong_magnitude", "uv_invalid")] + \
               [("ray_depth_counts", C.c_uint64 * RT_DEPTH_HIST_SIZE)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_[:-1]}
        d["ray_depth_counts"] = [int(x) for x in self.ray_depth_counts]
        return d


RT_K_NAMES = ("raygen", "closest", "shade", "shadow", "accumulate", "resolve")


class KernelProfile(C.Structure):
    _fields_ = [("launches", C.c_uint64 * 6), ("total_ms", C.c_double * 6)]

    def as_dict(self):
        return {n: {"launches": int(self.launches[i]), "total_ms": float(self.total_ms[i])} for i, n in enumerate(RT_K_NAMES)}


class SceneInfo(C.Structure):
    _fields_ = [("n_objects", C.c_uint32), ("n_unbounded", C.c_uint32), ("n_nodes", C.c_uint32), ("bvh_depth", C.c_uint32),
                ("tree_in_lds", C.c_uint32), ("lds_image_bytes", C.c_uint32), ("built_on_device", C.c_uint32), ("object_list", C.c_uint32),
                ("variant", C.c_char * 16), ("wide_nodes", C.c_uint32), ("wide_levels", C.c_uint32), ("device", C.c_int32), ("reserved", C.c_uint32)]


class Ray(C.Structure):
    _fields_ = [("origin", F3), ("direction", F3)]


class Hit(C.Structure):
    _fields_ = [("k", C.c_float), ("object", C.c_uint32), ("material", C.c_uint32), ("normal", F3), ("uv", C.c_float * 2),
                ("world_point", F3), ("outside_face", C.c_uint32)]


# every symbol include/rt_hip.h declares
SYMBOLS = {
    "rt_version": (C.c_int, []),
    "rt_last_error": (C.c_char_p, []),
    "rt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rt_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (C.c_int, [C.c_void_p]),
    "rt_scene_get_info": (C.c_int, [C.c_void_p, C.POINTER(SceneInfo)]),
    "rt_render_create": (C.c_int, [C.c_void_p, C.POINTER(RenderOpts), C.POINTER(C.c_void_p)]),
    "rt_render_destroy": (C.c_int, [C.c_void_p]),
    "rt_render_passes": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]),
    "rt_render_read_rgb24": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_render_read_accum": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "rt_render_resolve_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rt_render_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "rt_render_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_render_kernel_profile": (C.c_int, [C.c_void_p, C.POINTER(KernelProfile)]),
    "rt_render_validator_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "rt_trace_closest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_uint64, C.c_void_p]),
    "rt_trace_any": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "rt_trace_closest_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_uint64, C.c_void_p]),
}
RT_TRACE_NOALLOC_MAX_RAYS = 1 << 17


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt_hip error {code}: {message}")
        self.code = code


_lib = None


def load():
    """Load librt_hip.so.  There is no fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                if os.environ.get("RT_HIP_LIB"):
                    continue          # an older build under A/B test may lack newer entry points
                raise
            fn.restype = res
            fn.argtypes = args
        if lib.rt_version() != RT_ABI_VERSION:
            raise ImportError(f"librt_hip.so has ABI {lib.rt_version()}, binding expects {RT_ABI_VERSION}")
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise RtError(rc, load().rt_last_error().decode("utf-8", "replace"))
