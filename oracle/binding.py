"""ctypes binding of oracle/liboracle.so (oracle.cpp).  TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the raytracer_amd package."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from raytracer_amd import capi
from raytracer_amd.flatten import FlatScene, flatten_scene
from raytracer_amd.render_job import HIT_DTYPE, _pack_rays, make_opts

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make` at the repo root")
        lib = C.CDLL(LIB_PATH)
        lib.orc_last_error.restype = C.c_char_p
        lib.orc_scene_create.argtypes = [C.POINTER(capi.SceneDesc), C.POINTER(C.c_void_p)]
        lib.orc_scene_destroy.argtypes = [C.c_void_p]
        lib.orc_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_uint64, C.c_void_p]
        lib.orc_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        lib.orc_trace_any_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
        lib.orc_trace_any_culled.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
        lib.orc_scene_set_culled_any.argtypes = [C.c_void_p, C.c_int]
        lib.orc_render.argtypes = [C.c_void_p, C.POINTER(capi.RenderOpts), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(capi.Stats)]
        lib.orc_render_validated.argtypes = [C.c_void_p, C.POINTER(capi.RenderOpts), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(capi.Stats), C.c_void_p, C.c_uint32]
        lib.orc_resolve_rgb24.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.orc_camera_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(capi.Camera)]
        lib.orc_sphere_uv.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_pixel_centre_sky_rgb24.argtypes = [C.POINTER(capi.Camera), C.POINTER(capi.SkyBox), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}: {load().orc_last_error().decode()}")


class OracleScene:
    def __init__(self, scene_or_flat):
        self.flat: FlatScene = scene_or_flat if isinstance(scene_or_flat, FlatScene) else flatten_scene(scene_or_flat)
        self.handle = C.c_void_p()
        _check(load().orc_scene_create(C.byref(self.flat.desc), C.byref(self.handle)))

    def trace_closest(self, origins, directions, kMin=0.001, kMax=np.inf, seed=0):
        rays = _pack_rays(origins, directions)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        _check(load().orc_trace_closest(self.handle, rays.ctypes.data, len(rays), kMin, kMax, seed, hits.ctypes.data))
        return hits

    def trace_any(self, origins, directions, kMin, kMax, threads=1, culled=False):
        """RenderJob.AnyIntersectionFast on a batch of rays.  culled=False: the reference's scan of every object (RenderJob.cs:575);
        culled=True: the same boolean through the checker's own box tree (oracle.cpp CullTree; tests prove the two equal)."""
        rays = _pack_rays(origins, directions)
        n = len(rays)
        k0 = np.ascontiguousarray(np.broadcast_to(np.asarray(kMin, np.float32), (n,)))
        k1 = np.ascontiguousarray(np.broadcast_to(np.asarray(kMax, np.float32), (n,)))
        out = np.zeros(n, dtype=np.uint8)
        fn = load().orc_trace_any_culled if culled else load().orc_trace_any_mt
        _check(fn(self.handle, rays.ctypes.data, k0.ctypes.data, k1.ctypes.data, n, out.ctypes.data, threads))
        return out.astype(bool)

    def set_culled_any(self, enable: bool):
        """Shadow rays of render() through CullTree instead of the every-object scan (full-size scenes only; same booleans)."""
        _check(load().orc_scene_set_culled_any(self.handle, 1 if enable else 0))

    def render(self, options, first_pass=0, n_passes=None, stride=1, threads=1, accum=None):
        """RenderInternal on the CPU: returns (rgb_sum (H, W, 3) float32 in image-row order, stats dict)."""
        W, H = options.RenderWidth, options.RenderHeight
        if accum is None:
            accum = np.zeros((H, W, 3), dtype=np.float32)
        ro = make_opts(options)
        st = capi.Stats()
        n = options.Passes if n_passes is None else n_passes
        _check(load().orc_render(self.handle, C.byref(ro), first_pass, n, stride, threads, accum.ctypes.data, C.byref(st)))
        return accum, st.as_dict()

    def render_validated(self, options, threads=1):
        """render() plus GraphicsValidator.Errors (GraphicsValidator.cs:35-45): (rgb_sum, stats, normals_wrong[n_objects], uv_invalid[n_objects])."""
        W, H = options.RenderWidth, options.RenderHeight
        accum = np.zeros((H, W, 3), dtype=np.float32)
        ro = make_opts(options)
        st = capi.Stats()
        per = np.zeros((self.flat.n_objects, 2), dtype=np.uint64)
        _check(load().orc_render_validated(self.handle, C.byref(ro), 0, options.Passes, 1, threads, accum.ctypes.data, C.byref(st), per.ctypes.data, self.flat.n_objects))
        return accum, st.as_dict(), per[:, 0].copy(), per[:, 1].copy()

    def close(self):
        if self.handle:
            load().orc_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def resolve_rgb24(rgb_sum: np.ndarray, sample_count: int) -> np.ndarray:
    H, W, _ = rgb_sum.shape
    src = np.ascontiguousarray(rgb_sum, dtype=np.float32)
    out = np.zeros((H, W, 3), dtype=np.uint8)
    _check(load().orc_resolve_rgb24(src.ctypes.data, sample_count, W, H, out.ctypes.data))
    return out


def camera_create(lookFrom, lookTowards, up, fov, aspect, lens, focus) -> capi.Camera:
    a, b, c = (np.asarray(x, dtype=np.float32) for x in (lookFrom, lookTowards, up))
    cam = capi.Camera()
    _check(load().orc_camera_create(a.ctypes.data, b.ctypes.data, c.ctypes.data, fov, aspect, lens, focus, C.byref(cam)))
    return cam


def sphere_uv(p):
    p = np.asarray(p, dtype=np.float32)
    uv = np.zeros(2, dtype=np.float32)
    load().orc_sphere_uv(p.ctypes.data, uv.ctypes.data)
    return uv


def pixel_centre_sky_rgb24(cam: capi.Camera, sky: capi.SkyBox, x, y_from_bottom, width, height):
    rgb = np.zeros(3, dtype=np.uint8)
    load().orc_pixel_centre_sky_rgb24(C.byref(cam), C.byref(sky), x, y_from_bottom, width, height, rgb.ctypes.data)
    return rgb
