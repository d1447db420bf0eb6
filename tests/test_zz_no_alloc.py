"""rt_trace_closest_noalloc (include/rt_hip.h): the closest-hit hook for callers that cannot allocate device memory.  Its ray and hit
staging lives in a fixed array inside the library instead of two hipMalloc'ed buffers; everything else is rt_trace_closest, so the
hit records must be identical byte for byte — on two calls in a row, the second smaller than the first (the array then still holds
the first call's tail), on a scene whose tree lives in LDS and on one whose tree lives in HBM, and beyond the limit it must refuse.
Named test_zz_* so that it is collected last."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import capi
from raytracer_amd.flatten import flatten_scene
from raytracer_amd.render_job import HIT_DTYPE, _pack_rays

import helpers

pytestmark = pytest.mark.gpu


def _noalloc(dev, o, d, kmin, kmax, seed):
    rays = _pack_rays(o, d)
    hits = np.zeros(len(rays), dtype=HIT_DTYPE)
    capi.check(capi.load().rt_trace_closest_noalloc(dev.handle, rays.ctypes.data, len(rays), kmin, kmax, seed, hits.ctypes.data))
    return hits


def _same(a, b):
    """Byte equality of the records that hit; a miss only defines `object` and `material` (rt_hip.h: k is undefined then)."""
    assert np.array_equal(a["object"], b["object"]) and np.array_equal(a["material"], b["material"])
    hit = a["object"] != 0xFFFFFFFF
    assert a[hit].tobytes() == b[hit].tobytes()
    return int(hit.sum())


@pytest.mark.parametrize("name", ["Testing", "Zoo", "MiniManyLights"])      # tree in LDS / object list, every primitive type / tree in HBM
def test_noalloc_equals_rt_trace_closest_on_two_calls_in_a_row(name):
    sc = helpers.small_scenes()[name](); flat = flatten_scene(sc); dev = rt.DeviceScene(flat)
    rs = np.random.default_rng(17)
    n1, n2 = capi.RT_TRACE_NOALLOC_MAX_RAYS, 4_321
    o1, d1 = helpers.camera_rays(sc, 64, 64, n1 // 2, rs); o2, d2 = helpers.secondary_rays(flat, n1 - n1 // 2, rs)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    first = _noalloc(dev, o, d, 0.001, np.inf, 5)                           # the largest call the entry point takes
    assert _same(first, dev.TraceClosest(o, d, 0.001, np.inf, seed=5)) > n1 // 8
    p = rs.permutation(n1)[:n2]                                             # other rays, fewer of them, another range and seed
    second = _noalloc(dev, o[p], d[p], 0.5, 9.0, 6)
    _same(second, dev.TraceClosest(o[p], d[p], 0.5, 9.0, seed=6))
    again = _noalloc(dev, o, d, 0.001, np.inf, 5)                           # and the first input once more: nothing of call two lingers
    assert again.tobytes() == first.tobytes()
    dev.Dispose()


def test_noalloc_limits_and_empty_input():
    sc = helpers.small_scenes()["Sphere"](); dev = rt.DeviceScene(sc)
    lib = capi.load()
    n = capi.RT_TRACE_NOALLOC_MAX_RAYS + 1
    rays = np.zeros((n, 6), np.float32); rays[:, 5] = -1
    hits = np.zeros(n, dtype=HIT_DTYPE)
    assert lib.rt_trace_closest_noalloc(dev.handle, rays.ctypes.data, n, 0.001, np.inf, 0, hits.ctypes.data) == -6       # RT_ERR_TOO_LARGE
    assert b"at most" in lib.rt_last_error()
    assert lib.rt_trace_closest_noalloc(dev.handle, None, 0, 0.001, np.inf, 0, None) == 0                               # no rays: nothing to do
    assert lib.rt_trace_closest_noalloc(dev.handle, None, 5, 0.001, np.inf, 0, hits.ctypes.data) == -1
    assert lib.rt_trace_closest_noalloc(None, rays.ctypes.data, 5, 0.001, np.inf, 0, hits.ctypes.data) == -1
    dev.Dispose()
