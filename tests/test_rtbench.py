"""tools/rtbench: the C++ driver of the C ABI (no Python, no PyTorch, no oracle in it) and the .rtscene file it loads."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import builtin_scenes as bs
from raytracer_amd import capi
from raytracer_amd.flatten import flatten_scene, save_rtscene, _TABLES

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTBENCH = os.path.join(ROOT, "tools", "rtbench")


def test_rtscene_file_holds_every_table(tmp_path):
    flat = flatten_scene(helpers.zoo_scene())
    path = tmp_path / "zoo.rtscene"
    save_rtscene(flat, str(path))
    raw = path.read_bytes()
    assert raw[:8] == b"RTSCENE1" and np.frombuffer(raw, np.uint32, 1, 8)[0] == capi.RT_ABI_VERSION
    off = 12 + C.sizeof(capi.Camera) + C.sizeof(capi.SkyBox)
    assert raw[12:12 + C.sizeof(capi.Camera)] == bytes(flat.desc.camera)
    for name in _TABLES:
        n, elem = np.frombuffer(raw, np.uint32, 2, off); off += 8
        ptr = getattr(flat.desc, name)
        assert n == getattr(flat.desc, "n_" + name) and elem == C.sizeof(ptr._type_)
        if n:
            assert raw[off:off + n * elem] == C.string_at(ptr, int(n) * int(elem))
        off += int(n) * int(elem)
    assert off == len(raw)


def test_rtbench_is_built_and_links_only_the_library():
    assert os.access(RTBENCH, os.X_OK), "make builds tools/rtbench"
    needed = subprocess.run(["ldd", RTBENCH], capture_output=True, text=True).stdout
    assert "librt_hip.so" in needed and "python" not in needed and "torch" not in needed and "oracle" not in needed


def test_rtbench_rejects_bad_files(tmp_path):
    bad = tmp_path / "bad.rtscene"; bad.write_bytes(b"RTSCENE1" + b"\0" * 10)
    r = subprocess.run([RTBENCH, str(bad)], capture_output=True, text=True)
    assert r.returncode == 2 and "truncated" in r.stderr
    assert subprocess.run([RTBENCH], capture_output=True).returncode == 2


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,passes,depth", [("Demo", 160, 90, 6, 8), ("Testing", 128, 72, 5, 10), ("Zoo", 96, 54, 4, 6)])
def test_rtbench_picture_equals_the_ctypes_hosts(tmp_path, name, w, h, passes, depth):
    import raytracer_amd as rt
    from raytracer_amd.core import RenderOptions
    sc = helpers.small_scenes()[name]()
    flat = flatten_scene(sc)
    path, ppm = tmp_path / "s.rtscene", tmp_path / "s.ppm"
    save_rtscene(flat, str(path))
    r = subprocess.run([RTBENCH, str(path), "--width", str(w), "--height", str(h), "--passes", str(passes), "--depth", str(depth), "--seed", "77",
                        "--passes-per-call", "2", "--ppm", str(ppm)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    job = rt.RenderJob(sc, RenderOptions(RenderWidth=w, RenderHeight=h, Passes=passes, MaxBounceDepth=depth, Seed=77)); job.Wait()
    raw = ppm.read_bytes()
    header = f"P6\n{w} {h}\n255\n".encode()
    assert raw.startswith(header)
    img = np.frombuffer(raw, np.uint8, offset=len(header)).reshape(h, w, 3)
    assert np.array_equal(img, job.Image)                      # same library, same keys: not a byte differs
    st = job.RenderStats
    assert out["pixels_rendered"] == w * h * passes == st.PixelsRendered and out["closest_rays"] == st.RayCount and out["shadow_rays"] == st.ShadowRays
    assert out["objects"] == flat.n_objects
    job.Dispose()
