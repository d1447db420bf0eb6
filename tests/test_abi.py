"""The C ABI: ctypes structures match include/rt_hip.h, and librt_hip.so exports every declared symbol.
No compute calls — this runs without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

from raytracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_hip.h")

STRUCTS = {
    "rt_sphere": capi.Sphere, "rt_quad": capi.Quad, "rt_triangle": capi.Triangle, "rt_infinite_plane": capi.InfinitePlane,
    "rt_axis_plane": capi.AxisPlane, "rt_disk": capi.Disk, "rt_box": capi.Box, "rt_capsule": capi.Capsule, "rt_cylinder": capi.Cylinder,
    "rt_medium": capi.Medium, "rt_object": capi.Object, "rt_texture": capi.Texture, "rt_material": capi.Material, "rt_light": capi.Light,
    "rt_skybox": capi.SkyBox, "rt_camera": capi.Camera, "rt_scene_desc": capi.SceneDesc, "rt_render_opts": capi.RenderOpts,
    "rt_stats": capi.Stats, "rt_ray": capi.Ray, "rt_hit": capi.Hit, "rt_scene_info": capi.SceneInfo, "rt_kernel_profile": capi.KernelProfile,
}


def test_struct_sizes_match_header():
    src = '#include <stdio.h>\n#include "rt_hip.h"\nint main(void){\n' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));\n' for n in STRUCTS) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c"); exe = os.path.join(td, "s")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe])   # the header is plain C
        out = subprocess.check_output([exe], text=True)
    sizes = dict(line.split() for line in out.strip().splitlines())
    for name, ct in STRUCTS.items():
        assert int(sizes[name]) == C.sizeof(ct), name


OPT_FIELDS = ["seed", "device", "stream", "accum_device", "flags", "max_passes_per_call", "n_devices", "device_ids"]


def test_field_offsets_of_scene_desc_and_opts():
    fields = ["camera", "skybox", "objects", "spheres", "materials", "lights"]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "rt_hip.h"\nint main(void){\n' + "".join(
        f'printf("{f} %zu\\n", offsetof(rt_scene_desc, {f}));\n' for f in fields) + "".join(
        f'printf("o_{f} %zu\\n", offsetof(rt_render_opts, {f}));\n' for f in OPT_FIELDS) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c"); exe = os.path.join(td, "s")
        open(c, "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = dict(l.split() for l in subprocess.check_output([exe], text=True).strip().splitlines())
    for f in fields:
        assert int(out[f]) == getattr(capi.SceneDesc, f).offset, f
    for f in OPT_FIELDS:
        assert int(out["o_" + f]) == getattr(capi.RenderOpts, f).offset, f


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_binding_covers_every_declared_symbol():
    assert declared_symbols() == sorted(capi.SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    assert os.path.exists(capi.LIB_PATH), "build raytracer_amd/librt_hip.so with `make` first"
    # load in a child process so a missing GPU runtime cannot poison this interpreter
    code = ("import ctypes,sys; lib=ctypes.CDLL(sys.argv[1]); "
            "[getattr(lib,s) for s in sys.argv[2:]]; lib.rt_version.restype=ctypes.c_int; print(lib.rt_version())")
    out = subprocess.check_output([sys.executable, "-c", code, capi.LIB_PATH] + declared_symbols(), text=True)
    assert int(out.strip()) == capi.RT_ABI_VERSION


def test_null_arguments_are_rejected_without_a_device():
    lib = capi.load()
    assert lib.rt_device_count(None) == -1
    assert b"null" in lib.rt_last_error()
    h = C.c_void_p()
    assert lib.rt_scene_create(None, 0, C.byref(h)) == -1
    d = capi.SceneDesc()          # struct_size 0 -> version mismatch, detected before any HIP call
    assert lib.rt_scene_create(C.byref(d), 0, C.byref(h)) == -1
    assert b"mismatch" in lib.rt_last_error()
    assert lib.rt_render_passes(None, 0, 1, 1) == -1
