"""DeviceSceneBuilder: flattens a Scene graph (core.Scene + impl.*) into the SoA tables of include/rt_hip.h.

This is the Python twin of the C# `DeviceSceneBuilder` described in INTEGRATION.md: materials and textures get
their ids by OBJECT IDENTITY (RefractiveMaterial.cs:51-52 and EmissiveRefractiveMaterial.cs:53-63 compare
materials by reference), objects keep their index in Scene.SceneObjects, and any type without a `flatten`
raises NotSupportedError — the backend never falls back to a CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import numpy as np

from . import capi
from .core import NotSupportedError, Scene

# floats per record, include/rt_hip.h
_PRIM_FLOATS = {
    capi.HIT_SPHERE: 4, capi.HIT_QUAD: 14, capi.HIT_TRIANGLE: 20, capi.HIT_INFINITE_PLANE: 7, capi.HIT_XY_PLANE: 5, capi.HIT_XZ_PLANE: 5,
    capi.HIT_YZ_PLANE: 5, capi.HIT_DISK: 16, capi.HIT_BOX: 21, capi.HIT_CAPSULE: 16, capi.HIT_CYLINDER: 7,
}
_PRIM_FIELD = {
    capi.HIT_SPHERE: ("spheres", capi.Sphere), capi.HIT_QUAD: ("quads", capi.Quad), capi.HIT_TRIANGLE: ("triangles", capi.Triangle),
    capi.HIT_INFINITE_PLANE: ("infinite_planes", capi.InfinitePlane), capi.HIT_XY_PLANE: ("xy_planes", capi.AxisPlane),
    capi.HIT_XZ_PLANE: ("xz_planes", capi.AxisPlane), capi.HIT_YZ_PLANE: ("yz_planes", capi.AxisPlane), capi.HIT_DISK: ("disks", capi.Disk),
    capi.HIT_BOX: ("boxes", capi.Box), capi.HIT_CAPSULE: ("capsules", capi.Capsule), capi.HIT_CYLINDER: ("cylinders", capi.Cylinder),
}
_OBJECT_DTYPE = np.dtype([("type", "<u4"), ("index", "<u4"), ("material", "<u4"), ("aabb_min", "<f4", 3), ("aabb_max", "<f4", 3)])
assert _OBJECT_DTYPE.itemsize == C.sizeof(capi.Object)


class FlatScene:
    """The finished tables; keeps every numpy buffer the rt_scene_desc points into alive."""

    def __init__(self, desc: capi.SceneDesc, keep: list, n_objects: int, n_materials: int, n_lights: int):
        self.desc, self._keep = desc, keep
        self.n_objects, self.n_materials, self.n_lights = n_objects, n_materials, n_lights


# rt_scene_desc's tables, in order (the .rtscene file and tools/rtbench.cpp follow it)
_TABLES = ("objects", "spheres", "quads", "triangles", "infinite_planes", "xy_planes", "xz_planes", "yz_planes", "disks", "boxes", "capsules",
           "cylinders", "media", "materials", "textures", "lights")


def save_rtscene(flat: "FlatScene", path: str) -> None:
    """Write the flattened tables to a file a host in any language can hand to rt_scene_create (SURVEY §8d "scenes dumped to
    .rtscene"): "RTSCENE1", u32 abi_version, rt_camera, rt_skybox, then per table { u32 count, u32 element size, bytes }."""
    import ctypes as C
    d = flat.desc
    with open(path, "wb") as f:
        f.write(b"RTSCENE1")
        f.write(np.uint32(d.abi_version).tobytes())
        f.write(bytes(d.camera)); f.write(bytes(d.skybox))
        for name in _TABLES:
            n = int(getattr(d, "n_" + name)); ptr = getattr(d, name)
            elem = C.sizeof(ptr._type_)
            f.write(np.array([n, elem], np.uint32).tobytes())
            if n:
                f.write(C.string_at(ptr, n * elem))


class DeviceSceneBuilder:
    def __init__(self):
        self._prims: Dict[int, List[np.ndarray]] = {t: [] for t in _PRIM_FLOATS}
        self._prim_count: Dict[int, int] = {t: 0 for t in _PRIM_FLOATS}
        self._media: List[tuple] = []
        self._objects: List[np.ndarray] = []
        self._n_objects = 0
        self._materials: List[capi.Material] = []
        self._textures: List[capi.Texture] = []
        self._lights: List[capi.Light] = []
        self._mat_ids: Dict[int, int] = {}
        self._tex_ids: Dict[int, int] = {}
        self._pins: list = []            # keep id()-keyed objects alive while the dicts exist
        self._skybox = capi.SkyBox()
        self._camera = capi.Camera()

    # ---- identity-keyed tables --------------------------------------------------------------------------
    def texture(self, tex) -> int:
        key = id(tex)
        if key not in self._tex_ids:
            if not hasattr(tex, "flatten"):
                raise NotSupportedError(f"{type(tex).__name__} is not a Texture the device knows")
            self._pins.append(tex)
            self._tex_ids[key] = tex.flatten(self)
        return self._tex_ids[key]

    def material(self, mat) -> int:
        key = id(mat)
        if key not in self._mat_ids:
            if not hasattr(mat, "flatten"):
                raise NotSupportedError(f"{type(mat).__name__} is not a Material the device knows")
            self._pins.append(mat)
            self._mat_ids[key] = mat.flatten(self)
        return self._mat_ids[key]

    def add_texture(self, type_, f) -> int:
        t = capi.Texture(); t.type = type_
        for i, x in enumerate(np.asarray(f, dtype=np.float32).reshape(-1)):
            t.f[i] = float(x)
        self._textures.append(t)
        return len(self._textures) - 1

    def add_material(self, type_, tex=(0, 0), flags=0, f=()) -> int:
        m = capi.Material(); m.type = type_; m.tex[0], m.tex[1] = int(tex[0]), int(tex[1]); m.flags = flags
        for i, x in enumerate(np.asarray(f, dtype=np.float32).reshape(-1)):
            m.f[i] = float(x)
        self._materials.append(m)
        return len(self._materials) - 1

    # ---- primitives ------------------------------------------------------------------------------------------
    def add_prim(self, type_, row) -> int:
        row = np.asarray(row, dtype=np.float32).reshape(1, -1)
        if row.shape[1] != _PRIM_FLOATS[type_]:
            raise ValueError(f"primitive type {type_}: expected {_PRIM_FLOATS[type_]} floats, got {row.shape[1]}")
        self._prims[type_].append(row)
        self._prim_count[type_] += 1
        return self._prim_count[type_] - 1

    def add_medium(self, boundary_type, boundary_index, density, neg_inv_density) -> int:
        self._media.append((int(boundary_type), int(boundary_index), float(density), float(neg_inv_density)))
        return len(self._media) - 1

    def add_object(self, hittable):
        if not hasattr(hittable, "flatten"):
            raise NotSupportedError(f"{type(hittable).__name__} is not a Hittable the device knows")
        type_, index = hittable.flatten(self)
        mat = getattr(hittable, "Material", None)
        if mat is None:
            raise NotSupportedError(f"{type(hittable).__name__} exposes no Material")
        o = np.zeros(1, dtype=_OBJECT_DTYPE)
        o["type"], o["index"], o["material"] = type_, index, self.material(mat)
        o["aabb_min"], o["aabb_max"] = hittable.BoundingVolume.Min, hittable.BoundingVolume.Max
        self._objects.append(o)
        self._n_objects += 1

    def add_bulk(self, type_, rows, material_ids, aabb_min, aabb_max):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        n = len(rows)
        if rows.shape[1] != _PRIM_FLOATS[type_]:
            raise ValueError("bulk rows have the wrong width")
        first = self._prim_count[type_]
        self._prims[type_].append(rows)
        self._prim_count[type_] += n
        o = np.zeros(n, dtype=_OBJECT_DTYPE)
        o["type"] = type_
        o["index"] = np.arange(first, first + n, dtype=np.uint32)
        o["material"] = material_ids
        o["aabb_min"], o["aabb_max"] = aabb_min, aabb_max
        self._objects.append(o)
        self._n_objects += n

    # ---- lights, skybox, camera --------------------------------------------------------------------------------
    def add_light(self, type_, position, colour, attenuation_radius, cutoff_radius, diffusion_radius, atten_kind, atten_p,
                  shape_type, shape_index, shape_min, shape_max) -> int:
        l = capi.Light(); l.type = type_
        for i in range(3):
            l.position[i], l.colour[i], l.atten_p[i] = float(position[i]), float(colour[i]), float(atten_p[i])
            l.shape_aabb_min[i], l.shape_aabb_max[i] = float(shape_min[i]), float(shape_max[i])
        l.attenuation_radius, l.cutoff_radius, l.diffusion_radius = float(attenuation_radius), float(cutoff_radius), float(diffusion_radius)
        l.atten_kind, l.shape_type, l.shape_index = int(atten_kind), int(shape_type), int(shape_index)
        self._lights.append(l)
        return len(self._lights) - 1

    def set_skybox(self, type_, colour):
        self._skybox.type = type_
        for i in range(3):
            self._skybox.colour[i] = float(colour[i])

    def set_camera(self, cam):
        c = self._camera
        for dst, src in ((c.look_from, cam.LookFrom), (c.horizontal, cam.Horizontal), (c.vertical, cam.Vertical),
                         (c.lower_left_corner, cam.LowerLeftCorner), (c.u, cam.U), (c.v, cam.V)):
            for i in range(3):
                dst[i] = float(src[i])
        c.lens_radius = float(cam.LensRadius)

    # ---- finish -------------------------------------------------------------------------------------------------
    def finish(self) -> FlatScene:
        d = capi.SceneDesc()
        d.struct_size, d.abi_version = C.sizeof(capi.SceneDesc), capi.RT_ABI_VERSION
        d.camera, d.skybox = self._camera, self._skybox
        keep = []

        def np_table(name, ctype, arr):
            arr = np.ascontiguousarray(arr)
            keep.append(arr)
            setattr(d, "n_" + name, len(arr))
            setattr(d, name, C.cast(arr.ctypes.data, C.POINTER(ctype)) if len(arr) else None)

        def ct_table(name, ctype, items):
            arr = (ctype * max(1, len(items)))(*items)
            keep.append(arr)
            setattr(d, "n_" + name, len(items))
            setattr(d, name, C.cast(arr, C.POINTER(ctype)) if items else None)

        for t, (name, ctype) in _PRIM_FIELD.items():
            chunks = self._prims[t]
            arr = np.concatenate(chunks, axis=0) if chunks else np.zeros((0, _PRIM_FLOATS[t]), np.float32)
            assert arr.shape[1] * 4 == C.sizeof(ctype), name
            np_table(name, ctype, arr)
        objs = np.concatenate(self._objects) if self._objects else np.zeros(0, _OBJECT_DTYPE)
        np_table("objects", capi.Object, objs)
        media = []
        for bt, bi, dens, nid in self._media:
            m = capi.Medium(); m.boundary_type, m.boundary_index, m.density, m.neg_inv_density = bt, bi, dens, nid
            media.append(m)
        ct_table("media", capi.Medium, media)
        ct_table("materials", capi.Material, self._materials)
        ct_table("textures", capi.Texture, self._textures)
        ct_table("lights", capi.Light, self._lights)
        return FlatScene(d, keep, self._n_objects, len(self._materials), len(self._lights))


def flatten_scene(scene: Scene) -> FlatScene:
    """Scene -> FlatScene.  Object i of the result is scene.SceneObjects[i]; bulk arrays follow in order."""
    b = DeviceSceneBuilder()
    b.set_camera(scene.Camera)
    if not hasattr(scene.SkyBox, "flatten"):
        raise NotSupportedError(f"{type(scene.SkyBox).__name__} is not a SkyBox the device knows")
    scene.SkyBox.flatten(b)
    for so in scene.SceneObjects:
        b.add_object(so.Hittable)
    for bulk in getattr(scene, "BulkObjects", ()):
        bulk.flatten_bulk(b)
    for light in scene.Lights:
        if not hasattr(light, "flatten"):
            raise NotSupportedError(f"{type(light).__name__} is not a Light the device knows")
        light.flatten(b)
    return b.finish()
