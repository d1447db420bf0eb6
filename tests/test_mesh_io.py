"""OBJ import -> TriangleArray -> the device (Todos.md "Import 3D models")."""
import numpy as np
import pytest

from raytracer_amd import impl, mesh_io
from raytracer_amd.core import Camera, Scene
from raytracer_amd.flatten import flatten_scene

OBJ = """# a unit cube with one quad face, one polygon and relative indices
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 1
vn 0 0 1
f 1 2 3 4
f 5/1 6/2 7/3 8/4
f 1//1 2//1 6//1 5//1
f -8 -4 -1 -5
f 2 3 7 6
f 1 4 8 5
f 1 1 2
"""


def _scene(tmp_path):
    p = tmp_path / "cube.obj"; p.write_text(OBJ)
    mesh = mesh_io.load_obj(str(p), scale=2.0, translate=(-1, -1, -1))
    cam = Camera.Create((3, 2.5, 4), (0, 0, 0), (0, 1, 0), 40, 16 / 9, 0.0, 5.0)
    return mesh, Scene("cube.obj", cam, [], [impl.PointLight(Position=(3, 4, 2), AttenuationRadius=12)], impl.DefaultSkyBox(), BulkObjects=[mesh])


def test_obj_faces_are_fanned_and_degenerates_dropped(tmp_path):
    mesh, sc = _scene(tmp_path)
    assert len(mesh) == 12 and mesh.dropped == 1
    pts = np.concatenate([mesh.v0, mesh.v1, mesh.v2])
    assert pts.min() == -1 and pts.max() == 1
    flat = flatten_scene(sc)
    assert flat.n_objects == 12 and flat.desc.n_triangles == 12
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.obj"; bad.write_text("v 0 0 0\nf 1 2 3\n"); mesh_io.read_obj(str(bad))


@pytest.mark.gpu
def test_imported_mesh_traces_like_the_oracle(tmp_path):
    import raytracer_amd as rt
    from oracle import binding as ob
    import helpers
    mesh, sc = _scene(tmp_path)
    flat = flatten_scene(sc)
    dev, orc = rt.DeviceScene(flat), ob.OracleScene(flat)
    rs = np.random.default_rng(3)
    o, d = helpers.camera_rays(sc, 64, 36, 20_000, rs)
    g, r = dev.TraceClosest(o, d, 0.001, np.inf), orc.trace_closest(o, d, 0.001, np.inf)
    helpers.assert_closest_parity(g, r)
    assert (r["object"] != 0xFFFFFFFF).mean() > 0.1
