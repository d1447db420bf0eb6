"""Headless console runner: the role the reference's (missing) console front end plays beside its GUIs.

    python -m raytracer_amd --list
    python -m raytracer_amd --scene "Demo" --width 1920 --height 1080 --passes 256 --depth 10 --out image.png

Scene pick = BuiltinScenes.GetAll() by name (BuiltinScenes.cs:298-301) plus this build's two benchmark scenes; options =
RenderOptions' fields (Core/RenderOptions.cs:11-19); the image is saved as ./image.png the way MainForm does
(Display.EtoForms/MainForm.cs:124-129); progress lines carry the RenderStats counters the GUIs show.  Everything is traced by
librt_hip.so — no GPU, no picture.
"""
import argparse
import sys
import time

from . import builtin_scenes as bs
from . import core
from .render_job import RenderJob, DeviceScene


def scene_table():
    table = {s.Name: (lambda s=s: s) for s in bs.GetAll()}
    table["Triangle Mesh"] = bs.TriangleMesh
    table["Many Spheres Many Lights"] = bs.ManySpheresManyLights
    return table


def obj_scene(path, aspect):
    """An imported model on a ground plane under the default sky, seen from a corner of its bounding box."""
    import numpy as np
    from . import impl, mesh_io
    mesh = mesh_io.load_obj(path, Material=impl.StandardMaterial(core.Colour(0.7), 1.0))
    pts = np.concatenate([mesh.v0, mesh.v1, mesh.v2])
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    centre, size = (lo + hi) / 2, float(np.linalg.norm(hi - lo)) or 1.0
    eye = centre + np.array([0.9, 0.6, 1.2], np.float32) * size
    cam = core.Camera.Create(tuple(eye), tuple(centre), (0, 1, 0), 40, aspect, 0.0, float(np.linalg.norm(eye - centre)))
    ground = core.SceneObject("ground", impl.InfinitePlane((0, float(lo[1]), 0), (0, 1, 0), Material=impl.StandardMaterial(core.Colour(0.5), 0.9)))
    light = impl.PointLight(Position=tuple(centre + np.array([0.5, 1.5, 0.8], np.float32) * size), AttenuationRadius=4 * size)
    return core.Scene(path, cam, [ground], [light], impl.DefaultSkyBox(), BulkObjects=[mesh])


VIS_NAMES = {k.rstrip("_"): v for k, v in vars(core.GraphicsDebugVisualisation).items() if not k.startswith("_") and isinstance(v, int)}


def save_png(path, rgb):
    from PIL import Image                      # the reference encodes with ImageSharp; encoding is not on the traced path
    Image.fromarray(rgb, "RGB").save(path)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m raytracer_amd", description=__doc__.split("\n\n")[0])
    ap.add_argument("--list", action="store_true", help="list the scenes and exit")
    ap.add_argument("--scene", default="Demo")
    ap.add_argument("--obj", default=None, help="render a Wavefront OBJ file (grey diffuse, a ground plane, one light, camera fitted) instead of a builtin scene")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--passes", type=int, default=100)
    ap.add_argument("--depth", type=int, default=10, help="MaxBounceDepth")
    ap.add_argument("--light-samples", type=int, default=2, help="LightSampleCountHint")
    ap.add_argument("--kmin", type=float, default=0.001)
    ap.add_argument("--kmax", type=float, default=float("inf"))
    ap.add_argument("--vis", default="None", help="DebugVisualisation: " + ", ".join(sorted(VIS_NAMES, key=VIS_NAMES.get)))
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--device", type=int, default=-1)
    ap.add_argument("--progress-passes", type=int, default=0, help="passes between progress lines (0 = one batch)")
    ap.add_argument("--out", default="./image.png")
    a = ap.parse_args(argv)

    table = scene_table()
    if a.list:
        for name in table:
            print(name)
        return 0
    if a.obj:
        table = {"obj": lambda: obj_scene(a.obj, a.width / a.height)}
        a.scene = "obj"
    match = [n for n in table if n.lower() == a.scene.lower()] or [n for n in table if a.scene.lower() in n.lower()]
    if len(match) != 1:
        print(f"scene {a.scene!r}: {'ambiguous: ' + ', '.join(match) if match else 'unknown'}; --list shows the names", file=sys.stderr)
        return 2
    vis = VIS_NAMES.get(a.vis)
    if vis is None:
        print(f"unknown visualisation {a.vis!r}", file=sys.stderr)
        return 2
    scene = table[match[0]]()
    opt = core.RenderOptions(RenderWidth=a.width, RenderHeight=a.height, KMin=a.kmin, KMax=a.kmax, Passes=a.passes, MaxBounceDepth=a.depth,
                             DebugVisualisation=vis, LightSampleCountHint=a.light_samples, Seed=a.seed)
    t0 = time.perf_counter()
    dev = DeviceScene(scene, a.device)
    info = dev.Info()
    print(f"scene '{scene.Name}': {info['n_objects']} objects, {info['n_nodes']} BVH nodes (depth {info['bvh_depth']}), kernels '{info['variant']}', "
          f"tree in {'LDS' if info['tree_in_lds'] else 'HBM'}; built in {time.perf_counter() - t0:.2f} s", flush=True)
    job = RenderJob(scene, opt, device_scene=dev)
    job.passes_per_call = a.progress_passes or int(job._opts.passes_per_batch) or a.passes
    t0 = time.perf_counter()
    task = job.StartOrGetRenderAsync()
    finished = []
    task.add_done_callback(lambda _: finished.append(time.perf_counter()))      # the moment the render task completed, not the next poll
    try:
        while not task.done():
            time.sleep(0.05)
            done = job.PassesSoFar
            if done:
                el = time.perf_counter() - t0
                print(f"\r{done}/{a.passes} passes, {a.width * a.height * done / el / 1e6:.1f} Msamples/s", end="", flush=True)
        task.result()
    except KeyboardInterrupt:
        job._cancel.set()
        print("\ncancelled", file=sys.stderr)
        return 130
    el = (finished[0] if finished else time.perf_counter()) - t0
    st = job.RenderStats
    print(f"\r{a.passes} passes in {el:.2f} s: {a.width * a.height * a.passes / el / 1e6:.1f} Msamples/s, {st.RayCount / el / 1e6:.0f} M closest rays/s, "
          f"{st.ShadowRays / el / 1e6:.0f} M shadow rays/s; sky {st.SkyRays}, bounce limit {st.BounceLimitExceeded}")
    save_png(a.out, job.Image)
    print(f"saved {a.out}")
    job.Dispose(); dev.Dispose()
    return 0


if __name__ == "__main__":
    sys.exit(main())
