"""Object outlines of the reference's committed Demo render (commit f2d7ea0f) as a geometric pin of camera, Quad, Disk and Sphere
placement: the yellow quad's bounding box came out IDENTICAL to the pixel on the device (1185-1344 x 761-925), the magenta disk
within 4 px, the red emissive sphere within 1 px.  Brightness is not compared (tests/golden/make_demo_silhouette_fixture.py says why)."""
import os
import sys

import numpy as np
import pytest

from raytracer_amd import builtin_scenes as bs
from raytracer_amd.core import RenderOptions
from raytracer_amd.flatten import flatten_scene
from oracle import binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_demo_silhouette_fixture import silhouettes          # noqa: E402  (the same masks the fixture was made with)

G = np.load(os.path.join(HERE, "golden", "demo_silhouettes.npz"))
TOL = {"yellow_quad": 2, "magenta_disk": 6, "red_dot": 3}


def check(img, few_samples=False):
    got = silhouettes(img)
    for k, tol in TOL.items():
        # (at 8 spp the disk's pink glow on today's white ground still speckles the rows below it and the left edge: top and right only)
        d = np.abs(got[k] - G[k])[[1, 2]] if few_samples and k == "magenta_disk" else np.abs(got[k] - G[k])
        assert d.max() <= tol, f"{k}: {got[k]} vs the reference's {G[k]}"


def test_oracle_demo_outlines_match_the_reference_render():
    acc, _ = ob.OracleScene(flatten_scene(bs.Demo())).render(RenderOptions(RenderWidth=1920, RenderHeight=1080, Passes=8, MaxBounceDepth=4, Seed=3), threads=8)
    check(ob.resolve_rgb24(acc, 8), few_samples=True)


@pytest.mark.gpu
def test_device_demo_outlines_match_the_reference_render():
    import raytracer_amd as rt
    job = rt.RenderJob(bs.Demo(), RenderOptions(RenderWidth=1920, RenderHeight=1080, Passes=256, MaxBounceDepth=10, Seed=3)); job.Wait()
    check(job.Image)
    job.Dispose()
