"""The reference's four committed Cornell-box renders (Renders/Refractive Material Indirect Emission Comparison, commit 3970579a)
as a geometric pin.  Their lights were brighter than today's BuiltinScenes.CornellBox (the pictures' walls are 2.5 x brighter than a
render of the in-tree scene by either side here), so brightness is NOT compared; what is compared does not depend on it:
where the box's opening ends against the exactly-black sky (camera position, field of view, the y flip) and the silhouette of
the emissive blue sphere (its centre, radius, and the small box in front of it).  tests/golden/make_cornell_fixture.py made the numbers."""
import os

import numpy as np
import pytest

from raytracer_amd import builtin_scenes as bs
from raytracer_amd.core import RenderOptions
from raytracer_amd.flatten import flatten_scene
from oracle import binding as ob

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_block_means.npz"))


def silhouettes(img):
    a = img.astype(int)
    lit = a.sum(axis=2) > 0
    rows, cols = lit.any(axis=1).nonzero()[0], lit.any(axis=0).nonzero()[0]
    m = (a[..., 2] > 200) & (a[..., 2] > 2.5 * np.maximum(a[..., 0], a[..., 1]))
    ys, xs = np.nonzero(m)
    bx, by = (m.sum(axis=0) >= 15).nonzero()[0], (m.sum(axis=1) >= 15).nonzero()[0]      # columns / rows with a run of the mask, not a stray noisy pixel
    return np.array([rows[0], rows[-1], cols[0], cols[-1]]), np.array([bx[0], bx[-1], by[0], by[-1]]), \
        np.array([xs.mean() / a.shape[1], ys.mean() / a.shape[0], m.mean()])


def check(img, opening=True, sphere=True, first_hits_only=False):
    lit, bbox, ca = silhouettes(img)
    if opening:
        assert np.abs(lit - G["lit_bounds"]).max() <= 2, f"box opening {lit} vs the reference's {G['lit_bounds']}"
    if sphere and first_hits_only:      # (without bounces the sphere's underside, lit only by the box top below it, is not blue yet: left, right, top)
        assert np.abs(bbox[:3] - G["blue_bbox"][:3]).max() <= 3, f"blue sphere {bbox} vs the reference's {G['blue_bbox']}"
    elif sphere:                        # (top and bottom rows end where the brighter ceiling light of that commit washes the blue out: 6 px)
        assert np.abs(bbox[:2] - G["blue_bbox"][:2]).max() <= 3 and np.abs(bbox[2:] - G["blue_bbox"][2:]).max() <= 6, f"blue sphere {bbox} vs the reference's {G['blue_bbox']}"
        assert np.abs(ca[:2] - G["blue_centroid_area"][:2]).max() <= 0.004 and abs(ca[2] - G["blue_centroid_area"][2]) <= 0.1 * G["blue_centroid_area"][2]


def test_oracle_cornell_silhouettes_match_the_reference_render():
    """The CPU side can afford a few samples per pixel: the sphere's silhouette from first hits only (no blue fireflies on the walls
    yet), the opening from a 3-bounce render."""
    orc = ob.OracleScene(flatten_scene(bs.CornellBox(1.0, True)))
    acc, _ = orc.render(RenderOptions(RenderWidth=1080, RenderHeight=1080, Passes=2, MaxBounceDepth=0, Seed=5), threads=8)
    check(ob.resolve_rgb24(acc, 2), opening=False, first_hits_only=True)
    acc, _ = orc.render(RenderOptions(RenderWidth=1080, RenderHeight=1080, Passes=3, MaxBounceDepth=3, Seed=5), threads=8)
    check(ob.resolve_rgb24(acc, 3), sphere=False)


@pytest.mark.gpu
def test_device_cornell_silhouettes_match_the_reference_render():
    import raytracer_amd as rt
    job = rt.RenderJob(bs.CornellBox(1.0, True), RenderOptions(RenderWidth=1080, RenderHeight=1080, Passes=100, MaxBounceDepth=10, Seed=5)); job.Wait()
    check(job.Image)
    job.Dispose()
