"""Block means of the reference's four committed Cornell-box renders (Renders/Refractive Material Indirect Emission Comparison/*.png:
1080 x 1080, 100 passes, commit 3970579a) -> tests/golden/cornell_block_means.npz.  The pictures are noisy (100 spp) and from an older
commit, so they pin the picture loosely: the mean of every 45 x 45 block of the sRGB-ish bytes, per channel (24 x 24 x 3 per render).
Run here (needs /root/reference); the GPU box only reads the npz."""
import os
import numpy as np
from PIL import Image

SRC = "/root/reference/Renders/Refractive Material Indirect Emission Comparison"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cornell_block_means.npz")
CASES = {"direct_10": ("direct emission 10%.png", 0.1, True), "direct_100": ("direct emission 100%.png", 1.0, True),
         "indirect_10": ("indirect emission 10%.png", 0.1, False), "indirect_100": ("indirect emission 100%.png", 1.0, False)}


def block_means(img, n=24):
    h, w, _ = img.shape
    return img[: h // n * n, : w // n * n].reshape(n, h // n, n, w // n, 3).astype(np.float64).mean(axis=(1, 3)).astype(np.float32)


def silhouettes(a):
    """What does not depend on how bright the lights were at that commit: where the box's opening ends (outside it the black sky is
    exactly 0) and where the emissive blue sphere is (direct emission, intensity 1: saturated blue)."""
    a = a.astype(int)
    lit = a.sum(axis=2) > 0
    rows, cols = lit.any(axis=1).nonzero()[0], lit.any(axis=0).nonzero()[0]
    m = (a[..., 2] > 200) & (a[..., 2] > 2.5 * np.maximum(a[..., 0], a[..., 1]))
    ys, xs = np.nonzero(m)
    bx, by = (m.sum(axis=0) >= 15).nonzero()[0], (m.sum(axis=1) >= 15).nonzero()[0]      # columns / rows with a run of the mask, not a stray noisy pixel
    return np.array([rows[0], rows[-1], cols[0], cols[-1]], np.int32), np.array([bx[0], bx[-1], by[0], by[-1]], np.int32), \
        np.array([xs.mean() / a.shape[1], ys.mean() / a.shape[0], m.mean()], np.float32)


if __name__ == "__main__":
    out = {}
    a = np.asarray(Image.open(os.path.join(SRC, CASES["direct_100"][0])).convert("RGB"))
    out["lit_bounds"], out["blue_bbox"], out["blue_centroid_area"] = silhouettes(a)
    for key, (name, intensity, direct) in CASES.items():
        a = np.asarray(Image.open(os.path.join(SRC, name)).convert("RGB"))
        assert a.shape == (1080, 1080, 3)
        out[key] = block_means(a)
        out[key + "_params"] = np.array([intensity, float(direct)], np.float32)
    np.savez_compressed(OUT, **out)
    print(OUT, {k: v.shape for k, v in out.items()})
