"""Silhouettes in the reference's committed Demo render (Renders/Demo Scene/image.png: 1920 x 1080, 1000 passes, commit f2d7ea0f)
-> tests/golden/demo_silhouettes.npz.  The picture's marble ground is from another MarbleTexture than today's (dark with white
blobs; today's is white with thin dark veins) and SharpNoise is not in the tree anyway, so brightness is not compared; the
objects' outlines are: the yellow quad, the magenta disk and the small red emissive sphere are found by colour.
Run here (needs /root/reference)."""
import os
import numpy as np
from PIL import Image


def bbox(m, run):
    bx, by = (m.sum(axis=0) >= run).nonzero()[0], (m.sum(axis=1) >= run).nonzero()[0]
    return np.array([bx[0], bx[-1], by[0], by[-1]], np.int32)


def silhouettes(img):
    a = img.astype(int); R, G, B = a[..., 0], a[..., 1], a[..., 2]
    return {"yellow_quad": bbox((R > 140) & (G > 140) & (B < 80), 12), "magenta_disk": bbox((R > 120) & (B > 120) & (G < 70), 12),
            "red_dot": bbox((R > 200) & (G < 80) & (B < 80), 2)}


if __name__ == "__main__":
    a = np.asarray(Image.open("/root/reference/Renders/Demo Scene/image.png").convert("RGB"))
    assert a.shape == (1080, 1920, 3)
    s = silhouettes(a)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "demo_silhouettes.npz"), **s)
    print(s)
