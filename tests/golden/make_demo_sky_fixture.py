"""Extracts sky pixels of the reference's committed render `Renders/Demo Scene/image.png` (1920x1080, 1000 passes,
Renders/Demo Scene/info.md:3-4) into tests/golden/demo_sky_pixels.npz.  The PNG is a data file of the reference;
the fixture holds 4096 (x, y_top, r, g, b) samples from the top 300 rows, all of which are sky (no object of
BuiltinScenes.Demo projects there except the 0.05-radius light visualiser, which the sampler skips by requiring
blue == 255, the DefaultSkyBox's blue channel).  Run here only (needs /root/reference)."""
import numpy as np
from PIL import Image

im = np.array(Image.open("/root/reference/Renders/Demo Scene/image.png").convert("RGB"))
assert im.shape == (1080, 1920, 3)
rs = np.random.default_rng(20221110)
xs = rs.integers(0, 1920, 6000); ys = rs.integers(0, 300, 6000)
keep = im[ys, xs, 2] == 255
xs, ys = xs[keep][:4096], ys[keep][:4096]
np.savez_compressed("tests/golden/demo_sky_pixels.npz", x=xs.astype(np.int32), y_top=ys.astype(np.int32), rgb=im[ys, xs].astype(np.uint8),
                    width=np.int32(1920), height=np.int32(1080))
print(len(xs), "pixels")
