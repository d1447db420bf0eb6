"""3D model import (the reference's Todos.md "Import 3D models", Core/Todos.md:28-30): Wavefront OBJ -> TriangleArray.

Only what the Triangle primitive needs: `v x y z` positions and `f` faces (any of `i`, `i/t`, `i//n`, `i/t/n`, negative = relative
indices); polygons are fanned from their first vertex; normals, texture coordinates, groups and materials are ignored (Triangle
uses its geometric normal, rt_hip.h).  Degenerate faces (zero area: the plane's normal would be NaN) are dropped and counted.
"""
from typing import Optional, Tuple

import numpy as np

from .core import Material
from .impl import TriangleArray


def read_obj(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """-> (vertices float32 [V,3], triangles int64 [T,3], 0-based)."""
    verts, tris = [], []
    with open(path, "r", encoding="utf-8", errors="replace") as f:
        for ln, line in enumerate(f, 1):
            parts = line.split()
            if not parts or parts[0].startswith("#"):
                continue
            if parts[0] == "v":
                if len(parts) < 4:
                    raise ValueError(f"{path}:{ln}: vertex needs three coordinates")
                verts.append((float(parts[1]), float(parts[2]), float(parts[3])))
            elif parts[0] == "f":
                if len(parts) < 4:
                    raise ValueError(f"{path}:{ln}: face needs at least three vertices")
                idx = []
                for tok in parts[1:]:
                    i = int(tok.split("/")[0])
                    i = i - 1 if i > 0 else len(verts) + i
                    if not 0 <= i < len(verts):
                        raise ValueError(f"{path}:{ln}: vertex index {tok} out of range")
                    idx.append(i)
                for k in range(1, len(idx) - 1):
                    tris.append((idx[0], idx[k], idx[k + 1]))
    return np.asarray(verts, dtype=np.float32).reshape(-1, 3), np.asarray(tris, dtype=np.int64).reshape(-1, 3)


def load_obj(path: str, Material: Optional[Material] = None, scale: float = 1.0, translate=(0.0, 0.0, 0.0)) -> TriangleArray:
    """A TriangleArray (one Material) for Scene(BulkObjects=[...]); `dropped` = number of zero-area faces left out."""
    v, t = read_obj(path)
    v = (v * np.float32(scale) + np.asarray(translate, dtype=np.float32)).astype(np.float32)
    v0, v1, v2 = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    area2 = np.linalg.norm(np.cross((v1 - v0).astype(np.float64), (v2 - v0).astype(np.float64)), axis=1)
    keep = area2 > 0
    mesh = TriangleArray(v0[keep], v1[keep], v2[keep], Material=Material)
    mesh.dropped = int((~keep).sum())
    return mesh
