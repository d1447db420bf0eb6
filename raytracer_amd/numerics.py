"""System.Numerics / MathF in binary32, as the reference's constructors use them.

Scene construction is host-side work: the C# flattener copies fields the reference's constructors already
computed (Quad.localToQuadMatrix, Box.WorldToBoxTransform, Camera.LowerLeftCorner ...).  A caller that is
not C# has to produce the same fields, so this module restates the few BCL routines those constructors call,
one float32 rounding per operator (numpy float32 scalars; NEP 50 keeps Python literals weak).

Software definitions of .NET 7 (the hardware-accelerated paths may differ in the last ulp).
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
PI = f32(math.pi)      # MathF.PI
E = f32(math.e)        # MathF.E


def vec3(x, y=None, z=None) -> np.ndarray:
    if y is None:
        a = np.asarray(x, dtype=np.float32)
        return np.full(3, a, dtype=np.float32) if a.ndim == 0 else a.astype(np.float32).copy()
    return np.array([x, y, z], dtype=np.float32)


def dot(a, b) -> np.float32:
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def cross(a, b) -> np.ndarray:
    return vec3(f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]), f32(a[0] * b[1]) - f32(a[1] * b[0]))


def length(a) -> np.float32:
    return np.sqrt(dot(a, a), dtype=np.float32)


def normalize(a) -> np.ndarray:
    with np.errstate(all="ignore"):
        return (np.asarray(a, np.float32) / length(a)).astype(np.float32)


def distance(a, b) -> np.float32:
    d = np.asarray(a, np.float32) - np.asarray(b, np.float32)
    return np.sqrt(dot(d, d), dtype=np.float32)


def lerp3(a, b, t) -> np.ndarray:
    t = f32(t)
    return (a * (f32(1.0) - t) + b * t).astype(np.float32)


# ---- Matrix4x4 (row-vector convention, M[r][c] = M(r+1)(c+1)) --------------------------------------------

def identity4() -> np.ndarray:
    return np.eye(4, dtype=np.float32)


def create_scale(sx, sy, sz) -> np.ndarray:
    m = identity4(); m[0, 0] = sx; m[1, 1] = sy; m[2, 2] = sz
    return m


def create_translation(x, y, z) -> np.ndarray:
    m = identity4(); m[3, 0] = x; m[3, 1] = y; m[3, 2] = z
    return m


def create_from_yaw_pitch_roll(yaw, pitch, roll) -> np.ndarray:
    """Matrix4x4.CreateFromYawPitchRoll = CreateFromQuaternion(Quaternion.CreateFromYawPitchRoll)."""
    yaw, pitch, roll = f32(yaw), f32(pitch), f32(roll)
    hr, hp, hy = f32(roll * f32(0.5)), f32(pitch * f32(0.5)), f32(yaw * f32(0.5))
    sr, cr = f32(np.sin(hr, dtype=np.float32)), f32(np.cos(hr, dtype=np.float32))
    sp, cp = f32(np.sin(hp, dtype=np.float32)), f32(np.cos(hp, dtype=np.float32))
    sy, cy = f32(np.sin(hy, dtype=np.float32)), f32(np.cos(hy, dtype=np.float32))
    qx = f32(f32(f32(cy * sp) * cr) + f32(f32(sy * cp) * sr))
    qy = f32(f32(f32(sy * cp) * cr) - f32(f32(cy * sp) * sr))
    qz = f32(f32(f32(cy * cp) * sr) - f32(f32(sy * sp) * cr))
    qw = f32(f32(f32(cy * cp) * cr) + f32(f32(sy * sp) * sr))
    xx, yy, zz = f32(qx * qx), f32(qy * qy), f32(qz * qz)
    xy, wz, xz, wy, yz, wx = f32(qx * qy), f32(qz * qw), f32(qz * qx), f32(qy * qw), f32(qy * qz), f32(qx * qw)
    m = identity4()
    one, two = f32(1.0), f32(2.0)
    m[0, 0] = one - two * (yy + zz); m[0, 1] = two * (xy + wz); m[0, 2] = two * (xz - wy)
    m[1, 0] = two * (xy - wz); m[1, 1] = one - two * (zz + xx); m[1, 2] = two * (yz + wx)
    m[2, 0] = two * (xz + wy); m[2, 1] = two * (yz - wx); m[2, 2] = one - two * (yy + xx)
    return m


def matmul4(a, b) -> np.ndarray:
    """Matrix4x4 operator*: each element is a left-to-right sum of four products."""
    out = np.zeros((4, 4), dtype=np.float32)
    for r in range(4):
        for c in range(4):
            s = f32(a[r, 0] * b[0, c])
            for k in range(1, 4):
                s = f32(s + f32(a[r, k] * b[k, c]))
            out[r, c] = s
    return out


def invert4(mat) -> np.ndarray:
    """Matrix4x4.Invert (cofactor expansion, software path), for one matrix or a stack (..., 4, 4).
    Raises ArithmeticError when (any matrix is) singular."""
    mat = np.asarray(mat, dtype=np.float32)
    a, b, c, d = (mat[..., 0, q] for q in range(4)); e, f, g, h = (mat[..., 1, q] for q in range(4))
    i, j, k, l = (mat[..., 2, q] for q in range(4)); m, n, o, p = (mat[..., 3, q] for q in range(4))
    kp_lo = k * p - l * o; jp_ln = j * p - l * n; jo_kn = j * o - k * n
    ip_lm = i * p - l * m; io_km = i * o - k * m; in_jm = i * n - j * m
    a11 = +(f * kp_lo - g * jp_ln + h * jo_kn)
    a12 = -(e * kp_lo - g * ip_lm + h * io_km)
    a13 = +(e * jp_ln - f * ip_lm + h * in_jm)
    a14 = -(e * jo_kn - f * io_km + g * in_jm)
    det = a * a11 + b * a12 + c * a13 + d * a14
    # MathF.Abs(det) < float.Epsilon (the smallest subnormal); a non-finite determinant is rejected too
    if np.any(np.abs(det) < np.finfo(np.float32).smallest_subnormal) or not np.all(np.isfinite(det)):
        raise ArithmeticError("matrix is not invertible")
    inv = f32(1.0) / det
    r = np.zeros(mat.shape, dtype=np.float32)
    r[..., 0, 0] = a11 * inv; r[..., 1, 0] = a12 * inv; r[..., 2, 0] = a13 * inv; r[..., 3, 0] = a14 * inv
    r[..., 0, 1] = -(b * kp_lo - c * jp_ln + d * jo_kn) * inv
    r[..., 1, 1] = +(a * kp_lo - c * ip_lm + d * io_km) * inv
    r[..., 2, 1] = -(a * jp_ln - b * ip_lm + d * in_jm) * inv
    r[..., 3, 1] = +(a * jo_kn - b * io_km + c * in_jm) * inv
    gp_ho = g * p - h * o; fp_hn = f * p - h * n; fo_gn = f * o - g * n
    ep_hm = e * p - h * m; eo_gm = e * o - g * m; en_fm = e * n - f * m
    r[..., 0, 2] = +(b * gp_ho - c * fp_hn + d * fo_gn) * inv
    r[..., 1, 2] = -(a * gp_ho - c * ep_hm + d * eo_gm) * inv
    r[..., 2, 2] = +(a * fp_hn - b * ep_hm + d * en_fm) * inv
    r[..., 3, 2] = -(a * fo_gn - b * eo_gm + c * en_fm) * inv
    gl_hk = g * l - h * k; fl_hj = f * l - h * j; fk_gj = f * k - g * j
    el_hi = e * l - h * i; ek_gi = e * k - g * i; ej_fi = e * j - f * i
    r[..., 0, 3] = -(b * gl_hk - c * fl_hj + d * fk_gj) * inv
    r[..., 1, 3] = +(a * gl_hk - c * el_hi + d * ek_gi) * inv
    r[..., 2, 3] = -(a * fl_hj - b * el_hi + d * ej_fi) * inv
    r[..., 3, 3] = +(a * fk_gj - b * ek_gi + c * ej_fi) * inv
    return r


def transform_point(v, m) -> np.ndarray:
    """Vector3.Transform(position, matrix)."""
    return vec3(*(f32(f32(f32(f32(v[0] * m[0, c]) + f32(v[1] * m[1, c])) + f32(v[2] * m[2, c])) + m[3, c]) for c in range(3)))
