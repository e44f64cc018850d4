"""Independent NumPy (float32) derivation of the INPUT STAGE, used to emit
tests/golden/input_stage_*.npz.

Written from the reference's formulas, not from the C++ oracle or the HIP kernels, and sharing no
code with them: whole-image NumPy float32 arithmetic (one IEEE rounding per operation, no FMA), one
window tap at a time in the shader's loop order, so the integer outputs compare bit for bit.

  load_frame      StaticFusion::loadImageFromSequenceAssoc   reference FrontEnd.cpp:216-254
  bilateral_mm    Shaders/depth_bilateral.frag:34-74         (Reconstruction.cpp:337-346 filterDepth)
  metricise       Shaders/depth_metric.frag:32-39            (Reconstruction.cpp:327-335)
  exp_neg         the weight's exp(-a) as include/sf_detmath.h specifies it, operation by operation

Run (in the build container):  python tools/golden/make_golden_input.py  -> tests/golden/input_stage_*.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
f32 = np.float32


def fma32(a, b, c):
    """Correctly rounded float32 fused multiply-add, array-at-a-time, without an FMA instruction:
    the product of two float32 is exact in float64; the float64 sum is rounded once; TwoSum gives the
    exact rounding error, which decides the (rare) case where the float64 sum sits exactly on a
    float32 rounding boundary (double rounding)."""
    f64 = np.float64
    p = a.astype(f64) * b.astype(f64)
    c64 = c.astype(f64)
    s = p + c64
    bb = s - p
    e = (p - (s - bb)) + (c64 - bb)  # exact error of s
    r = s.astype(f32)
    r64 = r.astype(f64)
    up = np.nextafter(r, f32(np.inf))
    dn = np.nextafter(r, f32(-np.inf))
    tie_up = (s > r64) & (s == (r64 + up.astype(f64)) * 0.5)  # s on the boundary, rounded down to r
    tie_dn = (s < r64) & (s == (r64 + dn.astype(f64)) * 0.5)  # s on the boundary, rounded up to r
    r = np.where(tie_up & (e > 0), up, r)
    r = np.where(tie_dn & (e < 0), dn, r)
    return r.astype(f32)


def exp_neg(a):
    """exp(-a), a >= 0 float32 array: include/sf_detmath.h, each line one float32 operation."""
    a = a.astype(f32)
    log2e = f32(1.44269502162933349609375)
    ln2_hi = f32(0.693145751953125)
    ln2_lo = f32(1.42860676533018589e-06)
    live = a <= f32(87.0)
    aa = np.where(live, a, f32(0.0)).astype(f32)
    n = np.rint(aa * log2e).astype(f32)  # ties to even
    r = fma32(-n, np.full_like(n, ln2_hi), aa)
    r = fma32(-n, np.full_like(n, ln2_lo), r)
    x = -r
    p = np.full_like(x, f32(1.0) / f32(5040.0))
    for c in (f32(1.0) / f32(720.0), f32(1.0) / f32(120.0), f32(1.0) / f32(24.0), f32(1.0) / f32(6.0), f32(0.5), f32(1.0), f32(1.0)):
        p = fma32(p, x, np.full_like(x, c))
    scale = ((127 - n.astype(np.int64)).astype(np.uint32) << np.uint32(23)).view(f32)
    return np.where(live, p * scale, f32(0.0)).astype(f32)


def check_fma32():
    """fma32 against libm's fmaf on random and on constructed boundary cases."""
    import ctypes

    libm = ctypes.CDLL("libm.so.6")
    libm.fmaf.restype = ctypes.c_float
    libm.fmaf.argtypes = [ctypes.c_float] * 3
    rng = np.random.default_rng(7)
    a = rng.normal(0, 1, 20000).astype(f32)
    b = rng.normal(0, 1, 20000).astype(f32)
    cc = (-(a.astype(np.float64) * b.astype(np.float64)) * (1 + rng.normal(0, 1e-7, 20000))).astype(f32)  # cancellation
    cc[::3] = rng.normal(0, 1, cc[::3].size).astype(f32)
    # boundary cases: a*b + c exactly half an ulp above a float32, plus a tiny product term
    a[:64] = f32(1.0) + f32(2.0) ** -12
    b[:64] = f32(1.0) + f32(2.0) ** -12  # a*b = 1 + 2^-11 + 2^-24: the 2^-24 is exactly half an ulp of 1.0x
    cc[:64] = rng.choice([f32(0.0), f32(2.0) ** -40, -f32(2.0) ** -40], 64)
    got = fma32(a, b, cc)
    ref = np.array([libm.fmaf(float(x), float(y), float(z)) for x, y, z in zip(a, b, cc)], dtype=f32)
    assert np.array_equal(got, ref), int((got != ref).sum())


def load_frame(color_full, depth_full, res):
    """FrontEnd.cpp:216-254 -> intensityCurrent, depthCurrent (rows, cols) float32; depth_mm uint16; color uint8."""
    full_rows, full_cols = depth_full.shape
    rows, cols = full_rows // res, full_cols // res
    v = np.arange(rows)[:, None]
    u = np.arange(cols)[None, :]
    sr = rows * res - res * v - 1  # :231 vertical flip
    sc = res * u
    px = color_full[sr, sc].astype(f32)  # (rows, cols, 3)
    norm = f32(1.0) / f32(255.0)
    r, g, b = norm * px[..., 0], norm * px[..., 1], norm * px[..., 2]
    intensity = (f32(0.299) * r + f32(0.587) * g) + f32(0.114) * b  # :236, left to right
    color = np.clip(np.rint(np.stack([r * f32(255.0), g * f32(255.0), b * f32(255.0)], axis=-1)), 0, 255).astype(np.uint8)  # :237
    mm = depth_full[sr, sc]
    depth = mm.astype(f32) * f32(1.0 / 1000.0)  # :243,249
    return intensity.astype(f32), depth.astype(f32), mm.astype(np.uint16), color


def bilateral_mm(mm, max_d=4.5):
    """depth_bilateral.frag:34-74 on a (rows, cols) uint16 image -> uint16."""
    rows, cols = mm.shape
    gate_hi = int(f32(max_d) * f32(1000.0))
    val = mm.astype(f32)
    centre_ok = ~((mm.astype(np.int64) > gate_hi) | (mm < 300))
    s_space, s_color = f32(0.024691358), f32(0.000555556)
    sum1 = np.zeros((rows, cols), f32)
    sum2 = np.zeros((rows, cols), f32)
    yy, xx = np.mgrid[0:rows, 0:cols]
    for dy in range(-6, 7):  # cy = y + dy ascending (:57)
        for dx in range(-6, 7):  # cx = x + dx ascending (:59)
            cy, cx = yy + dy, xx + dx
            inside = (cy >= 0) & (cy < rows) & (cx >= 0) & (cx < cols)
            tmp = np.where(inside, val[np.clip(cy, 0, rows - 1), np.clip(cx, 0, cols - 1)], f32(0.0)).astype(f32)
            fdx = xx.astype(f32) - cx.astype(f32)
            fdy = yy.astype(f32) - cy.astype(f32)
            space2 = fdx * fdx + fdy * fdy  # :66
            dc = val - tmp
            color2 = dc * dc  # :67
            w = exp_neg(space2 * s_space + color2 * s_color)  # :69
            sum1 = np.where(inside, sum1 + tmp * w, sum1).astype(f32)  # :71
            sum2 = np.where(inside, sum2 + w, sum2).astype(f32)  # :72
    with np.errstate(invalid="ignore", divide="ignore"):
        q = (sum1 / sum2).astype(f32)
    # GLSL round(): halves away from zero for the non-negative value here
    rounded = np.floor(q.astype(np.float64) + 0.5)
    out = np.where(centre_ok, rounded, 0.0)
    return out.astype(np.uint16)


def metricise(mm, max_d=4.5):
    gate_hi = int(f32(max_d) * f32(1000.0))
    bad = (mm.astype(np.int64) > gate_hi) | (mm < 300)
    return np.where(bad, f32(0.0), mm.astype(f32) / f32(1000.0)).astype(f32)


def synth_frame(full_rows, full_cols, seed):
    """A decoded RGB-D frame with depth structure, sensor noise, holes, near and far out-of-range pixels."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:full_rows, 0:full_cols]
    depth = 1800.0 + 900.0 * np.sin(xx / (0.11 * full_cols)) + 500.0 * np.cos(yy / (0.09 * full_rows))
    depth[yy > 0.7 * full_rows] += 1600.0  # a depth edge
    depth += rng.normal(0.0, 6.0, depth.shape)
    depth = np.clip(depth, 0, 65535).astype(np.uint16)
    depth[rng.random(depth.shape) < 0.02] = 0  # dropouts
    depth[: full_rows // 12, : full_cols // 8] = 250  # closer than the 300 mm gate
    depth[-full_rows // 10 :, -full_cols // 9 :] = 6000  # beyond the 4.5 m cut-off
    color = rng.integers(0, 256, (full_rows, full_cols, 3), dtype=np.uint8)
    return color, depth


def main():
    check_fma32()
    out_dir = os.path.join(ROOT, "tests", "golden")
    color, depth = synth_frame(120, 160, seed=2024)  # -> 60 x 80 at res_factor 2
    inten, d0, mm, col = load_frame(color, depth, 2)
    filt = bilateral_mm(mm)
    np.savez_compressed(os.path.join(out_dir, "input_stage_80x60.npz"), color_full=color, depth_full=depth, res_factor=2, intensity=inten,
                        depth_loaded=d0, depth_mm=mm, color=col, filtered_mm=filt, depth_metric=metricise(mm),
                        depth_current=metricise(filt))
    a = np.concatenate([np.linspace(0, 90, 4001), np.array([0.0, 1e-8, 0.34657, 0.34658, 87.0, 87.000001, 100.0])]).astype(f32)
    np.savez_compressed(os.path.join(out_dir, "input_stage_exp.npz"), a=a, exp_neg=exp_neg(a))
    print("wrote input_stage fixtures; filtered != raw on", int((filt != mm).sum()), "of", mm.size, "pixels")


if __name__ == "__main__":
    main()
