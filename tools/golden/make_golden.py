"""Independent NumPy (float32) re-derivation of the per-pixel stages of the solver path, used to
emit the golden fixtures in tests/golden/.

This file was written from the reference's formulas (file:line cited per function), NOT from the
C++ oracle or the HIP kernels, and shares no code with them: array-at-a-time NumPy float32
arithmetic (IEEE single, no FMA contraction), so per-pixel stages can be compared bit for bit and
the reduced quantities (normal equations, first IRLS solution) to float64 accuracy.

Stages derived here:
  pyramid_level      createImagePyramid            reference FrontEnd.cpp:296-388
  linearise_first    calculateCoord + calculateDerivatives + computeWeights for the FIRST outer
                     iteration (Warped := Pred)    reference FrontEnd.cpp:393-510, 1103-1110
  jacobian_rows      rows of A and B               reference FrontEnd.cpp:539-586
  irls_first         first IRLS iteration with b == 1: Cauchy weights, AtA, AtB, solution
                                                   reference FrontEnd.cpp:588-642
  segm_image         buildSegmImage                reference SegmentationBackground.cpp:176-197

Run (in the build container):  python tools/golden/make_golden.py   -> tests/golden/*.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

f32 = np.float32
FOVH = f32(np.pi * 62.5 / 180.0)


def tan_half_fovh():
    # the reference calls the float overload of tan() from libm (FrontEnd.cpp:378 with `using namespace std`);
    # numpy's float32 tan is a different implementation and may differ in the last bit
    import ctypes

    libm = ctypes.CDLL("libm.so.6")
    libm.tanf.restype = ctypes.c_float
    libm.tanf.argtypes = [ctypes.c_float]
    return f32(libm.tanf(ctypes.c_float(float(f32(0.5) * FOVH))))


# --------------------------------------------------------------------------------------------
def pyramid_level(depth_prev, inten_prev):
    """One level of createImagePyramid (FrontEnd.cpp:296-374). Inputs (rows, cols) float32."""
    rp, cp = depth_prev.shape
    rows, cols = rp // 2, cp // 2
    mask1 = np.array([1, 2, 2, 1], dtype=f32)
    conv = (mask1[:, None] * mask1[None, :] / f32(36.0)).astype(f32)  # convMask(i,j) = v(i)*v(j)/36   (:146-149)
    d_out = np.zeros((rows, cols), f32)
    i_out = np.zeros((rows, cols), f32)

    # ---- inner pixels: 4x4 block at (2v-1, 2u-1)
    v = np.arange(1, rows - 1)
    u = np.arange(1, cols - 1)
    V, U = np.meshgrid(v, u, indexing="ij")
    blk_d = np.stack([[depth_prev[2 * V - 1 + r, 2 * U - 1 + c] for c in range(4)] for r in range(4)])  # [r][c]
    blk_i = np.stack([[inten_prev[2 * V - 1 + r, 2 * U - 1 + c] for c in range(4)] for r in range(4)])
    # central 2x2 in column-major order: (1,1), (2,1), (1,2), (2,2)   (:311)
    d0, d1, d2, d3 = blk_d[1, 1].copy(), blk_d[2, 1].copy(), blk_d[1, 2].copy(), blk_d[2, 2].copy()
    sw = d1 < d0
    d0, d1 = np.where(sw, d1, d0), np.where(sw, d0, d1)
    sw = d3 < d2
    d2, d3 = np.where(sw, d3, d2), np.where(sw, d2, d3)
    dcenter = np.where(d3 < d1, np.maximum(d3, d0), np.maximum(d1, d2)).astype(f32)  # second largest (:315-317)

    sum_d = np.zeros_like(dcenter)
    sum_c = np.zeros_like(dcenter)
    weight = np.zeros_like(dcenter)
    for k in range(16):  # column-major walk of the block (:323-334)
        r, c = k % 4, k // 4
        abs_dif = np.abs(blk_d[r, c] - dcenter)
        take = abs_dif < f32(0.1)
        aux_w = conv[r, c] * (f32(0.1) - abs_dif)
        weight = np.where(take, weight + aux_w, weight)
        sum_d = np.where(take, sum_d + aux_w * blk_d[r, c], sum_d)
        sum_c = np.where(take, sum_c + aux_w * blk_i[r, c], sum_c)
    with np.errstate(divide="ignore", invalid="ignore"):
        d_in = sum_d / weight
        i_in = sum_c / weight
    # zero-depth centre: plain mask-weighted intensity, Eigen SSE packet order (DESIGN.md [C2])
    m = [[conv[r, c] * blk_i[r, c] for c in range(4)] for r in range(4)]
    lane = [(m[r][0] + m[r][1]) + (m[r][2] + m[r][3]) for r in range(4)]
    i_zero = (lane[0] + lane[2]) + (lane[1] + lane[3])
    nz = dcenter != 0
    d_out[1:-1, 1:-1] = np.where(nz, d_in, f32(0))
    i_out[1:-1, 1:-1] = np.where(nz, i_in, i_zero)

    # ---- boundary pixels: 2x2 block at (2v, 2u)   (:347-373)
    border = np.ones((rows, cols), bool)
    border[1:-1, 1:-1] = False
    vb, ub = np.nonzero(border)
    b = [depth_prev[2 * vb + r, 2 * ub + c] for (r, c) in ((0, 0), (1, 0), (0, 1), (1, 1))]  # k = r + 2c
    bi = [inten_prev[2 * vb + r, 2 * ub + c] for (r, c) in ((0, 0), (1, 0), (0, 1), (1, 1))]
    i_out[vb, ub] = f32(0.25) * ((bi[0] + bi[2]) + (bi[1] + bi[3]))
    new_d = np.zeros(len(vb), f32)
    cont = np.zeros(len(vb), np.int32)
    for k in range(4):
        take = b[k] != 0
        new_d = np.where(take, new_d + b[k], new_d)
        cont = cont + take
    with np.errstate(divide="ignore", invalid="ignore"):
        d_out[vb, ub] = np.where(cont != 0, new_d / cont.astype(f32), f32(0))
    return d_out, i_out


def xx_yy(depth):
    """(:378-388)  xx = (inv_f*(u - disp_u))*depth, yy = (inv_f*(v - disp_v))*depth"""
    rows, cols = depth.shape
    inv_f = f32(2.0) * tan_half_fovh() / f32(cols)
    disp_u, disp_v = f32(0.5) * f32(cols - 1), f32(0.5) * f32(rows - 1)
    xs = inv_f * (np.arange(cols, dtype=f32) - disp_u)
    ys = inv_f * (np.arange(rows, dtype=f32) - disp_v)
    return (xs[None, :] * depth).astype(f32), (ys[:, None] * depth).astype(f32)


# --------------------------------------------------------------------------------------------
def linearise_first(d_new, i_new, d_old, i_old):
    """First outer iteration of a level: Warped := Pred (FrontEnd.cpp:1103-1110), then
    calculateCoord (:393-430), calculateDerivatives (:432-479), computeWeights (:481-510)."""
    rows, cols = d_new.shape
    xx_n, yy_n = xx_yy(d_new)
    xx_w, yy_w = xx_yy(d_old)
    null = ~((d_new != 0) & (d_old != 0))
    half = f32(0.5)
    d_int = np.where(null, f32(0), half * (d_new + d_old)).astype(f32)
    x_int = np.where(null, f32(0), half * (xx_n + xx_w)).astype(f32)
    y_int = np.where(null, f32(0), half * (yy_n + yy_w)).astype(f32)
    i_int = (half * (i_new + i_old)).astype(f32)
    valid = ~null
    valid[0, :] = valid[-1, :] = False
    valid[:, 0] = valid[:, -1] = False

    eps_i, eps_d = f32(1e-6), f32(0.005)
    rx = np.ones((rows, cols), f32)
    rxi = np.ones((rows, cols), f32)
    ry = np.ones((rows, cols), f32)
    ryi = np.ones((rows, cols), f32)
    nn = ~null
    rx[:, :-1] = np.where(nn[:, :-1], np.abs(d_int[:, 1:] - d_int[:, :-1]) + eps_d, f32(1))
    rxi[:, :-1] = np.where(nn[:, :-1], np.abs(i_int[:, 1:] - i_int[:, :-1]) + eps_i, f32(1))
    ry[:-1, :] = np.where(nn[:-1, :], np.abs(d_int[1:, :] - d_int[:-1, :]) + eps_d, f32(1))
    ryi[:-1, :] = np.where(nn[:-1, :], np.abs(i_int[1:, :] - i_int[:-1, :]) + eps_i, f32(1))

    def ctr(a):
        return a[1:-1, 1:-1]

    dcu = np.zeros((rows, cols), f32)
    dcv = np.zeros((rows, cols), f32)
    ddu = np.zeros((rows, cols), f32)
    ddv = np.zeros((rows, cols), f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        dcu[1:-1, 1:-1] = (rxi[1:-1, :-2] * (i_int[1:-1, 2:] - ctr(i_int)) + ctr(rxi) * (ctr(i_int) - i_int[1:-1, :-2])) / (ctr(rxi) + rxi[1:-1, :-2])
        ddu[1:-1, 1:-1] = (rx[1:-1, :-2] * (d_int[1:-1, 2:] - ctr(d_int)) + ctr(rx) * (ctr(d_int) - d_int[1:-1, :-2])) / (ctr(rx) + rx[1:-1, :-2])
        dcv[1:-1, 1:-1] = (ryi[:-2, 1:-1] * (i_int[2:, 1:-1] - ctr(i_int)) + ctr(ryi) * (ctr(i_int) - i_int[:-2, 1:-1])) / (ctr(ryi) + ryi[:-2, 1:-1])
        ddv[1:-1, 1:-1] = (ry[:-2, 1:-1] * (d_int[2:, 1:-1] - ctr(d_int)) + ctr(ry) * (ctr(d_int) - d_int[:-2, 1:-1])) / (ctr(ry) + ry[:-2, 1:-1])
    for a in (dcu, dcv, ddu, ddv):
        a[~valid] = 0
    dct = (i_new - i_old).astype(f32)
    ddt = (d_new - d_old).astype(f32)

    err_c = f32(10.0) * (np.abs(dct) + np.abs(dcu) + np.abs(dcv))
    err_d = f32(200.0) * (np.abs(ddt) + np.abs(ddu) + np.abs(ddv))
    wc = np.where(valid, np.sqrt(f32(1.0) / (f32(1.0) + err_c)), f32(0)).astype(f32)
    wd = np.where(valid, np.sqrt(f32(1.0) / (f32(0.01) + err_d)), f32(0)).astype(f32)
    wc = (f32(1.0) / wc.max()) * wc
    wd = (f32(1.0) / wd.max()) * wd
    return dict(null=null, valid=valid, d_int=d_int, x_int=x_int, y_int=y_int, i_int=i_int, dcu=dcu, dcv=dcv, dct=dct,
                ddu=ddu, ddv=ddv, ddt=ddt, wc=wc.astype(f32), wd=wd.astype(f32))


def jacobian_rows(lin, cols, k_photometric_res=f32(0.15)):
    """A (2N x 6) and B (2N) in validPixels (column-major) order, FrontEnd.cpp:539-586."""
    valid = lin["valid"]
    order = np.argwhere(valid.T)  # column-major: u outer, v inner
    u, v = order[:, 0], order[:, 1]
    g = lambda a: a[v, u]
    d, x, y = g(lin["d_int"]), g(lin["x_int"]), g(lin["y_int"])
    f = f32(cols) / (f32(2.0) * tan_half_fovh())
    inv_d = f32(1.0) / d
    rows_out = []
    for (gu, gv, gt, tw, depth_row) in ((g(lin["dcu"]), g(lin["dcv"]), g(lin["dct"]), g(lin["wc"]) * k_photometric_res, False),
                                        (g(lin["ddu"]), g(lin["ddv"]), g(lin["ddt"]), g(lin["wd"]), True)):
        dy = gu * f * inv_d
        dz = gv * f * inv_d
        one = f32(1.0) if depth_row else f32(0.0)
        a0 = tw * (-dy)
        a1 = tw * (-dz)
        if depth_row:
            a2 = tw * (f32(1.0) + dy * x * inv_d + dz * y * inv_d)
            a3 = tw * (y + dy * inv_d * y * x + dz * (y * y * inv_d + d))
            a4 = tw * (-x - dy * (x * x * inv_d + d) - dz * inv_d * y * x)
        else:
            a2 = tw * (dy * x * inv_d + dz * y * inv_d)
            a3 = tw * (dy * inv_d * y * x + dz * (y * y * inv_d + d))
            a4 = tw * (-dy * (x * x * inv_d + d) - dz * inv_d * y * x)
        a5 = tw * (dy * y - dz * x)
        bb = tw * (-gt)
        rows_out.append((np.stack([a0, a1, a2, a3, a4, a5], axis=1).astype(f32), bb.astype(f32)))
    N = len(u)
    A = np.zeros((2 * N, 6), f32)
    B = np.zeros(2 * N, f32)
    A[0::2], B[0::2] = rows_out[0]
    A[1::2], B[1::2] = rows_out[1]
    return A, B


def irls_first(A, B, kc=f32(0.5)):
    """First IRLS iteration with b == 1 (FrontEnd.cpp:588-642): res = -B, aver_res = mean|res|,
    Cauchy weights, AtA / AtB (float64 accumulation of the float32 weighted rows), solution."""
    res = -B
    aver_res = f32(np.abs(res).astype(np.float64).sum()) / f32(len(res))
    inv_c = f32(1.0) / (kc * aver_res)
    w = (f32(1.0) * np.sqrt(f32(1.0) / (f32(1.0) + (res * inv_c) ** 2))).astype(f32)
    Aw = (w[:, None] * A).astype(f32)
    Bw = (w * B).astype(f32)
    AtA = (Aw.astype(np.float64).T @ Aw.astype(np.float64))
    AtB = (Aw.astype(np.float64).T @ Bw.astype(np.float64))
    AtA32, AtB32 = AtA.astype(f32), AtB.astype(f32)
    var = np.linalg.solve(AtA32.astype(np.float64), AtB32.astype(np.float64))
    return dict(aver_res=aver_res, AtA=AtA32, AtB=AtB32, var=var.astype(f32))


def segm_image(labels0, b_segm, cluster_res):
    """buildSegmImage (SegmentationBackground.cpp:176-197)."""
    b = np.clip(np.append(b_segm, f32(1.0))[np.minimum(labels0, 24)], 0, 1).astype(f32)
    res = np.append(cluster_res, np.nan)[np.minimum(labels0, 24)]
    flip = res.astype(np.float64) < 0.017  # NaN compares false
    b = np.where(flip, np.maximum(b, f32(1.0) - b), b)
    return np.where(labels0 == 24, f32(1.0), b).astype(f32)


# --------------------------------------------------------------------------------------------
def main():
    from staticfusion_amd.synth import make_pair

    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    # 160x120 input with a sphere and a strip of invalid depth, so that the zero-centre and the
    # "mean of non-zero depths" branches of the pyramid are exercised
    pr = make_pair(seed=4242, sphere=True, out_rows=120, out_cols=160)
    d_new, i_new = pr["new"]
    d_old, i_old = pr["old"]
    d_new = d_new.copy()
    d_old = d_old.copy()
    d_new[10:40, 100:130] = 0
    d_new[60:63, :] = 0
    d_old[80:100, 20:50] = 0
    lv = {"d_new0": d_new, "i_new0": i_new, "d_old0": d_old, "i_old0": i_old}
    for L in (1, 2):
        lv["d_new%d" % L], lv["i_new%d" % L] = pyramid_level(lv["d_new%d" % (L - 1)], lv["i_new%d" % (L - 1)])
        lv["d_old%d" % L], lv["i_old%d" % L] = pyramid_level(lv["d_old%d" % (L - 1)], lv["i_old%d" % (L - 1)])
    for L in (0, 1, 2):
        lv["xx_new%d" % L], lv["yy_new%d" % L] = xx_yy(lv["d_new%d" % L])
    np.savez_compressed(os.path.join(out, "pyramid_160x120.npz"), **lv)

    # first outer iteration at the coarsest of 3 levels (40x30): linearisation, rows, first IRLS step
    lin = linearise_first(lv["d_new2"], lv["i_new2"], lv["d_old2"], lv["i_old2"])
    A, B = jacobian_rows(lin, cols=lv["d_new2"].shape[1])
    first = irls_first(A, B)
    np.savez_compressed(os.path.join(out, "linearise_40x30.npz"), A=A, B=B, n_valid=np.int32(len(B) // 2),
                        **{k: (v.astype(np.uint8) if v.dtype == bool else v) for k, v in lin.items()},
                        **{"irls_" + k: v for k, v in first.items()})

    # buildSegmImage on a synthetic label image
    rng = np.random.RandomState(7)
    labels0 = rng.randint(0, 25, size=(120, 160)).astype(np.int32)
    b_segm = rng.uniform(-1, 2, size=24).astype(f32)
    cres = rng.uniform(0, 0.04, size=24).astype(f32)
    cres[[3, 11]] = np.nan
    np.savez_compressed(os.path.join(out, "segm_image_160x120.npz"), labels0=labels0, b_segm=b_segm, cluster_res=cres,
                        b_image=segm_image(labels0, b_segm, cres))
    print("golden fixtures written to", out, [f for f in os.listdir(out)])


if __name__ == "__main__":
    main()
