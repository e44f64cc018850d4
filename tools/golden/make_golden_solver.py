"""Independent NumPy derivation of the WHOLE coarse-to-fine solve, used to emit
tests/golden/solver_160x120.npz: StaticFusion::runSolver on one 160 x 120 frame pair, every outer iteration traced.

Written from the reference's FrontEnd.cpp / SegmentationBackground.cpp, not from the C++ oracle or the HIP kernels, and
sharing no code with them. Whole-image NumPy float32 arithmetic for the per-pixel formulas; every cross-pixel sum, linear
solve, eigen-decomposition and matrix exponential / logarithm in float64 through NumPy / SciPy (the reference uses float
Eigen / MRPT code whose internal order is not visible) -- so the fixture is compared with a tolerance (twists to 2e-6, poses
to 2e-6, b to 1e-4) and exactly on everything discrete (valid-pixel and iteration counts, how many outer iterations run).

  warp              StaticFusion::warpImagesAccurateInverse    reference FrontEnd.cpp:792-890
  linearise         calculateCoord / calculateDerivatives / computeWeights   :393-510
  seg_prior         StaticFusion::computeSegPrior              SegmentationBackground.cpp:53-95
  joint_solve       StaticFusion::solveOdometryAndSegmJoint    FrontEnd.cpp:512-690 (+ buildSystemSegm, solveSegmIteration:
                                                               SegmentationBackground.cpp:97-170; Jacobian rows: make_golden.py)
  filter_and_update StaticFusion::filterEstimateAndComputeT    FrontEnd.cpp:712-762
  run_solver        StaticFusion::runSolver                    FrontEnd.cpp:1071-1146
  residuals_vs_previous  StaticFusion::computeResidualsAgainstPreviousImage  FrontEnd.cpp:896-1069

The pyramid and the cluster labels / connectivity of the frame come from the independent derivations make_golden.py and
make_golden_kmeans.py (fixtures pyramid_160x120.npz, kmeans_160x120.npz: the same `new` frame).

Run (in the build container):  python tools/golden/make_golden_solver.py  -> tests/golden/solver_160x120.npz
"""
import os
import sys

import numpy as np
from scipy.linalg import expm, logm

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import jacobian_rows, tan_half_fovh, xx_yy  # noqa: E402

f32 = np.float32
NC = 24
# the drivers' parameter block (StaticFusion-imagesequenceassoc.cpp:62-79) and the constructor's kb for a sparse map (:123)
P = dict(ctf_levels=3, max_iter_per_level=3, max_iter_irls=6, k_photometric_res=f32(0.15), irls_delta_threshold=f32(0.0015), kc_Cauchy=f32(0.5),
         kb=f32(1.05), kz=f32(1.5), lambda_reg=f32(0.35), lambda_prior=f32(0.5), use_motion_filter=True,
         previous_speed_const_weight=f32(0.1), previous_speed_eig_weight=f32(2.0))


def warp(d_pred, i_pred, xx_pred, yy_pred, T_odometry):
    """forward splat of the prediction with T_odometry^-1 (centi-pixel bilinear weights), then normalisation"""
    rows, cols = d_pred.shape
    f = f32(cols) / (f32(2.0) * tan_half_fovh())
    disp_u, disp_v = f32(0.5) * f32(cols - 1), f32(0.5) * f32(rows - 1)
    T = np.linalg.inv(T_odometry.astype(np.float64)).astype(f32)
    z = d_pred
    x_w = T[0, 0] * xx_pred + T[0, 1] * yy_pred + T[0, 2] * z + T[0, 3]
    y_w = T[1, 0] * xx_pred + T[1, 1] * yy_pred + T[1, 2] * z + T[1, 3]
    d_w = T[2, 0] * xx_pred + T[2, 1] * yy_pred + T[2, 2] * z + T[2, 3]
    with np.errstate(all="ignore"):
        uf = f32(100.0) * (f * x_w / d_w + disp_u)
        vf = f32(100.0) * (f * y_w / d_w + disp_v)
    ok = (z != 0) & np.isfinite(uf) & np.isfinite(vf) & (np.abs(uf) < 2e9) & (np.abs(vf) < 2e9)
    uw = np.where(ok, uf, 0).astype(np.int64)  # int(): truncation towards zero
    vw = np.where(ok, vf, 0).astype(np.int64)
    ok &= (uw >= 0) & (uw < 100 * (cols - 1)) & (vw >= 0) & (vw < 100 * (rows - 1))
    acc_d = np.zeros((rows, cols), np.float64)
    acc_i = np.zeros((rows, cols), np.float64)
    wacu = np.zeros((rows, cols), np.float64)
    uw, vw, dw, iw = uw[ok], vw[ok], d_w[ok].astype(np.float64), i_pred[ok].astype(np.float64)
    ul = uw - uw % 100
    vd = vw - vw % 100
    delta_r, delta_u = ul + 100 - uw, vd + 100 - vw
    delta_l, delta_d = 100 - delta_r, 100 - delta_u
    near = np.minimum(delta_r, delta_l) + np.minimum(delta_u, delta_d) < 5
    iu = np.where(delta_r > delta_l, ul // 100, (ul + 100) // 100)
    iv = np.where(delta_u > delta_d, vd // 100, (vd + 100) // 100)
    for sel, vv, uu, w in ((near, iv, iu, np.full(len(uw), 200)),
                           (~near, vd // 100 + 1, ul // 100 + 1, delta_l + delta_d), (~near, vd // 100 + 1, ul // 100, delta_r + delta_d),
                           (~near, vd // 100, ul // 100 + 1, delta_l + delta_u), (~near, vd // 100, ul // 100, delta_r + delta_u)):
        np.add.at(acc_d, (vv[sel], uu[sel]), w[sel] * dw[sel])
        np.add.at(acc_i, (vv[sel], uu[sel]), w[sel] * iw[sel])
        np.add.at(wacu, (vv[sel], uu[sel]), w[sel])
    hit = wacu != 0
    with np.errstate(all="ignore"):
        d_out = np.where(hit, acc_d / wacu, 0).astype(f32)
        i_out = np.where(hit, acc_i / wacu, 0).astype(f32)
    inv_f = f32(1.0) / f
    us = (np.arange(cols, dtype=f32) - disp_u)[None, :]
    vs = (np.arange(rows, dtype=f32) - disp_v)[:, None]
    xx = np.where(hit, us * d_out * inv_f, 0).astype(f32)
    yy = np.where(hit, vs * d_out * inv_f, 0).astype(f32)
    return d_out, i_out, xx, yy


def linearise(d_new, i_new, xx_new, yy_new, d_w, i_w, xx_w, yy_w):
    """calculateCoord + calculateDerivatives + computeWeights for the warped images (d_w, i_w, xx_w, yy_w)"""
    rows, cols = d_new.shape
    null = ~((d_new != 0) & (d_w != 0))
    half = f32(0.5)
    d_int = np.where(null, f32(0), half * (d_new + d_w)).astype(f32)
    x_int = np.where(null, f32(0), half * (xx_new + xx_w)).astype(f32)
    y_int = np.where(null, f32(0), half * (yy_new + yy_w)).astype(f32)
    i_int = (half * (i_new + i_w)).astype(f32)
    valid = ~null
    valid[0, :] = valid[-1, :] = False
    valid[:, 0] = valid[:, -1] = False
    eps_i, eps_d = f32(1e-6), f32(0.005)
    rx, rxi, ry, ryi = [np.ones((rows, cols), f32) for _ in range(4)]
    nn = ~null
    rx[:, :-1] = np.where(nn[:, :-1], np.abs(d_int[:, 1:] - d_int[:, :-1]) + eps_d, f32(1))
    rxi[:, :-1] = np.where(nn[:, :-1], np.abs(i_int[:, 1:] - i_int[:, :-1]) + eps_i, f32(1))
    ry[:-1, :] = np.where(nn[:-1, :], np.abs(d_int[1:, :] - d_int[:-1, :]) + eps_d, f32(1))
    ryi[:-1, :] = np.where(nn[:-1, :], np.abs(i_int[1:, :] - i_int[:-1, :]) + eps_i, f32(1))
    c = lambda a: a[1:-1, 1:-1]
    dcu, dcv, ddu, ddv = [np.zeros((rows, cols), f32) for _ in range(4)]
    with np.errstate(all="ignore"):
        dcu[1:-1, 1:-1] = (rxi[1:-1, :-2] * (i_int[1:-1, 2:] - c(i_int)) + c(rxi) * (c(i_int) - i_int[1:-1, :-2])) / (c(rxi) + rxi[1:-1, :-2])
        ddu[1:-1, 1:-1] = (rx[1:-1, :-2] * (d_int[1:-1, 2:] - c(d_int)) + c(rx) * (c(d_int) - d_int[1:-1, :-2])) / (c(rx) + rx[1:-1, :-2])
        dcv[1:-1, 1:-1] = (ryi[:-2, 1:-1] * (i_int[2:, 1:-1] - c(i_int)) + c(ryi) * (c(i_int) - i_int[:-2, 1:-1])) / (c(ryi) + ryi[:-2, 1:-1])
        ddv[1:-1, 1:-1] = (ry[:-2, 1:-1] * (d_int[2:, 1:-1] - c(d_int)) + c(ry) * (c(d_int) - d_int[:-2, 1:-1])) / (c(ry) + ry[:-2, 1:-1])
    for a in (dcu, dcv, ddu, ddv):
        a[~valid] = 0
    dct, ddt = (i_new - i_w).astype(f32), (d_new - d_w).astype(f32)
    err_c = f32(10.0) * (np.abs(dct) + np.abs(dcu) + np.abs(dcv))
    err_d = f32(200.0) * (np.abs(ddt) + np.abs(ddu) + np.abs(ddv))
    wc = np.where(valid, np.sqrt(f32(1.0) / (f32(1.0) + err_c)), f32(0)).astype(f32)
    wd = np.where(valid, np.sqrt(f32(1.0) / (f32(0.01) + err_d)), f32(0)).astype(f32)
    wc = ((f32(1.0) / wc.max()) * wc).astype(f32)
    wd = ((f32(1.0) / wd.max()) * wd).astype(f32)
    return dict(null=null, valid=valid, d_int=d_int, x_int=x_int, y_int=y_int, i_int=i_int, dcu=dcu, dcv=dcv, dct=dct, ddu=ddu, ddv=ddv,
                ddt=ddt, wc=wc, wd=wd)


def seg_prior(labels, null, ddt):
    b_prior = np.zeros(NC, f32)
    lam = np.zeros(NC, f32)
    for l in range(NC):
        m = labels == l
        size = int(m.sum())
        if size == 0:
            continue
        nn = m & ~null
        nonnull = int(nn.sum())
        ratio = f32(nonnull) / f32(size)
        if ratio < f32(0.1):
            lam[l], b_prior[l] = f32(0.1), f32(-1)
        else:
            s = f32((f32(1.0) - P["kz"] * np.abs(ddt[nn])).astype(np.float64).sum())
            lam[l] = ratio
            b_prior[l] = max(f32(-1), min(f32(2), s / f32(nonnull)))
    return b_prior, lam


def segm_iteration(A_seg, aver_res_label, aver_res_overall, lam, b_prior):
    kc, kb = P["kc_Cauchy"], P["kb"]
    repr_res = max(f32(0.001), aver_res_overall)
    fixed = f32(np.log(f32(1.0) + (kb * repr_res / (kc * aver_res_overall)) ** 2))
    mult = f32(1.0) / (kc * aver_res_overall)
    A = A_seg.copy()
    B = np.zeros(A.shape[0], f32)
    for l in range(NC):
        if lam[l] > f32(0.1):
            data = fixed - f32(np.log(f32(1.0) + (aver_res_label[l] * mult) ** 2))
            A[l, l] = f32(2.0) * lam[l] * P["lambda_prior"]
            B[l] = data + f32(2.0) * P["lambda_prior"] * lam[l] * b_prior[l]
        else:
            A[l, l] = f32(2.0) * lam[l]
            B[l] = f32(2.0) * lam[l] * b_prior[l]
    AtA = (A.astype(np.float64).T @ A.astype(np.float64)).astype(f32)
    AtB = (A.astype(np.float64).T @ B.astype(np.float64)).astype(f32)
    b = np.linalg.solve(AtA.astype(np.float64), AtB.astype(np.float64)).astype(f32)
    return np.clip(b, f32(-1), f32(2)).astype(f32)


def joint_solve(lin, labels, conn, level, b_segm, b_prior, lam, cols):
    A, B = jacobian_rows(lin, cols, P["k_photometric_res"])
    order = np.argwhere(lin["valid"].T)                        # validPixels: u outer, v inner
    lab = labels[order[:, 1], order[:, 0]]
    n = len(lab)
    pairs = [(l, lc) for l in range(NC) for lc in range(l + 1, NC) if conn[l, lc]]   # buildSystemSegm
    A_seg = np.zeros((NC + len(pairs), NC), f32)
    for r, (l, lc) in enumerate(pairs):
        A_seg[NC + r, l], A_seg[NC + r, lc] = f32(2.0) * P["lambda_reg"], -f32(2.0) * P["lambda_reg"]
    res = -B
    aver_res = f32(np.abs(res).astype(np.float64).sum()) / f32(len(res))
    var = np.zeros(6, f32)
    prev = var.copy()
    if level == 0:
        b_segm = b_prior.copy()
    iters = 0
    AtA = None
    for k in range(1, P["max_iter_irls"] + 1):
        iters += 1
        inv_c = f32(1.0) / (P["kc_Cauchy"] * aver_res)
        bw = np.repeat(np.clip(b_segm[lab], 0, 1).astype(f32), 2)
        w = (bw * np.sqrt(f32(1.0) / (f32(1.0) + (res * inv_c) ** 2))).astype(f32)
        Aw, Bw = (w[:, None] * A).astype(f32), (w * B).astype(f32)
        AtA = (Aw.astype(np.float64).T @ Aw.astype(np.float64)).astype(f32)
        AtB = (Aw.astype(np.float64).T @ Bw.astype(np.float64)).astype(f32)
        var = np.linalg.solve(AtA.astype(np.float64), AtB.astype(np.float64)).astype(f32)
        res = (-B).astype(f32)
        for q in range(6):
            res = (res + var[q] * A[:, q]).astype(f32)
        ress = (np.abs(res[0::2]) + np.abs(res[1::2])).astype(f32)
        label_sum = np.array([ress[lab == l].astype(np.float64).sum() for l in range(NC)]).astype(f32)
        num = np.array([1 + int((lab == l).sum()) for l in range(NC)])
        aver_old = aver_res
        aver_res = f32(label_sum.astype(np.float64).sum()) / f32(2 * n)
        aver_label = (label_sum / (2 * num).astype(f32)).astype(f32)
        b_segm = segm_iteration(A_seg, aver_label, aver_old, lam, b_prior)
        delta = np.abs(prev - var).max()
        prev = var.copy()
        if delta < P["irls_delta_threshold"] or k == P["max_iter_irls"]:
            break
    est_cov = (np.linalg.inv(AtA.astype(np.float64)) * float(f32((res.astype(np.float64) ** 2).sum()))).astype(f32)
    return var, b_segm, est_cov, aver_res, iters, n


def twist_to_matrix(t):
    m = np.zeros((4, 4), np.float64)
    m[0, 1], m[1, 0] = -t[5], t[5]
    m[0, 2], m[2, 0] = t[4], -t[4]
    m[1, 2], m[2, 1] = -t[3], t[3]
    m[0:3, 3] = t[0:3]
    return m


def matrix_to_twist(T):
    lg = np.real(logm(T.astype(np.float64)))
    return np.array([lg[0, 3], lg[1, 3], lg[2, 3], -lg[1, 2], lg[0, 2], -lg[0, 1]], f32)


def filter_and_update(twist, est_cov, level, T_odometry, twist_old):
    if P["use_motion_filter"]:
        evals, B = np.linalg.eigh(est_cov.astype(np.float64))
        kai_b = np.linalg.solve(B, twist.astype(np.float64))
        lg = np.real(logm(T_odometry.astype(np.float64)))
        sub = twist_old.astype(np.float64).copy()
        sub[0] -= lg[0, 3]; sub[1] -= lg[1, 3]; sub[2] -= lg[2, 3]
        sub[3] += lg[1, 2]; sub[4] -= lg[0, 2]; sub[5] += lg[0, 1]
        kai_b_old = np.linalg.solve(B, sub)
        cf = float(P["previous_speed_eig_weight"] * f32(np.exp(-float(level))))
        df = float(P["previous_speed_const_weight"] * f32(np.exp(-float(level))))
        fil = (kai_b + (cf * evals + df) * kai_b_old) / (1.0 + cf * evals + df)
        twist = (B @ fil).astype(f32)                                  # Bii.inverse().solve(x) = Bii x
    T = (expm(twist_to_matrix(twist.astype(np.float64))) @ T_odometry.astype(np.float64)).astype(f32)
    return twist.astype(f32), T


def run_solver(pyr_new, pyr_old, labels, conn):
    """pyr_*: lists of (depth, intensity) per level (0 = finest); labels per level; returns the outer-iteration traces"""
    L = P["ctf_levels"]
    T = np.eye(4, dtype=f32)
    twist_old = np.zeros(6, f32)
    b_segm = np.full(NC, f32(0.5))
    traces = []
    for i in range(L):
        il = L - i - 1
        d_new, i_new = pyr_new[il]
        d_old, i_old = pyr_old[il]
        rows, cols = d_new.shape
        xx_new, yy_new = xx_yy(d_new)
        xx_old, yy_old = xx_yy(d_old)
        for k in range(P["max_iter_per_level"]):
            if i == 0 and k == 0:
                d_w, i_w, xx_w, yy_w = d_old, i_old, xx_old, yy_old
            else:
                d_w, i_w, xx_w, yy_w = warp(d_old, i_old, xx_old, yy_old, T)
            lin = linearise(d_new, i_new, xx_new, yy_new, d_w, i_w, xx_w, yy_w)
            b_prior, lam = seg_prior(labels[il], lin["null"], lin["ddt"])
            var, b_segm, est_cov, aver_res, iters, n_valid = joint_solve(lin, labels[il], conn, i, b_segm, b_prior, lam, cols)
            twist_level, T = filter_and_update(var.copy(), est_cov, i, T, twist_old)
            traces.append(dict(level=i, k=k, n_valid=n_valid, irls_iters=iters, aver_res=aver_res, var=var, twist_level=twist_level,
                               b_segm=b_segm.copy(), T=T.copy(), b_prior=b_prior, lambda_t_w=lam))
            if f32(np.sqrt((twist_level.astype(np.float64) ** 2).sum())) < f32(0.04):
                break
    return traces, T, matrix_to_twist(T)


def residuals_vs_previous(d_buf, i_buf, d_cur, i_cur, odom, T_odometry, labels0):
    """the frame of five frames ago, warped into the current one with the product of the buffered odometries, against the
    current frame: mean |depth residual| + k |intensity residual| per cluster (NaN for clusters without a pixel)"""
    T = np.eye(4, dtype=np.float64)
    for Tb in odom:
        T = (T.astype(f32).astype(np.float64) @ Tb.astype(np.float64))
    T = (T.astype(f32).astype(np.float64) @ T_odometry.astype(np.float64)).astype(f32)
    xx_buf, yy_buf = xx_yy(d_buf)
    both = (d_buf != 0) & (d_cur != 0)
    d_w, i_w, _, _ = warp(np.where(both, d_buf, f32(0)).astype(f32), i_buf, xx_buf, yy_buf, T)
    i_diff = np.where(both, i_cur, f32(0)).astype(f32)
    cum = (np.abs(d_cur - d_w) + P["k_photometric_res"] * np.abs(i_diff - i_w)).astype(f32)
    sel = (d_w != 0) & (d_cur != 0)
    out = np.full(NC, np.nan, f32)
    for l in range(NC):
        m = sel & (labels0 == l)
        if m.any():
            out[l] = f32(cum[m].astype(np.float64).sum()) / f32(2 * (int(m.sum()) + 1))
    return out


def history_case():
    """six frames of a synthetic walk driven through the ORACLE (an input provider here: its poses and labels are stored as
    inputs); derived: the five-frame residuals of the last frame"""
    from oracle import binding
    import staticfusion_amd as sf
    from staticfusion_amd.synth import DEFAULT_XI, Scene, se3_exp

    binding.build()
    ora = binding.load()
    p = ora.default_params_struct()
    p.ctf_levels = 3
    s = sf.Solver(ora, 120, 160, 1, p)
    scene = Scene(seed=31, sphere=True)
    T, frames = np.eye(4), []
    for k in range(6):
        depth, inten = scene.render(T, 640, 480, sphere_offset=(0.03 * k, 0, 0))
        frames.append((depth[::4, ::4].astype(f32).copy(), inten[::4, ::4].astype(f32).copy()))
        T = T @ se3_exp(np.array(DEFAULT_XI) * 0.8)
    s.set_current(0, *frames[0])
    s.current_to_prediction()
    s.push_history(0)
    odo = []
    for k in range(1, 6):
        s.set_current(0, *frames[k])
        s.process_frame(k)
        odo.append(s.T().astype(f32))
        if k < 5:
            s.current_to_prediction()
    labels0 = s.labels(0)
    want = residuals_vs_previous(frames[0][0], frames[0][1], frames[5][0], frames[5][1], odo[:4], odo[4], labels0)
    return frames, odo, labels0, want, s.cluster_residuals()


def big_motion_case():
    """a second pair with four times the camera motion: the first outer iterations exceed the 0.04 twist norm, so that several
    outer iterations per level (and the warp between them) are exercised; pyramid and labels by the independent derivations"""
    import make_golden_kmeans as mk
    from make_golden import pyramid_level
    from staticfusion_amd.synth import DEFAULT_XI, make_pair

    pr = make_pair(seed=777, sphere=True, out_rows=120, out_cols=160, xi=tuple(4.0 * np.array(DEFAULT_XI)))
    pyr_new, pyr_old = [pr["new"]], [pr["old"]]
    for _ in (1, 2):
        pyr_new.append(pyramid_level(*pyr_new[-1]))
        pyr_old.append(pyramid_level(*pyr_old[-1]))
    depth = [p_[0] for p_ in pyr_new]
    xy = [xx_yy(d) for d in depth]
    labels1, centres0 = mk.initialise(depth[1])
    labels1, centres, _ = mk.lloyd(depth[1], xy[1][0], xy[1][1], labels1, centres0.copy())
    labels0 = mk.label_level0(depth[0], xy[0][0], xy[0][1], labels1, centres)
    conn = mk.connectivity(depth[0], xy[0][0], xy[0][1], labels0)
    labels2 = mk.label_pyramid(depth[2], xy[2][0], xy[2][1], centres)
    return pr, pyr_new, pyr_old, [labels0, labels1, labels2], conn


def pack(prefix, traces, T, twist, out):
    out[prefix + "n_outer"], out[prefix + "T"], out[prefix + "twist"] = np.int32(len(traces)), T, twist
    for key in ("level", "k", "n_valid", "irls_iters"):
        out[prefix + key] = np.array([t[key] for t in traces], np.int32)
    for key in ("aver_res", "var", "twist_level", "b_segm", "T", "b_prior", "lambda_t_w"):
        out[prefix + "trace_" + key] = np.stack([np.asarray(t[key], f32) for t in traces])
    for t in traces:
        print("%slevel %d k %d: %5d valid, %d IRLS iterations, |twist_level| %.5f, aver_res %.6f, b in [%.2f, %.2f]" %
              (prefix, t["level"], t["k"], t["n_valid"], t["irls_iters"], float(np.linalg.norm(t["twist_level"])), float(t["aver_res"]),
               float(t["b_segm"].min()), float(t["b_segm"].max())))


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "pyramid_160x120.npz"))
    km = np.load(os.path.join(ROOT, "tests", "golden", "kmeans_160x120.npz"))
    pyr_new = [(g["d_new%d" % L], g["i_new%d" % L]) for L in range(3)]
    pyr_old = [(g["d_old%d" % L], g["i_old%d" % L]) for L in range(3)]
    assert np.array_equal(km["depth0"], g["d_new0"])
    labels = [km["labels%d" % L] for L in range(3)]
    traces, T, twist = run_solver(pyr_new, pyr_old, labels, km["connectivity"])
    out = {}
    pack("", traces, T, twist, out)
    pr, pyr_new_b, pyr_old_b, labels_b, conn_b = big_motion_case()
    traces_b, T_b, twist_b = run_solver(pyr_new_b, pyr_old_b, labels_b, conn_b)
    pack("big_", traces_b, T_b, twist_b, out)
    out["big_d_new0"], out["big_i_new0"] = pr["new"]
    out["big_d_old0"], out["big_i_old0"] = pr["old"]
    frames, odo, labels0, res, res_oracle = history_case()
    out["hist_depth"], out["hist_intensity"] = np.stack([f_[0] for f_ in frames]), np.stack([f_[1] for f_ in frames])
    out["hist_T"], out["hist_labels0"], out["hist_cluster_res"] = np.stack(odo), labels0, res
    print("five-frame residuals per cluster:", np.round(res, 4).tolist())
    print("  (the oracle, for information: max difference %.2e)" % np.nanmax(np.abs(res - res_oracle)))
    path = os.path.join(ROOT, "tests", "golden", "solver_160x120.npz")
    np.savez_compressed(path, **out)
    print("twist", twist, "; wrote", path)


if __name__ == "__main__":
    main()
