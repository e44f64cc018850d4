"""The CPU oracle against the committed golden fixtures (tests/golden/*.npz), which come from an
INDEPENDENT NumPy float32 re-derivation of the reference's formulas (tools/golden/make_golden.py).
The reference itself holds no golden vectors."""
import os

import numpy as np

from conftest import GOLDEN, config2_params, driver_params, make_solver
from staticfusion_amd import capi


def check_pyramid(api):
    g = np.load(os.path.join(GOLDEN, "pyramid_160x120.npz"))
    s = make_solver(api, 120, 160, config2_params(api, levels=3))
    s.set_current(0, g["d_new0"], g["i_new0"])
    s.set_prediction(0, g["d_old0"], g["i_old0"])
    s.build_pyramid(True)
    s.build_pyramid(False)
    for L in (0, 1, 2):
        for name, pset, ch in (("d_new", capi.SET_NEW, capi.CH_DEPTH), ("i_new", capi.SET_NEW, capi.CH_INTENSITY),
                               ("d_old", capi.SET_PRED, capi.CH_DEPTH), ("i_old", capi.SET_PRED, capi.CH_INTENSITY),
                               ("xx_new", capi.SET_NEW, capi.CH_XX), ("yy_new", capi.SET_NEW, capi.CH_YY)):
            assert np.array_equal(s.plane(pset, ch, L), g["%s%d" % (name, L)]), (name, L)  # bit exact
    return s


def check_linearise_and_first_irls(api):
    g = np.load(os.path.join(GOLDEN, "pyramid_160x120.npz"))
    lin = np.load(os.path.join(GOLDEN, "linearise_40x30.npz"))
    # one IRLS iteration, one outer iteration per level: the trace of the first outer iteration (40x30)
    p = config2_params(api, levels=3, max_iter_irls=1, max_iter_per_level=1, debug_planes=1)
    s = make_solver(api, 120, 160, p)
    s.set_current(0, g["d_new0"], g["i_new0"])
    s.set_prediction(0, g["d_old0"], g["i_old0"])
    s.build_pyramid(True)
    s.run_solver(True)
    st = s.stats()
    t0 = st.outer[0]
    assert t0.level == 0 and t0.n_valid == int(lin["n_valid"]) and t0.irls_iters == 1
    assert np.allclose(np.array(t0.var[:]), lin["irls_var"], rtol=2e-5, atol=2e-8)
    # Inter planes of the coarsest level are those of its (only) outer iteration
    for name, ch in (("d_int", capi.CH_DEPTH), ("i_int", capi.CH_INTENSITY), ("x_int", capi.CH_XX), ("y_int", capi.CH_YY)):
        assert np.array_equal(s.plane(capi.SET_INTER, ch, 2), lin[name]), name
    return s, lin


def check_lin_planes_single_level(api, tol_rows=1e-6, tol_ata=2e-6):
    """Everything `linearise_40x30.npz` holds about the first outer iteration of a 40x30 level (Warped := Pred,
    FrontEnd.cpp:1103-1110), through a ONE-level solve on those images (ctf_levels = 1 is accepted without
    segmentation): Inter planes, Null, the six derivative planes and the normalised pre-weights bit for bit; the
    Jacobian rows A / B (FrontEnd.cpp:539-586) and the normal equations of the first IRLS iteration (:615-642) to
    rounding. tol_rows is relative to the largest entry of a column: the HIP path expands the rows from its factored
    form, which rounds in another association."""
    g = np.load(os.path.join(GOLDEN, "pyramid_160x120.npz"))
    lin = np.load(os.path.join(GOLDEN, "linearise_40x30.npz"))
    p = config2_params(api, levels=1, max_iter_irls=1, max_iter_per_level=1, debug_planes=1)
    s = make_solver(api, 30, 40, p)
    s.set_current(0, g["d_new2"], g["i_new2"])
    s.set_prediction(0, g["d_old2"], g["i_old2"])
    s.build_pyramid(True)
    s.run_solver(True)
    st = s.stats()
    t0 = st.outer[0]
    assert st.n_outer == 1 and t0.level == 0 and t0.n_valid == int(lin["n_valid"]) and t0.irls_iters == 1
    for name, ch in (("d_int", capi.CH_DEPTH), ("i_int", capi.CH_INTENSITY), ("x_int", capi.CH_XX), ("y_int", capi.CH_YY)):
        assert np.array_equal(s.plane(capi.SET_INTER, ch, 0), lin[name]), name
    assert np.array_equal(s.lin_plane(capi.LIN_NULL) != 0, lin["null"] != 0)
    for name, which in (("dcu", capi.LIN_DCU), ("dcv", capi.LIN_DCV), ("dct", capi.LIN_DCT), ("ddu", capi.LIN_DDU),
                        ("ddv", capi.LIN_DDV), ("ddt", capi.LIN_DDT), ("wc", capi.LIN_WC), ("wd", capi.LIN_WD)):
        assert np.array_equal(s.lin_plane(which), lin[name]), name
    A, B = s.jacobian_rows()
    assert A.shape == lin["A"].shape and B.shape == lin["B"].shape
    scale = np.abs(lin["A"]).max(axis=0)
    assert (np.abs(A - lin["A"]) / scale).max() < tol_rows, (np.abs(A - lin["A"]) / scale).max()
    assert np.abs(B - lin["B"]).max() < tol_rows * np.abs(lin["B"]).max()
    AtA, AtB = np.array(t0.AtA[:]).reshape(6, 6), np.array(t0.AtB[:])
    assert np.array_equal(AtA, AtA.T)
    dg = np.sqrt(np.diag(lin["irls_AtA"]).astype(np.float64))
    assert (np.abs(AtA - lin["irls_AtA"]) / np.outer(dg, dg)).max() < tol_ata  # relative to sqrt(a_ii a_jj)
    assert (np.abs(AtB - lin["irls_AtB"]) / (dg * np.abs(lin["irls_AtB"] / dg).max())).max() < tol_ata
    assert np.allclose(np.array(t0.var[:]), lin["irls_var"], rtol=2e-5, atol=2e-8)
    assert np.allclose(t0.aver_res, 0, atol=1) and np.isfinite(t0.aver_res)
    return s


def check_kmeans(api):
    """tests/golden/kmeans_160x120.npz: the independent Python derivation of KMeans.cpp (tools/golden/make_golden_kmeans.py).
    Labels of every level, centres, connectivity and the iteration count: exactly."""
    g = np.load(os.path.join(GOLDEN, "kmeans_160x120.npz"))
    s = make_solver(api, 120, 160, driver_params(api, ctf_levels=4))
    s.set_current(0, g["depth0"], g["intensity0"])
    s.build_pyramid(False)
    s.kmeans()
    for L in range(4):
        assert np.array_equal(s.labels(L), g["labels%d" % L]), L
    assert np.array_equal(s.kmeans_centres(), g["centres"])  # bit for bit: the sums run in the reference's pixel order
    assert np.array_equal(s.connectivity(), g["connectivity"])
    assert s.stats().kmeans_iters == int(g["iterations"])
    assert (g["labels1"] != g["init_labels1"]).mean() > 0.05 and (g["labels0"] == 24).sum() > 0  # the iterations moved pixels; invalid pixels exist
    return s


def check_full_solve(api, tol=2e-6, tol_b=5e-5, tol_prior=1e-5):
    """tests/golden/solver_160x120.npz: StaticFusion::runSolver derived independently in NumPy (tools/golden/make_golden_solver.py:
    warp, linearisation, segmentation prior, joint IRLS with the b-solve, motion filter, SE(3) update; float64 for every
    cross-pixel sum and every small linear-algebra step) -- discrete outcomes exactly, the floats to rounding. Two frame
    pairs: the pyramid fixture's, and one with four times the motion (two outer iterations at the coarsest level)."""
    from conftest import trace_array

    g = np.load(os.path.join(GOLDEN, "pyramid_160x120.npz"))
    w = np.load(os.path.join(GOLDEN, "solver_160x120.npz"))
    out = []
    for prefix, new, old in (("", (g["d_new0"], g["i_new0"]), (g["d_old0"], g["i_old0"])),
                             ("big_", (w["big_d_new0"], w["big_i_new0"]), (w["big_d_old0"], w["big_i_old0"]))):
        s = make_solver(api, 120, 160, driver_params(api, kb=1.05, ctf_levels=3))
        s.set_current(0, *new)
        s.set_prediction(0, *old)
        s.build_pyramid(True)
        s.run_solver(True)
        st = s.stats()
        n = int(w[prefix + "n_outer"])
        assert st.n_outer == n and st.status == 0, prefix
        for key in ("level", "k", "n_valid", "irls_iters"):
            assert np.array_equal(trace_array(st, key)[:n], w[prefix + key]), (prefix, key)
        assert np.allclose(trace_array(st, "aver_res")[:n], w[prefix + "trace_aver_res"], rtol=2e-4, atol=0), prefix
        assert np.abs(trace_array(st, "var")[:n] - w[prefix + "trace_var"]).max() < tol, prefix
        assert np.abs(trace_array(st, "twist_level")[:n] - w[prefix + "trace_twist_level"]).max() < tol, prefix
        assert np.abs(trace_array(st, "b_segm")[:n] - w[prefix + "trace_b_segm"]).max() < tol_b, prefix
        # computeSegPrior of every outer iteration (SegmentationBackground.cpp:53-103): lambda_t_w is a ratio of two pixel
        # counts (exact), b_prior a clamped mean of 1 - kz |ddt| over the cluster
        assert np.array_equal(trace_array(st, "lambda_t_w")[:n], w[prefix + "trace_lambda_t_w"]), prefix
        assert np.abs(trace_array(st, "b_prior")[:n] - w[prefix + "trace_b_prior"]).max() < tol_prior, prefix
        T_trace = trace_array(st, "T")[:n].reshape(n, 4, 4).transpose(0, 2, 1)  # column-major in the ABI
        assert np.abs(T_trace - w[prefix + "trace_T"]).max() < tol, prefix
        assert np.abs(s.T() - w[prefix + "T"]).max() < tol and np.abs(s.twist() - w[prefix + "twist"]).max() < tol, prefix
        out.append(s)
    assert np.linalg.norm(w["twist"]) > 0.015 and (w["trace_b_segm"][-1] < 0.5).sum() >= 1  # a real motion; the sphere's clusters are dynamic
    assert list(w["big_k"][:2]) == [0, 1] and np.linalg.norm(w["big_trace_twist_level"][0]) > 0.04   # a second outer iteration at the coarsest level
    return out


def check_history_residuals(api, tol=5e-6):
    """computeResidualsAgainstPreviousImage (FrontEnd.cpp:896-1069) derived independently for the last of six frames driven
    through sf_process_frame (carried solver state, the 5-frame ring, the product of the buffered odometries)"""
    w = np.load(os.path.join(GOLDEN, "solver_160x120.npz"))
    s = make_solver(api, 120, 160, driver_params(api, kb=1.5, ctf_levels=3))
    s.set_current(0, w["hist_depth"][0], w["hist_intensity"][0])
    s.current_to_prediction()
    s.push_history(0)
    for k in range(1, 6):
        s.set_current(0, w["hist_depth"][k], w["hist_intensity"][k])
        s.process_frame(k)
        assert np.abs(s.T() - w["hist_T"][k - 1]).max() < 1e-5, k   # the fixture's poses are the oracle's (inputs of the derivation)
        if k < 5:
            s.current_to_prediction()
    assert np.array_equal(s.labels(0), w["hist_labels0"])
    got, want = s.cluster_residuals(), w["hist_cluster_res"]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) < tol, np.nanmax(np.abs(got - want))
    assert np.nanmax(want) > 0.2 and np.nanmin(want) < 0.005   # the sphere's clusters stand out
    return s


def test_oracle_full_solve_golden(ora):
    check_full_solve(ora)


def test_oracle_history_residuals_golden(ora):
    check_history_residuals(ora)


def test_oracle_kmeans_golden(ora):
    check_kmeans(ora)


def test_oracle_pyramid_golden(ora):
    check_pyramid(ora)


def test_oracle_linearise_first_irls_golden(ora):
    check_linearise_and_first_irls(ora)


def test_oracle_linearisation_planes_rows_and_normal_equations_golden(ora):
    check_lin_planes_single_level(ora)
