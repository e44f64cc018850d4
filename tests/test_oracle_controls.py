"""What the reference itself leaves open, measured on the oracle alone (CPU).

The reference accumulates AtA / AtB with an Eigen float GEMM (FrontEnd.cpp:640-641) and the warped images with an
order-dependent float scatter (:840-867); neither order is visible in its sources. The oracle fixes conventions ([C1]: float
operands, fp64 sums; the scatter in the reference's loop order). These tests run the SAME oracle under the other plausible
readings (test hooks sfo_test_set_gemm_mode / sfo_test_set_exact_warp) and pin two statements the parity tolerances rest on:
  * discrete outcomes (cluster labels, the b > 0.5 decision, iteration counts away from a stopping-threshold tie) do not
    depend on the reading;
  * the 24 b values do, by MORE than the 1e-5 SURVEY.md section 8(c)(iv) wished for and by about as much as the HIP path differs
    from the oracle (4e-5 at QVGA, tools/diag/b_summary.py): 1e-4 is the resolution the algorithm has, not a slack of the port.
The same comparison over 1000 + 240 + 60 sequences: profiles/r03_control_*.json (tools/diag/sequence_hunt.py --control).
"""
import ctypes

import numpy as np

from sequence_cases import compare_frames, make_case, run_case


def _hook(ora, name, value):
    fn = getattr(ora.lib, name)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]

    def prepare(solver):
        assert fn(solver.h, value) == 0

    return prepare


def test_summation_order_of_the_normal_equations(ora):
    thr = float(ora.default_params_struct().irls_delta_threshold)
    worst_b, worst_pose, frames = 0.0, 0.0, 0
    for seed in range(20100, 20110):
        case = make_case(seed)
        ref = run_case(ora, case)
        for mode in (1, 2):  # one float accumulator per entry; four interleaved float partial sums (SSE packets)
            got = run_case(ora, case, prepare=_hook(ora, "sfo_test_set_gemm_mode", mode))
            for r in compare_frames(ref, got, thr):
                frames += 1
                assert r["label_px"] == 0 and r["decision_px"] == 0, (seed, mode, r)
                if "flip" in r:  # the stopping test decided on the last digit: only ever as a threshold tie
                    assert r["flip"]["kind"] in ("threshold", "level-exit", "outer-count"), (seed, mode, r)
                    break
                worst_b = max(worst_b, r["b24"])
                worst_pose = max(worst_pose, r["rot"], r["trans"])
        rev = run_case(ora, case, prepare=_hook(ora, "sfo_test_set_gemm_mode", 3))  # [C1] over the rows in reverse: fp64 sums do not care
        for r in compare_frames(ref, rev, thr):
            assert r["b24"] == 0.0 and max(r["rot"], r["trans"]) < 1e-12 and "flip" not in r
    assert frames >= 100
    assert 1e-5 < worst_b < 0.2, worst_b      # the reading moves b by more than 1e-5 ...
    assert worst_pose < 1e-4, worst_pose      # ... and the pose by less than the bar (away from threshold ties)


def test_float_scatter_of_the_warp(ora):
    from conftest import driver_params, make_solver
    from staticfusion_amd.synth import make_pair, pose_delta

    worst = 0.0
    for seed in (1234, 1236):
        pr = make_pair(seed=seed, sphere=True, out_rows=120, out_cols=160)
        res = []
        for exact in (0, 1):
            s = make_solver(ora, 120, 160, driver_params(ora), pr)
            _hook(ora, "sfo_test_set_exact_warp", exact)(s)
            s.build_pyramid(True)
            s.run_solver(True)
            st = s.stats()
            res.append((np.array([list(st.outer[i].b_segm[:]) for i in range(st.n_outer)]), s.T().copy(), (st.n_outer, st.n_irls),
                        [s.labels(L).copy() for L in range(s.levels)]))
        (b0, T0, c0, l0), (b1, T1, c1, l1) = res
        assert c0 == c1 and all(np.array_equal(x, y) for x, y in zip(l0, l1))
        rot, trans = pose_delta(T0, T1)
        assert rot < 1e-5 and trans < 1e-5
        worst = max(worst, float(np.abs(b0 - b1).max()))
    assert 2e-6 < worst < 1e-3, worst  # the reference's own scatter rounding is visible in b at the 1e-5 level
