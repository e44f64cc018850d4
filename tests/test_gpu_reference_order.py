"""The REFERENCE-ORDER build of the frame kernel (staticfusion_amd/csrc/sf_reforder.h -> libsf_hip_reforder.so) against the
oracle, BIT FOR BIT -- VERDICT round 3, "next round" item 1.

The product build replaces several of the reference's float operation sequences by faster, differently rounded ones (exact
integer warp sums, factored Jacobian rows with fmaf, 1-ulp reciprocals, fp32 lane sums, exact per-cluster sums). The
reference-order build puts every one of them back in the order the reference's SOURCE fixes; what the source does not fix --
the internal order of its Eigen GEMM -- follows the oracle's convention [C1]. If the restatement in oracle/ and the device
code agree on the algorithm, the two must then agree on every bit of every output, and they do: these tests hold pose, twist,
b, the b image, the labels, the per-cluster five-frame residuals, the warped images and the whole per-iteration trace to
np.array_equal, on the scenes the product's tests use to stress the splat (picket fence, odd geometry, points behind the
camera), on BASELINE configs[1] / [2] at QVGA and at VGA. (600 QVGA + 5000 160 x 120 random sequences: profiles/PARITY.md.)
"""
import os

import numpy as np
import pytest

from conftest import config2_params, driver_params, make_solver, trace_array
from staticfusion_amd import _capi as capi

pytestmark = pytest.mark.gpu

TRACE_FIELDS = ("level", "k", "n_valid", "irls_iters", "aver_res", "delta_sol_max", "var", "twist_level", "b_segm", "T", "b_prior",
                "lambda_t_w", "AtA", "AtB")


@pytest.fixture(scope="module", params=["throughput", "latency"])
def ro(request):
    import staticfusion_amd as sf

    lib = os.path.join(os.path.dirname(sf.LIB), "libsf_hip_reforder.so")
    api = sf.Api(lib, "sf_").with_variant(request.param)
    assert api.backend_name() == "hip:gfx950:reference-order"
    return api


def assert_state_identical(sg, so, levels=None, warped_levels=()):
    a, b = sg.stats(), so.stats()
    assert (a.n_outer, a.n_irls, a.kmeans_iters, a.status, a.pixel_iters) == (b.n_outer, b.n_irls, b.kmeans_iters, b.status, b.pixel_iters)
    for f in TRACE_FIELDS:
        assert np.array_equal(trace_array(a, f), trace_array(b, f)), f
    assert np.array_equal(sg.T(), so.T()) and np.array_equal(sg.b(), so.b()) and np.array_equal(sg.b_image(), so.b_image())
    assert np.array_equal(sg.twist(), so.twist()) and np.array_equal(sg.twist_old(), so.twist_old())
    for L in range(sg.levels if levels is None else levels):
        assert np.array_equal(sg.labels(L), so.labels(L)), L
    for L in warped_levels:
        for ch in (capi.CH_DEPTH, capi.CH_INTENSITY, capi.CH_XX, capi.CH_YY):
            assert np.array_equal(sg.plane(capi.SET_WARPED, ch, L), so.plane(capi.SET_WARPED, ch, L)), (L, ch)


@pytest.mark.parametrize("seed", (3, 7))
def test_full_solver_pair_at_qvga(ro, ora, pair, seed):
    """BASELINE configs[2]: moving sphere, K-means + b-field, driver parameters."""
    pr = pair(seed=seed, sphere=True, rows=240, cols=320)
    solvers = []
    for api in (ro, ora):
        s = make_solver(api, 240, 320, driver_params(api, debug_planes=1), pr)
        s.build_pyramid(True)
        s.run_solver(True)
        s.build_segm_image()
        solvers.append(s)
    assert_state_identical(*solvers, warped_levels=range(4))


def test_pure_odometry_pair_at_qvga(ro, ora, pair):
    """BASELINE configs[1]: static pair, three levels, segmentation disabled, constructor parameters (10 IRLS iterations, 1e-6)."""
    pr = pair(seed=11, rows=240, cols=320)
    solvers = []
    for api in (ro, ora):
        s = make_solver(api, 240, 320, config2_params(api), pr)
        s.build_pyramid(True)
        s.run_solver(True)
        solvers.append(s)
    assert_state_identical(*solvers, levels=0)


def test_picket_fence_warp(ro, ora, pair):
    """tests/test_gpu_parity.py::test_warp_with_targets_outside_the_tile_windows' scene: neighbouring source pixels land far apart
    and many share a target cell -- the per-cell source lists of the ordered splat are long and out of order here. Warped depth,
    intensity, xx, yy of every level: identical."""
    pr = pair(seed=21, rows=240, cols=320, xi=(0.05, 0.0, 0.0, 0.0, 0.0, 0.0))
    d_old = pr["old"][0].copy()
    patch = np.zeros((240, 320), bool)
    patch[90:150, 130:190] = True
    patch &= ((np.arange(320) // 3) % 2 == 0)[None, :]
    d_old[patch] *= 0.3
    fence = {"new": pr["new"], "old": (d_old, pr["old"][1])}
    solvers = []
    for api in (ro, ora):
        s = make_solver(api, 240, 320, driver_params(api, debug_planes=1), fence)
        s.build_pyramid(True)
        s.run_solver(True)
        s.build_segm_image()
        solvers.append(s)
    assert_state_identical(*solvers, warped_levels=range(4))


def test_many_sources_per_cell(ro, ora, pair):
    """A camera that backs away fast: the predicted image shrinks to a third of its size, a target cell collects far more
    source pixels than the 32 the per-cell lists hold -- those cells are summed by the scan over the level (ro_splat's
    fall-back). Still the reference's order: identical."""
    pr = pair(seed=5, rows=120, cols=160)
    d_old = (pr["old"][0] * 0.3).astype(np.float32)  # everything three times closer in the prediction: it warps into the image centre
    solvers = []
    for api in (ro, ora):
        s = make_solver(api, 120, 160, driver_params(api, debug_planes=1, ctf_levels=3), {"new": pr["new"], "old": (d_old, pr["old"][1])})
        s.build_pyramid(True)
        s.run_solver(True)
        s.build_segm_image()
        solvers.append(s)
    assert_state_identical(*solvers, warped_levels=range(2))


def test_points_behind_the_camera_follow_the_reference(ro, ora):
    """The product leaves a point warped behind the camera out of validPixels; the reference keeps it
    (FrontEnd.cpp:415-427 has no depth test), and so does this build: equal to the oracle WITHOUT its HIP-rule switch."""
    import ctypes

    from test_gpu_edge_rules import _near_patch_pair

    pr = _near_patch_pair()
    solvers = []
    for api in (ro, ora):
        s = make_solver(api, 120, 160, driver_params(api), pr)
        s.build_pyramid(True)
        s.run_solver(True)
        s.build_segm_image()
        solvers.append(s)
    lib = ora.lib
    lib.sfo_test_behind_camera_valid.restype = ctypes.c_longlong
    lib.sfo_test_behind_camera_valid.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.sfo_test_behind_camera_valid(solvers[1].h, 0) > 300  # the scene does what it was built for
    assert_state_identical(*solvers)


def test_odd_geometry_sequence_with_history(ro, ora):
    """tests/test_gpu_edge_rules.py::test_first_touch_splat_on_odd_geometry's sequence (200 x 264, holes, a strong roll) through the
    drivers' frame loop with the five-frame residuals: every frame's pose, b image and per-cluster residuals identical."""
    from staticfusion_amd.synth import Scene, quantise_and_decimate, se3_exp

    rows, cols = 200, 264
    scene = Scene(seed=31, sphere=True)
    xi = np.array((0.010, -0.005, 0.008, 0.03, -0.006, 0.003))
    frames, T = [], np.eye(4)
    for k in range(7):
        d, i = quantise_and_decimate(*scene.render(T, 2 * cols, 2 * rows, sphere_offset=(0.02 * k, 0, 0)))
        d = d.copy()
        d[:, 96:144] = 0
        d[0:120, 208:232] = 0
        frames.append((d, i))
        T = T @ se3_exp(xi)
    solvers = [make_solver(api, rows, cols, driver_params(api, kb=1.5, ctf_levels=3, debug_planes=1)) for api in (ro, ora)]
    for s in solvers:
        s.set_current(0, *frames[0])
        s.current_to_prediction()
        s.push_history(0)
    for k in range(1, 7):
        for s in solvers:
            s.set_prediction(0, *frames[k - 1])
            s.set_current(0, *frames[k])
            s.process_frame(k)
        assert_state_identical(*solvers, warped_levels=range(3))
        cg, co = solvers[0].cluster_residuals(), solvers[1].cluster_residuals()
        assert np.array_equal(cg, co, equal_nan=True), k


def test_vga_six_levels(ro, ora, pair):
    pr = pair(seed=9, sphere=True, rows=480, cols=640)
    solvers = []
    for api in (ro, ora):
        s = make_solver(api, 480, 640, driver_params(api), pr)
        assert s.levels == 6
        s.build_pyramid(True)
        s.run_solver(True)
        s.build_segm_image()
        solvers.append(s)
    assert_state_identical(*solvers)


@pytest.mark.parametrize("rows", (64, 124, 128, 188, 248, 252))
def test_strip_boundaries_of_the_linearisation(ro, ora, pair, rows):
    """The register-strip linearisation gives a wave 62 rows of a level: image heights whose levels end exactly on a
    strip (124 = 2 x 62, 248 = 4 x 62, level 1 of 124: one strip), two rows behind one (64, 126, 188), or need a fifth strip
    (252: the columns are then cut into segments) -- the rows next to a strip edge get their upper / lower neighbours from the
    halo lanes of ANOTHER wave's strip. Three levels each, full solver, every trace value and the warped planes bit for bit."""
    cols = 96
    pr = pair(seed=40 + rows, sphere=True, rows=rows, cols=cols)
    solvers = []
    for api in (ro, ora):
        s = make_solver(api, rows, cols, driver_params(api, ctf_levels=3, debug_planes=1), pr)
        s.build_pyramid(True)
        s.run_solver(True)
        s.build_segm_image()
        solvers.append(s)
    assert solvers[0].levels == 3
    assert_state_identical(*solvers, warped_levels=range(3))


@pytest.mark.parametrize("rows,cols,levels", [(40, 42, 3), (36, 116, 2), (20, 52, 2), (26, 104, 3), (34, 72, 4), (20, 44, 2)])
def test_last_wave_partly_filled(ro, ora, pair, rows, cols, levels):
    """Pixel counts that are no multiple of the wave size (n0 % 64 = 16 or 48), odd rows / columns from level 1 on: the last wave of every
    per-pixel loop is partly filled, and round 5's cross-lane code (ds_bpermute in stage_segm_image, DPP neighbours in the strip
    linearisation) depends on which lanes are still there. Every value bit for bit, the segm image included."""
    pr = pair(seed=5, sphere=True, rows=rows, cols=cols)
    d_new = pr["new"][0].copy()
    d_new[-3:, -1] = 0  # label 24 inside the last wave
    prm = {"new": (d_new, pr["new"][1]), "old": pr["old"]}
    solvers = []
    for api in (ro, ora):
        s = make_solver(api, rows, cols, driver_params(api, kb=1.5, ctf_levels=levels, debug_planes=1), prm)
        s.build_pyramid(True)
        s.run_solver(True)
        s.build_segm_image()
        solvers.append(s)
    assert (solvers[0].labels(0) == 24).any()
    assert_state_identical(*solvers, warped_levels=range(levels))


def test_fp64_sums_row_by_row(ro, ora):
    """The oracle's [C1] sums (AtA / AtB, sum |res|, ||res||^2) are fp64 sums of float terms, row after row. Summed per lane and
    then over the lanes they differ in the 16th digit -- which moved the float they are converted to in ONE frame of 73 600
    (160 x 120 hunt case 61826, frame 6: AtA(0,0) 253.37692 against 253.37694; pose 1.4e-9 apart). The build walks them row by row."""
    from sequence_cases import make_case, run_case

    case = make_case(61826, 320, 240, True)
    ref, got = run_case(ora, case), run_case(ro, case)
    for k, (r, g) in enumerate(zip(ref, got)):
        assert r["counts"] == g["counts"], k
        for f in ("T", "b", "b_img", "labels"):
            assert np.array_equal(r[f], g[f]), (k, f)


@pytest.mark.parametrize("roll,forward", [(0.25, 0.25), (0.4, 0.0), (0.4, 0.25)])
def test_strong_roll_and_approach(ro, ora, roll, forward):
    """A quarter of a radian and more about the optical axis, with and without a fast approach: at the coarse levels the targets of
    a tile of 16 whole columns no longer fit its LDS window (30 rows x sin 0.25 = 7 columns against a margin of 6), the ordered
    splat falls back from the tiles to the per-cell lists in the middle of a level, and the solver needs up to 14 outer and 84
    IRLS iterations. Identical all the same."""
    from staticfusion_amd.synth import make_pair

    pr = make_pair(seed=17, sphere=True, out_rows=240, out_cols=320, xi=(forward, 0.0, 0.0, roll, 0.0, 0.0))
    solvers = []
    for api in (ro, ora):
        s = make_solver(api, 240, 320, driver_params(api, debug_planes=1), pr)
        s.build_pyramid(True)
        s.run_solver(True)
        s.build_segm_image()
        solvers.append(s)
    assert solvers[1].stats().n_irls >= 30
    assert_state_identical(*solvers, warped_levels=range(4))


def test_cluster_variant_is_refused(ro):
    import staticfusion_amd as sf

    with pytest.raises(sf.SfError, match="reference-order"):
        sf.Solver(ro, 120, 160, 1, ro.default_params_struct(), variant="cluster")


def test_ordered_tile_splat_equals_the_list_splat(tmp_path):
    """The product's ordered float splat of the coarse levels (`ordered_tile_splat`: LDS-resident tiles, rounds of ds_min) against
    the per-cell source lists of the reference-order build (`ro_splat`), in isolation: one workgroup, synthetic levels from 8 x 8 to 64 x 100 pixels (odd sizes included) with holes under a rigid warp -- every cell bit for bit
    (tools/micro/ordered_splat_check.hip) -- and, under a strong roll, the fall-back of `ordered_splat` from the tiles to the lists."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ordered_splat_check")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                           "-Wno-unused-function", "-Wno-unused-value", "-I", os.path.join(root, "staticfusion_amd", "csrc"), "-o", exe,
                           os.path.join(root, "tools", "micro", "ordered_splat_check.hip")], timeout=600)
    out = subprocess.check_output([exe], timeout=120).decode()
    # seven levels the tiles serve, three under a roll of 0.3 - 0.5 rad that they give up on (the lists then run over what they left)
    assert out.strip().endswith("OK") and out.count("differing cells 0") == 10 and out.count("(tiles returned 1)") == 7 and out.count("(tiles returned 0)") == 3, out
