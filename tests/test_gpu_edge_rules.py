"""Two rules at the edges of the path, each on both sides of its threshold.

1. buildSegmImage's residual test `perClusterAverageResidual[label] < 0.017` (SegmentationBackground.cpp:190, a float
   compared with a double literal): clusters whose 5-frame residual sits AT the threshold, one float below and above it,
   1e-4 away on either side, and NaN (no sample: compares false). The image must be the oracle's bit for bit.
2. A point warped BEHIND the camera that still projects into the image (a diverged pose, or -- here -- a surface closer to
   the camera than the frame-to-frame motion): the reference keeps such a pixel in validPixels (FrontEnd.cpp:816-823 has no
   depth test). The HIP build leaves it out -- the sign of the stored warped depth is what marks validPixels for the
   streaming passes (sf_solver.h, solve_linearise) -- and gives computeSegPrior its magnitude. The oracle carries the same
   rule behind a switch (sfo_test_set_hip_behind_camera_rule): with it, HIP and oracle agree as on any other input; without
   it the valid-pixel counts differ by exactly the pixels the oracle reports as behind the camera.
"""
import ctypes

import numpy as np
import pytest

from conftest import driver_params, make_solver, trace_array
from staticfusion_amd.synth import make_pair, pose_delta

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cols,levels", [(60, 80, 0), (40, 42, 3), (20, 52, 2), (20, 44, 2)])
def test_segm_image_residual_threshold_from_both_sides(hip, ora, rows, cols, levels):
    """(40 x 42, 20 x 52: n0 % 64 = 16, 20 x 44: 48 -- the last wave of the image is partly filled, and the value of a label comes
    from the lane of that number: ADVICE round 5)"""
    rng = np.random.RandomState(3)
    labels = rng.randint(0, 25, size=(rows, cols)).astype(np.int32)  # 24 = invalid cluster
    b = rng.uniform(-0.3, 1.3, 24).astype(np.float32)
    b[:6] = [0.2, 0.8, 0.5, 0.49999997, 0.0, 1.0]
    t32 = np.float32(0.017)  # 0.017000000923871994 as a double: NOT below the double literal 0.017
    res = np.full(24, 0.05, np.float32)
    res[:10] = [t32, np.nextafter(t32, np.float32(0)), np.nextafter(t32, np.float32(1)), t32 - np.float32(1e-4), t32 + np.float32(1e-4),
                np.float32(0.0169999), np.float32(0.0170001), np.nan, 0.0, -1.0]
    res[10:] = rng.uniform(0.0165, 0.0175, 14).astype(np.float32)
    out = []
    for api in (hip, ora):
        s = make_solver(api, rows, cols, driver_params(api, ctf_levels=levels))
        s.set_segm_state(0, labels, b, res)
        s.build_segm_image()
        out.append(s.b_image().copy())
    assert np.array_equal(out[0], out[1])
    assert np.all(out[0][labels == 24] == 1.0)
    # the rule itself, on the clusters that straddle the threshold
    for l, below in ((0, False), (1, True), (2, False), (3, True), (4, False), (7, False), (8, True)):
        px = out[0][labels == l]
        bb = min(max(float(b[l]), 0.0), 1.0)
        want = max(bb, 1.0 - bb) if below else bb
        assert px.size and np.all(px == np.float32(want)), (l, below)


def test_kmeans_refuses_odd_image_sizes(hip, ora):
    """KMeans.cpp:267 starts the full-resolution search of pixel (v, u) at labels_lowres(v/2, u/2), a rows/2 x cols/2 matrix: with an
    odd size the reference reads past its end, and what it finds there decides labels. Nothing to be identical to: every call
    that runs K-means refuses such a handle on both sides of the ABI (found in round 6 by the oracle's bounds-checked containers
    at 48 x 43); pure odometry, the input stage, prediction and the map have no such read (test_map_fusion runs 117 x 160)."""
    import staticfusion_amd as sf
    from conftest import config2_params

    for api in (hip, ora):
        for rows, cols in ((48, 43), (45, 48)):
            s = sf.Solver(api, rows, cols, 1, driver_params(api, ctf_levels=2))
            s.build_pyramid(True)
            for call in (s.kmeans, lambda: s.run_solver(True), lambda: s.process_frame(0)):
                with pytest.raises(sf.SfError, match="even rows and cols"):
                    call()
            s.build_segm_image()  # a function of labels and b alone
            s.close()
        s = sf.Solver(api, 48, 43, 1, config2_params(api, levels=2))
        s.build_pyramid(True)
        s.run_solver(True)
        with pytest.raises(sf.SfError, match="even rows and cols"):
            s.kmeans()
        s.close()


def _near_patch_pair():
    pr = make_pair(seed=5, sphere=False, out_rows=120, out_cols=160, xi=(0.0, 0.0, 0.12, 0.0, 0.0, 0.0))
    d_old = pr["old"][0].copy()
    d_old[40:80, 60:100] = 0.06  # a surface 6 cm from the old camera; the camera then advances 12 cm
    return {"old": (d_old, pr["old"][1]), "new": pr["new"]}


def test_points_warped_behind_the_camera(hip, ora):
    lib = ora.lib
    lib.sfo_test_set_hip_behind_camera_rule.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sfo_test_behind_camera_valid.restype = ctypes.c_longlong
    lib.sfo_test_behind_camera_valid.argtypes = [ctypes.c_void_p, ctypes.c_int]
    pr = _near_patch_pair()
    runs = {}
    for name, api, rule in (("hip", hip, None), ("oracle+rule", ora, 1), ("reference", ora, 0)):
        s = make_solver(api, 120, 160, driver_params(api), pr)
        if rule is not None:
            assert lib.sfo_test_set_hip_behind_camera_rule(s.h, rule) == 0
        s.build_pyramid(True)
        s.run_solver(True)
        s.build_segm_image()
        runs[name] = s
    behind = lib.sfo_test_behind_camera_valid(runs["reference"].h, 0)
    assert behind > 300  # the input does what it was built for: the reference carries such pixels through validPixels
    g, o, r = runs["hip"].stats(), runs["oracle+rule"].stats(), runs["reference"].stats()
    # with the rule on both sides: the usual parity
    assert (g.n_outer, g.n_irls, g.status) == (o.n_outer, o.n_irls, o.status)
    assert np.array_equal(trace_array(g, "n_valid"), trace_array(o, "n_valid"))
    assert np.array_equal(trace_array(g, "lambda_t_w"), trace_array(o, "lambda_t_w"))
    # b_prior to 1e-3 only: where the near surface (warped to -6 cm) and the wall behind it (2.9 m) land in ONE cell, the cell's weighted
    # mean jumps by 0.2 m when a centi-pixel coordinate falls on the other side of an integer (a 1e-8 difference of T): 28 such cells
    # at level 0 here (tools/diag/behind_camera_case.py), which move one cluster's prior by 4e-4 -- the scene, not the rule
    assert np.abs(trace_array(g, "b_prior") - trace_array(o, "b_prior")).max() < 1e-3
    rot, trans = pose_delta(runs["oracle+rule"].T(), runs["hip"].T())
    assert rot <= 1e-4 and trans <= 1e-4, (rot, trans)
    for L in range(runs["hip"].levels):
        assert np.array_equal(runs["hip"].labels(L), runs["oracle+rule"].labels(L))
    assert np.array_equal(runs["hip"].b_image() > 0.5, runs["oracle+rule"].b_image() > 0.5)
    # against the reference's rule: the departure, as stated in DESIGN.md section 6 -- fewer valid pixels in the iterations
    # whose warp put the near surface behind the camera
    nv_g, nv_r = trace_array(g, "n_valid"), trace_array(r, "n_valid")
    n = min(len(nv_g), len(nv_r))
    first = int(np.argmax(nv_g[:n] != nv_r[:n]))
    assert (nv_g[:n] != nv_r[:n]).any() and nv_g[first] < nv_r[first]


def test_first_touch_splat_on_odd_geometry(hip, ora):
    """The one-workgroup builds' splat keeps no zeroed accumulator image: a window's flush STORES the cells nobody has written
    yet (a row watermark per column says which, sf_device_common.h) in groups of 16 rows. What that bookkeeping has to get
    right, all in one scene: rows that are not a multiple of 16 (200: the last group of a column is short), source tiles
    without a valid pixel (three strips of 16 predicted columns are holes: their windows do not exist, the columns they would
    have reached are zeroed at the end), a block of holes at the top of another strip (its first window starts far down: the
    rows above are skipped rows), and a motion with a strong roll, under which the windows of the tiles of one strip reach
    different columns. The warped images of every level and the five-frame residuals (the second user of the splat) equal the
    oracle's; the cluster build (zero pass + atomics on every cell) runs the same comparison."""
    from staticfusion_amd import _capi as capi
    from staticfusion_amd.synth import Scene, quantise_and_decimate, se3_exp
    from test_gpu_parity import POSE_TOL, assert_planes_close

    rows, cols = 200, 264
    scene = Scene(seed=31, sphere=True)
    xi = np.array((0.010, -0.005, 0.008, 0.03, -0.006, 0.003))  # the fourth component turns the image about the optical axis
    frames, T = [], np.eye(4)
    for k in range(7):
        d, i = quantise_and_decimate(*scene.render(T, 2 * cols, 2 * rows, sphere_offset=(0.02 * k, 0, 0)))
        d = d.copy()
        d[:, 96:144] = 0     # three source strips without a valid pixel
        d[0:120, 208:232] = 0  # a strip whose first valid row is far down
        frames.append((d, i))
        T = T @ se3_exp(xi)
    assert frames[0][0].shape == (rows, cols)
    solvers = [make_solver(api, rows, cols, driver_params(api, kb=1.5, ctf_levels=3, debug_planes=1)) for api in (hip, ora)]
    for s in solvers:
        s.set_current(0, *frames[0])
        s.current_to_prediction()
        s.push_history(0)
    for k in range(1, 7):
        for s in solvers:
            s.set_prediction(0, *frames[k - 1])
            s.set_current(0, *frames[k])
            s.process_frame(k)
        sg, so = solvers
        rot, trans = pose_delta(so.T(), sg.T())
        assert rot <= POSE_TOL and trans <= POSE_TOL, (k, rot, trans)
        assert np.array_equal(sg.labels(0), so.labels(0))
        a, b = sg.stats(), so.stats()
        assert (a.n_outer, a.n_irls) == (b.n_outer, b.n_irls), k
        for L in range(3):
            for ch in range(2):  # warped depth and intensity as the last linearisation of the level saw them
                assert_planes_close(sg.plane(capi.SET_WARPED, ch, L), so.plane(capi.SET_WARPED, ch, L), frac=0.97)
        if k >= 5:
            cg, co = sg.cluster_residuals(), so.cluster_residuals()
            assert np.array_equal(np.isnan(cg), np.isnan(co))
            assert np.allclose(cg[~np.isnan(cg)], co[~np.isnan(co)], rtol=1e-3, atol=1e-6)  # (tolerance: test_frame_sequence_with_history)
    if hip.default_variant != "cluster":
        assert solvers[0].splat_replays() > 0, "no tile had targets outside its window: the roll is too small for what this test wants"


@pytest.mark.parametrize("rows,cols,levels", [(32, 48, 2), (32, 40, 3)])
def test_tiny_images_take_the_ordered_splat_at_every_level(hip, ora, rows, cols, levels):
    """Level 0 of at most SF_ORDERED_SPLAT_MAX_PIXELS = 2048 pixels: the ordered float splat (the reference's sums, sf_reforder.h) runs at
    EVERY level and in the residual stage too, which takes the per-cell source lists (`ro_splat`; its LDS tiles serve the solve's warp
    only) out of the per-workgroup scratch -- a path no other test of the product reaches. With the reference's warp sums
    everywhere only the passes' arithmetic separates the product from the oracle: pose to 1e-5 (measured 1e-8 at 32 x 48, 3.4e-6 at
    32 x 40 whose coarsest level has 80 pixels), identical counts and labels, b to 1e-3 (2.4e-7 and 1.4e-4). Then 3000 streams of the same sequence on the throughput build -- more than its resident
    workgroups, so the scratch is indexed by workgroup, not by stream: first and last stream bit-identical to the single one."""
    from staticfusion_amd.synth import Scene, quantise_and_decimate, se3_exp

    scene = Scene(seed=41, sphere=True)
    xi = np.array((0.006, -0.004, 0.005, 0.01, -0.004, 0.003))
    frames, T = [], np.eye(4)
    for k in range(7):
        frames.append(quantise_and_decimate(*scene.render(T, 2 * cols, 2 * rows, sphere_offset=(0.02 * k, 0, 0))))
        T = T @ se3_exp(xi)

    def run(api, batch=1):
        s = make_solver(api, rows, cols, driver_params(api, kb=1.5, ctf_levels=levels), batch=batch)
        for b in range(batch):
            s.set_current(b, *frames[0])
        s.current_to_prediction()
        s.push_history(0)
        out = []
        for k in range(1, 7):
            for b in range(batch):
                s.set_prediction(b, *frames[k - 1])
                s.set_current(b, *frames[k])
            s.process_frame(k)
            st = s.stats()
            out.append(dict(T=[s.T(0), s.T(batch - 1)], labels=s.labels(0), counts=(st.n_outer, st.n_irls), status=st.status, b=s.b(),
                            b_img=s.b_image(), cr=s.cluster_residuals()))
        return out

    ref, got = run(ora), run(hip)
    for k, (r, g) in enumerate(zip(ref, got)):
        rot, trans = pose_delta(r["T"][0], g["T"][0])
        assert rot <= 1e-5 and trans <= 1e-5, (k, rot, trans)
        assert r["counts"] == g["counts"] and r["status"] == g["status"] == 0, k
        assert np.array_equal(r["labels"], g["labels"])
        assert np.abs(r["b"] - g["b"]).max() <= 1e-3 and np.abs(r["b_img"] - g["b_img"]).max() <= 1e-3, k
        assert np.array_equal(np.isnan(r["cr"]), np.isnan(g["cr"]))
        assert np.allclose(g["cr"][~np.isnan(g["cr"])], r["cr"][~np.isnan(r["cr"])], rtol=1e-3, atol=1e-6)
    if hip.default_variant == "throughput":
        many = run(hip, batch=3000)
        for g, m in zip(got, many):
            assert np.array_equal(g["T"][0], m["T"][0]) and np.array_equal(g["T"][0], m["T"][1])


@pytest.mark.parametrize("xi,tol,rolls", [((0.0, 0.0, 0.0, 0.25, 0.0, 0.0), 1e-5, False), ((0.25, 0.0, 0.0, 0.25, 0.0, 0.0), 1e-4, False),
                                          ((0.0, 0.0, 0.0, 0.0, 0.0, 0.3), 1e-4, True), ((0.0, 0.0, 0.1, 0.0, 0.0, 0.3), 1e-4, True)])
def test_strong_rotations_and_the_coarse_levels_tile_windows(hip, ora, xi, tol, rolls):
    """The product's ordered float splat of the coarse levels (LDS tiles, sf_reforder.h) gives up on a level whose taps leave a tile's
    window and takes the per-cell lists instead. An in-plane rotation of 0.3 rad does that to the 40 x 30 level in the throughput
    build (a tile of 16 columns x 30 rows turns into 24 columns of targets, its window holds 22): the counter of slot 25 of the
    stage profile says that it happened (round 5; round 4's two cases -- kept -- set xi[3], which is a PITCH in this camera frame
    (the optical axis is z: xi[5] rolls), stretch the image and replay tiles of the integer splat, but never left a coarse tile's
    window: profiles/r05c_roll_fallbacks_wx_is_pitch.txt, r05d_roll_fallbacks.txt). Same answer either way: pose to `tol`, identical
    iteration counts, labels and decisions."""
    pr = make_pair(seed=17, sphere=True, out_rows=240, out_cols=320, xi=xi)
    out = []
    for api in (hip, ora):
        s = make_solver(api, 240, 320, driver_params(api), pr)
        s.build_pyramid(True)
        s.run_solver(True)
        s.build_segm_image()
        out.append(s)
    sg, so = out
    rot, trans = pose_delta(so.T(), sg.T())
    assert rot <= tol and trans <= tol, (rot, trans)
    a, b = sg.stats(), so.stats()
    assert (a.n_outer, a.n_irls) == (b.n_outer, b.n_irls) and a.status == b.status == 0
    assert np.array_equal(sg.labels(0), so.labels(0)) and np.array_equal(sg.b_image() > 0.5, so.b_image() > 0.5)
    if rolls and hip.default_variant == "throughput":
        assert sg.ordered_fallbacks() > 0, "no coarse level left its tile windows: the list path did not run"


def test_list_fallback_where_the_scratch_blocks_are_shared(hip_auto, pair):
    """the per-cell source lists of the ordered coarse splat (`ro_splat`, the fall-back of a level whose
    taps leave a tile's window) live in scratch blocks handed out per STREAM while the handle has one for each, else per
    WORKGROUP of the launch (sf_reforder.h: ro_list_of, `blockIdx.x`). Here: one multi-frame launch of 2048 QVGA streams on the
    throughput build -- more streams than resident workgroups, frames of a stream on different workgroups -- in which every third
    stream is the strong-roll pair (its 40 x 30 level falls back, counted) and the others are ordinary pairs: every stream's pose of
    every frame, its labels, b and b image equal, bit for bit, the same stream solved in a handle of two."""
    import staticfusion_amd as sf

    api = hip_auto.with_variant("throughput")
    roll = make_pair(seed=17, sphere=True, out_rows=240, out_cols=320, xi=(0.0, 0.0, 0.0, 0.0, 0.0, 0.3))  # (xi[5]: about the optical axis)
    plain = pair(seed=7, sphere=True, rows=240, cols=320)
    K = 7  # frames 0 .. 6 of every stream in ONE launch: the last two with the five-frame residuals
    which = lambda b: roll if b % 3 == 0 else plain

    def run(batch, kinds):
        s = make_solver(api, 240, 320, driver_params(api), batch=batch)
        for b in range(batch):
            s.set_current(b, *kinds(b)["new"])
            s.set_prediction(b, *kinds(b)["old"])
        T = s.process_frames(0, K, trajectory=True)
        return s, T

    ref, T_ref = run(2, lambda b: (roll, plain)[b])
    assert ref.ordered_fallbacks() >= K, "the roll pair did not send a level to the lists: this test would test nothing"
    n_fb = ref.ordered_fallbacks()
    B = 2048
    big, T = run(B, which)
    assert big.resident_workgroups()[1] < B, "the batch must exceed the resident workgroups (scratch blocks per workgroup, not per stream)"
    assert big.ordered_fallbacks() == n_fb * ((B + 2) // 3)  # every roll stream took the lists exactly as often as the one solved in the handle of two
    for b in range(B):
        r = 0 if b % 3 == 0 else 1
        assert np.array_equal(T[:, b], T_ref[:, r]), b
    for b in list(range(0, 64)) + list(range(B - 64, B)) + list(range(700, 764)):
        r = 0 if b % 3 == 0 else 1
        assert np.array_equal(big.labels(0, b), ref.labels(0, r)) and np.array_equal(big.b_image(b), ref.b_image(r)), b
        assert np.array_equal(big.b(b), ref.b(r)) and np.array_equal(big.cluster_residuals(b), ref.cluster_residuals(r), equal_nan=True), b
    assert big.stats(0).status == ref.stats(0).status
