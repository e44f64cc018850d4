"""Known-answer tests of the CPU oracle (the reference holds no tests of its own: SURVEY.md §4).
Each case is hand-computable or has an analytic answer."""
import numpy as np
import pytest

from conftest import config2_params, driver_params, make_solver
from staticfusion_amd import capi
from staticfusion_amd.synth import DEFAULT_XI, pose_delta, se3_exp

f32 = np.float32


def _pyr(ora, depth, inten, levels=2):
    rows, cols = depth.shape
    p = config2_params(ora, levels=levels)
    s = make_solver(ora, rows, cols, p)
    s.set_current(0, depth, inten)
    s.build_pyramid(False)
    return s


def test_pyramid_second_largest_centre_and_bilateral_average(ora):
    # 16x16 level 0 -> 8x8 level 1. Inner pixel (v,u)=(2,3) reads the 4x4 block at rows 3..6, cols 5..8.
    d = np.full((16, 16), 2.0, f32)
    i = np.full((16, 16), 0.5, f32)
    d[4, 6], d[5, 6], d[4, 7], d[5, 7] = 1.0, 1.2, 1.25, 3.0   # central 2x2: second largest = 1.25
    i[4, 6], i[5, 6], i[4, 7], i[5, 7] = 0.1, 0.2, 0.3, 0.9
    s = _pyr(ora, d, i)
    # taps within 0.1 of dcenter=1.25: 1.2 (k=6, mask 4/36) and 1.25 (k=9, mask 4/36); 1.0, 3.0 and the 2.0 ring are out
    w1 = f32(4) / f32(36) * (f32(0.1) - abs(f32(1.2) - f32(1.25)))
    w2 = f32(4) / f32(36) * (f32(0.1) - f32(0))
    exp_d = (w1 * f32(1.2) + w2 * f32(1.25)) / (w1 + w2)
    exp_i = (w1 * f32(0.2) + w2 * f32(0.3)) / (w1 + w2)
    assert s.plane(capi.SET_NEW, capi.CH_DEPTH, 1)[2, 3] == pytest.approx(exp_d, rel=1e-6)
    assert s.plane(capi.SET_NEW, capi.CH_INTENSITY, 1)[2, 3] == pytest.approx(exp_i, rel=1e-6)
    # a uniform neighbourhood reproduces itself
    assert s.plane(capi.SET_NEW, capi.CH_DEPTH, 1)[5, 5] == pytest.approx(2.0, rel=1e-6)


def test_pyramid_zero_centre_and_border_means(ora):
    d = np.full((16, 16), 2.0, f32)
    i = np.linspace(0, 1, 256, dtype=f32).reshape(16, 16)
    d[4:6, 6:8] = 0.0  # central 2x2 of output pixel (2,3) all invalid -> depth 0, plain mask average of intensity
    d[0, 0] = 0.0      # border pixel (0,0): mean of the three non-zero depths
    d[0:2, 14:16] = 0  # border pixel (0,7): all invalid -> 0
    s = _pyr(ora, d, i)
    D, I = s.plane(capi.SET_NEW, capi.CH_DEPTH, 1), s.plane(capi.SET_NEW, capi.CH_INTENSITY, 1)
    assert D[2, 3] == 0.0
    mask = np.outer([1, 2, 2, 1], [1, 2, 2, 1]).astype(np.float64) / 36.0
    assert I[2, 3] == pytest.approx(float((mask * i[3:7, 5:9]).sum()), rel=1e-6)
    assert D[0, 0] == pytest.approx(2.0, rel=1e-7)
    assert D[0, 7] == 0.0
    assert I[0, 0] == pytest.approx(0.25 * float(i[0:2, 0:2].sum()), rel=1e-6)
    # xx / yy of level 1 (reference FrontEnd.cpp:378-388)
    inv_f = 2.0 * np.tan(0.5 * np.pi * 62.5 / 180.0) / 8.0
    XX = s.plane(capi.SET_NEW, capi.CH_XX, 1)
    assert XX[3, 6] == pytest.approx(inv_f * (6 - 3.5) * D[3, 6], rel=1e-5)


def test_identical_images_give_identity(ora, pair):
    pr = pair(seed=3, rows=60, cols=80)
    same = {"new": pr["new"], "old": pr["new"]}
    s = make_solver(ora, 60, 80, config2_params(ora, levels=3), same)
    s.build_pyramid(True)
    s.run_solver(True)
    assert np.abs(s.T() - np.eye(4)).max() < 1e-6
    assert np.abs(s.twist()).max() < 1e-6


def test_warp_identity_snaps_to_integer_pixels(ora, pair):
    """With T = I every old pixel projects onto itself: the 'weight 200' branch (FrontEnd.cpp:835-843)
    must reproduce the Pred level in the Warped level (second outer iteration of a level)."""
    pr = pair(seed=3, rows=60, cols=80)
    same = {"new": pr["new"], "old": pr["new"]}
    p = config2_params(ora, levels=2, debug_planes=1, max_iter_per_level=2)
    s = make_solver(ora, 60, 80, p, same)
    s.build_pyramid(True)
    s.run_solver(True)
    # level 0 (finest) is warped with an (almost) identity T_odometry
    P, W = s.plane(capi.SET_PRED, capi.CH_DEPTH, 0), s.plane(capi.SET_WARPED, capi.CH_DEPTH, 0)
    inner = np.zeros_like(P, bool)
    inner[1:-2, 1:-2] = True  # the reference's range test excludes the last row/column
    assert np.abs(P - W)[inner].max() < 1e-5


@pytest.mark.parametrize("xi", [DEFAULT_XI, (0.02, 0, 0, 0, 0, 0), (0, 0, 0, 0, 0.01, 0)])
def test_static_scene_recovers_known_motion(ora, pair, xi):
    pr = pair(seed=21, rows=240, cols=320, xi=xi)
    s = make_solver(ora, 240, 320, config2_params(ora, levels=3), pr)
    s.build_pyramid(True)
    s.run_solver(True)
    rot, trans = pose_delta(se3_exp(xi), s.T())
    assert rot < 1.5e-3 and trans < 4e-3, (rot, trans)
    st = s.stats()
    assert st.n_outer == 3 and st.status == 0 and 3 <= st.n_irls <= 30


def test_moving_sphere_is_segmented_dynamic(ora, pair):
    pr = pair(seed=1234, sphere=True, rows=240, cols=320)
    s = make_solver(ora, 240, 320, driver_params(ora), pr)
    s.build_pyramid(True)
    s.run_solver(True)
    s.build_segm_image()
    b, lab = s.b(), s.labels(0)
    # which clusters cover the sphere (centre (0.2,0.1,1.5) m, r=0.25 m -> image disc)
    f = 320 / (2 * np.tan(0.5 * np.pi * 62.5 / 180))
    v, u = np.mgrid[0:240, 0:320]
    disc = (u - (159.5 + f * 0.25 / 1.5)) ** 2 + (v - (119.5 + f * 0.1 / 1.5)) ** 2 < (0.7 * f * 0.25 / 1.5) ** 2
    sphere_labels = np.unique(lab[disc])
    assert len(sphere_labels) >= 1
    assert np.all(b[sphere_labels] < 0.5), b[sphere_labels]
    background = np.setdiff1d(np.unique(lab[~disc]), sphere_labels)
    assert np.mean(b[background] > 0.5) > 0.9
    bi = s.b_image()
    assert bi[disc].mean() < 0.3 and bi[~disc].mean() > 0.9
    assert 1e-4 < (bi < 0.5).mean() < 0.2
    rot, trans = pose_delta(pr["T_gt"], s.T())
    assert rot < 2e-3 and trans < 5e-3


def test_kmeans_centres_are_sequential_means_of_their_members(ora, pair):
    """Fixed point of the Lloyd update (KMeans.cpp:215-221): after kMeans3DCoord every centre is the
    float32 sum of its level-1 members, added one by one in column-major pixel order, divided by the
    member count.  Invalid pixels carry label 24 at every level; the L0 label of a pixel is a valid
    cluster; connectivity is symmetric with a true diagonal."""
    pr = pair(seed=1234, sphere=True, rows=240, cols=320)
    d_new = pr["new"][0].copy()
    d_new[100:140, 200:260] = 0  # a hole: invalid pixels
    s = make_solver(ora, 240, 320, driver_params(ora))
    s.set_current(0, d_new, pr["new"][1])
    s.build_pyramid(False)
    s.kmeans()
    lab1 = s.labels(1)
    z, x, y = (s.plane(capi.SET_NEW, ch, 1) for ch in (capi.CH_DEPTH, capi.CH_XX, capi.CH_YY))
    cent = s.kmeans_centres()  # (3, 24): rows z, x, y
    order = np.argsort(np.arange(lab1.size).reshape(lab1.shape).T.ravel())  # column-major walk
    flat = lambda a: a.T.ravel()
    L, Z, X, Y = flat(lab1), flat(z), flat(x), flat(y)
    assert np.array_equal(L == 24, Z == 0)
    for l in range(24):
        m = L == l
        if not m.any():
            assert np.all(cent[:, l] == 0)
            continue
        for r, P in enumerate((Z, X, Y)):
            acc = np.add.accumulate(P[m], dtype=np.float32)[-1]  # strictly sequential float32 sum
            assert cent[r, l] == np.float32(acc) / np.float32(m.sum()), (l, r)
    for lev in range(5):
        lab = s.labels(lev)
        dep = s.plane(capi.SET_NEW, capi.CH_DEPTH, lev)
        assert np.array_equal(lab == 24, dep == 0), lev
        assert lab.min() >= 0 and lab.max() <= 24
    conn = s.connectivity()
    assert conn.diagonal().all() and np.array_equal(conn, conn.T)
    assert 24 < conn.sum() < 24 * 24  # some, not all, clusters touch
    assert s.stats().kmeans_iters >= 1


def test_build_segm_image_semantics(ora):
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "segm_image_160x120.npz"))
    s = make_solver(ora, 120, 160, driver_params(ora))
    s.set_segm_state(0, g["labels0"], g["b_segm"], g["cluster_res"])
    s.build_segm_image()
    assert np.array_equal(s.b_image(), g["b_image"])


def test_empty_and_degenerate_inputs(ora, pair):
    z = np.zeros((60, 80), f32)
    s = make_solver(ora, 60, 80, driver_params(ora), {"new": (z, z), "old": (z, z)})
    s.build_pyramid(True)
    s.run_solver(True)
    s.build_segm_image()
    st = s.stats()
    assert st.status & capi.STATUS_EMPTY_LEVEL
    assert np.array_equal(s.T(), np.eye(4, dtype=f32))
    assert np.all(s.labels(0) == 24) and np.all(s.b_image() == 1.0)
    # half of the image invalid: still solves, invalid pixels keep label 24
    pr = pair(seed=5, rows=60, cols=80)
    d_new = pr["new"][0].copy()
    d_new[:, :40] = 0
    s = make_solver(ora, 60, 80, driver_params(ora), {"new": (d_new, pr["new"][1]), "old": pr["old"]})
    s.build_pyramid(True)
    s.run_solver(True)
    assert s.stats().status == 0 and np.isfinite(s.T()).all()
    assert np.all(s.labels(0)[:, :40] == 24) and np.all(s.labels(0)[:, 41:] < 24)


def test_kmeans_refuses_odd_image_sizes(ora):
    """KMeans.cpp:267: labels_lowres(v/2, u/2) lies outside the rows/2 x cols/2 label matrix for an odd image size -- the reference
    reads whatever is there. The oracle's bounds-checked containers threw on it (48 x 43, round 6); the calls that run K-means
    refuse such a handle, on both sides of the ABI."""
    import staticfusion_amd as sf

    for rows, cols in ((48, 43), (45, 48)):
        s = sf.Solver(ora, rows, cols, 1, driver_params(ora, ctf_levels=2))
        s.build_pyramid(True)
        for call in (s.kmeans, lambda: s.run_solver(True), lambda: s.process_frame(0)):
            with pytest.raises(sf.SfError, match="even rows and cols"):
                call()
    s = sf.Solver(ora, 48, 43, 1, config2_params(ora, levels=2))  # pure odometry has no such read
    s.build_pyramid(True)
    s.run_solver(True)
    assert np.isfinite(s.T()).all()


def test_frame_sequence_history_and_residuals(ora, pair):
    """process_frame over 7 frames of a static scene: the 5-frame residual check runs from frame 5 on and
    reports small residuals for static clusters (computeResidualsAgainstPreviousImage)."""
    from staticfusion_amd.synth import Scene, quantise_and_decimate

    scene = Scene(seed=77)
    xi = np.array(DEFAULT_XI) * 0.5
    frames = []
    T = np.eye(4)
    for k in range(8):
        frames.append(quantise_and_decimate(*scene.render(T, 320, 240)))
        T = T @ se3_exp(xi)
    s = make_solver(ora, 120, 160, driver_params(ora, kb=1.5))
    s.set_current(0, *frames[0])
    s.current_to_prediction()
    s.push_history(0)
    for k in range(1, 8):
        s.set_prediction(0, *frames[k - 1])  # frame-to-frame mode (bootstrap of the reference drivers)
        s.set_current(0, *frames[k])
        s.process_frame(k)
        rot, trans = pose_delta(se3_exp(xi), s.T())
        assert rot < 3e-3 and trans < 6e-3, (k, rot, trans)
        cr = s.cluster_residuals()
        if k < 5:
            assert np.isnan(cr).all()
        else:
            assert np.isfinite(cr).sum() >= 12 and np.nanmax(cr) < 0.2
    assert (s.b_image() > 0.5).mean() > 0.9


def test_behind_camera_rule_switch_changes_only_such_inputs(ora):
    """sfo_test_set_hip_behind_camera_rule (the HIP build's treatment of points warped behind the camera):
    identical results on an ordinary pair (no such point), fewer valid pixels on a pair built to have them."""
    import ctypes

    from conftest import driver_params, make_solver
    from staticfusion_amd.synth import make_pair

    lib = ora.lib
    lib.sfo_test_set_hip_behind_camera_rule.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sfo_test_behind_camera_valid.restype = ctypes.c_longlong
    lib.sfo_test_behind_camera_valid.argtypes = [ctypes.c_void_p, ctypes.c_int]
    ordinary = make_pair(seed=11, sphere=True, out_rows=60, out_cols=80)
    near = make_pair(seed=5, sphere=False, out_rows=60, out_cols=80, xi=(0.0, 0.0, 0.12, 0.0, 0.0, 0.0))
    d_old = near["old"][0].copy()
    d_old[20:40, 30:50] = 0.06
    near = {"old": (d_old, near["old"][1]), "new": near["new"]}
    for pr, expect_behind in ((ordinary, False), (near, True)):
        res = []
        for rule in (0, 1):
            s = make_solver(ora, 60, 80, driver_params(ora), pr)
            assert lib.sfo_test_set_hip_behind_camera_rule(s.h, rule) == 0
            s.build_pyramid(True)
            s.run_solver(True)
            res.append((s.T().copy(), [s.stats().outer[i].n_valid for i in range(s.stats().n_outer)], lib.sfo_test_behind_camera_valid(s.h, 0)))
        (T0, nv0, behind0), (T1, nv1, _) = res
        assert (behind0 > 0) == expect_behind
        if expect_behind:
            assert nv0 != nv1 and sum(nv1[:len(nv0)]) < sum(nv0[:len(nv1)]) + 1
        else:
            assert nv0 == nv1 and np.array_equal(T0, T1)
