"""Frame-to-model prediction without OpenGL: Reconstruction::getPredictedImages
(reference Reconstruction.cpp:628-720) = surfel point-sprite rendering at two confidence levels
(IndexMap.cpp:221-300, Shaders/splat.vert, combo_splat.frag) + density test + fill-in + depth extraction.
CPU part: the oracle on hand cases. GPU part: the HIP kernels against the oracle, bit exact."""
import numpy as np
import pytest

from conftest import driver_params, make_solver
from staticfusion_amd import capi
from staticfusion_amd.synth import DEFAULT_XI, Scene, se3_exp

ROWS, COLS = 240, 320


def encode_color(rgb):  # Shaders/color.glsl:19-25
    rgb = np.asarray(rgb, np.int64)
    return ((rgb[..., 0] << 16) + (rgb[..., 1] << 8) + rgb[..., 2]).astype(np.float32)


def surfels_from_frame(depth, rgb8, pose, mp, step=1, conf=None, seed=0):
    """One surfel per (step-th) valid pixel of a depth frame seen from `pose`, as GlobalModel's initialisation does
    (Shaders/init_unstable.vert): world position, normal from depth differences, radius by surfels.glsl getRadius."""
    rng = np.random.default_rng(seed)
    rows, cols = depth.shape
    yy, xx = np.mgrid[0:rows, 0:cols]
    z = depth.astype(np.float64)
    X = (xx - mp.cx) * z / mp.fx
    Y = (yy - mp.cy) * z / mp.fy
    P = np.stack([X, Y, z], -1)
    du = np.gradient(P, axis=1)
    dv = np.gradient(P, axis=0)
    nrm = np.cross(du, dv)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-12)
    nrm = np.where(nrm[..., 2:3] > 0, -nrm, nrm)  # towards the camera
    mean_focal = ((1.0 / abs(mp.fx)) + (1.0 / abs(mp.fy))) / 2.0
    radius = (z / (1.0 / mean_focal)) * 1.41421356237  # depth / focal * sqrt2 with focal = 1 / meanFocal
    radius_n = np.minimum(2.0 * radius, radius / np.maximum(np.abs(nrm[..., 2]), 1e-6))
    ok = (z > 0.45) & (z < 19.0)
    ok[::1, ::1] &= ((yy % step) == 0) & ((xx % step) == 0)
    R, t = pose[:3, :3].astype(np.float64), pose[:3, 3].astype(np.float64)
    Pw = P[ok] @ R.T + t
    Nw = nrm[ok] @ R.T
    n = Pw.shape[0]
    s = np.zeros((n, 12), np.float32)
    s[:, 0:3] = Pw
    s[:, 3] = rng.uniform(0.0, 0.6, n) if conf is None else conf
    s[:, 4] = encode_color(rgb8[ok])
    s[:, 6] = 0.0
    s[:, 7] = 0.0
    s[:, 8:11] = Nw
    s[:, 11] = radius_n[ok]
    return s


def synthetic_view(T, sphere=True):
    scene = Scene(seed=99, sphere=sphere)
    depth, inten = scene.render(T, 640, 480)
    depth = depth[::2, ::2]
    g = np.clip(np.rint(inten[::2, ::2] * 255), 1, 255).astype(np.uint8)  # >= 1: a drawn pixel is never black
    rgb = np.stack([g, np.clip(g.astype(int) + 7, 1, 255).astype(np.uint8), np.clip(g.astype(int) // 2 + 3, 1, 255).astype(np.uint8)], -1)
    return depth.astype(np.float32), rgb


def prime_stream(s, depth, rgb, b_value):
    """give the stream a previous frame: DEPTH_FILTERED / RGB of the input stage and a b image"""
    full_d = np.repeat(np.repeat(np.clip(np.rint(depth[::-1] * 1000), 0, 65535).astype(np.uint16), 2, 0), 2, 1)
    full_c = np.repeat(np.repeat(rgb[::-1], 2, 0), 2, 1)
    s.load_frame(0, full_c, full_d, 2)
    s.filter_depth()
    s.set_segm_state(0, np.zeros((ROWS, COLS), np.int32), np.full(24, b_value, np.float32), np.ones(24, np.float32))
    s.build_segm_image()


# ------------------------------------------------------------------------------------------------
def test_oracle_single_surfel_and_fill_in(ora):
    s = make_solver(ora, ROWS, COLS, driver_params(ora))
    mp = s.default_model_params()
    assert mp.cx == 160 and mp.cy == 120 and abs(mp.fx - 0.5 * 320 / np.tan(np.pi * 62.5 / 360)) < 1e-3 and mp.conf_high == 0.25
    depth, rgb = synthetic_view(np.eye(4), sphere=False)
    prime_stream(s, depth, rgb, b_value=0.9)  # b > 0.6: the raw depth may fill in
    filt = s.input_image(capi.IN_DEPTH_FILTERED_MM)
    # (1) empty model: not dense -> everything comes from the previous frame where 0 < z <= 4.5
    s.predict_from_model(0, np.zeros((0, 12), np.float32), np.eye(4))
    d, i = s.prediction()
    zr = filt.astype(np.float32) / np.float32(1000)
    assert np.array_equal(d, np.where((zr > 4.5) | (zr <= 0), 0, zr).astype(np.float32))
    col = s.input_image(capi.IN_COLOR).astype(np.float32) * (np.float32(1) / np.float32(255))
    assert np.array_equal(i, (np.float32(0.299) * col[..., 0] + np.float32(0.587) * col[..., 1]) + np.float32(0.114) * col[..., 2])
    # (2) one fronto-parallel surfel on the optical axis: a disc of radius fx r / z pixels at depth z, colour from the surfel
    surf = np.zeros((1, 12), np.float32)
    surf[0, :4] = (0, 0, 2.0, 0.5)
    surf[0, 4] = encode_color(np.array([200, 100, 50]))
    surf[0, 8:11] = (0, 0, -1)
    surf[0, 11] = 0.05
    s.set_segm_state(0, np.zeros((ROWS, COLS), np.int32), np.full(24, 0.1, np.float32), np.ones(24, np.float32))
    s.build_segm_image()  # b < 0.6 everywhere: no depth fill-in
    s.predict_from_model(0, surf, np.eye(4))
    d, i = s.prediction()
    hit = d > 0
    assert np.all(d[hit] == np.float32(2.0))
    yy, xx = np.mgrid[0:ROWS, 0:COLS]
    r_px = np.hypot(xx + 0.5 - mp.cx, yy + 0.5 - mp.cy)
    r_disc = mp.fx * 0.05 / 2.0
    assert np.all(hit[r_px < r_disc - 1.0]) and not np.any(hit[r_px > r_disc + 1.0])
    want = np.float32(0.299) * (np.float32(200) / np.float32(255)) + np.float32(0.587) * (np.float32(100) / np.float32(255))
    assert np.all(np.abs(i[hit] - (want + np.float32(0.114) * (np.float32(50) / np.float32(255)))) < 1e-6)
    # (3) below the high-confidence threshold the surfel is still drawn (low pass), below the low one it is not
    surf[0, 3] = 0.2
    s.predict_from_model(0, surf, np.eye(4))
    assert (s.prediction()[0] > 0).sum() == hit.sum()
    surf[0, 3] = 0.1
    s.predict_from_model(0, surf, np.eye(4))
    assert (s.prediction()[0] > 0).sum() == 0
    # (3b) GL_LESS against the cleared depth buffer: gl_FragDepth = z / (2 maxDepth) + 0.5 reaches 1.0 AT maxDepth -> not drawn
    near = mp.max_depth
    mp.max_depth = 2.0
    surf[0, 3] = 0.5
    s.predict_from_model(0, surf, np.eye(4), mp)  # z = 2.0 = maxDepth: passes the cull (z > maxDepth is false), fails the depth test
    assert (s.prediction()[0] > 0).sum() == 0
    surf[0, 2] = np.float32(1.9999)
    s.predict_from_model(0, surf, np.eye(4), mp)
    assert (s.prediction()[0] > 0).sum() > 0
    surf[0, 2] = 2.0
    mp.max_depth = near
    # (4) a nearer surfel hides a farther one; beyond extract_max_depth nothing is reported
    two = np.repeat(surf, 2, 0)
    two[:, 3] = 0.5
    two[1, 2] = 1.0
    s.predict_from_model(0, two, np.eye(4))
    assert s.prediction()[0][120, 160] == np.float32(1.0)
    two[:, 2] = (6.0, 7.0)
    s.predict_from_model(0, two, np.eye(4))
    assert (s.prediction()[0] > 0).sum() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("step,b_value,seed", [(1, 0.9, 1), (1, 0.3, 2), (3, 0.9, 3), (6, 0.9, 4)])
def test_hip_prediction_bit_exact_vs_oracle(hip, ora, step, b_value, seed):
    """a model built from view 0 predicted from a moved camera: dense (step 1) and sparse models (fill-in path),
    random confidences around both thresholds, hidden surfaces behind the sphere"""
    depth0, rgb0 = synthetic_view(np.eye(4))
    T1 = se3_exp(np.array(DEFAULT_XI) * 3.0)
    depth1, rgb1 = synthetic_view(T1)
    out = []
    for api in (hip, ora):
        s = make_solver(api, ROWS, COLS, driver_params(api))
        mp = s.default_model_params()
        prime_stream(s, depth1, rgb1, b_value)
        surf = surfels_from_frame(depth0, rgb0, np.eye(4), mp, step=step, seed=seed)
        s.predict_from_model(0, surf, T1.astype(np.float32), mp)
        out.append(s.prediction())
        # the prediction feeds the solver's old pyramid
        s.build_pyramid(True)
        out[-1] = out[-1] + (s.plane(capi.SET_PRED, capi.CH_DEPTH, 1),)
    (dg, ig, pg), (do, io, po) = out
    assert np.array_equal(dg, do) and np.array_equal(ig, io) and np.array_equal(pg, po)
    assert (do > 0).mean() > 0.5
    if step == 1:  # a dense model seen from nearby reproduces the scene
        err = np.abs(do - depth1)[(do > 0) & (depth1 > 0) & (depth1 < 4.4)]
        assert np.median(err) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["shuffled", "column", "reversed"])
def test_hip_prediction_any_buffer_order(hip, ora, order):
    """the splat resolves a workgroup's fragments in an LDS tile when its surfels are neighbours (tall box for the map's
    column order, wide box for a row-ordered buffer) and falls back to per-fragment global atomics when they are not
    (shuffled): the same images as the oracle in every case (ties go to the lower index OF THAT ORDER on both sides)"""
    depth0, rgb0 = synthetic_view(np.eye(4))
    T1 = se3_exp(np.array(DEFAULT_XI) * 2.0)
    depth1, rgb1 = synthetic_view(T1)
    out = []
    for api in (hip, ora):
        s = make_solver(api, ROWS, COLS, driver_params(api))
        mp = s.default_model_params()
        prime_stream(s, depth1, rgb1, 0.3)
        surf = surfels_from_frame(depth0, rgb0, np.eye(4), mp, step=1, seed=7)
        if order == "shuffled":
            surf = surf[np.random.default_rng(11).permutation(surf.shape[0])]
        elif order == "column":  # the reference's point order: x outer, y inner
            px = np.round(surf[:, 0] / surf[:, 2] * mp.fx + mp.cx - 0.0).astype(int)
            py = np.round(surf[:, 1] / surf[:, 2] * mp.fy + mp.cy - 0.0).astype(int)
            surf = surf[np.lexsort((py, px))]
        else:
            surf = surf[::-1].copy()
        s.predict_from_model(0, surf, T1.astype(np.float32), mp)
        out.append(s.prediction())
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert (out[1][0] > 0).mean() > 0.5


# ------------------------------------------------------------------------------------------------
#  GlobalModel::initialise: the surfel model of the first fused frame
# ------------------------------------------------------------------------------------------------
def load_view(s, depth, rgb, b_value):
    prime_stream(s, depth, rgb, b_value)  # sf_load_frame + sf_filter_depth + a b image


def test_oracle_init_model_hand_cases(ora):
    s = make_solver(ora, ROWS, COLS, driver_params(ora))
    mp = s.default_model_params()
    depth = np.full((ROWS, COLS), 2.0, np.float32)  # a fronto-parallel wall at 2 m
    depth[:, :40] = 0.0     # invalid band
    depth[:20, :] = 25.0    # beyond max_depth = 20 (and beyond the bilateral filter's 4.5 m gate)
    rgb = np.zeros((ROWS, COLS, 3), np.uint8)
    rgb[..., 0], rgb[..., 1], rgb[..., 2] = 10, 20, 30
    load_view(s, depth, rgb, b_value=0.8)
    surf = s.init_model_from_frame(0, np.eye(4), mp, time=7)
    mm = s.input_image(capi.IN_DEPTH_MM)
    assert surf.shape[0] == int(((mm >= 300) & (mm <= 4500)).sum())  # DEPTH_METRIC gates the raw list (depth_metric.frag)
    assert np.all(surf[:, 2] == np.float32(2.0)) and np.all(surf[:, 4] == np.float32((10 << 16) + (20 << 8) + 30))
    assert np.all(surf[:, 5] == 1) and np.all(surf[:, 6] == 1) and np.all(surf[:, 7] == 7)
    assert np.all(surf[:, 3] == np.float32(round(0.8 * 255) / 255.0))  # confidence = b through the 8-bit colour encoding
    inner = (surf[:, 0] > -0.5) & (surf[:, 0] < 0.5) & (np.abs(surf[:, 1]) < 0.3)
    assert np.allclose(surf[inner, 8:11], [0, 0, 1], atol=1e-6)         # geometry.glsl:36-39: cross(-x, -y) = +z for a fronto-parallel wall
    r = 2.0 / (0.5 * (mp.fx + mp.fy)) * np.sqrt(2.0)
    assert np.allclose(surf[inner, 11], r, rtol=1e-5)                   # surfels.glsl getRadius
    # first surfel: point order is x outer, y inner; x = i + 0.5
    i0, j0 = 40, 20
    assert surf[0, 0] == pytest.approx((i0 + 0.5 - mp.cx) * 2.0 / mp.fx, rel=1e-6) and surf[0, 1] == pytest.approx((j0 + 0.5 - mp.cy) * 2.0 / mp.fy, rel=1e-6)
    # a pose moves positions and normals
    T = se3_exp(np.array([0.1, -0.2, 0.3, 0.0, 0.0, np.pi / 2])).astype(np.float32)
    moved = s.init_model_from_frame(0, T, mp, time=7)
    assert np.allclose(moved[:, :3], surf[:, :3] @ T[:3, :3].T + T[:3, 3], atol=2e-6)
    assert np.allclose(moved[inner, 8:11], np.array([0, 0, 1]) @ T[:3, :3].T, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("b_value,seed", [(0.9, 1), (0.37, 2)])
def test_hip_init_model_bit_exact_and_closed_loop(hip, ora, b_value, seed):
    """GlobalModel::initialise on a synthetic view: HIP == oracle bit for bit (NaN normals at degenerate pixels
    included); the model predicted from a second pose resembles the second view."""
    T0 = np.eye(4)
    T1 = se3_exp(np.array(DEFAULT_XI) * 2.0)
    depth0, rgb0 = synthetic_view(T0)
    depth0 = depth0.copy()
    rng = np.random.default_rng(seed)
    depth0[rng.random(depth0.shape) < 0.01] = 0.0  # holes: raw and filtered validity differ near them
    depth1, rgb1 = synthetic_view(T1)
    models, preds = [], []
    for api in (hip, ora):
        s = make_solver(api, ROWS, COLS, driver_params(api))
        mp = s.default_model_params()
        load_view(s, depth0, rgb0, b_value)
        m = s.init_model_from_frame(0, T0.astype(np.float32), mp, time=1)
        models.append(m)
        mp.time = mp.max_time = 2  # the next tick: surfels stamped later than max_time are not drawn (splat.vert:57)
        s.predict_from_model(0, m, T1.astype(np.float32), mp)
        preds.append(s.prediction())
    assert models[0].shape == models[1].shape and models[0].tobytes() == models[1].tobytes()
    assert np.array_equal(preds[0][0], preds[1][0]) and np.array_equal(preds[0][1], preds[1][1])
    if b_value > 0.25:  # confidence above the high threshold: the model alone draws the prediction
        d = preds[1][0]
        both = (d > 0) & (depth1 > 0) & (depth1 < 4.4)
        assert both.mean() > 0.5 and np.median(np.abs(d - depth1)[both]) < 0.01
