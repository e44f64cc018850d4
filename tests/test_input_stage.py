"""Input stage: loadImageFromSequenceAssoc's decimation (reference
FrontEnd.cpp:216-254) and getFilteredDepth = bilateral filter + metricise (Reconstruction.cpp:722-732,
Shaders/depth_bilateral.frag:34-74, depth_metric.frag:32-39).

CPU part: the oracle against the golden fixture of the independent NumPy derivation
(tools/golden/make_golden_input.py) and against hand cases. GPU part (-m gpu): the HIP kernels
against the oracle through the C ABI -- every output BIT-EXACT (integer millimetres, bytes, and the
float images, whose arithmetic is specified operation by operation)."""
import ctypes as C
import math
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, driver_params, make_solver
from staticfusion_amd import capi

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "golden"))


def run_input_stage(api, color, depth, res, batch=1, cutoff=None):
    rows, cols = depth.shape[0] // res, depth.shape[1] // res
    s = make_solver(api, rows, cols, driver_params(api, ctf_levels=2), batch=batch)
    if cutoff is not None:
        s.set_depth_cutoff(cutoff)
    for b in range(batch):
        s.load_frame(b, color, depth, res)
    loaded = s.current(0)
    s.filter_depth()
    return s, loaded


def outputs(s, stream=0):
    d, i = s.current(stream)
    return {"depth_current": d, "intensity": i, "depth_mm": s.input_image(capi.IN_DEPTH_MM, stream),
            "filtered_mm": s.input_image(capi.IN_DEPTH_FILTERED_MM, stream), "depth_metric": s.input_image(capi.IN_DEPTH_METRIC, stream),
            "color": s.input_image(capi.IN_COLOR, stream)}


def assert_same(a, b):
    for k in a:
        assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k  # bit exact


# ------------------------------------------------------------------------------------------------
#  CPU: oracle vs golden fixture / hand cases
# ------------------------------------------------------------------------------------------------
def test_exp_neg_matches_the_numpy_derivation_and_libm(ora):
    g = np.load(os.path.join(GOLDEN, "input_stage_exp.npz"))
    a = np.ascontiguousarray(g["a"], np.float32)
    out = np.zeros_like(a)
    fn = ora.lib.sfo_test_exp_neg
    fn.restype, fn.argtypes = None, [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]
    fn(a.ctypes.data_as(C.POINTER(C.c_float)), a.size, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(out, g["exp_neg"])  # same float32 operations -> same bits
    live = a <= 87.0
    exact = np.array([math.exp(-float(x)) for x in a[live]])
    assert np.abs(out[live] - exact).max() / 1.0 <= 1.0  # sanity
    assert (np.abs(out[live].astype(np.float64) - exact) / exact).max() < 2.0 * 2.0 ** -23  # <= 2 ulp
    assert np.all(out[~live] == 0.0)


def test_oracle_matches_golden_fixture(ora):
    g = np.load(os.path.join(GOLDEN, "input_stage_80x60.npz"))
    s, loaded = run_input_stage(ora, g["color_full"], g["depth_full"], int(g["res_factor"]))
    assert np.array_equal(loaded[0], g["depth_loaded"]) and np.array_equal(loaded[1], g["intensity"])
    o = outputs(s)
    for k in ("intensity", "depth_mm", "color", "filtered_mm", "depth_metric", "depth_current"):
        assert np.array_equal(o[k], g[k]), k
    assert (o["filtered_mm"] != o["depth_mm"]).mean() > 0.5  # the filter does something on this frame


def test_loader_indexing_flip_and_decimation(ora):
    H, W, res = 48, 64, 2
    yy, xx = np.mgrid[0:H, 0:W]
    depth = (1000 + yy * 64 + xx).astype(np.uint16)  # value encodes its own position
    color = np.stack([yy % 256, xx % 256, (yy + xx) % 256], axis=-1).astype(np.uint8)
    s = make_solver(ora, H // res, W // res, driver_params(ora, ctf_levels=2))
    s.load_frame(0, color, depth, res)
    mm = s.input_image(capi.IN_DEPTH_MM)
    for v, u in ((0, 0), (5, 7), (23, 31)):
        sr, sc = (H // res) * res - res * v - 1, res * u  # FrontEnd.cpp:231
        assert mm[v, u] == depth[sr, sc]
        assert tuple(s.input_image(capi.IN_COLOR)[v, u]) == tuple(color[sr, sc])
        f32 = np.float32
        c = color[sr, sc].astype(f32) * (f32(1) / f32(255))
        assert s.current()[1][v, u] == (f32(0.299) * c[0] + f32(0.587) * c[1]) + f32(0.114) * c[2]
        assert s.current()[0][v, u] == f32(depth[sr, sc]) * f32(1.0 / 1000.0)


def test_bilateral_hand_cases(ora):
    H, W = 24, 32
    color = np.zeros((H, W, 3), np.uint8)
    flat = np.full((H, W), 1234, np.uint16)
    s, _ = run_input_stage(ora, color, flat, 1)
    assert np.all(s.input_image(capi.IN_DEPTH_FILTERED_MM) == 1234)  # a constant image is a fixed point
    assert np.all(s.current()[0] == np.float32(1234) / np.float32(1000))
    # range gate: < 300 mm and > cutoff are invalid, the bounds themselves are valid (depth_bilateral.frag:36)
    gate = flat.copy()
    gate[0, 0], gate[0, 1], gate[0, 2], gate[0, 3] = 299, 300, 4500, 4501
    s, _ = run_input_stage(ora, color, gate, 1)
    # the loader flips vertically: source row 0 is output row H - 1
    f = s.input_image(capi.IN_DEPTH_FILTERED_MM)[H - 1]
    assert f[0] == 0 and f[1] != 0 and f[2] != 0 and f[3] == 0
    m = s.input_image(capi.IN_DEPTH_METRIC)[H - 1]
    assert m[0] == 0 and m[1] == np.float32(300) / np.float32(1000) and m[2] == np.float32(4.5) and m[3] == 0
    # a far-away neighbour has weight exp(-color2 * 0.000555556) == 0 exactly: the outlier does not leak
    spike = flat.copy()
    spike[10, 10] = 3000
    s, _ = run_input_stage(ora, color, spike, 1)
    f = s.input_image(capi.IN_DEPTH_FILTERED_MM)
    assert f[H - 1 - 10, 10] == 3000 and np.all(np.delete(f.ravel(), (H - 1 - 10) * W + 10) == 1234)
    # a lower cut-off invalidates what lies beyond it
    s, _ = run_input_stage(ora, color, flat, 1, cutoff=1.2)
    assert np.all(s.input_image(capi.IN_DEPTH_FILTERED_MM) == 0) and np.all(s.current()[0] == 0)


def test_input_stage_errors(ora):
    s = make_solver(ora, 24, 32, driver_params(ora, ctf_levels=2))
    assert ora.filter_depth(s.h) == -3  # SF_ERR_STATE: nothing loaded yet
    color, depth = np.zeros((50, 64, 3), np.uint8), np.zeros((50, 64), np.uint16)
    with pytest.raises(capi.SfError):
        s.load_frame(0, color, depth, 2)  # 50 / 2 != 24
    with pytest.raises(capi.SfError):
        s.set_depth_cutoff(0.0)


# ------------------------------------------------------------------------------------------------
#  GPU: HIP kernels vs oracle, bit exact
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_matches_golden_fixture(hip):
    g = np.load(os.path.join(GOLDEN, "input_stage_80x60.npz"))
    s, loaded = run_input_stage(hip, g["color_full"], g["depth_full"], int(g["res_factor"]))
    assert np.array_equal(loaded[0], g["depth_loaded"]) and np.array_equal(loaded[1], g["intensity"])
    o = outputs(s)
    for k in ("intensity", "depth_mm", "color", "filtered_mm", "depth_metric", "depth_current"):
        assert np.array_equal(o[k], g[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("full,res,seed", [((480, 640), 2, 1), ((480, 640), 2, 2), ((240, 320), 1, 3), ((480, 640), 4, 4), ((96, 128), 1, 5)])
def test_hip_input_stage_bit_exact_vs_oracle(hip, ora, full, res, seed):
    from make_golden_input import synth_frame

    color, depth = synth_frame(full[0], full[1], seed)
    sg, lg = run_input_stage(hip, color, depth, res)
    so, lo = run_input_stage(ora, color, depth, res)
    assert np.array_equal(lg[0], lo[0]) and np.array_equal(lg[1], lo[1])
    assert_same(outputs(sg), outputs(so))


@pytest.mark.gpu
def test_hip_input_stage_batch_device_buffers_and_solver_handover(hip, ora):
    """sf_load_frame_device (frames already in HBM) for a batch, then the solver consumes depthCurrent."""
    from make_golden_input import synth_frame

    B, H, W, res = 3, 240, 320, 2
    frames = [synth_frame(H, W, 10 + b) for b in range(B)]
    hiprt = C.CDLL("libamdhip64.so")
    col = np.ascontiguousarray(np.stack([f[0] for f in frames]))
    dep = np.ascontiguousarray(np.stack([f[1] for f in frames]))
    ptrs = []
    for arr in (col, dep):
        ptr = C.c_void_p()
        assert hiprt.hipMalloc(C.byref(ptr), C.c_size_t(arr.nbytes)) == 0
        assert hiprt.hipMemcpy(ptr, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), 1) == 0
        ptrs.append(ptr)
    p = driver_params(hip, ctf_levels=3)
    sg = make_solver(hip, H // res, W // res, p, batch=B)
    hip.check(hip.load_frame_device(sg.h, ptrs[0], ptrs[1], H, W, res))
    sg.current_to_prediction()  # bootstrap: prediction := the unfiltered loaded frame
    sg.filter_depth()
    so = make_solver(ora, H // res, W // res, driver_params(ora, ctf_levels=3), batch=B)
    for b in range(B):
        so.load_frame(b, frames[b][0], frames[b][1], res)
    so.current_to_prediction()
    so.filter_depth()
    for b in range(B):
        assert_same(outputs(sg, b), outputs(so, b))
    # the filtered frame is what createImagePyramid(true) now sees
    sg.build_pyramid(True)
    so.build_pyramid(True)
    for b in range(B):
        assert np.array_equal(sg.plane(capi.SET_NEW, capi.CH_DEPTH, 1, b), so.plane(capi.SET_NEW, capi.CH_DEPTH, 1, b))
    ms = C.c_float()
    hip.check(hip.timed_input_stage(sg.h, ptrs[0], ptrs[1], H, W, res, 2, C.byref(ms)))
    assert ms.value > 0
    for ptr in ptrs:
        hiprt.hipFree(ptr)
