"""More GPU parity: a randomised parameter sweep, a 30-frame sequence through the input stage with
carried solver state, and the bench-size batch (every duplicate stream bit-identical)."""
import os

import numpy as np
import pytest

from conftest import config2_params, driver_params, make_solver
from staticfusion_amd import capi
from staticfusion_amd.synth import DEFAULT_XI, LCG64, Scene, pose_delta, se3_exp
from test_gpu_parity import POSE_TOL, assert_traces_match, solve_both

pytestmark = pytest.mark.gpu


def _sweep_cases():
    """10 cases in the suite; SF_SWEEP_CASES=first:last runs another range of the generator (one-off hunts)"""
    spec = os.environ.get("SF_SWEEP_CASES")
    if not spec:
        return range(10)
    first, last = (int(x) for x in spec.split(":"))
    return range(first, last)


def sweep_case(hip, ora, pair, case, tol_twist=5e-6, tol_b=5e-4, rtol_aver=1e-3):
    """parameters drawn from the ranges the drivers / constructor use (StaticFusion-datasets.cpp:79-94, FrontEnd.cpp:57-76)"""
    g = LCG64(9000 + case)
    u = lambda lo, hi: lo + (hi - lo) * g.uniform()
    sphere = case % 2 == 0
    over = dict(ctf_levels=2 + int(u(0, 2.999)), max_iter_per_level=1 + int(u(0, 2.999)), max_iter_irls=2 + int(u(0, 8.999)),
                use_motion_filter=int(u(0, 1.999)), segmentation_enabled=1 if sphere else int(u(0, 1.999)),
                irls_delta_threshold=float(10 ** u(-6, -2.5)), kc_Cauchy=float(u(0.3, 1.0)), kz=float(u(1.0, 2.0)),
                lambda_reg=float(u(0.1, 0.6)), lambda_prior=float(u(0.2, 0.8)), k_photometric_res=float(u(0.08, 0.3)),
                previous_speed_const_weight=float(u(0.02, 0.2)), previous_speed_eig_weight=float(u(0.3, 2.5)))
    kb = float(u(1.0, 1.6))
    xi = tuple(float(u(0.4, 1.6)) * np.array(DEFAULT_XI))
    pr = pair(seed=300 + case, sphere=sphere, rows=120, cols=160, xi=xi)
    sg, so = solve_both(hip, ora, 120, 160, lambda a: driver_params(a, kb=kb, **over), pr)
    assert_traces_match(sg, so, tol_twist=tol_twist, tol_b=tol_b, rtol_aver=rtol_aver)
    rot, trans = pose_delta(so.T(), sg.T())
    assert rot <= POSE_TOL and trans <= POSE_TOL
    for L in range(sg.levels):
        assert np.array_equal(sg.labels(L), so.labels(L))
    assert np.array_equal(sg.b_image() > 0.5, so.b_image() > 0.5)


@pytest.mark.parametrize("case", _sweep_cases())
def test_random_parameter_sweep(hip, ora, pair, case):
    sweep_case(hip, ora, pair, case)


def test_sequence_through_the_input_stage(hip, ora):
    """30 decoded VGA frames: loader + bilateral filter + the solver with carried state (twist_old, b, 5-frame
    ring, residual check from frame 5 on), frame-to-frame mode. Per-frame pose parity and the drift of the
    accumulated trajectory between the two implementations."""
    scene = Scene(seed=4242, sphere=True)
    g = LCG64(77)
    T, frames = np.eye(4), []
    xi = np.array(DEFAULT_XI) * 0.5
    for k in range(30):  # smooth random walk of the twist
        xi = 0.85 * xi + 0.15 * np.array(DEFAULT_XI) * np.array([g.uniform(-1.5, 1.5) for _ in range(6)])
        depth, inten = scene.render(T, 640, 480, sphere_offset=(0.015 * k, 0.0, 0.004 * k))
        d_mm = np.clip(np.rint(depth * 1000.0), 0, 65535).astype(np.uint16)
        g8 = np.clip(np.rint(inten * 255.0), 0, 255).astype(np.uint8)
        frames.append((np.ascontiguousarray(np.repeat(g8[::-1, :, None], 3, axis=2)), np.ascontiguousarray(d_mm[::-1])))
        T = T @ se3_exp(xi)
    solvers = [make_solver(api, 240, 320, driver_params(api, kb=1.05)) for api in (hip, ora)]
    acc = [np.eye(4), np.eye(4)]
    for s in solvers:
        s.load_frame(0, *frames[0], 2)
        s.filter_depth()
        s.current_to_prediction()
        s.push_history(0)
    for k in range(1, 30):
        for i, s in enumerate(solvers):
            s.load_frame(0, *frames[k], 2)
            s.filter_depth()
            s.process_frame(k)
            acc[i] = acc[i] @ s.T().astype(np.float64)
            s.current_to_prediction()
        sg, so = solvers
        rot, trans = pose_delta(so.T(), sg.T())
        assert rot <= POSE_TOL and trans <= POSE_TOL, (k, rot, trans)
        assert np.array_equal(sg.labels(0), so.labels(0)), k
        assert np.array_equal(sg.b_image() > 0.5, so.b_image() > 0.5), k
        assert (sg.stats().n_outer, sg.stats().n_irls) == (so.stats().n_outer, so.stats().n_irls), k
    rot, trans = pose_delta(acc[0], acc[1])
    assert rot <= 1e-4 and trans <= 1e-4, (rot, trans)  # 29 frames of accumulated difference


def test_bench_size_batch_duplicates_identical(hip, pair):
    """2048 QVGA streams (8 distinct pairs tiled), one launch: every copy of a pair gives bit-identical T, b and counters."""
    prs = [pair(seed=1234 + i, rows=240, cols=320) for i in range(8)]
    B = 2048
    s = make_solver(hip, 240, 320, config2_params(hip, levels=3), batch=B)
    for b in range(B):
        s.set_current(b, *prs[b % 8]["new"])
        s.set_prediction(b, *prs[b % 8]["old"])
    s.process_frame(0)
    T, n_irls, n_outer, pix = s.batch_results()
    for b in range(8, B):
        assert np.array_equal(T[b], T[b % 8]) and n_irls[b] == n_irls[b % 8] and pix[b] == pix[b % 8], b
    assert len({tuple(T[i].ravel()) for i in range(8)}) == 8  # and the distinct pairs give distinct poses


def test_stream_order_changes_nothing_but_the_schedule(hip, pair):
    """With more streams than resident workgroups a launch hands the streams out longest-expected-first (a device-side
    counting sort by the IRLS iterations of each stream's previous frame, sf_hip.hip). Streams of very different cost, two
    frames each: every result equals the run with the plain queue order (SF_NO_STREAM_ORDER)."""
    if hip.default_variant == "cluster":
        pytest.skip("every workgroup of a cluster launch is resident: no queue")
    B = 1600  # more than 5 x 256 resident workgroups
    easy = pair(seed=61, rows=60, cols=80, sphere=True)
    hard = pair(seed=62, rows=60, cols=80, sphere=True, xi=tuple(3.0 * np.array(DEFAULT_XI)))
    out = []
    for no_order in ("", "1"):
        if no_order:
            os.environ["SF_NO_STREAM_ORDER"] = "1"
        try:
            s = make_solver(hip, 60, 80, driver_params(hip), batch=B)
            for b in range(B):
                pr = hard if b % 7 == 3 else easy
                s.set_current(b, *pr["new"])
                s.set_prediction(b, *pr["old"])
            s.process_frame(0)
            s.process_frame(1)  # ordered by the counts of frame 0
            T, n_irls, n_outer, pix = s.batch_results()
            out.append((T.copy(), n_irls.copy(), pix.copy(), s.b_image(3).copy(), s.b_image(0).copy()))
        finally:
            os.environ.pop("SF_NO_STREAM_ORDER", None)
    assert len(set(out[0][1].tolist())) > 1, "the streams should differ in cost"
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


def test_overlapped_upload_equals_direct_upload(hip, pair):
    """sf_upload_current_async + sf_commit_upload (second HIP stream, page-locked host buffers) hands the solver the
    same frames as sf_set_current, also when the next upload is in flight during a solve."""
    import ctypes as C

    B, n0 = 6, 120 * 160
    prs = [pair(seed=40 + i, rows=120, cols=160) for i in range(3)]
    fp = C.POINTER(C.c_float)
    ref = make_solver(hip, 120, 160, config2_params(hip, levels=3), batch=B)
    s = make_solver(hip, 120, 160, config2_params(hip, levels=3), batch=B)
    host = []
    for which in ("old", "new"):
        pd, pi = C.c_void_p(), C.c_void_p()
        hip.check(hip.alloc_pinned(4 * n0 * B, C.byref(pd)))
        hip.check(hip.alloc_pinned(4 * n0 * B, C.byref(pi)))
        d = np.ctypeslib.as_array(C.cast(pd, fp), shape=(B, n0))
        i = np.ctypeslib.as_array(C.cast(pi, fp), shape=(B, n0))
        for b in range(B):
            d[b] = np.ascontiguousarray(prs[b % 3][which][0].T).ravel()
            i[b] = np.ascontiguousarray(prs[b % 3][which][1].T).ravel()
        host.append((C.cast(pd, fp), C.cast(pi, fp), pd, pi))
    # reference path: frame "old" then "new" through the per-stream setters
    for b in range(B):
        ref.set_current(b, *prs[b % 3]["old"])
    ref.current_to_prediction()
    for b in range(B):
        ref.set_current(b, *prs[b % 3]["new"])
    ref.process_frame(0)
    # overlapped path: the upload of "new" is issued while the (dummy) solve of "old" runs
    hip.check(hip.upload_current_async(s.h, host[0][0], host[0][1]))
    hip.check(hip.commit_upload(s.h))
    s.current_to_prediction()
    s.build_pyramid(True)  # some work on the compute stream
    hip.check(hip.upload_current_async(s.h, host[1][0], host[1][1]))
    assert hip.upload_current_async(s.h, host[1][0], host[1][1]) == -3  # SF_ERR_STATE: one upload at a time
    hip.check(hip.commit_upload(s.h))
    assert hip.commit_upload(s.h) == -3
    s.process_frame(0)
    Ta, _, _, _ = ref.batch_results()
    Tb, _, _, _ = s.batch_results()
    assert np.array_equal(Ta, Tb)
    s.synchronize()
    for hst in host:
        hip.check(hip.free_pinned(hst[2]))
        hip.check(hip.free_pinned(hst[3]))


def test_parameter_setters_and_external_hip_stream(hip, ora, pair):
    """sf_set_params / sf_set_kb / sf_set_twist_old change the next solve exactly like a handle created with those
    values (and like the oracle); sf_set_hip_stream runs the handle on a caller-owned hipStream_t."""
    import ctypes as C

    pr = pair(seed=21, sphere=True, rows=120, cols=160)
    tw = np.array([0.004, -0.002, 0.003, 0.001, -0.002, 0.0015], np.float32)
    res = {}
    for name, api in (("hip", hip), ("ora", ora)):
        # (a) created with the final values
        a = make_solver(api, 120, 160, driver_params(api, kb=1.3, max_iter_irls=4, ctf_levels=4), pr)
        a.set_twist_old(0, tw)
        a.build_pyramid(True); a.run_solver(True)
        # (b) created with other values, then changed through the setters
        b = make_solver(api, 120, 160, driver_params(api, kb=1.05, max_iter_irls=6, ctf_levels=4), pr)
        p = api.default_params_struct()
        api.check(api.get_params(b.h, C.byref(p)))
        p.max_iter_irls = 4
        b.set_params(p)
        b.set_kb(1.3)
        b.set_twist_old(0, tw)
        b.build_pyramid(True); b.run_solver(True)
        assert np.array_equal(a.T(), b.T()) and a.stats().n_irls == b.stats().n_irls
        assert np.array_equal(a.twist_old(), b.twist_old())
        res[name] = (a.T(), a.stats().n_irls)
    assert res["hip"][1] == res["ora"][1]
    rot, trans = pose_delta(res["ora"][0], res["hip"][0])
    assert rot <= 1e-4 and trans <= 1e-4
    # without the carried twist the motion filter gives a different answer: the setter is not a no-op
    c = make_solver(hip, 120, 160, driver_params(hip, kb=1.3, max_iter_irls=4, ctf_levels=4), pr)
    c.build_pyramid(True); c.run_solver(True)
    assert not np.array_equal(c.T(), res["hip"][0])
    assert c.last_solver_kernel_ms() > 0.0
    # caller-owned stream
    hiprt = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    assert hiprt.hipStreamCreate(C.byref(st)) == 0
    d = make_solver(hip, 120, 160, driver_params(hip, kb=1.3, max_iter_irls=4, ctf_levels=4), pr)
    hip.check(hip.set_hip_stream(d.h, st))
    d.build_pyramid(True); d.run_solver(True)
    assert hiprt.hipStreamSynchronize(st) == 0
    assert np.array_equal(d.T(), c.T())
    hip.check(hip.set_hip_stream(d.h, None))
    d.close()
    assert hiprt.hipStreamDestroy(st) == 0


def test_bench_two_rank_flow_on_one_gpu(tmp_path):
    """bench.py under torch.distributed.run with two ranks (both on cuda:0, gloo for the barrier / reductions):
    the N > 1 control flow -- per-rank batches, barrier-bracketed timing, MAX / SUM reductions, one JSON line from rank 0."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, SF_BENCH_BACKEND="gloo", SF_BENCH_SINGLE_GPU="1")
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                   "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "128", "--steps", "2",
                                   "--warmup", "1", "--distinct", "2"], env=env, cwd=ROOT, stderr=subprocess.STDOUT, timeout=600).decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["frames_per_s"] * d["ms_per_step"] * 1e-3 - 2 * 128) < 1e-6 * 256  # whole-job frames per step = both ranks' batches
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    assert d["world_size"] == 2 and [r["rank"] for r in d["per_rank"]] == [0, 1] and all(r["frames_per_s"] > 0 for r in d["per_rank"])
    fs = d["full_solver"]  # BASELINE configs[2] timed in the same run
    assert fs["value"] > 0 and fs["roofline"]["frac"] > 0 and fs["pose_delta_vs_cpu"]["cluster_labels_identical"]
    assert fs["pose_delta_vs_cpu"]["rot_rad"] < 1e-4 and d["pose_delta_vs_cpu"]["rot_rad"] < 1e-4
    assert len(d["devices"]) == 2 and [x["rank"] for x in d["devices"]] == [0, 1]
    # the sequences workload through the same two-rank flow
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                   "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "sequences", "--batch", "32",
                                   "--steps", "3", "--warmup", "1", "--seq-distinct", "2", "--seq-frames", "16"], env=env, cwd=ROOT,
                                  stderr=subprocess.STDOUT, timeout=900).decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    assert d["world_size"] == 2 and d["config"]["distinct_sequences_per_rank"] == 2 and d["value"] > 0
    assert abs(d["frames_per_s"] * d["ms_per_step"] * 1e-3 - 2 * 32) < 1e-6 * 64
    assert d["pose_delta_vs_cpu"]["tracking_error_vs_ground_truth"]["trans_m_max"] < 0.02


def test_bench_gpus_n_started_directly(tmp_path):
    """`python bench.py --gpus N` WITHOUT torch.distributed.run around it (what a driver that runs the N = 1 command as plain
    `python bench.py --gpus 1` will most likely do for N = 8 too): --gpus 2 launches its two ranks itself (both on cuda:0 through
    the one-GPU test hook) and the line says world_size 2 with two per-rank entries; --gpus 8 on this one-GPU box exits non-zero
    instead of printing a line with "n_gpus": 8 from one process; a rank count that contradicts --gpus is refused as well."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SF_BENCH_BACKEND="gloo", SF_BENCH_SINGLE_GPU="1", SF_BENCH_CACHE=str(tmp_path))
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "sequences", "--batch", "32", "--steps", "3",
                                   "--warmup", "1", "--seq-distinct", "2", "--seq-frames", "16"], env=env, cwd=ROOT, stderr=subprocess.STDOUT,
                                  timeout=900).decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and len(d["per_rank"]) == 2 and len(d["devices"]) == 2
    assert abs(d["frames_per_s"] * d["ms_per_step"] * 1e-3 - 2 * 32) < 1e-6 * 64
    assert len(list(tmp_path.glob("sf_seq_*.npz"))) == 4  # two ranks x two sequences, rendered once and cached

    import torch

    if torch.cuda.device_count() < 8:
        env8 = {k: v for k, v in env.items() if k != "SF_BENCH_SINGLE_GPU"}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env8, cwd=ROOT,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode != 0 and not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        assert "refusing" in r.stderr.decode()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
