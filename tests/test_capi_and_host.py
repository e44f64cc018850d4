"""CPU-only checks of the boundary and the host logic: the C-ABI library loads and exports every
symbol include/sf.h declares (no compute calls), the ctypes table matches the header, parameter
defaults match the reference's driver values, the synthetic generator and the bench's algorithmic
byte count behave."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "sf.h")).read()
    return sorted(set(re.findall(r"SF_FN\((\w+)\)\(", txt)))


def test_header_matches_ctypes_table():
    from staticfusion_amd import capi

    assert header_symbols() == sorted(capi.SIGNATURES.keys())


def test_hip_library_builds_and_exports_every_symbol():
    import staticfusion_amd as sf

    if not os.path.exists(sf.LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "staticfusion_amd", "csrc")])
    lib = ctypes.CDLL(sf.LIB)
    for name in header_symbols():
        assert hasattr(lib, "sf_" + name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", sf.LIB]).decode()
    exported = set(re.findall(r" T (sf_\w+)", out))
    assert exported == {"sf_" + n for n in header_symbols()}  # nothing undeclared leaks out either
    # the product must not depend on the oracle in any way
    needed = subprocess.check_output(["readelf", "-d", sf.LIB]).decode()
    assert "liboracle" not in needed
    assert lib.sf_backend is not None
    lib.sf_backend.restype = ctypes.c_char_p
    assert lib.sf_backend() == b"hip:gfx950"


def test_product_sources_never_touch_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "staticfusion_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def test_hip_library_fails_loudly_without_gpu():
    """No CPU fallback: without a device sf_create must fail with SF_ERR_DEVICE (skipped on the GPU box)."""
    import staticfusion_amd as sf

    api = sf.load()
    p = api.default_params_struct()
    h = ctypes.c_void_p()
    rc = api.create(ctypes.byref(p), 240, 320, 1, 0, ctypes.byref(h))
    if rc == 0:
        api.destroy(h)
        pytest.skip("a GPU is present")
    assert rc == -2 and b"no CPU fallback" in api.last_error()


def test_create_ex_rejects_unknown_variants(ora):
    """sf_create_ex: the build selector is validated before any device work (both libraries)."""
    import staticfusion_amd as sf

    for api in (sf.load(), ora):
        p = api.default_params_struct()
        h = ctypes.c_void_p()
        assert api.create_ex(ctypes.byref(p), 240, 320, 1, 0, 7, ctypes.byref(h)) == -1
        assert api.create_ex(ctypes.byref(p), 240, 320, 1, 0, -1, ctypes.byref(h)) == -1
    s = sf.Solver(ora, 60, 80, 1, ora.default_params_struct(), variant="throughput")  # accepted and ignored by the oracle
    assert s.variant() == ("auto", 1, 1)


def test_default_params_are_the_driver_values(ora):
    """StaticFusion-datasets.cpp:79-94 and FrontEnd.cpp:57-76; both libraries must agree byte for byte."""
    import staticfusion_amd as sf

    hip = sf.load()
    for getter in ("default_params_struct", "ctor_params_struct"):
        a, b = getattr(ora, getter)(), getattr(hip, getter)()
        assert bytes(a) == bytes(b), getter
    d = ora.default_params_struct()
    assert (d.max_iter_per_level, d.max_iter_irls, d.use_motion_filter) == (3, 6, 1)
    assert d.irls_delta_threshold == pytest.approx(0.0015) and d.lambda_reg == pytest.approx(0.35)
    assert d.previous_speed_const_weight == pytest.approx(0.1) and d.previous_speed_eig_weight == pytest.approx(2.0)
    c = ora.ctor_params_struct()
    assert (c.max_iter_per_level, c.max_iter_irls, c.use_motion_filter) == (2, 10, 0)
    assert c.kb == pytest.approx(1.25) and c.fovh == pytest.approx(np.pi * 62.5 / 180.0)


def test_error_behaviour(ora):
    import staticfusion_amd as sf

    p = ora.default_params_struct()
    with pytest.raises(sf.SfError):
        sf.Solver(ora, 4, 4, 1, p)  # too small
    p.ctf_levels = 9
    with pytest.raises(sf.SfError):
        sf.Solver(ora, 240, 320, 1, p)
    s = sf.Solver(ora, 60, 80, 2, ora.default_params_struct())
    with pytest.raises(sf.SfError):
        s.T(5)  # stream out of range
    with pytest.raises(sf.SfError):
        s.residuals_vs_history(3)  # index < 5
    assert s.levels == 3 and s.level_shape(2) == (15, 20)


def test_synthetic_generator_is_deterministic_and_consistent():
    from staticfusion_amd.synth import LCG64, make_pair, pose_delta, se3_exp

    g = LCG64(1234)
    assert [g.next_u64() for _ in range(2)] == [(6364136223846793005 * 1234 + 1442695040888963407) % 2**64,
                                                (6364136223846793005 * ((6364136223846793005 * 1234 + 1442695040888963407) % 2**64)
                                                 + 1442695040888963407) % 2**64]
    a, b = make_pair(seed=9, out_rows=60, out_cols=80), make_pair(seed=9, out_rows=60, out_cols=80)
    assert np.array_equal(a["new"][0], b["new"][0]) and np.array_equal(a["old"][1], b["old"][1])
    d = a["new"][0]
    assert d.dtype == np.float32 and d.shape == (60, 80) and 1.5 < d[d > 0].min() and d.max() < 3.6
    assert np.allclose(np.round(d.astype(np.float64) * 1000), d.astype(np.float64) * 1000, atol=1e-3)  # uint16 mm
    T = se3_exp([0.01, 0, 0, 0, 0.02, 0])
    r, t = pose_delta(np.eye(4), T)
    assert r == pytest.approx(0.02, rel=1e-6) and t == pytest.approx(np.linalg.norm(T[:3, 3]))


def test_bench_algorithmic_bytes():
    import bench
    from staticfusion_amd import SfFrameStats

    st = SfFrameStats()
    st.n_outer, st.pixel_iters, st.kmeans_iters = 2, 1000, 3
    st.outer[0].level, st.outer[0].k = 0, 0  # coarsest, first: no warp
    st.outer[1].level, st.outer[1].k = 1, 0
    levels_n = [400, 100]
    out = bench.algorithmic_bytes([st], levels_n, 100, True, True)
    assert out["irls"] == 60 * 1000
    assert out["linearise"] == 88 * (100 + 400) and out["warp"] == 32 * 400
    assert out["pyramid"] == 2 * 48 * 100 and out["kmeans"] == 20 * 100 * 3
    assert out["segm_image"] == 8 * 400 and out["residuals"] == 32 * 400
    # a launch of K sequence frames builds the old image's pyramid only where the buffers did not swap (sf_frame_kernels.hip)
    assert bench.sequence_pyramids_per_frame(20) == (1.1, 18) and bench.sequence_pyramids_per_frame(2) == (2.0, 0)
    assert bench.sequence_pyramids_per_frame(3) == (4.0 / 3.0, 2) and bench.sequence_pyramids_per_frame(1) == (2.0, 0)
    one = bench.algorithmic_bytes([st], levels_n, 100, True, True, pyramids_per_frame=1.1)
    assert one["pyramid"] == pytest.approx(1.1 * 48 * 100) and one["irls"] == out["irls"]
    moved = bench.algorithmic_bytes([st], levels_n, 100, True, True, per_unit=bench.MOVED_B)
    assert moved["irls"] == 58 * 1000 and moved["residuals"] == 72 * 400 and moved["linearise"] == 50 * 500


def test_frames_in_one_call_on_the_oracle(ora):
    """sf_process_frames through the binding: the trajectory block is [frame][stream] 4 x 4 (row, column) and the call
    equals the frames one by one (on the CPU oracle the entry point IS that loop; the GPU test compares launches)."""
    import numpy as np
    from conftest import driver_params, make_solver
    from staticfusion_amd.synth import make_pair

    pr = [make_pair(seed=5 + q, sphere=True, out_rows=30, out_cols=40) for q in range(2)]
    solvers = [make_solver(ora, 30, 40, driver_params(ora), batch=2) for _ in range(2)]
    for s in solvers:
        for b in range(2):
            s.set_current(b, *pr[b]["new"])
            s.set_prediction(b, *pr[b]["old"])
    one, many = solvers
    T = many.process_frames(0, 3, trajectory=True)
    assert T.shape == (3, 2, 4, 4)
    for k in range(3):
        one.process_frame(k)
        for b in range(2):
            assert np.array_equal(T[k, b], one.T(b))
    assert np.array_equal(many.b(1), one.b(1))


def test_bench_head_stamp_fallback(tmp_path, monkeypatch):
    """bench.git_head(): the checkout's HEAD, or -- on a GPU box, which gets a snapshot without .git -- the stamp build() left
    beside the libraries (staticfusion_amd/csrc/BUILD_HEAD)."""
    import subprocess

    import bench

    head = bench.git_head()
    assert head is None or len(head) >= 7
    stamp = os.path.join(ROOT, "staticfusion_amd", "csrc", "BUILD_HEAD")
    had = open(stamp).read() if os.path.exists(stamp) else None
    try:
        open(stamp, "w").write("abcdef012345+dirty\n")

        def no_git(*a, **k):
            raise OSError("no git here")

        monkeypatch.setattr(subprocess, "check_output", no_git)
        assert bench.git_head() == "abcdef012345+dirty"
    finally:
        if had is None:
            os.remove(stamp)
        else:
            open(stamp, "w").write(had)


def test_traffic_evidence_belongs_to_the_sources():
    """bench.py prints `roofline.traffic` only when profiles/traffic_<workload>_b<batch>.json was measured on the running device
    sources (their hash is in the file). After a change of those sources the files are stale until tools/refresh_evidence.sh has
    run on a GPU box: that is reported here as a skip with the reason -- visible in every CPU run -- instead of going unnoticed until
    the bench line carries null."""
    import glob
    import json

    import bench

    now = bench.source_sha()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*_b*.json")))
    assert files, "no PMC traffic evidence committed"
    stale = [os.path.basename(f) for f in files if json.load(open(f)).get("src_sha") != now]
    if stale:
        pytest.skip("PMC traffic evidence is stale for the sources %s (%s): run tools/refresh_evidence.sh on a GPU box" % (now, ", ".join(stale)))


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus 8` started directly must never print a line that claims 8 GPUs from one process: without 8
    devices it exits non-zero before anything runs (this container has none); with WORLD_SIZE set, a rank count that
    contradicts --gpus is refused (needs a GPU to get that far: tests/test_gpu_parity_sweep.py)."""
    import sys

    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("this node really has 8 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SF_BENCH_SINGLE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 2 and b"refusing" in r.stderr and not r.stdout.strip()


def test_abi_version_and_struct_sizes(ora):
    """sf_abi_version: the library's header version and the sizes it assumes for the structs and the stage-profile array callers
    hand over (three silent ABI changes -- 24 -> 32 profile slots, a trace field, an argument). Oracle and product
    implement it; the ctypes mirrors of this package must agree with both."""
    import staticfusion_amd as sf
    from staticfusion_amd import capi

    hdr = open(os.path.join(ROOT, "include", "sf.h")).read()
    version = int(re.search(r"#define SF_ABI_VERSION (\d+)", hdr).group(1))
    assert version == capi.ABI_VERSION, "staticfusion_amd/_capi.py mirrors another version of include/sf.h"
    libs = [ora.lib.sfo_abi_version]
    if os.path.exists(sf.LIB):
        libs.append(ctypes.CDLL(sf.LIB).sf_abi_version)
    for fn in libs:
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        fn.restype = ctypes.c_int
        assert fn(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == version
        assert a.value == ctypes.sizeof(capi.SfParams) and b.value == ctypes.sizeof(capi.SfFrameStats) and c.value == 32
        assert fn(None, None, None) == version


def test_binding_refuses_a_library_of_another_abi(ora, tmp_path, monkeypatch):
    """the check runs when a library is LOADED (Api.__init__), not only in a unit test."""
    import staticfusion_amd as sf
    from oracle import binding
    from staticfusion_amd import capi

    monkeypatch.setattr(capi, "ABI_VERSION", capi.ABI_VERSION + 1)
    with pytest.raises(sf.SfError, match="ABI mismatch"):
        capi.Api(binding.LIB, "sfo_")


def test_render_with_stride_two_is_the_decimated_full_render():
    """The sequence generator casts only the rays of the pixels the loaders' [::2, ::2] decimation keeps (synth.py: stride = 2):
    the same bits as rendering every pixel and dropping three quarters."""
    from staticfusion_amd.synth import Scene, quantise_and_decimate, sequence_trajectory

    for seed, k in ((1000, 0), (1001, 57), (1003, 199)):
        poses, offs = sequence_trajectory(seed, 200)
        sc = Scene(seed=seed, sphere=True, sphere_seed=seed + 4444)
        full = quantise_and_decimate(*sc.render(poses[k], 640, 480, sphere_offset=offs[k]))
        fast = quantise_and_decimate(*sc.render(poses[k], 640, 480, sphere_offset=offs[k], stride=2), decimate=False)
        assert np.array_equal(full[0], fast[0]) and np.array_equal(full[1], fast[1]), (seed, k)


def test_no_vector_instruction_in_front_of_an_exec_restore_at_a_barrier():
    """tools/diag/exec_lint.py over every built library: the miscompilation behind round 4's "address 0" fault (a register copy that
    the compiler placed in front of the `s_or_b64 exec` of a loop-exit block that holds a barrier: executed with EXEC = 0, it restores
    nothing) shows in the disassembly, and no library that ships may contain it. profiles/HISTORY.md, round 5."""
    import glob
    import importlib.util

    spec = importlib.util.spec_from_file_location("exec_lint", os.path.join(ROOT, "tools", "diag", "exec_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    if not os.path.exists(lint.OBJDUMP):
        pytest.skip("no llvm-objdump in this image")
    libs = sorted(glob.glob(os.path.join(ROOT, "staticfusion_amd", "csrc", "libsf_hip*.so")))
    assert any(os.path.basename(p) == "libsf_hip.so" for p in libs)
    for path in libs:
        assert len(lint.code_objects(path)) >= 5, path  # (the host objects' kernels + the frame-kernel builds)
        assert lint.lint(path) == [], path
    # the lint recognises the pattern (the block as the faulting build had it)
    bad = "\ts_cbranch_execz .LBB1_7\n.LBB1_7:\n\tv_mov_b64_e32 v[62:63], v[92:93]\n\ts_barrier\n\ts_or_b64 exec, exec, s[10:11]\n"
    good = "\ts_cbranch_execz .LBB1_7\n.LBB1_7:\n\ts_or_b64 exec, exec, s[10:11]\n\tv_mov_b64_e32 v[62:63], v[92:93]\n\ts_barrier\n"
    assert len(lint.lint_text(bad)) == 1 and lint.lint_text(good) == []


def test_the_linearisation_sweep_keeps_its_loads_in_flight():
    """the register-strip linearisation issues the loads of column u + 4 while it evaluates column u. That only holds
    while the compiler's wait counter stays exact in the sweep -- no spill, no memory operation that may or may not be issued, no
    consumer hoisted into the loop's latch: each of them turned every wait into `vmcnt(0)` (and the stage from - 37 % to + 10 %)
    when the sweep was written. Checked where it shows: the disassembly of the product library (tools/diag/loop_waits.py)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("loop_waits", os.path.join(ROOT, "tools", "diag", "loop_waits.py"))
    lw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lw)
    if not os.path.exists(lw.exec_lint.OBJDUMP):
        pytest.skip("no llvm-objdump in this image")
    lib = os.path.join(ROOT, "staticfusion_amd", "csrc", "libsf_hip.so")
    for kern in ("256", "256o5", "1024"):
        for seg in (0, 1):  # later linearisations of a frame (not the first, no debug planes), without and with segmentation
            loops = lw.innermost_loops(lib, kern, "solve_linearise_stripsILb0ELb0ELb%d" % seg)
            sweep = max(loops, key=lambda r: r["valu"])
            # three columns per trip; the only LDS traffic is the flush of the segmentation prior's running sums (atomics)
            assert sweep["loads"] >= 12 and sweep["dpp"] >= 15 and sweep["lds"] <= (16 if seg else 0), (kern, seg, sweep)
            assert sweep["scratch"] == 0 and sweep["flat"] == 0, (kern, seg, sweep)
            assert sweep["waits"] and min(sweep["waits"]) >= 4, (kern, seg, sweep)  # (at least one column's loads stay in flight at every wait)

