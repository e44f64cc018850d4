"""Several frames per launch (sf_process_frames / sf_process_sequence_frames_device): the queue hands out (frame, stream)
pairs and a stream's next frame starts as soon as ITS previous frame is done, on whatever workgroup -- another CU, another
XCD -- is free. Everything a frame leaves behind for the next one (pyramids, the 5-frame ring, the carried twist, b, the
pose ring) crosses between workgroups through an agent-scope release / acquire pair per stream and frame.

The results must be those of the same frames launched one by one, bit for bit, for every stream and every frame -- under
UNEVEN load (streams that cost 1x and 4x mixed, more streams than resident workgroups: fast streams run several frames ahead
of slow ones and consecutive frames of a stream land on different CUs), checking every value the next frame reads.
"""
import numpy as np
import pytest

from conftest import driver_params, make_solver
from staticfusion_amd import capi
from staticfusion_amd.synth import DEFAULT_XI, make_sequence, pose_delta

pytestmark = pytest.mark.gpu


def _everything(s, streams):
    out = []
    for b in streams:
        out += [s.T(b), s.twist(b), s.twist_old(b), s.b(b), s.kmeans_centres(b), s.cluster_residuals(b), s.labels(0, b), s.b_image(b),
                s.plane(capi.SET_PRED, capi.CH_DEPTH, 1, b), s.plane(capi.SET_NEW, capi.CH_INTENSITY, 2, b)]
        st = s.stats(b)
        out.append(np.array([st.n_outer, st.n_irls, st.kmeans_iters, st.status, st.pixel_iters]))
    return out


@pytest.mark.parametrize("first,frames", [(0, 8), (3, 5)])
def test_frames_of_unequal_streams_in_one_launch(hip, pair, first, frames):
    B = 1500 if hip.default_variant != "cluster" else 8
    easy = pair(seed=61, rows=60, cols=80, sphere=True)
    hard = pair(seed=62, rows=60, cols=80, sphere=True, xi=tuple(3.0 * np.array(DEFAULT_XI)))
    mid = pair(seed=63, rows=60, cols=80, sphere=True, xi=tuple(1.7 * np.array(DEFAULT_XI)))
    which = lambda b: (easy, hard, easy, mid, easy)[(b * 7 + b // 11) % 5]
    solvers = [make_solver(hip, 60, 80, driver_params(hip), batch=B) for _ in range(2)]
    for s in solvers:
        for b in range(B):
            s.set_current(b, *which(b)["new"])
            s.set_prediction(b, *which(b)["old"])
        for im in range(first):
            s.process_frame(im)
    one, many = solvers
    T_ref = []
    for k in range(frames):
        one.process_frame(first + k)
        T_ref.append(one.batch_results()[0].copy())
    T = many.process_frames(first, frames, trajectory=True)
    assert T.shape == (frames, B, 4, 4)
    for k in range(frames):
        assert np.array_equal(T[k], T_ref[k]), k
    probe = list(range(0, B, max(1, B // 37)))
    for x, y in zip(_everything(one, probe), _everything(many, probe)):
        assert np.array_equal(x, y, equal_nan=True)
    assert one.counters() == many.counters()
    # ... and the state the NEXT frame starts from (5-frame ring, carried twist): one more frame each, launched alone
    one.process_frame(first + frames)
    many.process_frame(first + frames)
    assert np.array_equal(one.batch_results()[0], many.batch_results()[0])
    assert all(np.array_equal(one.cluster_residuals(b), many.cluster_residuals(b), equal_nan=True) for b in probe)
    # twice the same launch: the schedule differs, the results do not
    again = make_solver(hip, 60, 80, driver_params(hip), batch=B)
    for b in range(B):
        again.set_current(b, *which(b)["new"])
        again.set_prediction(b, *which(b)["old"])
    for im in range(first):
        again.process_frame(im)
    assert np.array_equal(again.process_frames(first, frames, trajectory=True), T)


def test_sequences_from_the_pool_in_one_launch(hip, ora):
    """sf_process_sequence_frames_device = sf_advance_sequences_device + sf_process_frame per frame, one launch; staggered
    streams over two sequences; per-frame poses against the one-by-one loop (bit for bit) and against the oracle (<= 1e-4)."""
    import ctypes

    hiprt = ctypes.CDLL("libamdhip64.so")

    class Dev:  # a device copy of a host array (plain hipMalloc: the tests use no torch)
        def __init__(self, h):
            self.ptr = ctypes.c_void_p()
            assert hiprt.hipMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(h.nbytes)) == 0
            assert hiprt.hipMemcpy(self.ptr, h.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(h.nbytes), 1) == 0

        def data_ptr(self):
            return self.ptr.value

    D, F, K = 2, 12, 9
    B = 96 if hip.default_variant != "cluster" else 4
    seqs = [make_sequence(2000 + q, F, sphere=True, out_rows=60, out_cols=80) for q in range(D)]
    col = lambda a: np.ascontiguousarray(np.asarray(a, np.float32).T).ravel()
    pd_h = np.stack([col(f[0]) for sq in seqs for f in sq["frames"]])
    pi_h = np.stack([col(f[1]) for sq in seqs for f in sq["frames"]])
    pd, pi = Dev(pd_h), Dev(pi_h)
    phase = (np.arange(B) // D * 5) % (F - 1)
    index = np.stack([(np.arange(B) % D) * F + (phase + k) % F for k in range(K + 1)]).astype(np.int32)
    # some streams skip an advance in the middle (frame_index < 0: the images stay, the frame is solved again): inside the
    # launch that breaks the alternation of the pyramid-buffer swaps, up to a stream that ends the launch swapped
    index[4, 3::7] = -1
    index[K, 5::9] = -1
    index[K - 1, 1] = index[K, 1] = -1
    solvers = [make_solver(hip, 60, 80, driver_params(hip), batch=B) for _ in range(2)]
    for s in solvers:
        s.advance_sequences_device(pd.data_ptr(), pi.data_ptr(), index[0], D * F)
        s.push_history(0)
    one, many = solvers
    T_ref = []
    for k in range(1, K + 1):
        one.advance_sequences_device(pd.data_ptr(), pi.data_ptr(), index[k], D * F)
        one.process_frame(k)
        T_ref.append(one.batch_results()[0].copy())
    T = many.process_sequence_frames_device(pd.data_ptr(), pi.data_ptr(), index[1:], D * F, 1, trajectory=True)
    for k in range(K):
        assert np.array_equal(T[k], T_ref[k]), k
    for b in (0, 1, B // 2, B - 1):
        for pset in (capi.SET_NEW, capi.SET_PRED):
            assert np.array_equal(one.plane(pset, capi.CH_DEPTH, 0, b), many.plane(pset, capi.CH_DEPTH, 0, b))
        assert np.array_equal(one.b_image(b), many.b_image(b)) and np.array_equal(one.labels(0, b), many.labels(0, b))
    # sf_clear_sync_timeout on a build without rendezvous puts back the image layout the host assumes (a launch that gave up on a
    # stream may leave its pyramid buffers swapped and level 0 in the caller's pool); after a launch that completed it changes
    # nothing: the next frame of both handles is the same frame
    many.clear_sync_timeout()
    for s in (one, many):
        s.advance_sequences_device(pd.data_ptr(), pi.data_ptr(), index[0], D * F)
        s.process_frame(K + 1)
    assert np.array_equal(one.batch_results()[0], many.batch_results()[0])
    # the oracle through the same entry point (host pools), first streams of each sequence
    so = make_solver(ora, 60, 80, driver_params(ora), batch=D)
    idx_o = index[:, :D]
    so.advance_sequences_device(pd_h.ctypes.data, pi_h.ctypes.data, idx_o[0], D * F)
    so.push_history(0)
    To = so.process_sequence_frames_device(pd_h.ctypes.data, pi_h.ctypes.data, idx_o[1:], D * F, 1, trajectory=True)
    for k in range(K):
        for b in range(D):
            rot, trans = pose_delta(To[k, b], T[k, b])
            assert rot <= 1e-4 and trans <= 1e-4, (k, b, rot, trans)


def test_argument_checks(hip, pair):
    import staticfusion_amd as sf

    s = make_solver(hip, 60, 80, driver_params(hip), pair(seed=3, rows=60, cols=80, sphere=True), batch=2)
    with pytest.raises(sf.SfError):
        s.process_frames(0, 0)
    with pytest.raises(sf.SfError):
        s.process_frames(-1, 2)
    T = s.process_frames(0, 1, trajectory=True)  # one frame: the plain call
    assert np.array_equal(T[0, 1], s.T(1))


def test_shader_clock_of_the_stream_frames(hip, pair):
    """Slot 26 of the stage profile: shader cycles over the intervals of slot 13 (the bench line's `shader_clock_mhz`). A
    plausible clock, and counters that only move when frames run."""
    s = make_solver(hip, 60, 80, driver_params(hip), pair(seed=3, rows=60, cols=80, sphere=True), batch=64)
    c0 = s.shader_clock_counters()
    assert s.shader_clock_counters() == c0
    s.process_frames(0, 4)
    c1 = s.shader_clock_counters()
    assert c1[0] > c0[0] and c1[1] > c0[1]
    assert 300.0 < s.shader_clock_mhz(c0, c1) < 3500.0, (c0, c1)


def test_pool_arguments_are_checked(hip, pair):
    """a frame number outside the pool or a misaligned pool must be refused on the host, nothing launched."""
    import ctypes

    import staticfusion_amd as sf

    hiprt = ctypes.CDLL("libamdhip64.so")
    s = make_solver(hip, 60, 80, driver_params(hip), pair(seed=3, rows=60, cols=80, sphere=True), batch=2)
    n0 = 60 * 80
    ptr = ctypes.c_void_p()
    assert hiprt.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(4 * n0 * 3 + 64)) == 0
    base = ptr.value
    ok = np.array([0, 2], np.int32)
    s.advance_sequences_device(base, base, ok, 3)
    before = s.plane(capi.SET_NEW, capi.CH_DEPTH, 0, 1).copy()
    with pytest.raises(sf.SfError):
        s.advance_sequences_device(base, base, np.array([0, 3], np.int32), 3)   # frame 3 of a pool of 3
    with pytest.raises(sf.SfError):
        s.advance_sequences_device(base + 4, base, ok, 3)                        # not 16-byte aligned
    with pytest.raises(sf.SfError):
        s.process_sequence_frames_device(base, base, np.array([[0, 1], [1, 7]], np.int32), 3, 1)
    with pytest.raises(sf.SfError):
        s.advance_sequences_device(base, base, ok, 0)
    assert np.array_equal(s.plane(capi.SET_NEW, capi.CH_DEPTH, 0, 1), before)  # nothing was launched by the refused calls
    s.advance_sequences_device(base, base, np.array([-1, 1], np.int32), 3)     # a negative entry leaves that stream alone
    s.synchronize()
    hiprt.hipFree(ptr)


def test_a_launch_that_gave_up_on_streams_leaves_nothing_dangling(hip, monkeypatch):
    """a sequence launch that gives up on a stream (SF_STATUS_SYNC_TIMEOUT: the wait for the stream's previous
    frame ran into its bound -- cannot happen, so a test hook gives up frame 3 of every third stream) skips materialise_level0
    for it: the stream's state keeps naming frames in the caller's pool, which frames 1 and 2 read in place. The pool is released
    after the launch, then a frame is launched on the handle BEFORE sf_clear_sync_timeout: the first frame of that launch puts
    the stream back into the host's layout (nothing reads the pool any more), and after the clear + fresh images the handle
    computes what a fresh handle computes. The streams that were not given up are those of the undisturbed launch."""
    import ctypes

    if hip.default_variant == "cluster":
        pytest.skip("the cluster build runs such calls frame by frame: no frame waits for another inside a launch")
    hiprt = ctypes.CDLL("libamdhip64.so")
    D, F, K, B = 2, 8, 6, 64
    seqs = [make_sequence(2100 + q, F, sphere=True, out_rows=60, out_cols=80) for q in range(D)]
    col = lambda a: np.ascontiguousarray(np.asarray(a, np.float32).T).ravel()
    pd_h = np.stack([col(f[0]) for sq in seqs for f in sq["frames"]])
    pi_h = np.stack([col(f[1]) for sq in seqs for f in sq["frames"]])
    ptrs = []
    for h in (pd_h, pi_h):
        p = ctypes.c_void_p()
        assert hiprt.hipMalloc(ctypes.byref(p), ctypes.c_size_t(h.nbytes)) == 0
        assert hiprt.hipMemcpy(p, h.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(h.nbytes), 1) == 0
        ptrs.append(p)
    index = np.stack([(np.arange(B) % D) * F + k for k in range(K + 1)]).astype(np.int32)
    s, ref = (make_solver(hip, 60, 80, driver_params(hip), batch=B) for _ in range(2))
    for x in (s, ref):
        x.advance_sequences_device(ptrs[0].value, ptrs[1].value, index[0], D * F)
        x.push_history(0)
    T_ref = ref.process_sequence_frames_device(ptrs[0].value, ptrs[1].value, index[1:], D * F, 1, trajectory=True)
    monkeypatch.setenv("SF_DEBUG_GIVE_UP_AT_FRAME", "3")
    T = s.process_sequence_frames_device(ptrs[0].value, ptrs[1].value, index[1:], D * F, 1, trajectory=True)
    monkeypatch.delenv("SF_DEBUG_GIVE_UP_AT_FRAME")
    gave_up = list(range(0, B, 3))
    for b in range(B):
        if b in gave_up:  # frames 0 .. 2 as usual, the rest skipped: NaN rows, and the stream says so
            assert np.array_equal(T[:3, b], T_ref[:3, b]) and np.isnan(T[3:, b]).all(), b
            assert s.stats(b).status & capi.STATUS_SYNC_TIMEOUT, b
        else:
            assert np.array_equal(T[:, b], T_ref[:, b]), b
            assert s.stats(b).status & capi.STATUS_SYNC_TIMEOUT == 0, b
    # the pools go away (poisoned first: whoever still reads them reads NaN, should the allocator keep the pages mapped)
    s.synchronize()
    ref.synchronize()
    for p, h in zip(ptrs, (pd_h, pi_h)):
        assert hiprt.hipMemset(p, 0xFF, ctypes.c_size_t(h.nbytes)) == 0
        assert hiprt.hipFree(p) == 0
    s.process_frame(K + 1)  # before the clear: must not touch the pool
    s.synchronize()
    s.clear_sync_timeout()
    fresh = make_solver(hip, 60, 80, driver_params(hip), batch=B)
    carried = [(s.twist_old(b), s.cluster_residuals(b)) for b in range(B)]
    for x in (s, fresh):
        for b in range(B):
            f0, f1 = seqs[b % D]["frames"][0], seqs[b % D]["frames"][1]
            x.set_prediction(b, *f0)
            x.set_current(b, *f1)
            if x is fresh:  # what runSolver / buildSegmImage carry over from the frames before (FrontEnd.cpp:1134-1144)
                x.set_twist_old(b, carried[b][0])
                x.set_segm_state(b, cluster_res=carried[b][1])
        x.build_pyramid(True)
        x.run_solver(True)
        x.build_segm_image()
    for b in gave_up[:5] + [1, B - 1]:
        assert np.array_equal(s.T(b), fresh.T(b)) and np.array_equal(s.b(b), fresh.b(b)), b
        assert np.array_equal(s.b_image(b), fresh.b_image(b)) and np.array_equal(s.labels(0, b), fresh.labels(0, b)), b
        for pset in (capi.SET_NEW, capi.SET_PRED):
            assert np.array_equal(s.plane(pset, capi.CH_DEPTH, 1, b), fresh.plane(pset, capi.CH_DEPTH, 1, b)), b
        assert s.stats(b).status & capi.STATUS_SYNC_TIMEOUT == 0
