"""Long sequences against the oracle, frame by frame -- shared by tests/test_long_sequences.py, the fixture generator
tools/golden/make_golden_long_sequences.py, the hunt tools/diag/long_sequence_hunt.py and bench.py's sequences blocks.

The reference's image-sequence driver (StaticFusion-imagesequenceassoc.cpp:140-191) runs hundreds of frames with state carried
from frame to frame: the previous frame's twist (`twist_odometry_old`, FrontEnd.cpp:1134-1144), the b field, the K-means
centres, the five-frame ring of images and the chain of their poses (FrontEnd.cpp:896-915). A frame whose stopping test is
decided within rounding of its threshold (a TIE) therefore does not end with that frame: what it leaves behind differs, and
the question this module answers is what that does to the frames after it.

`Runner` drives one backend (the HIP library in any build, or the oracle: same ABI) through that loop for D sequences held in a
frame pool, `frame_records` reads back what the comparisons need, `compare` / `after_event_curves` turn two runs into per-frame
distances, the events (count mismatches, frames past the pose bar) and what follows each of them.
"""
import ctypes
import zlib

import numpy as np

from sequence_cases import TIE_REL_MARGIN, classify_flip  # noqa: F401
from staticfusion_amd.synth import pose_delta  # noqa: F401 (re-exported)

SEEDS = (1000, 1001, 1002, 1003)  # the sequences bench.py's `sequences` blocks play on rank 0
# ... and one with an EVENT (profiles/PARITY.md, "long sequences"): frame 190 of sequence 2059 is a tie of the oracle's stopping test
# (delta_sol_max within 1.3 % of irls_delta_threshold) that the throughput and the latency build decide the other way
EVENT_SEEDS = (2059,)
FRAMES = 200
ROWS, COLS = 240, 320
POSE_BAR = 1e-4  # rad / m per frame: BASELINE.json's north star


class DevicePool:
    """A host array copied into plain hipMalloc memory (the tests use no torch)."""

    _rt = None

    def __init__(self, host):
        if DevicePool._rt is None:
            DevicePool._rt = ctypes.CDLL("libamdhip64.so")
        rt = DevicePool._rt
        self.ptr = ctypes.c_void_p()
        host = np.ascontiguousarray(host)
        assert rt.hipMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(host.nbytes)) == 0
        assert rt.hipMemcpy(self.ptr, host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(host.nbytes), 1) == 0

    def data_ptr(self):
        return self.ptr.value

    def free(self):
        if self.ptr:
            DevicePool._rt.hipFree(self.ptr)
            self.ptr = ctypes.c_void_p()


class HostPool:
    def __init__(self, host):
        self.a = np.ascontiguousarray(host)

    def data_ptr(self):
        return self.a.ctypes.data

    def free(self):
        pass


def sequence_params(api):
    """The drivers' parameters (bench.py make_params for the sequences workload)."""
    p = api.default_params_struct()
    p.kb = 1.05
    return p


class Runner:
    """D sequences of F frames each in one frame pool ([D * F][n0] column-major images); stream q plays sequence q from its
    frame 0: frame 0 is the first prediction, frame k >= 1 is solved against frame k - 1 (the bench's step; reference
    StaticFusion-imagesequenceassoc.cpp:140-191 without the map)."""

    def __init__(self, api, pool_d, pool_i, D, F, rows=ROWS, cols=COLS, variant=None, params=None, streams=None):
        import staticfusion_amd as sf

        self.api, self.D, self.F = api, D, F
        self.streams = list(range(D)) if streams is None else list(streams)  # which sequence each stream plays
        self.B = len(self.streams)
        self.pool_d, self.pool_i = pool_d, pool_i
        self.s = sf.Solver(api, rows, cols, self.B, params if params is not None else sequence_params(api), variant=variant)
        self.k = 0
        self.s.advance_sequences_device(pool_d.data_ptr(), pool_i.data_ptr(), self.index(0), D * F)
        self.s.push_history(0)

    def index(self, k):
        return (np.asarray(self.streams) * self.F + k).astype(np.int32)

    def step(self):
        """The next frame of every stream: advance + sf_process_frame."""
        self.k += 1
        self.s.advance_sequences_device(self.pool_d.data_ptr(), self.pool_i.data_ptr(), self.index(self.k), self.D * self.F)
        self.s.process_frame(self.k)
        return self.k

    def run_in_one_launch(self, n_frames):
        """The next n_frames of every stream in ONE launch (sf_process_sequence_frames_device); -> T [n_frames][B][4][4]."""
        idx = np.stack([self.index(self.k + 1 + j) for j in range(n_frames)])
        T = self.s.process_sequence_frames_device(self.pool_d.data_ptr(), self.pool_i.data_ptr(), idx, self.D * self.F, self.k + 1, trajectory=True)
        self.k += n_frames
        return T

    def frame_records(self, images=True):
        """One record per stream for the frame just solved. images=False keeps only checksums of the label and decision images."""
        T_all, n_irls, n_outer, _ = self.s.batch_results()
        out = []
        for b in range(self.B):
            st = self.s.stats(b)
            lab, bi = self.s.labels(0, b), self.s.b_image(b)
            rec = {
                "T": T_all[b].copy(), "b": self.s.b(b).copy(), "counts": (int(st.n_outer), int(st.n_irls)), "status": int(st.status),
                "outer": [(int(o.level), int(o.k), int(o.irls_iters), float(o.delta_sol_max), int(o.n_valid),
                           float(np.sqrt(np.sum(np.array(o.twist_level[:], dtype=np.float64) ** 2))))
                          for o in (st.outer[i] for i in range(st.n_outer))],
                "label_crc": zlib.crc32(np.ascontiguousarray(lab, dtype=np.int32).tobytes()),
                "decisions": np.packbits(bi > 0.5),
            }
            if images:
                rec["labels"], rec["b_img"] = lab.copy(), bi.copy()
            out.append(rec)
        return out

    def close(self):
        self.s.close()


def compare(ref, got, thr):
    """One record for a frame of one stream: pose distance, discrete mismatches, b distances, and -- when the IRLS counts
    differ -- what kind of tie it is (sequence_cases.classify_flip)."""
    rot, trans = pose_delta(ref["T"], got["T"])
    dec = np.unpackbits(ref["decisions"]) != np.unpackbits(got["decisions"])
    rec = {
        "rot": rot, "trans": trans, "label_equal": ref["label_crc"] == got["label_crc"], "decision_px": int(dec.sum()),
        "b24": float(np.abs(ref["b"] - got["b"]).max()), "counts": list(got["counts"]), "counts_ref": list(ref["counts"]),
        "identical": bool(np.array_equal(ref["T"], got["T"]) and np.array_equal(ref["b"], got["b"]) and ref["counts"] == got["counts"]
                          and ref["label_crc"] == got["label_crc"] and np.array_equal(ref["decisions"], got["decisions"])),
    }
    if "b_img" in ref and "b_img" in got:
        rec["b_img"] = float(np.abs(ref["b_img"] - got["b_img"]).max())
        rec["identical"] = rec["identical"] and bool(np.array_equal(ref["b_img"], got["b_img"]) and np.array_equal(ref["labels"], got["labels"]))
    # the IRLS counts PER OUTER ITERATION, not their totals: two ties in one frame can cancel in the sum (seed 3098, frame 95:
    # 3 + 2 + 3 + 2 + 1 against 3 + 2 + 2 + 2 + 2 iterations, 4.7e-3 m apart -- filed under "no count mismatch" until round 5 looked)
    if ref["counts"] != got["counts"] or [o[:3] for o in ref["outer"]] != [o[:3] for o in got["outer"]]:
        rec["flip"] = classify_flip(ref["outer"], got["outer"], thr)
        rec["per_level_counts"] = [[o[2] for o in ref["outer"]], [o[2] for o in got["outer"]]]
    return rec


def chain(Ts):
    """The accumulated pose of a list of per-frame T (T_k = pose of camera k in frame k - 1): T_1 T_2 ... in float64."""
    A = np.eye(4)
    for T in Ts:
        A = A @ np.asarray(T, dtype=np.float64)
    return A


def events_and_curves(recs, bar=POSE_BAR):
    """recs: the compare() records of ONE stream, frame 1 first. An EVENT is a frame whose iteration counts differ from the
    oracle's (a tie, classified) or whose pose leaves the bar without a count mismatch at or before it in the window. Returns
    (events, curve): events = [{frame, kind, dist}], curve = for the FIRST event the per-frame distance max(rot, trans) of the
    frames after it (index 0 = the event's own frame) -- what the carried state does with the difference."""
    dist = [max(r["rot"], r["trans"]) for r in recs]
    events = []
    for k, r in enumerate(recs):
        if "flip" in r:
            events.append({"frame": k + 1, "kind": r["flip"]["kind"], "dist": dist[k], "rel_margin": r["flip"].get("rel_margin")})
        elif dist[k] > bar and not any(e["frame"] <= k + 1 and k + 1 - e["frame"] <= 1 for e in events):
            # a frame past the bar with identical counts, not directly at or after a tie: ill-conditioned on its own
            if not events or dist[k - 1] <= bar:
                events.append({"frame": k + 1, "kind": "no-count-mismatch", "dist": dist[k], "rel_margin": None})
    curve = dist[events[0]["frame"] - 1:] if events else []
    return events, curve


def frames_back_under(curve, bar=POSE_BAR):
    """N such that every frame from N frames after the event on is under the bar (0: the event's own frame already is)."""
    over = [j for j, v in enumerate(curve) if v > bar]
    return 0 if not over else over[-1] + 1


def episodes(dist, event_frames, bar=POSE_BAR, gap=2):
    """What an event does to the frames after it, per EPISODE: a maximal run of disturbed frames -- past the bar, or with an
    iteration-count mismatch -- that are at most `gap` frames apart (a tie is often followed by a frame or two that start from the
    state the tie left behind). -> [{first, last, frames_past_bar, length, peak, after}]: `length` = frames from the first
    disturbed frame to the last one past the bar (0: no frame left the bar), `after` = the 8 distances behind the episode."""
    disturbed = sorted(set(k for k, v in enumerate(dist) if v > bar) | set(f - 1 for f in event_frames))
    out, run = [], []
    for k in disturbed + [None]:
        if run and (k is None or k - run[-1] > gap + 1):
            over = [q for q in range(run[0], run[-1] + 1) if dist[q] > bar]
            out.append({"first": run[0] + 1, "last": run[-1] + 1, "frames_past_bar": len(over),
                        "length": (over[-1] - run[0] + 1) if over else 0, "peak": float(max(dist[run[0]:run[-1] + 1])),
                        "after": [float(v) for v in dist[run[-1] + 1:run[-1] + 9]]})
            run = []
        if k is not None:
            run.append(k)
    return out


def summarise_stream(recs, bar=POSE_BAR):
    dist = np.array([max(r["rot"], r["trans"]) for r in recs])
    events, curve = events_and_curves(recs, bar)
    return {
        "frames": len(recs),
        "frames_past_bar": int((dist > bar).sum()), "worst": float(dist.max()), "median": float(np.median(dist)),
        "count_mismatches": sum(1 for r in recs if "flip" in r),  # per outer iteration (compare()), not only in the totals
        "label_mismatch_frames": sum(1 for r in recs if not r["label_equal"]),
        "decision_mismatch_frames": sum(1 for r in recs if r["decision_px"]),
        "b24_over_1e-5": sum(1 for r in recs if r["b24"] > 1e-5), "b24_over_1e-4": sum(1 for r in recs if r["b24"] > 1e-4),
        "b24_worst": max(r["b24"] for r in recs),
        "bit_identical_frames": sum(1 for r in recs if r["identical"]),
        "events": events, "after_first_event": [float(v) for v in curve[:60]],
        "frames_until_back_under_bar": frames_back_under(curve, bar) if events else None,
        "episodes": episodes(list(dist), [e["frame"] for e in events], bar),
    }


def oracle_sequence(job):
    """(seed, frames, rows, cols, cache_dir[, gemm_mode]) -> compact frame records of the ORACLE on that synthetic sequence (one
    process per sequence: the hunt and the fixture generator map this over a multiprocessing pool). gemm_mode != 0 selects one of
    the oracle's other readings of the Eigen GEMM order (sfo_test_set_gemm_mode, tests/test_oracle_controls.py): the control
    that says what ANOTHER faithful build of the reference would see. Test infrastructure only."""
    seed, F, rows, cols, cache_dir = job[:5]
    gemm_mode = job[5] if len(job) > 5 else 0
    from oracle import binding
    from staticfusion_amd.synth import sequence_arrays

    d, i, T_gt = sequence_arrays(seed, F, rows, cols, cache_dir=cache_dir)
    ora = binding.load()
    r = Runner(ora, HostPool(d), HostPool(i), 1, F, rows, cols)
    if gemm_mode:
        fn = ora.lib.sfo_test_set_gemm_mode
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
        assert fn(r.s.h, gemm_mode) == 0
    recs = []
    for _ in range(F - 1):
        r.step()
        recs.append(r.frame_records(images=False)[0])
    r.close()
    return {"seed": seed, "recs": recs, "T_gt": T_gt, "gemm_mode": gemm_mode}
