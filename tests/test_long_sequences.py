"""Sequences of the length BASELINE's configs have, against the oracle, frame by frame ("next round" item 1).

configs[0] / [3] / [4] are TUM sequences of hundreds of frames driven by StaticFusion-imagesequenceassoc.cpp:140-191; their
stand-in here is the `sequences` workload of bench.py: four 200-frame synthetic QVGA walks (seeds 1000-1003, a smooth random-walk
camera, a swinging sphere), frame-to-frame prediction, full solver with the drivers' parameters. Everything the reference carries
from frame to frame is in play: `twist_odometry_old` (FrontEnd.cpp:1134-1144), b, the K-means centres, the five-frame ring and the
pose chain of its images (FrontEnd.cpp:896-915).

* CPU: the oracle on a prefix of sequence 1000 reproduces tests/golden/long_sequences_qvga.npz bit for bit (the fixture is the
  oracle's own output, tools/golden/make_golden_long_sequences.py: it pins the restatement + generator + libm, not the reference).
* GPU, every build of the frame kernel: all 5 x 199 frames (the four bench sequences + one with a known tie, seed 2059) through the C ABI (sf_advance_sequences_device + sf_process_frame)
  in lock step with the oracle through the same entry points on host pools -- labels and (b > 0.5) decisions identical in every
  frame, pose <= 1e-4 rad / m per frame outside EVENTS (a frame whose iteration counts differ from the oracle's: a tie of a
  stopping test, classified), after an event back under the bar within AFTER_EVENT_FRAMES frames and never above AFTER_EVENT_BOUND
  in between; the accumulated trajectory T_1 ... T_199 of product and oracle against each other and both against the generator's
  ground truth; the same 199 frames in ONE launch (sf_process_sequence_frames_device) give the loop's poses bit for bit.
* GPU, libsf_hip_reforder.so: np.array_equal with the oracle on every frame (pose, b, counts, labels, b image).
"""
import json
import os

import numpy as np
import pytest

import long_sequences as ls
from conftest import GOLDEN, ROOT

FIXTURE = os.path.join(GOLDEN, "long_sequences_qvga.npz")

# What follows an event, as MEASURED (profiles/PARITY.md, "long sequences"; profiles/r05a_long_*): the frame after a tie starts from
# a slightly different twist_odometry_old / b / K-means state and the solver contracts the difference by 20 - 100 x per frame.
# 2 x 25 472 frames of the throughput / latency build against the oracle: 17 episodes, 15 of them ONE frame past the bar, the longest
# 2 frames (tie at t: 7.4e-5, t + 1: 7.6e-4, t + 2: 3.5e-5), peak 8.1e-4, first frame behind an episode <= 7.9e-5. The oracle against
# its own gemm2 reading on 50 944 frames: 28 episodes, the longest 4 frames, peak 2.0e-3 -- the bounds below sit between the two.
# (A later sample of 512 sequences = 101 888 frames has three frames beyond them, 2.1e-3 ... 4.7e-3: bifurcations of the ORACLE's own
# trajectory, which the gemm2 control reproduces to four digits; none of them lies in the five sequences of this test.)
AFTER_EVENT_FRAMES = 3     # frames t + 1 ... t + 3 after an event at t may still be past the bar ...
AFTER_EVENT_BOUND = 1.5e-3  # ... but not past this (an event's own frame included); from t + 4 on the bar holds again
# the 24 b values: carried from frame to frame, the product's integer splat is their whole distance. Measured on
# these sequences: > 1e-4 in 28 % of the frames, worst 0.015 (seed 2059 around its tie: 0.19); the bar below is 3 x the worst
B24_BOUND = {1000: 0.05, 1001: 0.05, 1002: 0.05, 1003: 0.05, 2059: 0.6}
TRAJECTORY_BOUND = 2e-4    # T_1 ... T_199 accumulated, product against oracle, without an event (measured <= 5.3e-5); with one: 2e-3
ALL_SEEDS = ls.SEEDS + ls.EVENT_SEEDS


def _fixture():
    with np.load(FIXTURE) as z:
        return {k: z[k] for k in z.files}


def test_fixture_shape_and_ground_truth():
    fx = _fixture()
    assert tuple(fx["seeds"]) == ALL_SEEDS and int(fx["frames"]) == ls.FRAMES
    assert fx["T"].shape == (5, ls.FRAMES - 1, 4, 4) and fx["decisions"].shape == (5, ls.FRAMES - 1, ls.ROWS * ls.COLS // 8)
    assert (fx["status"] == 0).all()
    # the oracle tracks the generator's camera: per frame and over the whole walk
    for q in range(len(ALL_SEEDS)):
        per = [ls.pose_delta(fx["T_gt"][q][k + 1], fx["T"][q][k]) for k in range(ls.FRAMES - 1)]
        assert max(p[0] for p in per) < 5e-3 and max(p[1] for p in per) < 1e-2, q
        rot, trans = ls.pose_delta(ls.chain(list(fx["T_gt"][q][1:])), ls.chain(list(fx["T"][q])))
        assert rot < 0.05 and trans < 0.1, (q, rot, trans)  # frame-to-frame odometry without a map drifts (measured: 20 mrad, 4 cm)


def test_oracle_prefix_reproduces_the_fixture(ora, tmp_path):
    """24 frames of sequence 1000 on the CPU: the oracle's poses, b, counts, labels and decisions are the fixture's, bit for bit
    (the rendering is a prefix of the 200-frame walk: the trajectory generator is causal)."""
    from staticfusion_amd.synth import sequence_arrays

    fx = _fixture()
    F = 25
    d, i, T_gt = sequence_arrays(ls.SEEDS[0], F)
    assert np.array_equal(T_gt, fx["T_gt"][0][:F])
    r = ls.Runner(ora, ls.HostPool(d), ls.HostPool(i), 1, F)
    for k in range(F - 1):
        r.step()
        rec = r.frame_records(images=False)[0]
        assert np.array_equal(rec["T"], fx["T"][0][k]) and np.array_equal(rec["b"], fx["b"][0][k]), k
        assert rec["counts"] == tuple(fx["counts"][0][k]) and rec["label_crc"] == int(fx["label_crc"][0][k]), k
        assert np.array_equal(rec["decisions"], fx["decisions"][0][k]), k
    r.close()


def test_event_bookkeeping():
    """events_and_curves / frames_back_under on hand-made records."""
    def rec(d, flip=None):
        r = {"rot": d, "trans": d / 2, "counts": [9, 40], "counts_ref": [9, 40]}
        if flip:
            r["flip"] = {"kind": flip, "rel_margin": 0.001}
            r["counts"] = [9, 41]
        return r

    quiet = [rec(1e-6) for _ in range(10)]
    assert ls.events_and_curves(quiet) == ([], [])
    run = [rec(1e-6)] * 3 + [rec(4e-4, "threshold"), rec(2e-4), rec(3e-5), rec(1e-6), rec(1e-6)]
    ev, curve = ls.events_and_curves(run)
    assert [e["frame"] for e in ev] == [4] and ev[0]["kind"] == "threshold" and curve[:3] == [4e-4, 2e-4, 3e-5]
    assert ls.frames_back_under(curve) == 2
    alone = [rec(1e-6)] * 2 + [rec(3e-4)] + [rec(1e-6)] * 3  # past the bar with identical counts
    ev, curve = ls.events_and_curves(alone)
    assert [(e["frame"], e["kind"]) for e in ev] == [(3, "no-count-mismatch")] and ls.frames_back_under(curve) == 1


# ---------------------------------------------------------------------------------------------------------------- GPU


@pytest.fixture(scope="module")
def pools():
    """The five sequences, rendered once per session on the box's cores (0.03 s per frame and core), in host and device pools."""
    import multiprocessing as mp

    from staticfusion_amd.synth import sequence_arrays

    with mp.get_context("spawn").Pool(min(8, len(os.sched_getaffinity(0)))) as pool:
        arrs = [sequence_arrays(s, ls.FRAMES, pool=pool, cache_dir=os.environ.get("SF_BENCH_CACHE", "/tmp")) for s in ALL_SEEDS]
    d, i = np.concatenate([a[0] for a in arrs]), np.concatenate([a[1] for a in arrs])
    out = {"host": (ls.HostPool(d), ls.HostPool(i)), "dev": (ls.DevicePool(d), ls.DevicePool(i)), "T_gt": [a[2] for a in arrs]}
    yield out
    for p in out["dev"]:
        p.free()


@pytest.fixture(scope="module")
def oracle_frames(ora, pools):
    """The oracle's 5 x 199 frames with their label and b images (labels as bytes), checked against the fixture on the way."""
    fx = _fixture()
    r = ls.Runner(ora, *pools["host"], len(ALL_SEEDS), ls.FRAMES)
    frames = []
    for k in range(ls.FRAMES - 1):
        r.step()
        recs = r.frame_records(images=True)
        for q, rec in enumerate(recs):
            assert np.array_equal(rec["T"], fx["T"][q][k]) and np.array_equal(rec["b"], fx["b"][q][k]), (q, k)
            assert rec["counts"] == tuple(fx["counts"][q][k]) and rec["label_crc"] == int(fx["label_crc"][q][k]), (q, k)
            rec["labels"] = rec["labels"].astype(np.uint8)
        frames.append(recs)
    r.close()
    return frames


def _dump(name, payload):
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "long_sequences_%s.json" % name), "w") as f:
            json.dump(payload, f)


def _lockstep(api, pools, oracle_frames, thr, variant=None):
    r = ls.Runner(api, *pools["dev"], len(ALL_SEEDS), ls.FRAMES, variant=variant)
    per_stream = [[] for _ in ALL_SEEDS]
    Ts = []
    for k in range(ls.FRAMES - 1):
        r.step()
        recs = r.frame_records(images=True)
        Ts.append(np.stack([x["T"] for x in recs]))
        for q, got in enumerate(recs):
            got["labels"] = got["labels"].astype(np.uint8)
            per_stream[q].append(ls.compare(oracle_frames[k][q], got, thr))
    r.close()
    return per_stream, np.stack(Ts)


@pytest.mark.gpu
def test_200_frames_against_the_oracle(hip, ora, pools, oracle_frames):
    thr = float(ora.default_params_struct().irls_delta_threshold)
    per_stream, T_loop = _lockstep(hip, pools, oracle_frames, thr)
    summaries = []
    for q, recs in enumerate(per_stream):
        s = ls.summarise_stream(recs)
        A_ref, A_got = ls.chain([f[q]["T"] for f in oracle_frames]), ls.chain(list(T_loop[:, q]))
        A_gt = ls.chain(list(pools["T_gt"][q][1:]))
        s["trajectory"] = {"got_vs_ref": ls.pose_delta(A_ref, A_got), "ref_vs_gt": ls.pose_delta(A_gt, A_ref), "got_vs_gt": ls.pose_delta(A_gt, A_got)}
        s["seed"] = ALL_SEEDS[q]
        s["b_img_worst"] = max(r["b_img"] for r in recs)
        s["dist"] = [float("%.3g" % max(r["rot"], r["trans"])) for r in recs]
        summaries.append(s)
    _dump(hip.default_variant, summaries)
    for q, (recs, s) in enumerate(zip(per_stream, summaries)):
        assert s["label_mismatch_frames"] == 0 and s["decision_mismatch_frames"] == 0, (q, s)
        assert s["b24_worst"] <= B24_BOUND[s["seed"]], (q, s["b24_worst"])
        dist = np.array([max(r["rot"], r["trans"]) for r in recs])
        allowed = np.full(len(recs), ls.POSE_BAR)
        for e in s["events"]:
            # a count mismatch as a tie within 2 % of its threshold -- or in the frames right behind one (they start from the state
            # the tie left: seed 2071 in the hunt has a second mismatch at t + 1)
            assert e["kind"] in ("threshold", "level-exit") or any(0 < e["frame"] - x["frame"] <= AFTER_EVENT_FRAMES for x in s["events"]), (q, e)
            allowed[e["frame"] - 1: e["frame"] + AFTER_EVENT_FRAMES] = AFTER_EVENT_BOUND
        assert (dist <= allowed).all(), (q, s["events"], [(int(k) + 1, float(dist[k])) for k in np.nonzero(dist > allowed)[0]])
        # the whole walk: product and oracle end within the sum of what single frames may differ by, and equally far from the truth
        rot, trans = s["trajectory"]["got_vs_ref"]
        bound = 2e-3 if s["events"] else TRAJECTORY_BOUND
        assert rot <= bound and trans <= bound, (q, rot, trans)
        assert abs(s["trajectory"]["got_vs_gt"][1] - s["trajectory"]["ref_vs_gt"][1]) <= bound
    # the same 199 frames of every stream in ONE launch: the loop's poses, bit for bit
    r = ls.Runner(hip, *pools["dev"], len(ALL_SEEDS), ls.FRAMES)
    T_one = r.run_in_one_launch(ls.FRAMES - 1)
    r.close()
    assert np.array_equal(T_one, T_loop)
    # ... also with round 4's form of the in-launch advance (the new frame copied into the pyramid buffer instead of read in the pool)
    os.environ["SF_NO_POOL_IN_PLACE"] = "1"
    try:
        r = ls.Runner(hip, *pools["dev"], len(ALL_SEEDS), ls.FRAMES)
        T_copy = r.run_in_one_launch(ls.FRAMES - 1)
        r.close()
    finally:
        del os.environ["SF_NO_POOL_IN_PLACE"]
    assert np.array_equal(T_copy, T_loop)


@pytest.mark.gpu
@pytest.mark.parametrize("build", ["throughput", "latency"])
def test_reference_order_build_is_bit_identical_for_200_frames(ora, pools, oracle_frames, build):
    import staticfusion_amd as sf

    api = sf.Api(os.path.join(os.path.dirname(sf.LIB), "libsf_hip_reforder.so"), "sf_").with_variant(build)
    assert api.backend_name() == "hip:gfx950:reference-order"
    thr = float(ora.default_params_struct().irls_delta_threshold)
    per_stream, _ = _lockstep(api, pools, oracle_frames, thr)
    for q, recs in enumerate(per_stream):
        bad = [k + 1 for k, r in enumerate(recs) if not r["identical"]]
        assert not bad, (q, bad[:10])
