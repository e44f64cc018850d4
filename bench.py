#!/usr/bin/env python3
"""bench.py — throughput of the StaticFusion solver hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
For N > 1 it runs one rank per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* from the environment); rank 0 prints ONE JSON line. Started DIRECTLY with --gpus N > 1
(no WORLD_SIZE in the environment) it launches those N ranks itself -- it re-executes under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` -- or
exits non-zero when the node has fewer than N GPUs; a line never claims more GPUs than ranks ran
(world_size == n_gpus == len(per_rank), one distinct device per rank, asserted before printing).

A "step" is ONE pass of the hot path over ONE batch of synthetic RGB-D pairs that is already
resident in HBM: the reference drivers' per-frame sequence
    createImagePyramid(true); runSolver(true); computeResidualsAgainstPreviousImage; buildSegmImage
(reference StaticFusion-datasets.cpp:171-184) for every stream of the batch = one launch of
`sf_frame_kernel` through the C ABI (sf_process_frame).

Workloads
  static     BASELINE.json configs[1] (the configuration `metric` is quoted on): synthetic static RGB-D pairs (VGA
             render decimated to QVGA 320x240), 3-level pyramid, segmentation disabled (pure 6-DoF Cauchy IRLS),
             constructor parameters (max_iter_irls 10, delta 1e-6, no motion filter).  THE DEFAULT: `value`.
             The default run ALSO times configs[2] in the same process and reports it as the `full_solver` block.
  sphere     BASELINE.json configs[2]: moving sphere, full solver, K-means(24), b-field, driver parameters, 5 levels.
  sequences  SURVEY.md section 8(d) config 5 / BASELINE configs[4] shape: per rank, independent synthetic SEQUENCES (seeds
             1000 + rank ..., smooth random-walk camera, a sphere swinging through the room), resident in HBM; every
             step advances every stream by one frame with the previous frame as prediction (frame-to-frame, the
             dataset drivers' loop without the map), so iteration counts are data dependent and streams diverge.

metric  = solver iterations/s: executions of the IRLS loop body (reference FrontEnd.cpp:611-684)
          summed over all streams, steps and GPUs, divided by the wall time of the K timed steps
          (max over ranks).  frames/s is reported beside it.
Multi-GPU: independent streams per GPU (no data-path collective), weak scaling; the line carries world_size and the
per-rank rates so that the rank count RCCL saw can be checked from the output.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOAD_TEXT = {
    "static": ("BASELINE.json configs[1]: synthetic static RGB-D pairs, VGA->QVGA 320x240, 3-level pyramid, "
               "segmentation disabled (pure 6-DoF Cauchy IRLS)"),
    "sphere": ("BASELINE.json configs[2]: synthetic QVGA pairs with a moving sphere, full solver, "
               "K-means(24) + b-field, driver parameters, 5 levels"),
    "sequences": ("SURVEY 8(d) config 5: independent synthetic QVGA sequences per rank (seeds 1000 + rank ..., smooth random-walk "
                  "camera, swinging sphere), frame-to-frame prediction, full solver, K-means(24) + b-field, driver parameters"),
    "tum": ("BASELINE.json configs[0] / configs[3] in frame-to-frame mode (SURVEY 8(d) config 1 substitute): a TUM-format directory "
            "(rgb/ depth/ rgbd_assoc.txt, reference README.md:67-89) through the loader + bilateral depth filter of the input stage, "
            "every stream plays the sequence from its own start frame, full solver, driver parameters"),
}

# BASELINE.json configs this environment cannot exercise, and why; `--workload tum --dataset DIR` runs [0] / [3] in
# frame-to-frame mode the moment a dataset is mounted (tests/test_gpu_parity_hunt.py::test_tum_dataset_bench skips likewise)
CONFIGS_UNAVAILABLE = [
    "configs[0] TUM fr1/360 rawlog: dataset absent (no network); the rawlog container itself needs MRPT -- the PNG + rgbd_assoc.txt "
    "form of the sequence runs with --workload tum --dataset DIR",
    "configs[3] TUM fr3/walking_xyz feeding the OpenGL surfel model: dataset absent, no OpenGL / Pangolin in the image -- frame-to-frame "
    "with --workload tum --dataset DIR, headless fusion with tools/run_sequence.py --mode fusion",
    "configs[4] 8 sequences on 8 GPUs: this process sees the GPUs the driver gives it (python -m torch.distributed.run ... bench.py --gpus N)",
]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16384, help="independent streams per GPU (8.3 MB of HBM each: 136 GB); 16 rounds of the 1024 resident workgroups")
    ap.add_argument("--workload", choices=["static", "sphere", "sequences", "tum"], default="static")
    ap.add_argument("--dataset", default=os.environ.get("SF_TUM_DATASET"), help="tum workload: directory with rgb/ depth/ rgbd_assoc.txt")
    ap.add_argument("--variant", choices=["auto", "throughput", "latency", "cluster"], default="auto", help="frame-kernel build (sf_create_ex)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic pairs (tiled over the batch)")
    ap.add_argument("--seq-distinct", type=int, default=4, help="sequences workload: distinct sequences per rank")
    ap.add_argument("--seq-frames", type=int, default=200, help="sequences workload: frames per sequence")
    ap.add_argument("--no-full-solver", action="store_true", help="static workload: skip the configs[2] block")
    ap.add_argument("--no-sequences", action="store_true", help="static workload: skip the sequences blocks")
    ap.add_argument("--seq-batches", default="16384,4096", help="static workload: stream counts of the sequences blocks")
    ap.add_argument("--launch-per-frame", action="store_true", help="A/B: one kernel launch per step instead of one launch for the K timed steps")
    ap.add_argument("--pass-reps", type=int, default=4, help="repetitions of each isolated IRLS pass (roofline.irls_passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


def make_params(api, workload):
    if workload == "static":
        p = api.ctor_params_struct()
        p.ctf_levels = 3
        p.segmentation_enabled = 0
    else:
        p = api.default_params_struct()
        p.kb = 1.05
    return p


# bytes per unit: SURVEY.md §8(d)'s algorithmic figures, and what the kernels of this design really move per unit (DESIGN.md §5,
# right-hand column; measured per stage group in profiles/*traffic_by_stage*): the IRLS passes stream 2 x 29 B records, the
# linearisation reads ~29 B and writes 25 B instead of materialising A | B, ...
ALGORITHMIC_B = {"irls": 60, "linearise": 88, "warp": 32, "pyramid": 48, "kmeans": 20, "segm_image": 8, "residuals": 32}
MOVED_B = {"irls": 58, "linearise": 50, "warp": 25, "pyramid": 40, "kmeans": 6, "segm_image": 5, "residuals": 72}  # linearise: 25 read (strips: no halo re-reads) + 25 written


def algorithmic_bytes(stats_list, levels_n, n_l1, seg, with_residuals, pyramids_per_frame=2.0, per_unit=ALGORITHMIC_B):
    """Per-unit figures x the units one launch processed. pyramids_per_frame: createImagePyramid calls that
    really ran per frame of a stream -- 2 (old + new image) unless a launch of sequence frames swapped the pyramid buffers
    instead of rebuilding the old image's pyramid (sequence_pyramids_per_frame)."""
    total = {"irls": 0, "linearise": 0, "warp": 0, "pyramid": 0, "kmeans": 0, "segm_image": 0, "residuals": 0}
    n_levels = len(levels_n)
    for st in stats_list:
        total["irls"] += per_unit["irls"] * int(st.pixel_iters)
        for i in range(st.n_outer):
            L = n_levels - 1 - st.outer[i].level  # image level
            total["linearise"] += per_unit["linearise"] * levels_n[L]
            if not (st.outer[i].level == 0 and st.outer[i].k == 0):
                total["warp"] += per_unit["warp"] * levels_n[L]
        total["pyramid"] += pyramids_per_frame * per_unit["pyramid"] * sum(levels_n[1:])
        if seg:
            total["kmeans"] += per_unit["kmeans"] * n_l1 * int(st.kmeans_iters)
        total["segm_image"] += per_unit["segm_image"] * levels_n[0]
        if with_residuals:
            total["residuals"] += per_unit["residuals"] * levels_n[0]
    return total


def sequence_pyramids_per_frame(frames_per_launch):
    """createImagePyramid calls per frame of a stream in ONE launch of K sequence frames (sf_frame_kernels.hip, `swap`): frame 0
    builds both pyramids; frames 1 .. K - 2 swap the two pyramid buffers of the stream (the old image's pyramid IS the
    previous frame's new one) and build one; the last frame swaps only if that returns the buffers to the host's layout
    (K - 2 odd). Returns (pyramids per frame, frames that swapped)."""
    K = int(frames_per_launch)
    swapped = 0 if K < 2 else (K - 2) + ((K - 2) % 2)
    return (2.0 * K - swapped) / K, swapped


def reduce_over_ranks(dist, device, elapsed, iters, frames):
    """MAX of the elapsed time, SUM of the counters over ranks (dist is None for one process)."""
    if dist is None:
        return elapsed, iters, frames
    import torch

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(iters), float(frames)], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), float(c[0].item()), float(c[1].item())


def gather_per_rank(dist, device, elapsed, iters, frames):
    """[(elapsed, iters, frames)] of every rank, in rank order: evidence of the ranks the collective saw."""
    if dist is None:
        return [(elapsed, float(iters), float(frames))]
    import torch

    mine = torch.tensor([elapsed, float(iters), float(frames)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [tuple(float(x) for x in t.tolist()) for t in out]


def source_sha():
    """Identity of the device code a measurement belongs to: sha256 over the kernel sources and headers."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "staticfusion_amd", "csrc")
    files = sorted(f for f in os.listdir(base) if f.endswith((".h", ".hip", ".cpp")) or f == "Makefile")
    for f in files + ["../../include/sf.h", "../../include/sf_detmath.h"]:
        with open(os.path.join(base, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def git_head():
    """HEAD of the checkout, or -- on a GPU box, which receives a snapshot without .git -- what __graft_entry__.build() /
    tools/stamp_head.sh recorded in staticfusion_amd/csrc/BUILD_HEAD when the libraries were built."""
    try:
        return subprocess.check_output(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, "staticfusion_amd", "csrc", "BUILD_HEAD")) as f:
            return f.read().strip() or None
    except Exception:
        return None


def measured_traffic(workload, batch, variant, frames_per_launch=None):
    """HBM bytes per launch from the PMC counters (tools/measure_traffic.sh writes profiles/traffic_*.json together with the
    identity of the sources it measured). A file that belongs to other sources, another batch or another build of the
    kernel is NOT reported: null plus the reason."""
    path = os.path.join(ROOT, "profiles", "traffic_%s_b%d.json" % (workload, batch))
    if not os.path.exists(path):
        return None, "no PMC measurement for this workload and batch (tools/measure_traffic.sh)"
    with open(path) as f:
        t = json.load(f)
    now = source_sha()
    if t.get("src_sha") != now or t.get("batch") != batch or t.get("variant", variant) != variant:
        return None, "stale: measured on sources %s (batch %s, %s), running %s" % (t.get("src_sha"), t.get("batch"), t.get("variant"), now)
    if frames_per_launch is not None and t.get("frames_per_launch", 1) != frames_per_launch:
        # a launch of K sequence frames skips pyramids and copies a launch per frame does not: only like is compared with like
        return None, "measured with %d frame(s) per launch, this run has %d (tools/measure_traffic.sh %s %d %s %d)" % (
            t.get("frames_per_launch", 1), frames_per_launch, workload, batch, variant, frames_per_launch)
    return {"hbm_bytes_per_launch": t["hbm_bytes_per_launch"], "src_sha": t["src_sha"], "head": t.get("head"), "batch": t["batch"],
            "variant": t.get("variant"), "frames_per_launch": t.get("frames_per_launch", 1), "per": t.get("per"), "ratio_to_algorithmic": None,
            "valu_issue": t.get("valu_issue")}, None


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch_if_needed(args):
    """`python bench.py --gpus N` started directly (WORLD_SIZE unset) with N > 1: one process cannot be N ranks. Re-execute
    under torch.distributed.run with N ranks on this node, or fail loudly when the node does not have N GPUs.
    (SF_BENCH_SINGLE_GPU=1, the one-GPU test hook that puts every rank on cuda:0, lifts the device-count check.)"""
    if "WORLD_SIZE" in os.environ or args.gpus <= 1:
        return
    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not os.environ.get("SF_BENCH_SINGLE_GPU"):
        sys.stderr.write("bench.py: --gpus %d requested but this node exposes %d GPU(s); refusing to print a line for GPUs that did not run\n"
                         % (args.gpus, have))
        sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


class Harness:
    """Device selection, process group and the barrier of the contract."""

    def __init__(self, args):
        import torch  # plumbing only: device sync + torch.distributed (RCCL) barrier / reductions

        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:  # a line must never report GPUs that no rank ran on
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python -m torch.distributed.run "
                             "--nproc-per-node %d ... bench.py --gpus %d), or run `python bench.py --gpus %d` directly and let it "
                             "launch the ranks" % (args.gpus, self.world, args.gpus, args.gpus, args.gpus))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
        # Test hooks (tests/test_gpu_parity_sweep.py runs the N > 1 flow on a one-GPU box): SF_BENCH_BACKEND=gloo does the
        # barrier / reductions over gloo on CPU tensors, SF_BENCH_SINGLE_GPU=1 puts every rank on cuda:0. The driver's runs
        # use neither: one rank per GPU, RCCL ("nccl").
        self.backend = os.environ.get("SF_BENCH_BACKEND", "nccl")
        self.dev_index = 0 if os.environ.get("SF_BENCH_SINGLE_GPU") else self.local_rank
        torch.cuda.set_device(self.dev_index)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.dev_index))
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
            self.dist = dist
        self.reduce_device = torch.device("cuda", self.dev_index) if self.backend == "nccl" else torch.device("cpu")
        # which physical device every rank sits on (gathered: the line carries it, and main() asserts they are distinct)
        prop = torch.cuda.get_device_properties(self.dev_index)
        mine = {"rank": self.rank, "device_index": int(torch.cuda.current_device()), "name": prop.name,
                "uuid": str(getattr(prop, "uuid", "")), "pci_bus_id": int(getattr(prop, "pci_bus_id", -1)), "host": os.uname().nodename}
        self.devices = [mine]
        if self.dist is not None:
            got = [None] * self.world
            self.dist.all_gather_object(got, mine)
            self.devices = got

    def barrier(self, solver):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        solver.synchronize()

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def oracle_check(pairs, workload, T_first, solver, n=4):
    """Parity of the first frame against the CPU oracle on the first n distinct pairs (the checker: never timed)."""
    import staticfusion_amd as sf
    from oracle import binding
    from staticfusion_amd.synth import pose_delta

    ora = binding.load()
    n = min(n, len(pairs))
    o = sf.Solver(ora, 240, 320, n, make_params(ora, workload))
    for b in range(n):
        o.set_current(b, *pairs[b]["new"])
        o.set_prediction(b, *pairs[b]["old"])
    o.process_frame(0)
    rot_max = trans_max = 0.0
    labels_equal = decisions_equal = True
    for b in range(n):
        r, t = pose_delta(o.T(b), T_first[b])
        rot_max, trans_max = max(rot_max, r), max(trans_max, t)
        if workload != "static":
            labels_equal = labels_equal and all(np.array_equal(solver.labels(L, b), o.labels(L, b)) for L in range(solver.levels))
            decisions_equal = decisions_equal and np.array_equal(solver.b_image(b) > 0.5, o.b_image(b) > 0.5)
    out = {"rot_rad": rot_max, "trans_m": trans_max, "frames_checked": n}
    if workload != "static":
        out["cluster_labels_identical"] = bool(labels_equal)
        out["static_dynamic_decision_identical"] = bool(decisions_equal)
    return out


def run_pairs_workload(hx, args, workload, batch, with_parity):
    """configs[1] / configs[2]: one launch of the frame kernel per step over `batch` resident pairs. Returns the block."""
    import staticfusion_amd as sf
    from staticfusion_amd.synth import make_batch

    api = sf.load()
    params = make_params(api, workload)
    rows, cols, B = 240, 320, batch
    pairs = make_batch(args.distinct, base_seed=1234 + 100000 * hx.rank, sphere=(workload == "sphere"), distinct=args.distinct)
    solver = sf.Solver(api, rows, cols, B, params, device=hx.dev_index, variant=args.variant)
    for b in range(B):
        pr = pairs[b % len(pairs)]
        solver.set_current(b, *pr["new"])
        solver.set_prediction(b, *pr["old"])
    solver.process_frame(0)
    solver.synchronize()
    T_first, _, _, _ = solver.batch_results()
    parity = oracle_check(pairs, workload, T_first, solver) if (with_parity and hx.rank == 0) else None

    for im in range(1, 5):  # prime the 5-frame history so that the timed frames include computeResiduals
        solver.process_frame(im)
    im = 5
    for _ in range(args.warmup):
        solver.process_frame(im)
        im += 1
    solver.synchronize()
    copy_gbs = copy_bandwidth(solver)

    # ---- the timed region: exactly K steps = K launches of sf_frame_kernel, back to back, bracketed by barrier + device
    #      synchronisation. HIP events recorded on the handle's stream around the same K launches give the average kernel
    #      duration; the device-side counters give the IRLS iterations those K steps executed (read outside the region).
    c0 = solver.counters()
    clk0 = solver.shader_clock_counters()
    hx.barrier(solver)
    t0 = time.perf_counter()
    region_ms = solver.timed_process_frames(im, args.steps)
    hx.barrier(solver)
    elapsed = time.perf_counter() - t0
    im += args.steps
    c1 = solver.counters()
    shader_mhz = solver.shader_clock_mhz(clk0, solver.shader_clock_counters())
    frames_timed, iters_total, pix_total = c1[0] - c0[0], c1[1] - c0[1], c1[3] - c0[3]
    assert frames_timed == B * args.steps, (frames_timed, B, args.steps)
    k_ms = region_ms / args.steps
    solver.process_frame(im)  # per-stage unit counts of ONE launch (for the algorithmic byte count): one more, un-timed step
    stats_last = [solver.stats(b) for b in range(min(B, args.distinct))]
    status_or = 0
    for st in stats_last:
        status_or |= int(st.status)
    assert status_or & sf.STATUS_SYNC_TIMEOUT == 0, "a cluster rendezvous timed out: the frames of this run are not valid"
    variant = solver.variant()
    irls_passes = isolated_passes(solver, B, rows * cols, args.pass_reps, copy_gbs) if variant[0] != "cluster" else None
    resident = solver.resident_workgroups()
    levels_n = [solver.level_shape(L)[0] * solver.level_shape(L)[1] for L in range(solver.levels)]
    levels = int(solver.levels)
    solver.close()

    t_max, iters_all, frames_all = reduce_over_ranks(hx.dist, hx.reduce_device, elapsed, iters_total, B * args.steps)
    per_rank = gather_per_rank(hx.dist, hx.reduce_device, elapsed, iters_total, B * args.steps)
    seg = bool(params.segmentation_enabled)
    per_stream = algorithmic_bytes(stats_last, levels_n, levels_n[1], seg, True)
    moved_per_stream = algorithmic_bytes(stats_last, levels_n, levels_n[1], seg, True, per_unit=MOVED_B)
    alg_bytes_launch = sum(per_stream.values()) * B / float(len(stats_last))
    achieved = alg_bytes_launch / (k_ms * 1e-3) / 1e9
    traffic, why = measured_traffic(workload, B, variant[0])
    if traffic:
        traffic["ratio_to_algorithmic"] = traffic["hbm_bytes_per_launch"] / alg_bytes_launch
    return {
        "workload": workload,
        "value": iters_all / t_max,
        "frames_per_s": frames_all / t_max,
        "ms_per_step": 1e3 * t_max / args.steps,
        "iterations_per_frame": iters_total / float(B * args.steps),
        "pixel_iterations_per_s": pix_total / elapsed * hx.world,
        "parity": parity,
        "per_rank": [{"rank": r, "elapsed_s": e, "iterations_per_s": i / e, "frames_per_s": f / e} for r, (e, i, f) in enumerate(per_rank)],
        "config": {
            "workload": WORKLOAD_TEXT[workload],
            "streams_per_gpu": B, "distinct_pairs": len(pairs), "rows": rows, "cols": cols, "ctf_levels": levels,
            "max_iter_irls": int(params.max_iter_irls), "max_iter_per_level": int(params.max_iter_per_level),
            "kernel_build": "%s (%d threads per workgroup, %d workgroup(s) per stream)" % variant + ", %d workgroups resident per CU (grid %d)" % resident,
            "parallelism": "independent streams, %d GPU(s)" % hx.world,
            "step": "the frame sequence of sf_process_frame -- pyramid(old) + runSolver(true) + residuals + segm image -- for every stream; "
                    + ("one launch per step" if os.environ.get("SF_TIMED_LAUNCH_PER_FRAME") else "the K timed steps are ONE launch of sf_frame_kernel "
                       "(sf_process_frames): a stream starts its next frame when ITS previous one is done, no barrier over the batch between steps"),
            "launches_in_timed_region": args.steps if os.environ.get("SF_TIMED_LAUNCH_PER_FRAME") else 1,
        },
        "roofline": {
            "kernel": "sf_frame_kernel (%s build)" % variant[0],
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            # the second fraction SURVEY 8(d) asks for: against a plain copy kernel measured on this box right before the timed region
            "copy_gbs": copy_gbs, "frac_of_copy": (achieved / copy_gbs) if copy_gbs else None,
            # the shader clock the timed stream-frames ran at (in-kernel: clock64 over the 100 MHz wall clock): the package's
            # power management restarts low at every launch and climbs for several hundred ms
            "shader_clock_mhz": shader_mhz,
            # HBM bytes per launch from the PMC counters (measured per frame of every stream, tools/measure_traffic.sh, times the
            # frames of this launch), or null when no measurement of THESE sources exists
            "traffic": traffic["hbm_bytes_per_launch"] * (1 if os.environ.get("SF_TIMED_LAUNCH_PER_FRAME") else args.steps) if traffic else None,
            "traffic_provenance": traffic, "traffic_note": why,
            # one launch covers frames_per_launch steps (sf_process_frames); bytes and duration per STEP, i.e. per frame of
            # every stream, are what `achieved` divides -- the same ratio as bytes per launch / launch duration
            "frames_per_launch": 1 if os.environ.get("SF_TIMED_LAUNCH_PER_FRAME") else args.steps,
            "algorithmic_bytes_per_launch": alg_bytes_launch * (1 if os.environ.get("SF_TIMED_LAUNCH_PER_FRAME") else args.steps),
            "kernel_launch_ms": k_ms * (1 if os.environ.get("SF_TIMED_LAUNCH_PER_FRAME") else args.steps),
            "algorithmic_bytes_per_step": alg_bytes_launch,
            "algorithmic_bytes_breakdown_per_stream": {k: v / float(len(stats_last)) for k, v in per_stream.items()},
            # what the kernels of this design move for the same units (records instead of A | B, 16-byte
            # accumulator cells out and back, ...); the PMC total above is the measurement these design figures add up to
            "moved_bytes_breakdown_per_stream": {k: v / float(len(stats_last)) for k, v in moved_per_stream.items()},
            "bytes_per_unit": {"algorithmic": ALGORITHMIC_B, "moved": MOVED_B},
            "kernel_ms_avg": k_ms,
            # the north star's "residual / Jacobian kernel" on its own: the two streaming passes of one IRLS iteration
            # launched alone (sf_irls_pass_kernel) over the level-0 records of every stream of THIS handle
            "irls_passes": irls_passes,
        },
        "pairs": pairs,
    }


COPY_BYTES = 1 << 30  # sf_microbench_copy: 1 GiB read + 1 GiB written per repetition


def copy_bandwidth(solver):
    """"fraction = achieved / measured-peak copy bandwidth and / 8 TB/s (both quoted)". A 2 GiB device copy
    (16-byte loads and stores, sf_microbench_copy) on the handle's stream right before the timed region of a block: what a plain
    streaming kernel reaches on THIS box at THIS moment (the package's power-limited clock moves it by +- 2-4 % from box to box)."""
    import staticfusion_amd as sf

    # two scratch blocks beside the solver's own allocations: at the largest batches they may not fit -- the field is optional,
    # the block is not (a quarter of the size still streams 1 GiB per repetition; null when that does not fit either)
    for nbytes in (COPY_BYTES, COPY_BYTES // 4):
        try:
            solver.microbench_copy(nbytes, 2)
            return max(solver.microbench_copy(nbytes, 4) for _ in range(3))  # (the best of three: single runs scatter by 3 %)
        except sf.SfError as e:
            if "scratch blocks" not in str(e):
                raise
    return None


def isolated_passes(solver, B, n0, reps, copy_gbs=None):
    """Each IRLS pass alone over level 0 of all B streams: ms per repetition, GB/s against the algorithmic 30 B per pixel and
    pass (60 B per pixel and IRLS iteration), fraction of the 8 TB/s peak; `iteration` = both passes."""
    out = {}
    px = float(B) * n0
    for which, name in ((1, "pass1_weights_normal_equations"), (2, "pass2_residuals_label_sums")):
        solver.microbench_pass(which, 0, 1)
        ms = solver.microbench_pass(which, 0, reps) / reps
        out[name] = {"ms": ms, "gpx_per_s": px / ms / 1e6, "bytes": 30.0 * px, "achieved": 30.0 * px / ms / 1e6, "frac": 30.0 * px / ms / 1e6 / HBM_PEAK_GBS,
                     "frac_of_copy": (30.0 * px / ms / 1e6 / copy_gbs) if copy_gbs else None}
    ms = sum(v["ms"] for v in out.values())
    out["iteration"] = {"ms": ms, "bytes": 60.0 * px, "achieved": 60.0 * px / ms / 1e6, "unit": "GB/s", "frac": 60.0 * px / ms / 1e6 / HBM_PEAK_GBS,
                        "frac_of_copy": (60.0 * px / ms / 1e6 / copy_gbs) if copy_gbs else None,
                        "pixels": px, "repetitions": reps, "kernel": "sf_irls_pass_kernel (level 0 of every stream, one pass per launch)"}
    return out


def _cached_sequence(seed, F, pool):
    """One synthetic sequence as (d [F][n0], i [F][n0], T_gt) -- from /tmp when a run on this node has rendered it before (the
    driver runs N = 1, 2, 4, 8 back to back: rank r's seeds at N recur at 2 N), else rendered and stored. The cache key
    carries a hash of the generator and the resolution (staticfusion_amd/synth.py: sequence_arrays)."""
    from staticfusion_amd.synth import sequence_arrays

    d, i, T_gt = sequence_arrays(seed, F, pool=pool, cache_dir=os.environ.get("SF_BENCH_CACHE", "/tmp"))
    return d, i, list(T_gt)


def synthetic_sequence_pool(hx, args):
    """The frames of D synthetic sequences per rank as two HBM-resident [D * F][n0] arrays (column-major images). The
    renderer processes of all ranks of the node share the CPUs the container may use (8 ranks x 16 processes on a 16-CPU
    quota only contend), and a sequence rendered once on this node is read back from /tmp."""
    import multiprocessing as mp

    D, F = args.seq_distinct, args.seq_frames
    cap = cgroup_cpu_limit()
    cpus = len(os.sched_getaffinity(0)) if cap is None else max(1, min(len(os.sched_getaffinity(0)), int(cap)))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(hx.world)))
    with mp.get_context("spawn").Pool(max(1, min(16, cpus // max(1, local_world)))) as pool:
        seqs = [_cached_sequence(1000 + hx.rank * D + q, F, pool) for q in range(D)]
    dev = "cuda:%d" % hx.dev_index
    return {
        "d": hx.torch.from_numpy(np.concatenate([s[0] for s in seqs])).to(dev),
        "i": hx.torch.from_numpy(np.concatenate([s[1] for s in seqs])).to(dev),
        "D": D, "F": F, "T_gt": [s[2] for s in seqs], "rows": 240, "cols": 320, "workload": "sequences",
        "what": {"distinct_sequences_per_rank": D, "frames_per_sequence": F, "sequence_seeds": [1000 + hx.rank * D, 1000 + hx.rank * D + D - 1]},
    }


def tum_sequence_pool(hx, args):
    """A TUM-format directory through the input stage (loader decimation / flip / RGB -> intensity, bilateral depth filter +
    metricise: reference FrontEnd.cpp:216-254, Reconstruction.cpp:722-732) into the same kind of pool: one sequence."""
    import staticfusion_amd as sf
    from staticfusion_amd import io as sfio

    if not args.dataset or not os.path.isdir(args.dataset):
        raise SystemExit("dataset absent: %r (no dataset ships with this repository; --dataset DIR or SF_TUM_DATASET)" % (args.dataset,))
    io = sfio.Io()
    directory = args.dataset if args.dataset.endswith("/") else args.dataset + "/"
    ts, files_depth, files_color = io.load_assoc(directory)
    F = min(len(ts), args.seq_frames)
    assert F >= args.warmup + args.steps + 12, "the sequence is too short for the requested steps (%d frames)" % F
    first = io.imread_depth16(files_depth[0])
    rows, cols = first.shape[0] // 2, first.shape[1] // 2
    api = sf.load()
    conv = sf.Solver(api, rows, cols, 1, make_params(api, "tum"), device=hx.dev_index)
    frames_d, frames_i = [], []
    col = lambda a: np.ascontiguousarray(np.asarray(a, np.float32).T).ravel()
    for k in range(F):
        conv.load_frame(0, io.imread_color(files_color[k]), io.imread_depth16(files_depth[k]), 2)
        if k:
            conv.filter_depth()  # the bootstrap frame is used unfiltered (StaticFusion-imagesequenceassoc.cpp:117-123)
        d, i = conv.current(0)
        frames_d.append(col(d))
        frames_i.append(col(i))
    conv.close()
    dev = "cuda:%d" % hx.dev_index
    return {"d": hx.torch.from_numpy(np.stack(frames_d)).to(dev), "i": hx.torch.from_numpy(np.stack(frames_i)).to(dev), "D": 1, "F": F,
            "T_gt": None, "rows": rows, "cols": cols, "workload": "tum", "what": {"dataset": os.path.abspath(args.dataset), "frames": F}}


LONG_FIXTURE = os.path.join(ROOT, "tests", "golden", "long_sequences_qvga.npz")


def sequences_parity(hx, api, params, pool, variant, first_timed, last_timed, T_last_big):
    """pose_delta_vs_cpu of a sequences block: the TIMED frames of the first stream of every distinct sequence against the CPU
    oracle's frame-by-frame poses for the same sequences (tests/golden/long_sequences_qvga.npz, the oracle's own output,
    tools/golden/make_golden_long_sequences.py; tests/test_long_sequences.py holds all 199 frames of it against every build).
    Nothing is added to the timed region: streams 0 .. D-1 of the batch play sequences 0 .. D-1 from frame 0, and a handle of D
    streams of the same build replays exactly those frames in one launch with its trajectory -- the results of a stream do not
    depend on the batch it runs in, which is CHECKED here: the replay's pose of the last timed frame must equal the big batch's
    bit for bit. None when the fixture does not cover this pool (other seeds, resolution or length)."""
    import staticfusion_amd as sf
    from staticfusion_amd.synth import pose_delta

    D, F = pool["D"], pool["F"]
    seeds = pool["what"].get("sequence_seeds")
    if pool["workload"] != "sequences" or not os.path.exists(LONG_FIXTURE) or seeds is None:
        return None
    with np.load(LONG_FIXTURE) as z:
        fx_seeds, fx_T, fx_frames = list(z["seeds"]), z["T"], int(z["frames"])
    want = list(range(seeds[0], seeds[1] + 1))
    if fx_frames != F or fx_seeds[:len(want)] != want or (pool["rows"], pool["cols"]) != (240, 320) or last_timed >= F:
        return {"note": "the oracle fixture covers seeds %s, %d frames at 240 x 320: not this pool" % (fx_seeds, fx_frames)}
    replay = sf.Solver(api, pool["rows"], pool["cols"], D, params, device=hx.dev_index, variant=variant)
    idx = lambda k: (np.arange(D) * F + k).astype(np.int32)
    replay.advance_sequences_device(pool["d"].data_ptr(), pool["i"].data_ptr(), idx(0), D * F)
    replay.push_history(0)
    T = replay.process_sequence_frames_device(pool["d"].data_ptr(), pool["i"].data_ptr(), np.stack([idx(k) for k in range(1, last_timed + 1)]), D * F, 1, trajectory=True)
    replay.close()
    same = bool(np.array_equal(T[last_timed - 1], T_last_big[:D]))
    timed = [pose_delta(fx_T[q][k - 1], T[k - 1][q]) for k in range(first_timed, last_timed + 1) for q in range(D)]
    whole = [pose_delta(fx_T[q][k - 1], T[k - 1][q]) for k in range(1, last_timed + 1) for q in range(D)]
    return {
        "rot_rad_max": max(p[0] for p in timed), "trans_m_max": max(p[1] for p in timed), "frames_compared": len(timed),
        "frames_past_1e-4": sum(1 for p in timed if max(p) > 1e-4), "streams": D, "timed_frames": [first_timed, last_timed],
        "from_frame_1": {"rot_rad_max": max(p[0] for p in whole), "trans_m_max": max(p[1] for p in whole), "frames_compared": len(whole),
                         "frames_past_1e-4": sum(1 for p in whole if max(p) > 1e-4)},
        "replay_reproduces_the_timed_streams_bit_for_bit": same,
        "reference": "CPU oracle, tests/golden/long_sequences_qvga.npz (seeds %d .. %d)" % (want[0], want[-1]),
    }


def run_sequences_workload(hx, args, B, pool):
    """Independent sequences per rank, resident in HBM, frame-to-frame prediction (see the module docstring)."""
    import staticfusion_amd as sf
    from staticfusion_amd.synth import pose_delta

    api = sf.load()
    params = make_params(api, pool["workload"])
    rows, cols = pool["rows"], pool["cols"]
    D, F = pool["D"], pool["F"]
    assert args.warmup + args.steps + 10 <= F, "sequence too short for the requested steps"
    n0 = rows * cols
    pool_d, pool_i = pool["d"], pool["i"]
    assert pool_d.shape == (D * F, n0)
    solver = sf.Solver(api, rows, cols, B, params, device=hx.dev_index, variant=args.variant)
    # stream b plays sequence b % D starting at frame phase[b]; frames wrap to 1 (never to 0) so that a wrap is a large
    # jump for one frame rather than a repeated bootstrap
    phase = (np.arange(B) // D * 7) % (F - 1)
    seq_of = np.arange(B) % D

    def index_at(step):
        return (seq_of * F + (phase + step) % F).astype(np.int32)

    solver.advance_sequences_device(pool_d.data_ptr(), pool_i.data_ptr(), index_at(0), D * F)
    solver.push_history(0)
    step = 1
    for _ in range(5 + args.warmup):  # bootstrap + the 5-frame ring + warm-up
        solver.advance_sequences_device(pool_d.data_ptr(), pool_i.data_ptr(), index_at(step), D * F)
        solver.process_frame(step)
        step += 1
    solver.synchronize()
    copy_gbs = copy_bandwidth(solver)
    c0 = solver.counters()
    clk0 = solver.shader_clock_counters()
    hx.barrier(solver)
    t0 = time.perf_counter()
    if args.launch_per_frame:
        for _ in range(args.steps):
            solver.advance_sequences_device(pool_d.data_ptr(), pool_i.data_ptr(), index_at(step), D * F)
            solver.process_frame(step)
            step += 1
    else:  # the K timed steps = K frames of every stream in ONE launch (sf_process_sequence_frames_device)
        solver.process_sequence_frames_device(pool_d.data_ptr(), pool_i.data_ptr(), np.stack([index_at(step + q) for q in range(args.steps)]), D * F, step)
        step += args.steps
    hx.barrier(solver)
    elapsed = time.perf_counter() - t0
    c1 = solver.counters()
    shader_mhz = solver.shader_clock_mhz(clk0, solver.shader_clock_counters())
    frames_timed, iters_total = c1[0] - c0[0], c1[1] - c0[1]
    assert frames_timed == B * args.steps
    T_after_timed = solver.batch_results()[0].copy()  # pose of the last timed frame of every stream (sequences_parity)
    first_timed, last_timed = step - args.steps, step - 1
    # kernel duration: the timed launch itself (HIP events on the handle's stream around it) / its frames; launch-per-frame
    # mode and the unit counts: three more, un-timed launches
    k_ms, stats_last = ([] if args.launch_per_frame else [solver.last_solver_kernel_ms() / args.steps]), None
    for _ in range(3):
        solver.advance_sequences_device(pool_d.data_ptr(), pool_i.data_ptr(), index_at(step), D * F)
        solver.process_frame(step)
        solver.synchronize()
        if args.launch_per_frame:
            k_ms.append(solver.last_solver_kernel_ms())
        step += 1
    ncheck = min(B, 4 * D)
    stats_last = [solver.stats(b) for b in range(ncheck)]
    # tracking check against the ground truth of the generator for the last frame (streams 0 .. D-1 did not wrap)
    T_all, n_irls, _, _ = solver.batch_results()
    status_or = 0
    for st in stats_last:
        status_or |= int(st.status)
    assert status_or & sf.STATUS_SYNC_TIMEOUT == 0, "a cluster rendezvous timed out: the frames of this run are not valid"
    err = []
    if pool["T_gt"] is not None:
        for b in range(min(B, D)):
            k = int((phase[b] + step - 1) % F)
            err.append(pose_delta(pool["T_gt"][b % D][k], T_all[b]))
    variant = solver.variant()
    resident = solver.resident_workgroups()
    levels_n = [solver.level_shape(L)[0] * solver.level_shape(L)[1] for L in range(solver.levels)]
    levels = int(solver.levels)
    solver.close()
    vs_cpu = None
    if hx.rank == 0 and not args.launch_per_frame and B >= D:
        vs_cpu = sequences_parity(hx, api, params, pool, variant[0], first_timed, last_timed, T_after_timed)
        assert vs_cpu is None or "note" in vs_cpu or vs_cpu["replay_reproduces_the_timed_streams_bit_for_bit"], vs_cpu

    t_max, iters_all, frames_all = reduce_over_ranks(hx.dist, hx.reduce_device, elapsed, iters_total, B * args.steps)
    per_rank = gather_per_rank(hx.dist, hx.reduce_device, elapsed, iters_total, B * args.steps)
    # the timed launch of K sequence frames builds FEWER pyramids than the single-frame launches the unit counts come from:
    # count what ran (round 3 counted two per frame: 3.8 % too many bytes)
    pyr_per_frame, swapped = (2.0, 0) if args.launch_per_frame else sequence_pyramids_per_frame(args.steps)
    per_stream = algorithmic_bytes(stats_last, levels_n, levels_n[1], True, True, pyramids_per_frame=pyr_per_frame)
    moved_per_stream = algorithmic_bytes(stats_last, levels_n, levels_n[1], True, True, pyramids_per_frame=pyr_per_frame, per_unit=MOVED_B)
    alg_bytes_launch = sum(per_stream.values()) * B / float(len(stats_last))
    kms = float(np.mean(k_ms))
    achieved = alg_bytes_launch / (kms * 1e-3) / 1e9
    # the in-launch advance (prediction := current, current := pool frame). Round 5: a frame that swaps its pyramid buffers reads
    # level 0 of both images IN THE POOL and copies nothing; a frame that does not swap (the first of a launch, the last when
    # it has to leave the host's layout) copies both images, 8 B read + 8 B written per pixel each; a last frame that did swap
    # leaves both images in the buffers afterwards (the same 32 B). Its own line: no SURVEY figure covers it and it is NOT part
    # of `achieved`. (SF_NO_POOL_IN_PLACE=1: round 4's form, 16 B per pixel for every swapping frame.)
    in_place = not (os.environ.get("SF_NO_POOL_IN_PLACE") or os.environ.get("SF_NO_PYRAMID_FLIP"))
    if args.launch_per_frame:
        advance_copy = 0.0
    elif in_place:
        last_swapped = args.steps >= 2 and (args.steps - 2) % 2 == 1
        advance_copy = n0 * (32.0 * (args.steps - swapped) + (32.0 if last_swapped else 0.0)) / args.steps
    else:
        advance_copy = n0 * (16.0 * swapped + 32.0 * (args.steps - swapped)) / args.steps
    traffic, why = measured_traffic(pool["workload"], B, variant[0], frames_per_launch=1 if args.launch_per_frame else args.steps)
    if traffic:
        traffic["ratio_to_algorithmic"] = traffic["hbm_bytes_per_launch"] / alg_bytes_launch
    cfg = {"workload": WORKLOAD_TEXT[pool["workload"]], "streams_per_gpu": B}
    cfg.update(pool["what"])
    return {
        "workload": pool["workload"],
        "value": iters_all / t_max,
        "frames_per_s": frames_all / t_max,
        "ms_per_step": 1e3 * t_max / args.steps,
        "iterations_per_frame": iters_total / float(B * args.steps),
        "iterations_per_frame_spread": [int(n_irls.min()), int(n_irls.max())],
        "parity": ({"tracking_error_vs_ground_truth": {"rot_rad_max": max(e[0] for e in err), "trans_m_max": max(e[1] for e in err), "streams": len(err)},
                    "pose_delta_vs_cpu": vs_cpu} if err else None),
        "per_rank": [{"rank": r, "elapsed_s": e, "iterations_per_s": i / e, "frames_per_s": f / e} for r, (e, i, f) in enumerate(per_rank)],
        "config": dict(cfg, **{
            "rows": rows, "cols": cols, "ctf_levels": levels,
            "max_iter_irls": int(params.max_iter_irls), "max_iter_per_level": int(params.max_iter_per_level),
            "kernel_build": "%s (%d threads per workgroup, %d workgroup(s) per stream)" % variant + ", %d workgroups resident per CU (grid %d)" % resident,
            "parallelism": "independent sequences, %d GPU(s)" % hx.world,
            "step": "prediction := current, current := next frame from the HBM pool, then the frame sequence of sf_process_frame, for every stream; "
                    + ("one launch per step" if args.launch_per_frame else "the K timed steps are ONE launch of sf_frame_kernel (sf_process_sequence_frames_device): "
                       "a stream starts its next frame when ITS previous one is done"),
            "launches_in_timed_region": args.steps if args.launch_per_frame else 1,
        }),
        "roofline": {
            "kernel": "sf_frame_kernel (%s build)" % variant[0],
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "copy_gbs": copy_gbs, "frac_of_copy": (achieved / copy_gbs) if copy_gbs else None,
            # the shader clock the timed stream-frames ran at (in-kernel: clock64 over the 100 MHz wall clock): the package's
            # power management restarts low at every launch and climbs for several hundred ms
            "shader_clock_mhz": shader_mhz,
            "traffic": traffic["hbm_bytes_per_launch"] * (1 if args.launch_per_frame else args.steps) if traffic else None,
            "traffic_provenance": traffic, "traffic_note": why,
            "frames_per_launch": 1 if args.launch_per_frame else args.steps,
            "algorithmic_bytes_per_launch": alg_bytes_launch * (1 if args.launch_per_frame else args.steps),
            "kernel_launch_ms": kms * (1 if args.launch_per_frame else args.steps),
            "algorithmic_bytes_per_step": alg_bytes_launch,
            "algorithmic_bytes_breakdown_per_stream": {k: v / float(len(stats_last)) for k, v in per_stream.items()},
            "moved_bytes_breakdown_per_stream": {k: v / float(len(stats_last)) for k, v in moved_per_stream.items()},
            "bytes_per_unit": {"algorithmic": ALGORITHMIC_B, "moved": MOVED_B},
            "pyramids_built_per_frame": pyr_per_frame, "frames_that_swapped_pyramid_buffers": swapped,
            "advance_copy_bytes_per_stream_frame": advance_copy,  # in the launch and in `traffic`, not in `achieved`
            "level0_read_in_pool": bool(in_place and not args.launch_per_frame),
            "kernel_ms_avg": kms,
        },
        "pairs": None,
    }


def main():
    t_main = time.time()
    args = parse()
    if args.launch_per_frame:
        os.environ["SF_TIMED_LAUNCH_PER_FRAME"] = "1"  # sf_timed_process_frames honours it (the pairs workloads)
    self_launch_if_needed(args)  # `python bench.py --gpus N` directly: becomes N ranks under torch.distributed.run, or exits 2
    hx = Harness(args)
    want_parity = not args.no_cpu_baseline
    seq_blocks = []
    if args.workload in ("sequences", "tum"):
        pool = synthetic_sequence_pool(hx, args) if args.workload == "sequences" else tum_sequence_pool(hx, args)
        blk = run_sequences_workload(hx, args, args.batch, pool)
        del pool
    else:
        blk = run_pairs_workload(hx, args, args.workload, args.batch, want_parity)
    full = None
    if args.workload == "static" and not args.no_full_solver:
        full = run_pairs_workload(hx, args, "sphere", args.batch, want_parity)
    if args.workload == "static" and not args.no_sequences:
        # the realistic shape of the many-streams use: unequal streams (8 ... 66 IRLS iterations per frame), at a batch
        # that is many rounds of the resident workgroups and at one that is only three (the tail of a launch shows)
        pool = synthetic_sequence_pool(hx, args)
        for nb in [int(x) for x in args.seq_batches.split(",") if x]:
            seq_blocks.append(run_sequences_workload(hx, args, min(nb, args.batch), pool))
        del pool

    if hx.rank == 0:
        # the line reports exactly the ranks that ran: one per GPU asked for, each on its own device
        assert hx.world == args.gpus == len(blk["per_rank"]) == len(hx.devices), (hx.world, args.gpus, len(blk["per_rank"]), len(hx.devices))
        dev_keys = {(d["host"], d["uuid"] or d["pci_bus_id"], d["device_index"]) for d in hx.devices}
        assert len(dev_keys) == hx.world or os.environ.get("SF_BENCH_SINGLE_GPU"), "two ranks share a GPU: %r" % (hx.devices,)
        out = {
            "metric": "solver iterations/s (IRLS loop bodies, reference FrontEnd.cpp:611-684) at QVGA",
            "value": blk["value"],
            "unit": "iterations/s",
            "frames_per_s": blk["frames_per_s"],
            "n_gpus": args.gpus,
            "world_size": hx.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": blk["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": blk["config"],
            "iterations_per_frame": blk["iterations_per_frame"],
            "pixel_iterations_per_s": blk.get("pixel_iterations_per_s"),
            "pose_delta_vs_cpu": blk["parity"],
            "per_rank": blk["per_rank"],
            "devices": hx.devices,
            "roofline": blk["roofline"],
            "build": {"head": git_head(), "src_sha": source_sha()},
        }
        out["configs_unavailable"] = CONFIGS_UNAVAILABLE
        if "iterations_per_frame_spread" in blk:
            out["iterations_per_frame_spread"] = blk["iterations_per_frame_spread"]
        if seq_blocks:
            out["sequences"] = [{
                "workload": q["config"]["workload"], "streams_per_gpu": q["config"]["streams_per_gpu"],
                "value": q["value"], "unit": "iterations/s", "frames_per_s": q["frames_per_s"], "ms_per_step": q["ms_per_step"],
                "steps": args.steps, "warmup": args.warmup, "iterations_per_frame": q["iterations_per_frame"],
                "iterations_per_frame_spread": q["iterations_per_frame_spread"], "tracking": q["parity"],
                "pose_delta_vs_cpu": (q["parity"] or {}).get("pose_delta_vs_cpu"),
                "config": q["config"], "roofline": q["roofline"], "per_rank": q["per_rank"],
            } for q in seq_blocks]
        if full is not None:
            out["full_solver"] = {
                "workload": full["config"]["workload"],
                "value": full["value"], "unit": "iterations/s", "frames_per_s": full["frames_per_s"], "ms_per_step": full["ms_per_step"],
                "steps": args.steps, "warmup": args.warmup,
                "iterations_per_frame": full["iterations_per_frame"], "pose_delta_vs_cpu": full["parity"],
                "config": full["config"], "roofline": full["roofline"], "per_rank": full["per_rank"],
            }
        if not args.no_cpu_baseline and hx.world == 1 and args.workload in ("static", "sphere"):  # rank 0 at N = 1 only
            lib = native_oracle()
            out["cpu_baseline"] = cpu_baseline(args.workload, blk["pairs"], args.cpu_seconds, lib)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.workload, min(args.cpu_seconds, 8.0), lib)
            if full is not None:
                out["full_solver"]["cpu_baseline"] = cpu_baseline("sphere", full["pairs"], min(args.cpu_seconds, 8.0), lib)
        out["wall_s"] = round(time.time() - t_main, 1)  # this process, argument parsing to this line (set-up, every block, the CPU legs)
        print(json.dumps(out))
    hx.close()


# ---------------------------------------------------------------------------------------------------------------------
#  CPU comparator: the oracle (kind "port": the reference itself cannot be built here), timed on the GPU box's host
# ---------------------------------------------------------------------------------------------------------------------
def native_oracle():
    """The reference is built -O3 -msse2 -msse3 -mtune=native (reference CMakeLists.txt:100-105). The committed oracle
    library is tuned for the build container; for the timing leg the same sources are compiled once more ON THIS HOST with
    -mtune=native (a few seconds, /tmp). Falls back to the shipped library if no compiler is present."""
    from oracle import binding

    out = "/tmp/sf_oracle_native_%d/liboracle.so" % os.getuid()
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        src = os.path.join(ROOT, "oracle")
        files = [os.path.join(src, f) for f in sorted(os.listdir(src)) if f.endswith(".cpp")]
        subprocess.check_call(["g++", "-O3", "-msse2", "-msse3", "-mtune=native", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
                               "-o", out] + files, stderr=subprocess.DEVNULL, timeout=180)
        return out
    except Exception:
        return binding.LIB


def _cpu_leg(job):
    """One pinned process: `seconds` of the workload on one stream. Returns (iterations, frames, elapsed)."""
    workload, seconds, seed, lib, cpu = job
    if cpu is not None:
        os.sched_setaffinity(0, {cpu})
    import staticfusion_amd as sf
    from staticfusion_amd.synth import make_batch

    ora = sf.Api(lib, "sfo_")
    pairs = make_batch(1, base_seed=1234 + seed, sphere=(workload == "sphere"), distinct=1)
    s = sf.Solver(ora, 240, 320, 1, make_params(ora, workload))
    s.set_current(0, *pairs[0]["new"])
    s.set_prediction(0, *pairs[0]["old"])
    for im in range(5):
        s.process_frame(im)
    im, iters, frames = 5, 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        s.process_frame(im)
        im += 1
        frames += 1
        iters += s.stats(0).n_irls
    return iters, frames, time.perf_counter() - t0


def cpu_baseline(workload, pairs, seconds, lib):
    """One host core (pinned), a bounded sample of the same workload."""
    import staticfusion_amd as sf

    allowed = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, {allowed[-1]})
    try:
        ora = sf.Api(lib, "sfo_")
        n = min(4, len(pairs))
        s = sf.Solver(ora, 240, 320, n, make_params(ora, workload))
        for b in range(n):
            s.set_current(b, *pairs[b]["new"])
            s.set_prediction(b, *pairs[b]["old"])
        for im in range(5):
            s.process_frame(im)
        im, iters, frames = 5, 0, 0
        t0 = time.perf_counter()
        while True:
            s.process_frame(im)
            im += 1
            _, n_irls, _, _ = s.batch_results()
            iters += int(n_irls.sum())
            frames += n
            dt = time.perf_counter() - t0
            if dt > seconds:
                break
    finally:
        os.sched_setaffinity(0, set(allowed))
    return {
        "value": iters / dt, "unit": "iterations/s", "frames_per_s": frames / dt, "cores": 1, "kind": "port",
        "sample": "%d frames of the same workload (%d distinct pairs, repeated), single thread pinned to CPU %d, %.1f s" % (frames, n, allowed[-1], dt),
        "build": "g++ -O3 -msse2 -msse3 -mtune=native -ffp-contract=off on this host" if lib.startswith("/tmp/") else "shipped liboracle.so (no compiler on this host)",
        "host_cpus": os.cpu_count(), "usable_cpus": len(allowed),
    }


def cgroup_cpu_limit():
    """CPUs the container may use at once (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        return None if quota == "max" else float(quota) / float(period)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else float(q) / p
    except Exception:
        return None


def cpu_baseline_all_cores(workload, seconds, lib):
    """Embarrassingly parallel: one independent stream per CPU this container may use, one process each.
    The GPU boxes expose all host threads in the affinity mask but cap the container's CPU time (cpu.max): more
    processes than the cap only contend, so the leg runs min(affinity, cap) processes, pinned when there is no cap."""
    import multiprocessing as mp

    allowed = sorted(os.sched_getaffinity(0))
    cap = cgroup_cpu_limit()
    n = len(allowed) if cap is None else max(1, min(len(allowed), int(cap)))
    pin = cap is None  # under a quota the scheduler places the processes; pinning 16 of 256 threads would pick SMT siblings
    with mp.get_context("spawn").Pool(n) as pool:
        res = pool.map(_cpu_leg, [(workload, seconds, k % 8, lib, allowed[k] if pin else None) for k in range(n)])
    return {
        "value": sum(r[0] / r[2] for r in res), "unit": "iterations/s", "frames_per_s": sum(r[1] / r[2] for r in res),
        "cores": n, "kind": "port",
        "sample": "one independent stream per process, %d processes (%s), %.1f s each" % (n, "pinned" if pin else "container CPU quota %.1f" % cap, seconds),
        "host_cpus": os.cpu_count(), "usable_cpus": len(allowed), "cgroup_cpu_limit": cap,
    }


if __name__ == "__main__":
    main()
