#!/usr/bin/env python3
"""bench.py — throughput of the StaticFusion solver hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
For N > 1 it is launched under torch.distributed.run, one rank per GPU (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment); rank 0 prints ONE JSON line.

A "step" is ONE pass of the hot path over ONE batch of synthetic RGB-D pairs that is already
resident in HBM: the reference drivers' per-frame sequence
    createImagePyramid(true); runSolver(true); computeResidualsAgainstPreviousImage; buildSegmImage
(reference StaticFusion-datasets.cpp:171-184) for every stream of the batch = one launch of
`sf_frame_kernel` through the C ABI (sf_process_frame).

Workload (default): BASELINE.json configs[1] — synthetic static RGB-D pairs (VGA render decimated
to QVGA 320x240), 3-level pyramid, segmentation disabled (pure 6-DoF Cauchy IRLS), constructor
parameters (max_iter_irls 10, delta 1e-6, no motion filter).  `--workload sphere` runs configs[2]
(moving sphere, full solver, K-means 24, b-field, driver parameters, 5 levels).

metric  = solver iterations/s: executions of the IRLS loop body (reference FrontEnd.cpp:611-684)
          summed over all streams, steps and GPUs, divided by the wall time of the K timed steps
          (max over ranks).  frames/s is reported beside it.
Multi-GPU: independent streams per GPU (no data-path collective), weak scaling.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16384, help="independent streams per GPU (8.3 MB of HBM each: 136 GB); 16 rounds of the 1024 resident workgroups")
    ap.add_argument("--workload", choices=["static", "sphere"], default="static")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic pairs (tiled over the batch)")
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


def algorithmic_bytes(stats_list, levels_n, n_l1, seg, with_residuals):
    """SURVEY.md §8(d) per-unit figures x the units one launch processed."""
    total = {"irls": 0, "linearise": 0, "warp": 0, "pyramid": 0, "kmeans": 0, "segm_image": 0, "residuals": 0}
    n_levels = len(levels_n)
    for st in stats_list:
        total["irls"] += 60 * int(st.pixel_iters)
        for i in range(st.n_outer):
            L = n_levels - 1 - st.outer[i].level  # image level
            total["linearise"] += 88 * levels_n[L]
            if not (st.outer[i].level == 0 and st.outer[i].k == 0):
                total["warp"] += 32 * levels_n[L]
        total["pyramid"] += 2 * 48 * sum(levels_n[1:])
        if seg:
            total["kmeans"] += 20 * n_l1 * int(st.kmeans_iters)
        total["segm_image"] += 8 * levels_n[0]
        if with_residuals:
            total["residuals"] += 32 * levels_n[0]
    return total


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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch  # plumbing only: device sync + torch.distributed (RCCL) barrier / reductions

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # Test hooks (tests/test_gpu_parity_sweep.py runs the N > 1 flow on a one-GPU box): SF_BENCH_BACKEND=gloo does the
    # barrier / reductions over gloo on CPU tensors, SF_BENCH_SINGLE_GPU=1 puts every rank on cuda:0. The driver's runs
    # use neither: one rank per GPU, RCCL ("nccl").
    backend = os.environ.get("SF_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("SF_BENCH_SINGLE_GPU") else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    reduce_device = torch.device("cuda", dev_index) if backend == "nccl" else torch.device("cpu")

    import staticfusion_amd as sf
    from staticfusion_amd.synth import make_batch, pose_delta

    api = sf.load()
    params = make_params(api, args.workload)
    rows, cols, B = 240, 320, args.batch
    sphere = args.workload == "sphere"
    pairs = make_batch(args.distinct, base_seed=1234 + 100000 * rank, sphere=sphere, distinct=args.distinct)

    solver = sf.Solver(api, rows, cols, B, params, device=dev_index)
    for b in range(B):
        pr = pairs[b % len(pairs)]
        solver.set_current(b, *pr["new"])
        solver.set_prediction(b, *pr["old"])

    # ---- parity of the first frame against the CPU oracle (rank 0, the distinct pairs only)
    parity = None
    solver.process_frame(0)
    solver.synchronize()
    T_first, _, _, _ = solver.batch_results()
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import binding  # the checker: never on the measured path

        ora = binding.load()
        po = make_params(ora, args.workload)
        rot_max = trans_max = 0.0
        ncheck = min(4, len(pairs))
        osolver = sf.Solver(ora, rows, cols, ncheck, po)
        for b in range(ncheck):
            osolver.set_current(b, *pairs[b]["new"])
            osolver.set_prediction(b, *pairs[b]["old"])
        osolver.process_frame(0)
        for b in range(ncheck):
            r, t = pose_delta(osolver.T(b), T_first[b])
            rot_max, trans_max = max(rot_max, r), max(trans_max, t)
        parity = {"rot_rad": rot_max, "trans_m": trans_max, "frames_checked": ncheck}

    # ---- prime the 5-frame history so that the timed frames include computeResiduals
    for im in range(1, 5):
        solver.process_frame(im)
    im = 5
    for _ in range(args.warmup):
        solver.process_frame(im)
        im += 1
    solver.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        solver.synchronize()

    # ---- the timed region: exactly K steps = K launches of sf_frame_kernel, back to back, bracketed by
    #      barrier + device synchronisation.  HIP events recorded on the handle's stream around the
    #      same K launches give the average kernel duration; the device-side counters give the
    #      number of IRLS iterations those K steps executed (read outside the region).
    c0 = solver.counters()
    barrier()
    t0 = time.perf_counter()
    region_ms = solver.timed_process_frames(im, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    im += args.steps
    c1 = solver.counters()
    frames_timed = c1[0] - c0[0]
    iters_total = c1[1] - c0[1]
    pix_total = c1[3] - c0[3]
    assert frames_timed == B * args.steps, (frames_timed, B, args.steps)
    kernel_ms = [region_ms / args.steps]
    # per-stage unit counts of ONE launch (for the algorithmic byte count): one more, un-timed step
    solver.process_frame(im)
    im += 1
    stats_last = [solver.stats(b) for b in range(min(B, args.distinct))]

    t_max, iters_all, frames_all = reduce_over_ranks(dist, reduce_device, elapsed, iters_total, B * args.steps)

    if rank == 0:
        levels_n = [solver.level_shape(L)[0] * solver.level_shape(L)[1] for L in range(solver.levels)]
        seg = bool(params.segmentation_enabled)
        per_stream = algorithmic_bytes(stats_last, levels_n, levels_n[1], seg, True)
        scale = B / float(len(stats_last))
        alg_bytes_launch = sum(per_stream.values()) * scale
        k_ms = float(np.mean(kernel_ms))
        achieved = alg_bytes_launch / (k_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic_%s_b%d.json" % (args.workload, B))
        if os.path.exists(tfile):
            with open(tfile) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": "solver iterations/s (IRLS loop bodies, reference FrontEnd.cpp:611-684) at QVGA",
            "value": iters_all / t_max,
            "unit": "iterations/s",
            "frames_per_s": frames_all / t_max,
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("BASELINE.json configs[1]: synthetic static RGB-D pairs, VGA->QVGA 320x240, 3-level pyramid, "
                             "segmentation disabled (pure 6-DoF Cauchy IRLS)") if not sphere else
                            ("BASELINE.json configs[2]: synthetic QVGA pairs with a moving sphere, full solver, "
                             "K-means(24) + b-field, driver parameters, 5 levels"),
                "streams_per_gpu": B,
                "distinct_pairs": len(pairs),
                "rows": rows, "cols": cols, "ctf_levels": int(solver.levels),
                "max_iter_irls": int(params.max_iter_irls), "max_iter_per_level": int(params.max_iter_per_level),
                "parallelism": "independent streams, %d GPU(s), one workgroup per stream" % args.gpus,
                "step": "sf_process_frame: pyramid(old)+runSolver(true)+residuals+segm image for every stream, one launch",
            },
            "iterations_per_frame": iters_total / float(B * args.steps),
            "pixel_iterations_per_s": pix_total / elapsed * args.gpus,
            "pose_delta_vs_cpu": parity,
            "roofline": {
                "kernel": "sf_frame_kernel",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes_launch,
                "algorithmic_bytes_breakdown_per_stream": {k: v / float(len(stats_last)) for k, v in per_stream.items()},
                "kernel_ms_avg": k_ms,
            },
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args, pairs)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, pairs):
    """The CPU restatement (oracle/, kind 'port': the reference itself cannot be built here) on one
    host core, on a bounded sample of the same workload."""
    import staticfusion_amd as sf
    from oracle import binding

    ora = binding.load()
    p = make_params(ora, args.workload)
    n = min(4, len(pairs))
    s = sf.Solver(ora, 240, 320, n, p)
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
        if dt > args.cpu_seconds:
            break
    return {
        "value": iters / dt,
        "unit": "iterations/s",
        "frames_per_s": frames / dt,
        "cores": 1,
        "kind": "port",
        "sample": "%d frames of the same workload (%d distinct pairs, repeated), single thread, %.1f s" % (frames, n, dt),
        "host_cpus": os.cpu_count(),
    }


if __name__ == "__main__":
    main()
