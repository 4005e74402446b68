#!/usr/bin/env python3
"""Benchmark of the RANSAC pose-hypothesis hot path on MI355X.

Workload (BASELINE.json configs[1]): one synthetic query with 4 000 2D-3D matches, 10 %
inliers, unknown focal (P4Pf), 1 M hypotheses per step.  One step = sample 1 M minimal
4-tuples on the device, solve P4Pf for each, score every returned pose against all 4 000
correspondences, select the best pose.  Inputs are resident in HBM before the timed region.

N > 1 GPUs (launched by torch.distributed.run, one rank per GPU): the hypothesis range is
sharded, every rank evaluates its own 1 M hypotheses of the same query per step (weak
scaling) and the packed (inlier count, ~hypothesis id) keys are max-all-reduced over RCCL.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X fp64 vector peak: 256 CU x 4 SIMD x 16 FMA lanes/clk x 2 x 2.4 GHz
HBM_PEAK_GBS = 8000.0
# Algorithmic fp64 work per unit of each kernel (derivations in DESIGN.md):
SCORE_FLOP_PER_PAIR = 27           # reference formula per (pose, correspondence): 3x4 product, 2 div, distance, compare
ELIM_FLOP_PER_HYP = 2.34e4         # sparsity-exploiting elimination of the 78x88 template (22 920) + coefficient formulas
FINISH_FLOP_PER_HYP = 8.9e4        # executed fp64 lane-operations per hypothesis, from SQ_INSTS_VALU_*_F64 (profiles/)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--hyps", type=int, default=1 << 20, help="hypotheses per GPU per step")
    ap.add_argument("--matches", type=int, default=4000)
    ap.add_argument("--inlier-frac", type=float, default=0.10)
    ap.add_argument("--cpu-hyps", type=int, default=0, help="CPU baseline sample size (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.gpus > 1:
        print("bench.py: --gpus > 1 must be launched through torch.distributed.run", file=sys.stderr)
        return 2

    from pose_estimation_amd import Context, shard, synth, SELECT_INLIERS, SAMPLE_BASIC
    ctx = Context(local_rank)
    q = synth.make_query(args.matches, args.inlier_frac, seed=2)
    ctx.set_query(q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], 0.0)
    H = args.hyps

    def step(i):
        # hypothesis ids are global: rank r owns [ (i*world + r) * H, +H )
        ctx.sample(SAMPLE_BASIC, 1002, 0, ((i * world + rank) * H) % (1 << 31), H, export=False)
        r = ctx.ransac_run(SELECT_INLIERS, 0.5, early_exit=False)
        if world > 1:
            # the only exchange of the hypothesis-sharded path: one packed (count, id) key
            gh = ((i * world + rank) * H + r["best_hyp"]) & shard.KEY_ID_MASK
            r["global_best"] = shard.unpack_key(shard.reduce_best(shard.pack_key(r["inliers"], gh), dist, f"cuda:{local_rank}"))
        return r

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    tim = {}
    poses = 0
    last = None
    for i in range(args.steps):
        last = step(args.warmup + i)
        poses += last["poses_scored"]
        for k, v in ctx.last_timing().items():
            tim[k] = tim.get(k, 0.0) + v
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_hyps = H * args.steps * world
        info = ctx.device_info()
        # Roofline of the dominant kernel of the step.  Durations are HIP-event times taken by the
        # library on its own stream around each kernel, averaged over the launches of the timed steps.
        launches = max(int(tim["launches"]), 1)
        hyps_rank = H * args.steps
        work = {"k_p4pf_elim": (tim["elim_ms"], hyps_rank * ELIM_FLOP_PER_HYP),
                "k_p4pf_finish": (tim["finish_ms"], hyps_rank * FINISH_FLOP_PER_HYP),
                "k_score": (tim["score_ms"], poses * args.matches * SCORE_FLOP_PER_PAIR)}
        dom = max(work, key=lambda k: work[k][0])
        dom_ms, dom_flop = work[dom]
        tflops = dom_flop / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        roofline = {"kernel": dom, "bound": "valu_fp64", "achieved": round(tflops, 3), "peak": FP64_VECTOR_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(tflops / FP64_VECTOR_PEAK_TFLOPS, 4), "traffic": recorded_traffic(dom),
                    "launches": launches, "avg_launch_ms": round(dom_ms / launches, 4),
                    "kernel_ms_per_step": {k: round(v / args.steps, 3) for k, v in tim.items() if k.endswith("_ms")},
                    "kernel_tflops": {k: round(f / (ms * 1e-3) / 1e12, 3) for k, (ms, f) in work.items() if ms > 0},
                    "poses_per_hypothesis": round(poses / hyps_rank, 3)}
        out = {"metric": "RANSAC hypotheses/sec (P4Pf solve+score)", "value": round(total_hyps / dt, 1), "unit": "hypotheses/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE configs[1]: P4Pf RANSAC, single query, %d matches, %d%% inliers, %d hypotheses per GPU per step"
                          % (args.matches, round(args.inlier_frac * 100), H),
                          "select": "inlier count", "parallelism": "hypothesis range sharded over %d GPU(s)" % world,
                          "device": info["name"], "best_inliers": int(last["inliers"])},
               "roofline": roofline}
        if world == 1:
            # the same step with the host->device hand-over of the query inside the timed region
            t1 = time.perf_counter()
            for i in range(2):
                ctx.set_query(q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], 0.0)
                step(args.warmup + args.steps + i)
            ctx.synchronize()
            out["config"]["pcie_inclusive_hypotheses_per_s"] = round(2 * H / (time.perf_counter() - t1), 1)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ctx, q, args)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


def recorded_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed counter passes (FETCH_SIZE doubled
    as measured for gfx950, plus WRITE_SIZE), or None when no record exists for it."""
    path = os.path.join(ROOT, "profiles", "r1_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def cpu_baseline(ctx, q, args):
    """The CPU oracle (a port of the reference's algorithm; the reference itself needs Eigen,
    FLANN and OpenCV, which are not installed) on a bounded sample of the same workload: the
    first `count` hypotheses of the benchmark's sample stream, OpenMP over hypotheses."""
    import oracle_lib as O
    from pose_estimation_amd import SAMPLE_BASIC
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")), 16)
    kw = (q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], 0.0)
    probe = ctx.sample(SAMPLE_BASIC, 1002, 0, 0, 4096)
    probe[:, 3] = np.where(probe[:, 3] == 0xFFFFFFFF, 0, probe[:, 3])
    t0 = time.perf_counter()
    O.ransac_fixed_budget(*kw, probe, threads=cores)
    per = (time.perf_counter() - t0) / len(probe)
    count = args.cpu_hyps or int(min(max(15.0 / max(per, 1e-8), 4096), 1 << 22))
    samples = ctx.sample(SAMPLE_BASIC, 1002, 0, 0, count)
    samples[:, 3] = np.where(samples[:, 3] == 0xFFFFFFFF, 0, samples[:, 3])
    t0 = time.perf_counter()
    O.ransac_fixed_budget(*kw, samples, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": round(count / dt, 1), "unit": "hypotheses/s", "cores": cores, "kind": "port",
            "sample": "first %d hypotheses of the same query and sample stream (tuples exported by the device sampler), "
                      "oracle/pose_oracle.c with OpenMP over hypotheses, %.1f s" % (count, dt)}


if __name__ == "__main__":
    sys.exit(main())
