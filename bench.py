#!/usr/bin/env python3
"""Benchmark of the RANSAC pose-hypothesis hot path on MI355X.

Default workload (BASELINE.json configs[1]): one synthetic query with 4 000 2D-3D matches, 10 %
inliers, unknown focal (P4Pf), 1 M hypotheses per step.  One step = sample 1 M minimal 4-tuples
on the device, solve P4Pf for each, score every returned pose against all 4 000 correspondences,
select the best pose.  Inputs are resident in HBM before the timed region.

Other workloads (one JSON line each, same fields): --known-focal / --matches / --inlier-frac
reshape the single-query run (configs[0] and configs[4] shapes); --workload dubrovnik runs the
800-query batch of configs[2] (fixed budget, or --early-exit for the reference's loop) and
configs[3] (--advanced).  After the timed region of the default workload, short passes of those other
configs run as well and are reported under config.other_workloads of the same line (--no-extras
skips them), so that one run carries every BASELINE config.

N > 1 GPUs: one rank per GPU.  Launched by torch.distributed.run the ranks find RANK / WORLD_SIZE
in the environment; started plainly as `python bench.py --gpus N`, this script starts the N
ranks itself as child processes (before anything touches the GPU) and relays rank 0's line.
The single-query workload shards the hypothesis range (every rank evaluates its own 1 M
hypotheses of the same query per step, weak scaling) and max-all-reduces the packed
(inlier count, ~hypothesis id) keys over RCCL; the batch workload gives every rank its own list of
queries (--scaling weak, the default) or partitions one list over the ranks (--scaling strong), with
no collective in the data path.  --gpus 1 --rehearse-collectives runs that path's collectives with one
rank (the process group over RCCL, the per-step reduction, barriers, gathers of the results).

Prints ONE JSON line on rank 0, and nothing else on standard output: file descriptor 1 is pointed
at the error stream until that line is due (RCCL prints a banner there when a group is set up).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X fp64 vector peak: 256 CU x 4 SIMD x 16 FMA lanes/clk x 2 x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 matrix peak
HBM_PEAK_GBS = 8000.0
# Algorithmic fp64 work per unit of each kernel:
SCORE_FLOP_PER_PAIR = 27           # reference formula per (pose, correspondence): 3x4 product, 2 div, distance, compare
ELIM_FLOP_PER_HYP = 2.28e4         # sparsity-exploiting elimination of the 78x88 template down to the 5x15 tail (22 285) + coefficient formulas
P3P_FLOP_PER_HYP = 5.0e2           # Kneip P3P + Ferrari quartic per sample
FINISH_FLOP_PER_HYP = 1.91e4       # executed fp64 lane-operations per hypothesis of the eigen step / pose recovery (counters; no closed-form count)
SCREEN_MFMA_FLOP_PER_PAIR = 96     # what the matrix-core screen executes: three v_mfma_f32_32x32x16_bf16 per 32 x 32 pairs


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--hyps", type=int, default=1 << 20, help="hypotheses per GPU per step")
    ap.add_argument("--matches", type=int, default=4000)
    ap.add_argument("--inlier-frac", type=float, default=0.10)
    ap.add_argument("--cpu-hyps", type=int, default=0, help="CPU baseline sample size (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--known-focal", action="store_true", help="single: P3P instead of P4Pf (BASELINE configs[0] shape with --matches 2000 --inlier-frac 0.3)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend; gloo (CPU tensors) lets several ranks rehearse on one GPU")
    ap.add_argument("--workload", choices=["single", "dubrovnik", "knn"], default="single",
                    help="single: BASELINE configs[1] (default, the headline line); dubrovnik: configs[2], 800 queries sharded by query; "
                         "knn: the descriptor matching in front of the path (2-NN + ratio test of --queries descriptors against --points model points)")
    ap.add_argument("--points", type=int, default=1 << 20, help="knn: model points")
    ap.add_argument("--queries", type=int, default=800)
    ap.add_argument("--budget", type=int, default=10000, help="dubrovnik: hypotheses per query")
    ap.add_argument("--early-exit", action="store_true", help="dubrovnik: the reference's loop (coverage quality, stop at 12 inliers)")
    ap.add_argument("--advanced", action="store_true",
                    help="dubrovnik: configs[3], the advanced processor (co-occurrence sampling over camera sets, 100 steps, no early exit)")
    ap.add_argument("--host-queries", action="store_true", help="dubrovnik: every pass hands the queries over as host arrays (pose_amd_ransac_batch: "
                                                                "packing and the PCIe copy inside the timed region) instead of running a list resident in HBM")
    ap.add_argument("--python-threads", action="store_true", help="dubrovnik: drive the contexts from Python threads instead of pose_amd_ransac_batch")
    ap.add_argument("--host-threads", type=int, default=16, help="dubrovnik: contexts (streams) per GPU driven concurrently")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="dubrovnik with --gpus N: weak = every GPU runs its own list of --queries queries (default); strong = one list of "
                         "--queries queries partitioned over the GPUs")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="with --gpus 1: initialise the process group all the same and run every collective of the N > 1 path "
                         "with one rank (on a one-GPU box this is the only way RCCL's part of the path can be run)")
    ap.add_argument("--no-extras", action="store_true",
                    help="default workload only: skip the short passes of the other BASELINE configs that fill config.other_workloads")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children through
    torch.distributed.run (nothing in this process has touched the GPU), relay rank 0's JSON line,
    return the children's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [l for l in p.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1])
    return p.returncode


def setup_ranks(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev
    red_dev = f"cuda:{dev_index}" if args.backend == "nccl" else "cpu"
    if world > 1 or args.rehearse_collectives:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        kw = {}
        if "RANK" not in os.environ:                 # the one-rank rehearsal outside torch.distributed.run
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            kw = dict(rank=0, world_size=1)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), **kw)
        else:
            dist.init_process_group("gloo", **kw)
    return rank, world, dev_index, red_dev, torch, dist


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    return min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")), 16)


class KeyReducer:
    """The one exchange of the hypothesis-sharded path: a max-all-reduce of one packed (count, id)
    key per step, started without waiting and collected before the clock stops, so that it runs
    beside the next step's kernels.  The keys live in a ring of preallocated slots of one device
    tensor (a slot is filled in place; nothing is allocated inside the timed region).  A step's
    reduction is enqueued by a helper thread (always the same one, so every rank enqueues in step
    order): filling the slot and handing the collective to RCCL is ~0.1 ms of host time, which would
    otherwise sit between two steps with the GPU idle -- the library call of the next step releases
    the interpreter lock while it runs."""
    SLOTS = 8

    def __init__(self, torch, dist, device):
        self.dist = dist
        self.buf = torch.zeros(self.SLOTS, dtype=torch.int64, device=device) if dist is not None else None
        self.views = [self.buf[i:i + 1] for i in range(self.SLOTS)] if dist is not None else []
        self.pending = [None] * self.SLOTS                   # futures of the enqueueing thread; their results are the collectives' handles
        self.n = 0
        self.keys = []
        self.pool = None
        if dist is not None:
            from concurrent.futures import ThreadPoolExecutor
            on_gpu = str(device).startswith("cuda")
            self.pool = ThreadPoolExecutor(1, initializer=(lambda: torch.cuda.set_device(device)) if on_gpu else None)

    def _enqueue(self, slot, key):
        self.views[slot].fill_(key)
        return self.dist.all_reduce(self.views[slot], op=self.dist.ReduceOp.MAX, async_op=True)

    def _finish(self, slot):
        if self.pending[slot] is not None:
            self.pending[slot].result().wait()
            self.keys.append(int(self.views[slot].item()))
            self.pending[slot] = None

    def begin(self, key):
        if self.dist is None:
            self.keys.append(int(key))
            return
        slot = self.n % self.SLOTS
        self.n += 1
        self._finish(slot)                                   # the reduction that used this slot SLOTS steps ago
        self.pending[slot] = self.pool.submit(self._enqueue, slot, int(key))

    def collect(self):
        """Waits for every reduction still in flight; -> the reduced keys in step order."""
        if self.dist is not None:
            for i in range(self.SLOTS):
                self._finish((self.n + i) % self.SLOTS)
        out, self.keys = self.keys, []
        return out

    def close(self):
        if self.pool is not None:
            self.pool.shutdown(wait=True)
            self.pool = None


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    # Libraries write to file descriptor 1 behind Python's back (RCCL prints a version banner there when
    # a process group is initialised): until the one JSON line is due, descriptor 1 is the error stream.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    env = setup_ranks(args)
    rank, world, dist = env[0], env[1], env[5]
    if args.workload == "dubrovnik":
        out = bench_dubrovnik(args, env)
    elif args.workload == "knn":
        out = bench_knn(args, env)
    else:
        out = bench_single(args, env)
        if not args.no_extras and is_default_workload(args):
            extras = other_workloads(args, env)
            if rank == 0:
                out["config"]["other_workloads"] = extras
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        os.dup2(2, 1)                                # whatever the shutdown prints is no part of the line either
        dist.destroy_process_group()
    return 0


def is_default_workload(args):
    return (args.matches, args.inlier_frac, args.known_focal, args.hyps) == (4000, 0.10, False, 1 << 20)


def other_workloads(args, env):
    """Short passes of the other BASELINE configs after the timed region of the default workload,
    so that one run carries every config: value, ms per step and the dominant kernel with its
    roofline fraction.  No CPU legs; every entry is a full bench line of its own workload
    (`python bench.py <flags>` reproduces it with more steps)."""
    runs = [("configs[0]", ["--known-focal", "--matches", "2000", "--inlier-frac", "0.3", "--steps", "10", "--warmup", "2"]),
            ("configs[2] reference loop", ["--workload", "dubrovnik", "--early-exit", "--steps", "10", "--warmup", "2"]),
            ("configs[2] fixed budget", ["--workload", "dubrovnik", "--steps", "3", "--warmup", "1"]),
            ("configs[3]", ["--workload", "dubrovnik", "--advanced", "--steps", "10", "--warmup", "2"]),
            ("configs[4] shape, one GPU's share per step", ["--matches", "100000", "--inlier-frac", "0.01", "--hyps", str(1 << 21), "--steps", "2", "--warmup", "1"])]
    out = []
    for name, flags in runs:
        a = parse_args(flags + ["--gpus", str(args.gpus), "--backend", args.backend, "--no-cpu-baseline", "--no-extras"])
        a.is_extra = True
        t0 = time.perf_counter()
        line = (bench_dubrovnik if a.workload == "dubrovnik" else bench_single)(a, env)
        if env[0] == 0:
            rl = line.get("roofline") or {}
            out.append({"config": name, "flags": " ".join(flags), "metric": line["metric"], "value": line["value"], "unit": line["unit"],
                        "ms_per_step": line["ms_per_step"], "steps": line["steps"], "scaling": line["scaling"], "n_gpus": line["n_gpus"],
                        "dominant_kernel": rl.get("kernel"), "bound": rl.get("bound"), "frac": rl.get("frac"),
                        "kernel_ms_per_step": rl.get("kernel_ms_per_step"),
                        **{k: line["config"][k] for k in ("hypotheses_per_s", "mean_steps_per_query", "best_inliers") if k in line["config"]},
                        **{k: rl[k] for k in ("steps_computed", "poses_painted") if k in rl},
                        "wall_s_incl_setup": round(time.perf_counter() - t0, 2)})
    return out


def bench_single(args, env):
    """One query, hypothesis range sharded over the ranks (weak scaling: --hyps per GPU per step).
    -> the JSON line as a dict on rank 0 (None elsewhere)."""
    rank, world, dev_index, red_dev, torch, dist = env

    from pose_estimation_amd import Context, shard, synth, SELECT_INLIERS, SAMPLE_BASIC
    ctx = Context(dev_index)
    q = synth.make_query(args.matches, args.inlier_frac, seed=2, known_focal=args.known_focal)
    ctx.set_query(q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], q["focal"])
    H = args.hyps
    red = KeyReducer(torch, dist, red_dev)

    def step(i):
        # hypothesis ids are global: rank r owns [ (i*world + r) * H, +H )
        ctx.sample(SAMPLE_BASIC, 1002, 0, ((i * world + rank) * H) % (1 << 31), H, export=False)
        r = ctx.ransac_run(SELECT_INLIERS, 0.5, early_exit=False)
        if dist is not None:
            gh = ((i * world + rank) * H + r["best_hyp"]) & shard.KEY_ID_MASK
            red.begin(shard.pack_key(r["inliers"], gh))
        return r

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    red.collect()
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
    global_best = [shard.unpack_key(k) for k in red.collect()] if dist is not None else []   # every step's reduction has completed inside the timed region
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        total_hyps = H * args.steps * world
        info = ctx.device_info()
        hyps_rank = H * args.steps
        launches = max(int(tim["launches"]), 1)
        if args.known_focal:
            tim["p3p_ms"] = tim["solve_ms"]
        roofline = roofline_of(tim, hyps_rank, poses, args.matches, args.known_focal, launches, args.steps)
        solver = "P3P" if args.known_focal else "P4Pf"
        shape = ("BASELINE configs[0] shape: " if (args.matches, args.known_focal) == (2000, True) else
                 "BASELINE configs[1]: " if (args.matches, args.inlier_frac, args.known_focal) == (4000, 0.10, False) else
                 "BASELINE configs[4] shape: " if args.matches == 100000 else "")
        out = {"metric": "RANSAC hypotheses/sec (%s solve+score)" % solver, "value": round(total_hyps / dt, 1), "unit": "hypotheses/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s%s RANSAC, single query, %d matches, %g%% inliers, %d hypotheses per GPU per step"
                                      % (shape, solver, args.matches, args.inlier_frac * 100, H),
                          "select": "inlier count", "parallelism": "hypothesis range sharded over %d GPU(s)" % world,
                          "device": info["name"], "best_inliers": int(last["inliers"]),
                          **({"global_best_inliers_last_step": int(global_best[-1][0])} if global_best else {})},
               "roofline": roofline}
        if world == 1 and not getattr(args, "is_extra", False):
            # the same step with the host->device hand-over of the query inside the timed region
            t1 = time.perf_counter()
            for i in range(2):
                ctx.set_query(q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], q["focal"])
                step(args.warmup + args.steps + i)
            ctx.synchronize()
            out["config"]["pcie_inclusive_hypotheses_per_s"] = round(2 * H / (time.perf_counter() - t1), 1)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_single(ctx, q, args)
    red.collect()
    red.close()
    ctx.close()
    return out


def roofline_of(tim, hyps, poses, matches_x_poses_or_n, known_focal, launches, steps):
    """Roofline object of the dominant kernel.  `tim`: HIP-event times (ms) summed by the library
    on its own stream around each kernel over `hyps` hypotheses and `poses` scored poses;
    matches_x_poses_or_n: correspondences per query (single workload) or, for batches of queries of
    different sizes, the summed (pose, correspondence) pairs as a float."""
    pairs = matches_x_poses_or_n if isinstance(matches_x_poses_or_n, float) else poses * matches_x_poses_or_n
    maxsol = 4 if known_focal else 10
    per_launch_slots = hyps // max(launches, 1) * maxsol
    score_kernel = ("k_score_mfma" if per_launch_slots >= int(os.environ.get("POSE_AMD_MFMA_MIN_SLOTS", "8192")) and
                    os.environ.get("POSE_AMD_SCORE") not in ("packed", "fp64") else "k_score_screen")
    work = {}
    if tim.get("p3p_ms", 0) > 0:
        work["k_solve_p3p"] = (tim["p3p_ms"], tim.get("p3p_hyps", hyps) * P3P_FLOP_PER_HYP)
    if tim.get("elim_ms", 0) > 0:
        ph = tim.get("p4pf_hyps", hyps)
        work["k_p4pf_elim"] = (tim["elim_ms"], ph * ELIM_FLOP_PER_HYP)
        work["k_p4pf_finish"] = (tim["finish_ms"], ph * FINISH_FLOP_PER_HYP)
    work[score_kernel] = (tim["score_ms"], pairs * SCORE_FLOP_PER_PAIR)
    dom = max(work, key=lambda k: work[k][0])
    dom_ms, dom_flop = work[dom]
    tflops = dom_flop / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    rec = recorded_counters(dom)
    traffic = rec["hbm_bytes_per_hypothesis"] * hyps / launches if rec and "hbm_bytes_per_hypothesis" in rec else None
    roofline = {"kernel": dom, "bound": "valu_fp64", "achieved": round(tflops, 3), "peak": FP64_VECTOR_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(tflops / FP64_VECTOR_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_source": rec.get("source") if rec else None,
                # true when the device sources changed after the counter passes were collected (tools/r3_profiles.sh renews them)
                "traffic_stale": rec.get("stale") if rec else None,
                "launches": launches, "avg_launch_ms": round(dom_ms / launches, 4),
                "kernel_ms_per_step": {k: round(v / steps, 3) for k, v in tim.items() if k.endswith("_ms")},
                "kernel_tflops": {k: round(f / (ms * 1e-3) / 1e12, 3) for k, (ms, f) in work.items()
                                  if ms > 0 and k not in ("k_score_mfma", "k_p4pf_finish")},
                "kernel_executed_tflops": {k: round(f / (ms * 1e-3) / 1e12, 3) for k, (ms, f) in work.items() if ms > 0 and k == "k_p4pf_finish"},
                "poses_per_hypothesis": round(poses / max(hyps, 1), 3)}
    # executed float64 lane-operations (SQ_INSTS_VALU_*_F64 x 64 from the recorded counter passes)
    # beside the algorithmic figure: the difference is idle lanes, pivots and retired rows
    if rec and "executed_fp64_flop_per_hypothesis" in rec and dom_ms > 0:
        ex = rec["executed_fp64_flop_per_hypothesis"] * (tim.get("p4pf_hyps", hyps) if dom.startswith("k_p4pf") else hyps) / (dom_ms * 1e-3) / 1e12
        roofline["executed"] = {"tflops": round(ex, 3), "frac": round(ex / FP64_VECTOR_PEAK_TFLOPS, 4),
                                "flop_per_hypothesis": rec["executed_fp64_flop_per_hypothesis"]}
    if score_kernel == "k_score_mfma" and tim["score_ms"] > 0:
        # The matrix-core screen does not do the float64 work of the algorithmic count: it certifies
        # pairs outside with bf16 products and decides the few others in float64.  Its own roofs:
        sec = tim["score_ms"] * 1e-3
        mf = pairs * SCREEN_MFMA_FLOP_PER_PAIR / sec / 1e12
        roofline["k_score_mfma"] = {"pairs_per_s": round(pairs / sec, 1),
                                    "algorithmic_fp64_equivalent_tflops": round(pairs * SCORE_FLOP_PER_PAIR / sec / 1e12, 2),
                                    "mfma_bf16": {"achieved": round(mf, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
                                                  "unit": "TFLOP/s", "frac": round(mf / BF16_MFMA_PEAK_TFLOPS, 4)},
                                    "note": "vector pipe is its limiter (one fma + one compare per pair, queueing of uncertified pairs)"}
        if dom == "k_score_mfma":
            # the dominant kernel is the screen: state the roofline against the pipe it runs on
            roofline.update({"bound": "mfma", "achieved": round(mf, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "frac": round(mf / BF16_MFMA_PEAK_TFLOPS, 4),
                             "executed": None})
    if traffic and dom_ms > 0:
        gbps = traffic / (dom_ms / launches * 1e-3) / 1e9
        roofline["hbm"] = {"achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBS, 4)}
    return roofline


def kernel_source_hash():
    """SHA-256 over the device sources the library is built from (csrc/*.h, *.hip, in name order):
    the counter records under profiles/ carry the hash of the sources they were collected with."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pose_estimation_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.hip"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def recorded_counters(kernel):
    """Per-hypothesis HBM bytes (FETCH_SIZE doubled as measured for gfx950, plus WRITE_SIZE) and
    executed float64 lane-operations of `kernel` from the newest committed counter passes
    (profiles/r*_traffic.json, collected with rocprofv3 --pmc on this benchmark); None when no
    record exists.  Counters cannot be read inside the timed run itself."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
            n = rec["hypotheses_per_launch"]
            out = {"source": os.path.relpath(path, ROOT), "stale": rec.get("kernel_source_sha256") != kernel_source_hash()}
            if kernel in rec:
                out["hbm_bytes_per_hypothesis"] = rec[kernel]["hbm_bytes_per_launch"] / n
            ph = rec.get("_per_hypothesis", {}).get(kernel)
            if ph:
                out["executed_fp64_flop_per_hypothesis"] = ph["executed_fp64_lane_flop_per_hypothesis"]
            if len(out) > 2:
                return out
        except (OSError, ValueError, KeyError):
            continue
    return None


def cpu_baseline_single(ctx, q, args):
    """The CPU oracle (a port of the reference's algorithm; the reference itself needs Eigen,
    FLANN and OpenCV, which are not installed) on a bounded sample of the same workload: the
    first `count` hypotheses of the benchmark's sample stream, OpenMP over hypotheses."""
    import oracle_lib as O
    from pose_estimation_amd import SAMPLE_BASIC
    cores = host_cores()
    kw = (q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], q["focal"])
    probe = ctx.sample(SAMPLE_BASIC, 1002, 0, 0, 4096)
    probe[:, 3] = np.where(probe[:, 3] == 0xFFFFFFFF, 0, probe[:, 3])
    t0 = time.perf_counter()
    O.ransac_fixed_budget(*kw, probe, threads=cores)
    per = (time.perf_counter() - t0) / len(probe)
    count = args.cpu_hyps or int(min(max(12.0 / max(per, 1e-8), 4096), 1 << 22))
    samples = ctx.sample(SAMPLE_BASIC, 1002, 0, 0, count)
    samples[:, 3] = np.where(samples[:, 3] == 0xFFFFFFFF, 0, samples[:, 3])
    t0 = time.perf_counter()
    O.ransac_fixed_budget(*kw, samples, threads=cores)
    dt = time.perf_counter() - t0
    # the reference's own shape of the loop: one step after the other on one core, coverage-quality selection
    ns = min(len(samples), max(256, int(3.0 / max(per * cores, 1e-8))))
    t1 = time.perf_counter()
    O.ransac(*kw, samples[:ns], select_coverage=True, early_exit=False, max_steps=ns)
    dts = time.perf_counter() - t1
    # and as the reference executes it: 4 OpenMP threads inside projection / map painting, two h x w float maps per pose
    nf = 300
    t2 = time.perf_counter()
    O.ransac_reference_faithful(*kw, samples[:nf], early_exit=False, max_steps=nf, threads=4)
    dtf = time.perf_counter() - t2
    return {"value": round(count / dt, 1), "unit": "hypotheses/s", "cores": cores, "cpu": cpu_model(), "kind": "port",
            "sample": "first %d hypotheses of the same query and sample stream (tuples exported by the device sampler), "
                      "oracle/pose_oracle.c with OpenMP over hypotheses, %.1f s" % (count, dt),
            "serial_reference_loop": {"value": round(ns / dts, 1), "unit": "hypotheses/s", "cores": 1,
                                      "sample": "first %d hypotheses, sequential loop with coverage-quality selection, %.1f s" % (ns, dts)},
            "reference_faithful": {"value": round(nf / dtf, 1), "unit": "hypotheses/s", "cores": 4,
                                   "sample": "first %d hypotheses, sequential loop, TestHypothesis as the reference runs it (OMP_NUM_THREADS=4 inside "
                                             "projection and painting, two height x width float maps per pose), %.1f s" % (nf, dtf)}}


def bench_dubrovnik(args, env):
    """BASELINE configs[2]: 800 Dubrovnik6K-shaped synthetic queries (image sizes and known /
    unknown focal lengths of the real query list, 4 000-6 000 matches, 10-40 % inliers), basic
    RANSAC per query, queries partitioned over the GPUs (no collective in the data path).
    A step is one pass over all queries.  Each GPU is driven by a few host threads, one
    context (HIP stream) each, so that the small per-query launches of different queries overlap."""
    from concurrent.futures import ThreadPoolExecutor
    rank, world, dev_index, red_dev, torch, dist = env
    from pose_estimation_amd import Context, shard, synth, SELECT_INLIERS, SELECT_COVERAGE, SAMPLE_BASIC, SAMPLE_ADVANCED
    if args.advanced:
        args.budget, args.early_exit = 100, False
    with open(os.path.join(ROOT, "tests", "golden", "dubrovnik_queries.json")) as f:
        shapes = json.load(f)["queries"]
    # weak scaling: every rank runs its own list of --queries queries (ids rank*nq ...); strong: one list partitioned
    weak = args.scaling == "weak"
    nq = args.queries * (world if weak else 1)
    rng = np.random.default_rng(3)
    sizes = rng.integers(4000, 6001, nq)
    fracs = rng.uniform(0.10, 0.40, nq)
    mine = list(range(rank * args.queries, (rank + 1) * args.queries)) if weak else shard.query_shard(sizes, rank, world)
    queries = []
    for qi in mine:
        w, h, f = shapes[qi % len(shapes)]
        q = synth.make_query(int(sizes[qi]), float(fracs[qi]), 3000 + qi, width=w, height=h, focal=f if f > 0 else 0.9 * max(w, h),
                             known_focal=f > 0)
        if args.advanced:
            synth.add_camsets_vec(q, seed=4000 + qi)
        queries.append((qi, q))
    T = max(1, args.host_threads)
    select = SELECT_COVERAGE if (args.early_exit or args.advanced) else SELECT_INLIERS
    sample_mode = SAMPLE_ADVANCED if args.advanced else SAMPLE_BASIC
    # one pose_amd_ransac_batch call: T worker threads inside the library, one stream each
    batched = not args.python_threads
    os.environ["POSE_AMD_BATCH_LANES"] = str(T)
    ctxs = [Context(dev_index) for _ in range(1 if batched else T)]
    for i, (qi, q) in enumerate(queries):
        q["query_id"] = qi

    def run_one(ctx, qi, q):
        ctx.set_query(q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], q["focal"])
        if args.advanced:
            ctx.set_camsets(q["cam_off"], q["cam_ids"])
        ctx.sample(sample_mode, 3000, qi, 0, args.budget, export=False)
        return ctx.ransac_run(select, 0.5, early_exit=args.early_exit)

    def run_part(t):
        out = []
        for qi, q in queries[t::T]:
            r = run_one(ctxs[t], qi, q)
            out.append((qi, r["steps"], r["inliers"], float(np.linalg.norm(r["C"] - q["C_true"])) if r["found"] else float("inf")))
        return out

    # the descriptor array of the batch is built once: a pass is the library call alone, and its
    # result records are turned into (steps, inliers, camera error) after the timed region
    prep = ctxs[0].prepare_batch([q for _, q in queries]) if batched else None
    # ... and uploaded once: the timed region starts with the queries resident in HBM
    resident = ctxs[0].resident_batch(prep, with_camsets=args.advanced) if batched and not args.host_queries else None

    def convert(raw):
        from pose_estimation_amd.api import result_dict
        res = []
        for (qi, q), rr in zip(queries, raw):
            r = result_dict(rr)
            res.append((qi, r["steps"], r["inliers"], float(np.linalg.norm(r["C"] - q["C_true"])) if r["found"] else float("inf")))
        return res

    def one_pass(pool):
        res = []
        if resident is not None:
            return resident.run(sample_mode, seed=3000, budget=args.budget, select_mode=select, early_exit=args.early_exit, raw=True)
        if batched:
            return ctxs[0].ransac_batch(prep, sample_mode, seed=3000, budget=args.budget, select_mode=select,
                                        early_exit=args.early_exit, raw=True)
        for part in pool.map(run_part, range(T)):
            res.extend(part)
        return res

    def sync():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    with ThreadPoolExecutor(T) as pool:
        for _ in range(args.warmup):
            one_pass(pool)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = one_pass(pool)
        sync()
        dt = time.perf_counter() - t0
    if batched:
        res = convert(res)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        stats = [None] * world
        dist.all_gather_object(stats, res)
        res = [x for part in stats for x in part]
    out = None
    if rank == 0:
        steps_total = sum(r[1] for r in res)
        located = sum(1 for r in res if r[3] < 2.0)
        mode = ("advanced processor: co-occurrence sampling over camera sets, 100 steps, coverage quality" if args.advanced
                else "reference loop: coverage quality, early exit, <= %d steps" % args.budget if args.early_exit
                else "fixed budget of %d hypotheses per query, inlier-count selection" % args.budget)
        out = {"metric": "queries/sec (Dubrovnik-shaped batch, %s RANSAC)" % ("advanced" if args.advanced else "basic"), "value": round(nq * args.steps / dt, 1), "unit": "queries/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE configs[%d]: %d Dubrovnik6K-shaped queries (real image sizes, %d known / %d unknown focal), "
                                      "4000-6000 matches, 10-40%% inliers, %s"
                                      % (3 if args.advanced else 2, nq, sum(1 for i in range(nq) if shapes[i % len(shapes)][2] > 0),
                                         sum(1 for i in range(nq) if shapes[i % len(shapes)][2] == 0), mode),
                          "parallelism": "%s, %s" % (
                              ("every one of %d GPU(s) runs its own list of %d queries" % (world, args.queries)) if weak
                              else "one list partitioned over %d GPU(s) (longest first)" % world, "%d host threads (contexts) per GPU" % T if not batched
                              else "all queries of a GPU share their launches (rounds of 128, 640, 3200, ... steps, list of running queries kept on the device)"),
                          "queries": "resident in HBM (pose_amd_batch_create before the timed region, pose_amd_batch_run timed)" if resident is not None
                                     else "host arrays handed over in every pass (packing + PCIe copy inside the timed region)",
                          "hypotheses_per_s": round(steps_total * args.steps / dt, 1),
                          "mean_steps_per_query": round(steps_total / max(len(res), 1), 1),
                          "queries_within_2_units_of_true_position": located}}
        if resident is not None:
            # (with several GPUs: rank 0's own list)
            out["roofline"] = roofline_shared_launches(resident, queries, res, sample_mode, select, args)
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_batch(ctxs[0], queries, args, select, sample_mode)
        elif world == 1:
            # Per-kernel times of this workload: a profiling pass outside the timed region, one
            # context with timing events, a sample of the queries one after the other.
            prof = Context(dev_index)
            tim = {}
            hyps = poses = 0
            pairs = 0.0
            sample_q = queries[::max(1, len(queries) // 64)][:64]
            for qi, q in sample_q:
                run_one(prof, qi, q)                                   # warm the context's buffers
            for qi, q in sample_q:
                r = run_one(prof, qi, q)
                t_ = prof.last_timing()
                for k, v in t_.items():
                    tim[k] = tim.get(k, 0.0) + v
                if q["focal"] > 0:
                    tim["p3p_ms"] = tim.get("p3p_ms", 0.0) + t_["solve_ms"]
                    tim["p3p_hyps"] = tim.get("p3p_hyps", 0) + r["steps"]
                else:
                    tim["p4pf_hyps"] = tim.get("p4pf_hyps", 0) + r["steps"]
                hyps += r["steps"]; poses += r["poses_scored"]; pairs += float(r["poses_scored"]) * len(q["kx"])
            prof.close()
            tim["select_coverage_ms"] = tim.pop("select_ms", 0.0)
            rl = roofline_of(tim, hyps, poses, pairs, False, max(int(tim.get("launches", 1)), 1), len(sample_q))
            rl["note"] = ("per-kernel HIP-event times of %d of the queries run one after the other on one stream (outside the timed region); "
                          "kernel_ms_per_step is per query; speculative chunks make the kernels process more hypotheses than the steps counted"
                          % len(sample_q))
            out["roofline"] = rl
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline_batch(ctxs[0], queries, args, select, sample_mode)
    if resident is not None:
        resident.close()
    for c in ctxs:
        c.close()
    return out


def bench_knn(args, env):
    """The matching step in front of the path: one pass = pose_amd_knn_match of --queries uint8
    descriptors (default 8192) against a model of --points averaged descriptors resident in HBM
    (the reference's FLANN search, query_processor_basic.cpp:60-96, done exactly).  Replicas only:
    with --gpus N every rank searches its own query descriptors against its own copy of the model."""
    rank, world, dev_index, red_dev, torch, dist = env
    from pose_estimation_amd import Context, synth_desc
    nq = args.queries if args.queries != 800 else 8192
    rng = np.random.default_rng(17)
    # model descriptors straight as floats (SIFT-like norms); a share of the queries are noisy copies
    model = synth_desc.make_model(min(args.points, 1 << 16), seed=21)
    proto = model["proto"].astype(np.float32)
    feat = proto[rng.integers(0, len(proto), args.points)] + rng.normal(0.0, 12.0, (args.points, 128)).astype(np.float32)
    feat = np.clip(feat, 0, 255).astype(np.float32)
    pick = rng.integers(0, args.points, nq)
    q = np.clip(np.rint(feat[pick] + rng.normal(0.0, 6.0, (nq, 128))), 0, 255).astype(np.uint8)
    q[nq // 2:] = np.clip(np.rint(synth_desc._sift_like(rng, nq - nq // 2)), 0, 255).astype(np.uint8)
    ctx = Context(dev_index)
    ctx.knn_set_model(feat)

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        r = ctx.knn_match(q, 0.7)
    sync()
    t0 = time.perf_counter()
    ms = {}
    for _ in range(args.steps):
        r = ctx.knn_match(q, 0.7)
        for k, v in ctx.knn_last_timing().items():
            ms[k] = ms.get(k, 0.0) + v
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    out = None
    if rank == 0:
        pairs = float(nq) * args.points
        scan_s = ms["scan_ms"] / args.steps * 1e-3
        executed = pairs * 2 * 128 * 2 / scan_s / 1e12               # two bfloat16 parts of every model descriptor
        out = {"metric": "query descriptors matched per second (exact 2-NN + ratio test)", "value": round(nq * world * args.steps / dt, 1), "unit": "descriptors/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16 x2 (matrix-core scan), f64 (decision)", "data": "synthetic",
               "config": {"workload": "matching step in front of the path ((f)4): %d uint8 query descriptors against %d averaged float descriptors of 128 "
                                      "dimensions, 2 nearest by squared L2 and ratio 0.7; replicas only" % (nq, args.points),
                          "host_transfer": "query descriptors up (%d KB) and results down inside the timed region; the model is resident" % (nq * 128 // 1024),
                          "pairs_per_s": round(pairs * world * args.steps / dt, 1), "good_matches": int(r["good"].sum()),
                          "rows_searched_exhaustively": int(r["exhaustive_rows"])},
               "roofline": {"kernel": "k_knn_scan", "bound": "mfma", "achieved": round(executed, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(executed / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                            "algorithmic_tflops": round(pairs * 2 * 128 / scan_s / 1e12, 1), "avg_launch_ms": round(scan_s * 1e3, 3),
                            "kernel_ms_per_step": {k: round(v / args.steps, 3) for k, v in ms.items()},
                            "note": "executed = 2 x 128 x 2 flop per (query, point) pair: both bfloat16 parts of the model descriptor; algorithmic = one part"}}
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import knn_oracle as K
            ns = 1536
            t1 = time.perf_counter()
            K.knn2(feat, q[:ns], 0.7)
            dtc = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(ns / dtc, 2), "unit": "descriptors/s", "cores": host_cores(), "cpu": cpu_model(), "kind": "port",
                                   "sample": "the first %d query descriptors against the whole model, oracle/knn_oracle.py (NumPy float64 matrix product "
                                             "with the BLAS threads of the box, exact re-evaluation of a shortlist), %.1f s" % (ns, dtc)}
    ctx.close()
    return out


LDS_PEAK_GBS = 256 * 128 * 2.4          # 256 CUs x 128 B per clock x 2.4 GHz


def roofline_shared_launches(resident, queries, res, sample_mode, select, args):
    """Roofline object of the batch path whose queries share their launches: one more run of the
    resident list outside the timed region with a HIP event after every launch (on the library's
    stream), summed by kernel.  The rounds are speculative, so the kernels process more RANSAC
    steps than the results report; the algorithmic work is counted on the steps actually computed
    (device counters), with the poses per step and correspondences per pose of the results."""
    resident.set_timing(True)
    raw = resident.run(sample_mode, seed=3000, budget=args.budget, select_mode=select, early_exit=args.early_exit, raw=True)
    t = resident.last_timing()
    resident.set_timing(False)
    from pose_estimation_amd.api import result_dict
    pairs = 0.0
    pix = 0.0
    stat = {True: [0, 0, 0.0, 0], False: [0, 0, 0.0, 0]}          # known focal: steps, poses, sum of n, queries
    for (qi, q), rr in zip(queries, raw):
        r = result_dict(rr)
        st = stat[q["focal"] > 0]
        st[0] += r["steps"]; st[1] += r["poses_scored"]; st[2] += len(q["kx"]); st[3] += 1
        pix += float(q["width"]) * q["height"]
    for known, computed in ((True, t["p3p_steps_computed"]), (False, t["p4pf_steps_computed"])):
        st = stat[known]
        if st[0]:
            pairs += computed * (st[1] / st[0]) * (st[2] / st[3])
    mean_pix = pix / max(len(queries), 1)
    from pose_estimation_amd import SAMPLE_BASIC
    # the library's rule: matrix-core screen for the basic sampler's (mostly junk) poses, packed float32 screen otherwise
    mfma_used = sample_mode == SAMPLE_BASIC and os.environ.get("POSE_AMD_SEG_SCORE") != "packed" and os.environ.get("POSE_AMD_SCORE") not in ("packed", "fp64")
    score_kernel = "k_seg_score_mfma" if mfma_used else "k_seg_score_screen"
    work = {score_kernel: (t["score_ms"], pairs * SCORE_FLOP_PER_PAIR),
            "k_seg_p4pf_elim": (t["elim_ms"], t["p4pf_steps_computed"] * ELIM_FLOP_PER_HYP),
            "k_seg_p4pf_finish": (t["finish_ms"], t["p4pf_steps_computed"] * FINISH_FLOP_PER_HYP),
            "k_seg_solve_p3p": (t["p3p_ms"], t["p3p_steps_computed"] * P3P_FLOP_PER_HYP)}
    # coverage areas: no floating-point work; a pose's bitmap (one bit per pixel) is cleared and counted in LDS
    cov_bytes = t["poses_painted"] * mean_pix / 8 * 2
    times = dict((k, v[0]) for k, v in work.items())
    times["k_seg_cover_pose"] = t["coverage_ms"]
    dom = max(times, key=lambda k: times[k])
    ms = {k: round(v, 3) for k, v in t.items() if k.endswith("_ms")}
    if dom == "k_seg_cover_pose":
        ach = cov_bytes / (t["coverage_ms"] * 1e-3) / 1e9
        rl = {"kernel": dom, "bound": "lds", "achieved": round(ach, 1), "peak": round(LDS_PEAK_GBS, 1), "unit": "GB/s", "frac": round(ach / LDS_PEAK_GBS, 4),
              "traffic": None, "algorithmic": "2 x width x height / 8 bytes of LDS per pose painted (clear the bitmap, count it)"}
    elif dom == "k_seg_score_mfma":
        mf = pairs * SCREEN_MFMA_FLOP_PER_PAIR / (t["score_ms"] * 1e-3) / 1e12
        rl = {"kernel": dom, "bound": "mfma", "achieved": round(mf, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(mf / BF16_MFMA_PEAK_TFLOPS, 4),
              "traffic": None, "pairs_per_s": round(pairs / (t["score_ms"] * 1e-3), 1)}
    else:
        tf = work[dom][1] / (work[dom][0] * 1e-3) / 1e12 if work[dom][0] > 0 else 0.0
        rl = {"kernel": dom, "bound": "valu_fp64", "achieved": round(tf, 3), "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
              "frac": round(tf / FP64_VECTOR_PEAK_TFLOPS, 4), "traffic": None}
    rl.update({"launches": t["launches"], "kernel_ms_per_step": ms,
               "kernel_tflops": {k: round(f / (m * 1e-3) / 1e12, 3) for k, (m, f) in work.items() if m > 0 and k != "k_seg_p4pf_finish"},
               "kernel_executed_tflops": {k: round(f / (m * 1e-3) / 1e12, 3) for k, (m, f) in work.items() if m > 0 and k == "k_seg_p4pf_finish"},
               "k_seg_cover_pose_lds_gbs": round(cov_bytes / (t["coverage_ms"] * 1e-3) / 1e9, 1) if t["coverage_ms"] > 0 else None,
               "steps_computed": t["p3p_steps_computed"] + t["p4pf_steps_computed"], "poses_painted": t["poses_painted"],
               "pairs_scored": round(pairs, 1),
               "note": "HIP events after every launch of one run of the resident list (outside the timed region); kernel_ms_per_step is that run's "
                       "sum per kernel; steps_computed counts the speculative rounds (128, 640, 3200, ... steps for every query still running)"})
    return rl


def cpu_baseline_batch(ctx, queries, args, select, sample_mode):
    """The CPU oracle on a bounded sample of the same queries and the same tuples (exported by the
    device sampler): all host cores over queries (each query's loop is sequential, as in the
    reference), plus the reference-faithful mode (4 OpenMP threads inside TestHypothesis, two
    h x w float maps per pose) on a few of them."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle_lib as O
    from pose_estimation_amd import SELECT_COVERAGE
    cores = host_cores()
    coverage = select == SELECT_COVERAGE

    def tuples(qi, q):
        ctx.set_query(q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], q["focal"])
        if args.advanced:
            ctx.set_camsets(q["cam_off"], q["cam_ids"])
        s = ctx.sample(sample_mode, 3000, qi, 0, args.budget)
        s[:, 3] = np.where(s[:, 3] == 0xFFFFFFFF, 0, s[:, 3])
        return s

    def run(item):
        (qi, q), s = item
        kw = (q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], q["focal"])
        if not coverage:
            O.ransac_fixed_budget(*kw, s, threads=1)
            return len(s)
        return O.ransac(*kw, s, select_coverage=True, early_exit=args.early_exit, max_steps=args.budget)["steps"]

    # probe one query to size the sample for about 10 s of CPU work
    probe = (queries[0], tuples(*queries[0]))
    t0 = time.perf_counter()
    run(probe)
    per = time.perf_counter() - t0
    count = int(min(len(queries), max(cores, 10.0 * cores / max(per, 1e-6))))
    items = [(qq, tuples(*qq)) for qq in queries[:count]]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        steps = sum(pool.map(run, items))
    dt = time.perf_counter() - t0
    out = {"value": round(count / dt, 2), "unit": "queries/s", "cores": cores, "cpu": cpu_model(), "kind": "port",
           "hypotheses_per_s": round(steps / dt, 1),
           "sample": "the first %d of the queries on the same tuples (exported by the device sampler), oracle/pose_oracle.c, one sequential "
                     "loop per query, %d queries at a time, %.1f s" % (count, cores, dt)}
    if coverage:
        nf = 3
        t1 = time.perf_counter()
        fsteps = 0
        for (qi, q), s in items[:nf]:
            kw = (q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], q["focal"])
            fsteps += O.ransac_reference_faithful(*kw, s[:min(len(s), 600)], early_exit=args.early_exit, max_steps=min(args.budget, 600), threads=4)["steps"]
        dtf = time.perf_counter() - t1
        out["reference_faithful"] = {"value": round(nf / dtf, 3), "unit": "queries/s", "cores": 4, "hypotheses_per_s": round(fsteps / dtf, 1),
                                     "sample": "the first %d queries (at most 600 steps each), TestHypothesis as the reference runs it "
                                               "(OMP_NUM_THREADS=4, two height x width float maps per pose), %.1f s" % (nf, dtf)}
    return out


if __name__ == "__main__":
    sys.exit(main())
