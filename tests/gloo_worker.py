"""Worker of tests/test_shard_gloo.py: one rank of a CPU process group (gloo).  Each rank scores
its own hypothesis range with the CPU oracle, the packed keys are max-all-reduced, and rank 0
compares the winner with the oracle's sequential loop over the whole range."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib as O                                   # noqa: E402
from pose_estimation_amd import shard, synth             # noqa: E402


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total, n = 96, 400
    q = synth.make_query(n, 0.3, seed=21)
    args = (q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], 0.0)
    begin, end = shard.hypothesis_shard(total, rank, world)
    samples = np.array([O.sample_basic(77, 0, h, 4, n)[0] for h in range(begin, end)], np.uint32)
    r = O.ransac(*args, samples, select_coverage=False, early_exit=False, max_steps=len(samples))
    key = shard.pack_key(r["hyposet"], begin + r["best_step"]) if r["found"] else 0
    best = shard.reduce_best(key, dist)
    # the same reduction started without waiting, several in flight (what bench.py does per step)
    handles = [shard.reduce_best_begin(key + i, dist) for i in range(3)]
    best_async = [shard.reduce_best_end(h) for h in handles]
    # bench.py's reducer: a ring of eight preallocated slots, every step's reduction enqueued by a helper
    # thread; eleven steps reuse slots, and the keys come back in step order
    import torch
    import bench
    red = bench.KeyReducer(torch, dist, "cpu")
    for i in range(11):
        red.begin(1000 * i + rank)
    reduced = red.collect()
    red.begin(7 + rank)
    reduced += red.collect()
    red.close()
    reducer_ok = reduced == [1000 * i + world - 1 for i in range(11)] + [7 + world - 1]
    # by-query partition: every query is owned by exactly one rank
    costs = [4000 + 37 * i % 2000 for i in range(50)]
    mine = shard.query_shard(costs, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        allsamples = np.array([O.sample_basic(77, 0, h, 4, n)[0] for h in range(total)], np.uint32)
        ref = O.ransac(*args, allsamples, select_coverage=False, early_exit=False, max_steps=total)
        count, hyp = shard.unpack_key(best)
        flat = sorted(i for part in gathered for i in part)
        loads = [sum(costs[i] for i in part) for part in gathered]
        with open(out_path, "w") as f:
            json.dump({"world": world, "count": count, "hyp": hyp, "ref_count": int(ref["hyposet"]), "ref_hyp": int(ref["best_step"]),
                       "queries_ok": flat == list(range(50)), "loads": loads,
                       "async_ok": best_async == [best + i for i in range(3)], "reducer_ok": reducer_ok}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
