"""Partitioning of the RANSAC work over the GPUs of one node (one process per GPU).

Two partitions exist, both without any exchange of correspondence data:

* by query (independent queries, the loop of benchmark.cpp:24-36): static assignment, no
  collective, rank 0 gathers the small per-query results;
* by hypothesis range (one very large query): rank r evaluates the hypothesis ids
  [begin_r, end_r) of the same counter-based sample stream, and one max-all-reduce of a packed
  64-bit key names the winner.  The key orders poses exactly as the sequential loop does
  (query_processor_basic.cpp:214-225: a later pose replaces the kept one only if strictly
  better), i.e. larger inlier count first, then the lower hypothesis id.
"""
import numpy as np

KEY_ID_MASK = 0xFFFFFFFF


def hypothesis_shard(total, rank, world):
    """Contiguous id range [begin, end) of `rank`; ranges differ by at most one id."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(int(total), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def query_shard(costs, rank, world):
    """Indices of the queries `rank` processes.  `costs` (one number per query, e.g. its match
    count) balances the load: longest-processing-time-first assignment, deterministic (the rule of
    pose_amd_partition_queries, which spreads a batch over the devices of one process)."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world)
    owner = np.zeros(len(costs), dtype=np.int64)
    for q in order:
        r = int(np.argmin(load))
        owner[q] = r
        load[r] += costs[q]
    return [int(i) for i in np.nonzero(owner == rank)[0]]


def pack_key(count, hyp_id):
    """(score, hypothesis id) -> int64 key whose maximum is the sequential loop's choice."""
    if not (0 <= int(count) < (1 << 31)) or not (0 <= int(hyp_id) <= KEY_ID_MASK):
        raise ValueError("key field out of range")
    return (int(count) << 32) | (KEY_ID_MASK - int(hyp_id))


def unpack_key(key):
    key = int(key)
    return key >> 32, KEY_ID_MASK - (key & KEY_ID_MASK)


def reduce_best(key, dist=None, device="cpu"):
    """Max-all-reduce of one packed key over the process group (RCCL on GPUs, gloo on CPU)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(key)
    import torch
    t = torch.tensor([int(key)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def reduce_best_begin(key, dist=None, device="cpu"):
    """The same reduction started without waiting for it: the collective runs beside the next
    step's kernels (the library computes on a stream of its own).  -> handle for reduce_best_end."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return (None, int(key))
    import torch
    t = torch.tensor([int(key)], dtype=torch.int64, device=device)
    return (dist.all_reduce(t, op=dist.ReduceOp.MAX, async_op=True), t)


def reduce_best_end(handle):
    work, t = handle
    if work is None:
        return t
    work.wait()
    return int(t.item())
