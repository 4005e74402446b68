"""The allocation-free entry point pose_amd_p4pf_batch_noalloc against pose_amd_p4pf_batch.

Same kernels, so the bar is bit-exact: solution counts, focal lengths, rotations and
translations of every problem, on two calls in a row (the second one smaller, so it runs over
whatever the first left in the static scratch array), and no device memory may be taken by the
calls.  Device buffers for inputs and outputs come from plain hipMalloc calls made by the test."""
import ctypes as C

import numpy as np
import pytest

from hip_mem import DeviceArray, free_bytes
from pose_estimation_amd import synth, PoseAmdError

pytestmark = pytest.mark.gpu
MAX_COUNT = 16384


def run_noalloc(ctx, m, M):
    b = len(m)
    m_d = DeviceArray(np.ascontiguousarray(m, np.float64).reshape(b, 8))
    M_d = DeviceArray(np.ascontiguousarray(M, np.float64).reshape(b, 12))
    # outputs pre-filled with garbage: the entry point must define every element
    nsol = DeviceArray(np.full(b, -77, np.int32))
    f = DeviceArray(np.full((b, 10), 3.25))
    R = DeviceArray(np.full((b, 90), 3.25))
    t = DeviceArray(np.full((b, 30), 3.25))
    free0 = free_bytes()
    rc = ctx._lib.pose_amd_p4pf_batch_noalloc(ctx._h, b, m_d.ptr, M_d.ptr, nsol.ptr, f.ptr, R.ptr, t.ptr)
    ctx._chk(rc, "p4pf_batch_noalloc")
    ctx.synchronize()
    assert free_bytes() >= free0, "the call took device memory"
    out = nsol.get(), f.get(), R.get().reshape(b, 10, 3, 3), t.get().reshape(b, 10, 3)
    for a in (m_d, M_d, nsol, f, R, t):
        a.free()
    return out


def test_noalloc_entry_point_equals_p4pf_batch(ctx):
    m, M = synth.random_p4pf_problems(5000, 123)
    M[7] = M[7, [0, 0, 2, 3]]                      # a degenerate problem (two identical world points): -1
    # first call: large; second call: smaller, different problems, over the first call's leftovers;
    # then a count that does not fill the last pass of the elimination, and the largest allowed one
    for lo, hi in ((0, 5000), (5000 - 1237, 5000), (100, 103), (0, 1)):
        exp = ctx.p4pf_batch(m[lo:hi], M[lo:hi])
        got = run_noalloc(ctx, m[lo:hi], M[lo:hi])
        assert np.array_equal(got[0], exp[0])
        for g, e in zip(got[1:], exp[1:]):
            assert np.array_equal(g, e)
        if lo == 0 and hi == 5000:
            assert exp[0][7] == -1 and (exp[0] > 0).sum() > 4000
    mb, Mb = synth.random_p4pf_problems(64, 9)
    reps = MAX_COUNT // 64
    mb, Mb = np.tile(mb, (reps, 1, 1)), np.tile(Mb, (reps, 1, 1))
    exp = ctx.p4pf_batch(mb, Mb)
    got = run_noalloc(ctx, mb, Mb)
    assert all(np.array_equal(g, e) for g, e in zip(got, exp))
    # and a small one again after the largest
    exp = ctx.p4pf_batch(m[:33], M[:33])
    got = run_noalloc(ctx, m[:33], M[:33])
    assert all(np.array_equal(g, e) for g, e in zip(got, exp))


def test_noalloc_calls_from_two_contexts_are_ordered_on_the_device(ctx):
    """Two contexts (two streams) of one GPU share the static array.  Their calls are issued back to
    back without any synchronisation in between; the library makes each call's stream wait for the
    previous call's kernels, so both results are those of pose_amd_p4pf_batch."""
    from pose_estimation_amd import Context
    m, M = synth.random_p4pf_problems(12000, 321)
    halves = [(m[:8000], M[:8000]), (m[8000:], M[8000:])]
    exp = [ctx.p4pf_batch(a, b) for a, b in halves]
    with Context(0) as other:
        for rep in range(3):
            bufs = []
            for c, (a, b) in zip((ctx, other), halves):
                n = len(a)
                d = [DeviceArray(np.ascontiguousarray(a, np.float64).reshape(n, 8)), DeviceArray(np.ascontiguousarray(b, np.float64).reshape(n, 12)),
                     DeviceArray(np.full(n, -77, np.int32)), DeviceArray(np.full((n, 10), 3.25)), DeviceArray(np.full((n, 90), 3.25)), DeviceArray(np.full((n, 30), 3.25))]
                bufs.append(d)
            for c, d, (a, b) in zip((ctx, other), bufs, halves):           # both enqueued before either is waited for
                c._chk(c._lib.pose_amd_p4pf_batch_noalloc(c._h, len(a), *[x.ptr for x in d]), "p4pf_batch_noalloc")
            ctx.synchronize(); other.synchronize()
            for d, e, (a, b) in zip(bufs, exp, halves):
                n = len(a)
                assert np.array_equal(d[2].get(), e[0]) and np.array_equal(d[3].get(), e[1])
                assert np.array_equal(d[4].get().reshape(n, 10, 3, 3), e[2]) and np.array_equal(d[5].get().reshape(n, 10, 3), e[3])
                for x in d:
                    x.free()


def test_noalloc_entry_point_rejects_sizes_beyond_its_scratch(ctx):
    z = DeviceArray(np.zeros(8))
    p = z.ptr
    rc = ctx._lib.pose_amd_p4pf_batch_noalloc(ctx._h, MAX_COUNT + 1, p, p, p, p, p, p)
    assert rc == -1 and b"POSE_AMD_NOALLOC_MAX_COUNT" in ctx._lib.pose_amd_last_error(ctx._h)
    with pytest.raises(PoseAmdError):
        ctx._chk(ctx._lib.pose_amd_p4pf_batch_noalloc(ctx._h, -1, p, p, p, p, p, p), "p4pf_batch_noalloc")
    assert ctx._lib.pose_amd_p4pf_batch_noalloc(ctx._h, 0, None, None, None, None, None, None) == 0
