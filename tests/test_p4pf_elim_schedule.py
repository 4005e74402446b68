"""The generated elimination schedule of the P4Pf template, interpreted with NumPy, against
the oracle's partial-pivot LU (reference P4pf.cpp:1084-1109).  Runs without a GPU."""
import os
import sys

import numpy as np

import oracle_lib as O
from pose_estimation_amd import synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import gen_p4pf_elim as G   # noqa: E402


def test_schedule_shape():
    s = G.build_schedule()
    assert len(s["steps"]) == 49                       # constant pivots
    assert not any(st["materialised"] for st in s["steps"])
    assert len(s["lanes_rows"]) == 29 and s["lanes_rows"] == s["survivors"]
    assert len(s["stage2"]) == 24
    done = sorted([st["col"] for st in s["steps"]] + [st["col"] for st in s["stage2"]] + G.KEEP)
    assert done == list(range(78))                     # every left-hand column is eliminated once


def test_stage1_order_is_valid_and_slots_hold_the_two_groups():
    """A stage-1 pivot row must never have been a target of an earlier step (its entries are read
    as plain coefficients), whatever order the pivots are taken in; positions are a permutation
    of 0..15 and 16..; the candidate rows of a split-phase step all sit in the slot that runs it,
    and a slot skips a stage-1 step only when none of its rows is a target."""
    s = G.build_schedule()
    touched = set()
    for st in s["steps"]:
        assert st["row"] not in touched
        touched |= set(st["targets"])
        for slot in (0, 1):
            assert st["slots"][slot] == any(s["slot_of"][t] == slot for t in st["targets"])
    pos = s["lane_of"]
    assert len(set(pos.values())) == 29 and all(pos[r] // 16 == s["slot_of"][r] for r in pos)
    for slot in (0, 1):
        n = sum(1 for r in pos if s["slot_of"][r] == slot)
        assert sorted(pos[r] for r in pos if s["slot_of"][r] == slot) == list(range(16 * slot, 16 * slot + n))
    assert len(s["split"][0]) + len(s["split"][1]) == 12 and len(s["joint"]) == 12
    for slot in (0, 1):
        for i in s["split"][slot]:
            assert all(s["slot_of"][r] == slot for r in s["stage2"][i]["cand"])
    assert sorted(s["split"][0] + s["split"][1] + s["joint"]) == list(range(24))


def test_schedule_reproduces_action_matrix_rows():
    s = G.build_schedule()
    m, M = synth.random_p4pf_problems(60, seed=9, contaminate=(0, 1), noise=0.2)
    errs = []
    for i in range(60):
        n, k, A, _ = O.p4pf_stages(m[i], M[i])
        if n < 0:
            continue
        st, A5 = G.simulate(s, k)
        assert st == 0
        ref = A[5:10]
        errs.append(np.max(np.abs(A5 - ref)) / np.max(np.abs(ref)))
    errs = np.array(errs)
    assert len(errs) >= 55
    # two different pivot orders: the difference scales with the conditioning of the sample
    assert np.median(errs) < 1e-10 and np.quantile(errs, 0.9) < 1e-8 and np.max(errs) < 1e-4, (np.median(errs), np.max(errs))


def test_generated_updates_equal_step_by_step_elimination():
    """The kernel skips the stage-1 steps of a slot none of whose rows takes part, eliminates the
    two independent groups of the split phase side by side (per-slot pivots, columns and trailing
    columns), gathers the 16 rows that are active after the first joint step in slot 0, and takes
    its pivots in blocks: panel steps on the block's own columns, raw trailing rows of the
    pivots, effective multipliers m' = m L^-1, one update per block.  Same pivots,
    same result as the plain row-by-row elimination (up to rounding that scales with the
    conditioning of the sample); blocks cover every step exactly once, the rows a block publishes
    fit the LDS scratch and no published row exceeds its LDS slot."""
    s = G.build_schedule()
    sb, jb, gb = G.split_blocks(s, G.BLOCK), G.joint_blocks(s, G.BLOCK), G.single_blocks(s, G.BLOCK)
    steps = [i for parts in sb + gb for p in parts for i in p["steps"]] + [i for b in jb for i in b["steps"]]
    assert sorted(steps) == list(range(24))
    # the rows are gathered in slot 0 after the first joint step: 16 are active then, and every
    # column one of them can still hold is moved
    assert G.COMPACT and len(jb) == 1 and len(jb[0]["steps"]) == 1 and not any(parts[1]["steps"] for parts in gb)
    assert set(G.compact_cols(s)) == set(jb[0]["trail"]) and set(G.KEEP + G.RHS) <= set(G.compact_cols(s))
    assert len(s["survivors"]) - len(s["split"][0]) - len(s["split"][1]) - 1 == 16
    for parts in sb + gb:
        # a block serves one slot (a round of masked stores costs the same for one row as for five)
        assert bool(parts[0]["steps"]) != bool(parts[1]["steps"])
        nb = len(parts[0]["steps"]) + len(parts[1]["steps"])
        n_items = len(G.row_items(nb, (parts[0]["trail"] or parts[1]["trail"]))[0])
        assert ((nb - 1) * G.ROW_BYTES) // 16 + (32 if n_items > 32 else 16) <= G.PBUF_VEC      # every reader stays inside the scratch
        for p in parts:
            assert not (set(p["cols"]) & set(p["trail"]))
            if p["cols"]:
                G.row_items(len(p["cols"]), p["trail"])       # asserts the row fits
    for b in jb:
        assert not (set(b["cols"]) & set(b["trail"])) and set(G.KEEP + G.RHS) <= set(b["trail"])
        G.row_items(len(b["cols"]), b["trail"])
    m, M = synth.random_p4pf_problems(120, seed=19, contaminate=(0, 1, 2), noise=0.3)
    errs = []
    for i in range(120):
        n, k, A, _ = O.p4pf_stages(m[i], M[i])
        if n < 0:
            continue
        st1, A1 = G.simulate(s, k)
        st2, A2 = G.simulate_kernel(s, k, G.BLOCK)
        assert st1 == 0 and st2 == 0
        errs.append(np.max(np.abs(A1 - A2)) / np.max(np.abs(A1)))
    errs = np.array(errs)
    assert len(errs) >= 110 and np.median(errs) < 1e-11 and np.quantile(errs, 0.9) < 1e-9 and np.max(errs) < 1e-5, (np.median(errs), np.max(errs))


def test_generated_header_is_current():
    """The committed header equals what the generator emits for the default settings."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "pose_estimation_amd", "csrc", "p4pf_elim_gen.h")) as f:
        committed = f.read()
    assert committed == G.emit(G.build_schedule())
