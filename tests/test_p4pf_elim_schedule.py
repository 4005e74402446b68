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


def test_stage1_order_is_valid_and_in_lane_steps_are_co_located():
    """A stage-1 pivot row must never have been a target of an earlier step (its entries are read
    as plain coefficients), whatever order the pivots are taken in; the stage-2 steps marked
    in-lane have exactly two candidate rows, in the two slots of one lane, which no other in-lane
    step shares; positions are a permutation of 0..28."""
    s = G.build_schedule()
    touched = set()
    for st in s["steps"]:
        assert st["row"] not in touched
        touched |= set(st["targets"])
    pos = s["lane_of"]
    assert sorted(pos.values()) == list(range(29))
    used = set()
    in_lane = [st for st in s["stage2"] if st.get("in_lane")]
    assert len(in_lane) >= 2
    for st in in_lane:
        a, b = (pos[r] for r in st["cand"])
        assert len(st["cand"]) == 2 and abs(a - b) == 16 and not ({a, b} & used)
        used |= {a, b}
    # in-lane steps come first: later steps may involve the rows they retire
    flags = [bool(st.get("in_lane")) for st in s["stage2"]]
    assert flags == sorted(flags, reverse=True)


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


def test_blocked_stage2_equals_step_by_step_elimination():
    """The kernel takes the stage-2 pivots four at a time: panel steps on the block's own columns,
    raw trailing rows of the pivots, effective multipliers m' = m L^-1, one update per block.
    Same pivots, same result as publishing every pivot row at once (up to rounding that scales
    with the conditioning of the sample); blocks cover every non-leading step exactly once and
    no published row exceeds its LDS slot."""
    s = G.build_schedule()
    lead, blocks = G.stage2_blocks(s, G.BLOCK)
    assert sorted(lead + [i for b in blocks for i in b["steps"]]) == list(range(24))
    for b in blocks:
        assert not (set(b["cols"]) & set(b["trail"])) and set(G.KEEP + G.RHS) <= set(b["trail"])
        assert (len(b["trail"]) + len(b["cols"])) * 8 <= G.ROW_BYTES
    m, M = synth.random_p4pf_problems(120, seed=19, contaminate=(0, 1, 2), noise=0.3)
    errs = []
    for i in range(120):
        n, k, A, _ = O.p4pf_stages(m[i], M[i])
        if n < 0:
            continue
        st1, A1 = G.simulate(s, k)
        st2, A2 = G.simulate_blocked(s, k, G.BLOCK)
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
