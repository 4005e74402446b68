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


def test_generated_header_is_current():
    """The committed header equals what the generator emits for the default settings."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "pose_estimation_amd", "csrc", "p4pf_elim_gen.h")) as f:
        committed = f.read()
    assert committed == G.emit(G.build_schedule())
