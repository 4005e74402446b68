"""One-off parity survey on an MI355X: 6 000 random minimal P4Pf problems through pose_amd_p4pf_batch
against the oracle, with the gate and tolerance of tests/test_gpu_parity.py.  Prints the agreement
count and the first disagreements (index, device solutions, oracle solutions, oracle self-error)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as T
from pose_estimation_amd import Context, synth
ctx = Context(0)
N = 6000
m, M = synth.random_p4pf_problems(N, 77)
nsol, f, R, t = ctx.p4pf_batch(m, M)
gated = agree = near = 0
bad = []
for i in range(N):
    g = T.p4pf_gate(m[i], M[i])
    if g is None: continue
    n0, f0 = g[0], g[1]
    if n0 >= 2 and np.any(np.abs(np.diff(f0)) < 1e-9 * np.abs(f0[1:])) and nsol[i] == n0 - 2:
        near += 1; continue
    gated += 1
    ok = T.p4pf_agrees(nsol[i], f[i], R[i], t[i], g)
    agree += ok
    if not ok: bad.append((i, int(nsol[i]), int(n0), float(g[4])))
print("gated", gated, "agree", agree, "near_double", near, "disagree", len(bad))
print(bad[:20])
