import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from pose_estimation_amd import api, synth
from gates import p4pf_gate, relerr, RTOL
ctx = api.Context(0) if hasattr(api, "Context") else None
m, M = synth.random_p4pf_problems(600, 5)
nsol, f, R, t = ctx.p4pf_batch(m, M)
np.save(sys.argv[1], np.concatenate([nsol.astype(float).ravel(), f.ravel(), R.ravel(), t.ravel()]))
for i in range(600):
    g = p4pf_gate(m[i], M[i])
    if g is None: continue
    n, f0, R0, t0, e_ref = g
    if nsol[i] != n:
        print(i, "count", nsol[i], n, "e_ref", e_ref, "f_dev", f[i,:max(nsol[i],0)], "f_ref", f0); continue
    if n <= 0: continue
    e = max(relerr(f[i,:n], f0), relerr(R[i,:n], R0, 1e-3), relerr(t[i,:n], t0, 1e-3))
    if not e < max(RTOL, 20*e_ref): print(i, "err", e, "e_ref", e_ref)
