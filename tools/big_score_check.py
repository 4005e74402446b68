"""One-off survey on an MI355X: per-pose inlier counts of the production scoring path (matrix-core
screen for these launch sizes) against the float64-only kernel, over queries of several sizes,
inlier fractions, solvers and world-coordinate offsets.  Prints one line per case."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pose_estimation_amd import Context, synth, SAMPLE_BASIC, SELECT_INLIERS

ctx = Context(0)
os.environ["POSE_AMD_SCORE"] = "fp64"
ref = Context(0)
del os.environ["POSE_AMD_SCORE"]
cases = [(4000, 0.10, False, 0.0, 1 << 18), (4000, 0.10, True, 0.0, 1 << 18), (2000, 0.30, True, 0.0, 1 << 18), (977, 0.5, False, 3000.0, 1 << 17),
         (6001, 0.2, False, -800.0, 1 << 17), (100000, 0.01, False, 0.0, 1 << 16), (33, 0.9, True, 0.0, 1 << 17), (5000, 0.4, True, 1e5, 1 << 17)]
bad = 0
for ci, (n, frac, known, shift, H) in enumerate(cases):
    q = synth.make_query(n, frac, 100 + ci, known_focal=known)
    q["X"] = (q["X"] + np.float32(shift)).astype(np.float32)
    out = []
    for c in (ctx, ref):
        c.set_query(q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], q["focal"])
        c.sample(SAMPLE_BASIC, 7 + ci, ci, 0, H, export=False)
        out.append(c.ransac_run(SELECT_INLIERS, 0.5, early_exit=False, hyp_count=H, want_per_hyp=True))
    a, b = out
    same = np.array_equal(a["nsol"], b["nsol"]) and np.array_equal(a["counts"], b["counts"]) and (a["best_hyp"], a["best_sol"], a["inliers"]) == (b["best_hyp"], b["best_sol"], b["inliers"])
    bad += not same
    if not same:
        dn = np.flatnonzero(a["nsol"] != b["nsol"])
        valid = np.arange(a["counts"].shape[1])[None, :] < np.maximum(a["nsol"], 0)[:, None]
        dc = np.argwhere((a["counts"] != b["counts"]) & valid)
        print("   nsol differs at", len(dn), "hypotheses; counts differ at", len(dc), "valid slots; first:",
              [(int(h), int(k), int(a["counts"][h, k]), int(b["counts"][h, k])) for h, k in dc[:6]],
              "best", (a["best_hyp"], a["best_sol"], a["inliers"]), (b["best_hyp"], b["best_sol"], b["inliers"]))
    print("n %6d frac %.2f known_focal %d shift %8g H %7d: %s (best %d inliers)" % (n, frac, known, shift, H, "identical" if same else "DIFFERENT", a["inliers"]))
print("cases that differ:", bad)
