"""Wall-time breakdown of one query through the per-query API (set_query / sample / ransac_run)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pose_estimation_amd import Context, synth, SELECT_INLIERS, SELECT_COVERAGE, SAMPLE_BASIC
ctx = Context(0)
for known in (False, True):
    for frac in (0.1, 0.25, 0.4):
        q = synth.make_query(5000, frac, 9, known_focal=known)
        for mode, ee in ((SELECT_COVERAGE, True), (SELECT_INLIERS, True), (SELECT_INLIERS, False)):
            ts = []
            for rep in range(4):
                t0 = time.perf_counter()
                ctx.set_query(q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], q["focal"])
                t1 = time.perf_counter()
                ctx.sample(SAMPLE_BASIC, 3, rep, 0, 10000, export=False)
                t2 = time.perf_counter()
                r = ctx.ransac_run(mode, 0.5, early_exit=ee)
                t3 = time.perf_counter()
                ts.append((t1 - t0, t2 - t1, t3 - t2, r["steps"]))
            a = np.array(ts[1:]).mean(0)
            print("known_focal=%d inliers=%.2f %s early_exit=%d: set_query %.3f ms, sample %.3f ms, ransac_run %.3f ms (steps %.0f) kernels %s"
                  % (known, frac, "coverage" if mode == SELECT_COVERAGE else "inliers ", ee, a[0] * 1e3, a[1] * 1e3, a[2] * 1e3, a[3],
                     {k: round(v, 3) for k, v in ctx.last_timing().items()}))
