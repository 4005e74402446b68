"""Wall-time breakdown of the advanced path for one query."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pose_estimation_amd import Context, synth, SELECT_INLIERS, SELECT_COVERAGE, SAMPLE_ADVANCED
ctx = Context(0)
for known in (False, True):
    q = synth.make_query(5000, 0.25, 9, known_focal=known)
    synth.add_camsets(q, seed=4)
    for mode in (SELECT_COVERAGE, SELECT_INLIERS):
        ts = []
        for rep in range(4):
            t0 = time.perf_counter()
            ctx.set_query(q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], q["focal"])
            t1 = time.perf_counter()
            ctx.set_camsets(q["cam_off"], q["cam_ids"])
            t2 = time.perf_counter()
            ctx.sample(SAMPLE_ADVANCED, 3, rep, 0, 100, export=False)
            t3 = time.perf_counter()
            r = ctx.ransac_run(mode, 0.5, early_exit=False)
            t4 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
        a = np.array(ts[1:]).mean(0) * 1e3
        print("known=%d mode=%d: set_query %.3f set_camsets %.3f sample %.3f ransac_run %.3f ms; inliers %d; kernels %s"
              % (known, mode, a[0], a[1], a[2], a[3], r["inliers"], {k: round(v, 3) for k, v in ctx.last_timing().items()}))
