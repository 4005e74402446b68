import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pose_estimation_amd import Context, synth, SELECT_INLIERS, SAMPLE_BASIC
ctx = Context(0)
q = synth.make_query(4000, 0.1, 2)
ctx.set_query(q["kx"], q["ky"], q["X"], q["Y"], q["Z"], q["width"], q["height"], 0.0)
H = 20000
ctx.sample(SAMPLE_BASIC, 1002, 0, 0, H, export=False)
r = ctx.ransac_run(SELECT_INLIERS, 0.5, early_exit=False, hyp_count=H, want_per_hyp=True)
c = []
for h in range(H):
    for k in range(max(r["nsol"][h], 0)):
        c.append(r["counts"][h, k])
c = np.array(c)
print("poses", len(c), "count histogram", np.bincount(np.minimum(c, 12)))
