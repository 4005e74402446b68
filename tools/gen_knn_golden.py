#!/usr/bin/env python3
"""Writes tests/golden/knn_small.npz: a small descriptor-matching case (model views, query
descriptors) with the answer of oracle/knn_oracle.py (averaged descriptors, two nearest
neighbours, float distances, ratio test)."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import knn_oracle as K                       # noqa: E402
from pose_estimation_amd import synth_desc   # noqa: E402

model = synth_desc.make_model(1500, seed=5)
# a point without views (NaN descriptor in the reference's arithmetic) and two points with identical views (an exact tie)
off = model["desc_off"].astype(np.int64)
desc = model["descriptors"]
dup = desc[off[10]:off[11]]
off2 = np.concatenate([off, [off[-1], off[-1] + len(dup)]]).astype(np.uint32)      # point 1500: no views; point 1501: copy of point 10
desc2 = np.concatenate([desc, dup])
q, truth = synth_desc.make_queries(model, 400, seed=6)
q[7] = np.clip(np.rint(model["proto"][10]), 0, 255).astype(np.uint8)               # nearest neighbours: points 10 and 1501, tied
feat = K.average_views(off2, desc2)
idx, dist, good = K.knn2(feat, q, 0.7)
assert idx[7, 0] == 10 and idx[7, 1] == 1501 and dist[7, 0] == dist[7, 1]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "knn_small.npz"), desc_off=off2, descriptors=desc2, queries=q, features=feat, indices=idx,
                    dists=dist, good=good)
print("points", len(off2) - 1, "views", len(desc2), "queries", len(q), "good", int(good.sum()))
