"""CPU check of the descriptor matching in front of the RANSAC path.  TEST INFRASTRUCTURE ONLY:
tests/ and bench.py's cpu_baseline import it, the product never does.

Restates what the reference computes around its FLANN index, with the search itself exact:
  * average_views: query_processor_basic.cpp:30-46 -- per 3-D point the float sum of the uint8 SIFT
    descriptors of its views divided by their number (float arithmetic; the sums are integers below
    2^24, so any summation order gives the reference's value, the division rounds once);
  * knn2: query_processor_basic.cpp:60-96 -- nn = 2 by squared L2 distance (flann::L2<float>), then
    Lowe's ratio test dists[0] / dists[1] < ratio on the float distances.  The reference searches
    4 randomized kd-trees with 128 checks (approximate, not reproducible); this is the exact answer
    that search approximates.  Ties are broken by the lower index.
Parity unpinned by any vector of the reference (it holds none for the matching step); pinned by the
committed fixture tests/golden/knn_small.npz, which tools/gen_knn_golden.py generates with this file.
"""
import numpy as np

DIM = 128


def average_views(desc_off, descriptors):
    off = np.asarray(desc_off, dtype=np.int64)
    d = np.asarray(descriptors, dtype=np.uint8).reshape(-1, DIM)
    out = np.empty((len(off) - 1, DIM), dtype=np.float32)
    for i in range(len(off) - 1):
        b, e = off[i], off[i + 1]
        s = d[b:e].astype(np.float32).sum(axis=0, dtype=np.float32)
        with np.errstate(invalid="ignore", divide="ignore"):
            out[i] = s / np.float32(e - b)
    return out


def knn2(features, query_desc, ratio=0.7, chunk=512, shortlist=6):
    """indices [nq, 2] int32, dists [nq, 2] float32, good [nq] bool."""
    f = np.asarray(features, dtype=np.float32).reshape(-1, DIM).astype(np.float64)
    q = np.asarray(query_desc, dtype=np.uint8).reshape(-1, DIM).astype(np.float64)
    m, nq = len(f), len(q)
    fn = np.einsum("ij,ij->i", f, f)
    idx = np.full((nq, 2), -1, dtype=np.int32)
    dist = np.full((nq, 2), np.inf, dtype=np.float32)
    k = min(shortlist, m)
    for r0 in range(0, nq, chunk):
        qc = q[r0:r0 + chunk]
        s = fn[None, :] - 2.0 * (qc @ f.T)                        # ranks like the distance; NaN rows of the model never win
        s = np.where(np.isnan(s), np.inf, s)
        cand = np.argpartition(s, k - 1, axis=1)[:, :k] if k < m else np.tile(np.arange(m), (len(qc), 1))
        for r in range(len(qc)):
            c = np.sort(cand[r])
            diff = qc[r][None, :] - f[c]
            d = np.einsum("ij,ij->i", diff, diff)                 # exact re-evaluation of the shortlist
            d = np.where(np.isnan(d), np.inf, d)
            order = np.lexsort((c, d))[:2]
            for j, o in enumerate(order):
                if np.isfinite(d[o]):
                    idx[r0 + r, j] = c[o]
                    dist[r0 + r, j] = np.float32(d[o])
    with np.errstate(invalid="ignore", divide="ignore"):
        good = (dist[:, 0] / dist[:, 1]) < np.float32(ratio)
    return idx, dist, good


def margins(features, query_desc, idx):
    """Relative gap between the second and the third nearest distance of every row (a test only
    insists on the index pair where this gap is not at rounding level)."""
    f = np.asarray(features, dtype=np.float32).reshape(-1, DIM).astype(np.float64)
    q = np.asarray(query_desc, dtype=np.uint8).reshape(-1, DIM).astype(np.float64)
    fn = np.einsum("ij,ij->i", f, f)
    out = np.empty(len(q))
    for r0 in range(0, len(q), 512):
        qc = q[r0:r0 + 512]
        d = fn[None, :] - 2.0 * (qc @ f.T) + np.einsum("ij,ij->i", qc, qc)[:, None]
        d = np.where(np.isnan(d), np.inf, d)
        part = np.sort(np.partition(d, min(3, d.shape[1] - 1), axis=1)[:, :min(4, d.shape[1])], axis=1)
        gap12 = part[:, 1] - part[:, 0]
        gap23 = part[:, 2] - part[:, 1] if part.shape[1] > 2 else np.full(len(qc), np.inf)
        out[r0:r0 + 512] = np.minimum(gap12, gap23) / np.maximum(part[:, 1], 1.0)
    return out
