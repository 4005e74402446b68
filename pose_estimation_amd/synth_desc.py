"""Synthetic SIFT-like descriptors for the matching tests and benchmark: uint8 vectors of 128
dimensions with the norm of real SIFT descriptors (about 512), a model of points seen in a few
views each (a view's descriptor is the point's prototype plus noise), and query descriptors of which
a share are noisy views of model points (they pass the ratio test) and the rest are unrelated."""
import numpy as np


def _sift_like(rng, n):
    v = rng.gamma(0.6, 1.0, size=(n, 128))
    v *= 512.0 / np.linalg.norm(v, axis=1, keepdims=True)
    return np.clip(np.rint(v), 0, 255)


def make_model(npoints, seed, max_views=6, noise=6.0):
    rng = np.random.default_rng(seed)
    proto = _sift_like(rng, npoints)
    views = rng.integers(2, max_views + 1, npoints)
    off = np.zeros(npoints + 1, dtype=np.uint32)
    off[1:] = np.cumsum(views)
    rep = np.repeat(np.arange(npoints), views)
    desc = np.clip(np.rint(proto[rep] + rng.normal(0.0, noise, (len(rep), 128))), 0, 255).astype(np.uint8)
    return dict(desc_off=off, descriptors=desc, proto=proto)


def make_queries(model, nq, seed, match_frac=0.4, noise=8.0):
    rng = np.random.default_rng(seed)
    npoints = len(model["proto"])
    nm = int(nq * match_frac)
    which = rng.choice(npoints, nm, replace=nm > npoints)
    matched = np.clip(np.rint(model["proto"][which] + rng.normal(0.0, noise, (nm, 128))), 0, 255)
    other = _sift_like(rng, nq - nm)
    q = np.concatenate([matched, other]).astype(np.uint8)
    truth = np.concatenate([which, np.full(nq - nm, -1)])
    perm = rng.permutation(nq)
    return q[perm], truth[perm]
