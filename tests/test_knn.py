"""Descriptor matching in front of the RANSAC path (query_processor_basic.cpp:22-96): the NumPy
check against the committed fixture (CPU), and the matrix-core search against both (GPU)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import knn_oracle as K   # noqa: E402

from pose_estimation_amd import synth_desc   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "knn_small.npz")


def fixture():
    return np.load(GOLD)


def test_oracle_reproduces_fixture():
    g = fixture()
    feat = K.average_views(g["desc_off"], g["descriptors"])
    assert np.array_equal(feat, g["features"], equal_nan=True)
    assert np.isnan(feat[1500]).all()                                  # the point without views
    idx, dist, good = K.knn2(feat, g["queries"], 0.7)
    assert np.array_equal(idx, g["indices"]) and np.array_equal(dist, g["dists"]) and np.array_equal(good, g["good"])
    assert tuple(idx[7]) == (10, 1501) and dist[7, 0] == dist[7, 1]    # exact tie: lower index first
    assert 100 < good.sum() < 300


def test_oracle_against_direct_loops():
    """The chunked matrix form against the definition, two nearest by (distance, index)."""
    rng = np.random.default_rng(3)
    f = rng.uniform(0, 255, (40, 128)).astype(np.float32)
    f[17] = f[3]                                                       # a tie
    q = rng.integers(0, 256, (9, 128)).astype(np.uint8)
    q[0] = np.rint(f[3]).astype(np.uint8)
    idx, dist, good = K.knn2(f, q, 0.7)
    for r in range(len(q)):
        d = ((q[r].astype(np.float64)[None, :] - f.astype(np.float64)) ** 2).sum(axis=1)
        order = np.lexsort((np.arange(len(f)), d))[:2]
        assert tuple(idx[r]) == tuple(order)
        assert np.allclose(dist[r], d[order].astype(np.float32), rtol=1e-6)
        assert good[r] == (np.float32(d[order[0]]) / np.float32(d[order[1]]) < np.float32(0.7))


def assert_same_matches(got, idx, dist, good, margin=None, ratio=0.7):
    sure = np.ones(len(idx), dtype=bool) if margin is None else margin > 1e-9
    assert sure.mean() > 0.99
    assert np.array_equal(got["indices"][sure], idx[sure])
    assert np.allclose(got["dists"], dist, rtol=3e-7, atol=0.0)        # float32 roundings of float64 sums in two summation orders
    with np.errstate(invalid="ignore", divide="ignore"):
        edge = np.abs(dist[:, 0] / dist[:, 1] - np.float32(ratio)) < 1e-5
    assert edge.sum() <= 2
    assert np.array_equal(got["good"][~edge], good[~edge])


@pytest.mark.gpu
def test_fixture_on_the_device(ctx):
    g = fixture()
    feat = ctx.knn_set_model_views(g["desc_off"], g["descriptors"])
    assert np.array_equal(feat, g["features"], equal_nan=True)         # float sum / float count, bit for bit
    got = ctx.knn_match(g["queries"], 0.7)
    assert_same_matches(got, g["indices"], g["dists"], g["good"])
    assert tuple(got["indices"][7]) == (10, 1501)
    # the same model handed over as float descriptors
    ctx.knn_set_model(np.nan_to_num(g["features"], nan=0.0)[:1500])
    again = ctx.knn_match(g["queries"][:130], 0.7)
    idx, dist, good = K.knn2(np.nan_to_num(g["features"], nan=0.0)[:1500], g["queries"][:130], 0.7)
    assert_same_matches(again, idx, dist, good)


@pytest.mark.gpu
def test_large_model_against_the_oracle(ctx):
    """60 000 points (not a multiple of the tile), 1 500 query descriptors (not a multiple of the
    block): indices, float distances and ratio-test outcomes equal the exact CPU search; nearly every
    row is certified by the matrix-core scan."""
    model = synth_desc.make_model(60000 + 13, seed=11)
    q, truth = synth_desc.make_queries(model, 1500 + 7, seed=12)
    feat = ctx.knn_set_model_views(model["desc_off"], model["descriptors"])
    assert np.array_equal(feat, K.average_views(model["desc_off"], model["descriptors"]))
    got = ctx.knn_match(q, 0.7)
    idx, dist, good = K.knn2(feat, q, 0.7)
    assert_same_matches(got, idx, dist, good, K.margins(feat, q, idx))
    planted = truth >= 0
    assert (got["indices"][planted, 0] == truth[planted]).mean() > 0.98 and got["good"][planted].mean() > 0.9
    assert got["good"][~planted].mean() < 0.05
    assert got["exhaustive_rows"] <= len(q) // 50


@pytest.mark.gpu
def test_ties_beyond_the_kept_scores_are_searched_exhaustively(ctx, monkeypatch):
    """Twenty identical model descriptors in one slice: the scan keeps eight and cannot certify
    which two are nearest; the exhaustive search returns the two lowest indices."""
    monkeypatch.setenv("POSE_AMD_KNN_SLICES", "1")
    rng = np.random.default_rng(8)
    f = rng.uniform(0, 200, (5000, 128)).astype(np.float32)
    clones = rng.choice(5000, 20, replace=False)
    f[clones] = f[clones[0]]
    q = rng.integers(0, 200, (70, 128)).astype(np.uint8)
    q[5] = np.rint(f[clones[0]]).astype(np.uint8)
    ctx.knn_set_model(f)
    got = ctx.knn_match(q, 0.7)
    idx, dist, good = K.knn2(f, q, 0.7, shortlist=40)
    assert tuple(got["indices"][5]) == tuple(np.sort(clones)[:2]) == tuple(idx[5])
    assert got["exhaustive_rows"] >= 1
    assert_same_matches(got, idx, dist, good, K.margins(f, q, idx) + (np.arange(70) == 5))


@pytest.mark.gpu
def test_edges(ctx):
    rng = np.random.default_rng(9)
    with pytest.raises(Exception):
        ctx.knn_set_model(np.zeros((0, 128), dtype=np.float32))
    f = rng.uniform(0, 255, (1, 128)).astype(np.float32)
    ctx.knn_set_model(f)                                               # one point: no second neighbour
    got = ctx.knn_match(rng.integers(0, 256, (3, 128)).astype(np.uint8))
    assert (got["indices"][:, 0] == 0).all() and (got["indices"][:, 1] == -1).all() and np.isinf(got["dists"][:, 1]).all()
    assert len(ctx.knn_match(np.zeros((0, 128), dtype=np.uint8))["indices"]) == 0
    f = rng.uniform(0, 255, (33, 128)).astype(np.float32)              # one full tile and one point
    q = rng.integers(0, 256, (129, 128)).astype(np.uint8)              # one full block and one row
    ctx.knn_set_model(f)
    idx, dist, good = K.knn2(f, q, 0.7)
    assert_same_matches(ctx.knn_match(q), idx, dist, good)
