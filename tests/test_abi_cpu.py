"""CPU tests of the boundary: the shared library loads, exports every symbol the header
declares, and reports a clean error (not a crash, not a CPU fallback) when no GPU exists."""
import ctypes as C
import os
import re

import pytest

from pose_estimation_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "pose_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pose_amd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/pose_amd.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms      # the Python binding covers the whole ABI


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.Params) == 32
    assert C.sizeof(_lib.Result) == 4 + 4 + 8 + 24 + 72 + 4 + 4 + 8 + 8 + 4 + 4 + 4 + 4 + 8


def test_no_gpu_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by the gpu tests")
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.pose_amd_create(0, C.byref(h))
    assert rc != 0 and not h.value
    assert b"HIP" in lib.pose_amd_last_error(None) or b"device" in lib.pose_amd_last_error(None)
    from pose_estimation_amd import Context, PoseAmdError
    with pytest.raises(PoseAmdError):
        Context(0)


def test_query_partition_is_the_sharding_rule():
    """pose_amd_partition_queries (host code, spreads a batch over the devices of one process) assigns
    exactly as shard.query_shard assigns queries to ranks; loads are balanced."""
    import numpy as np
    from pose_estimation_amd import shard
    lib = _lib.load()
    rng = np.random.default_rng(4)
    for nq, parts in ((0, 3), (1, 4), (50, 2), (800, 8), (37, 5)):
        costs = np.ascontiguousarray(rng.integers(4000, 6001, nq), dtype=np.int32)
        if nq > 10:
            costs[::7] = costs[0]                                   # equal costs: ties go to the lower index / lower part
        part = np.full(max(nq, 1), -1, dtype=np.int32)
        rc = lib.pose_amd_partition_queries(nq, costs.ctypes.data_as(_lib._ip), parts, part.ctypes.data_as(_lib._ip))
        assert rc == 0
        for r in range(parts):
            assert [int(i) for i in np.nonzero(part[:nq] == r)[0]] == shard.query_shard(costs, r, parts)
        if nq >= 50:
            loads = [int(costs[part[:nq] == r].sum()) for r in range(parts)]
            assert max(loads) - min(loads) <= costs.max()
    assert lib.pose_amd_partition_queries(3, None, 2, None) != 0 and lib.pose_amd_partition_queries(3, None, 0, None) != 0


def test_facades_have_the_reference_processor_interface():
    """query_processor.hpp:33-73: Process, Project, TestHypothesis and the (empty) BundleAdjust."""
    from pose_estimation_amd import api
    for name in ("Project", "TestHypothesis", "BundleAdjust"):
        assert callable(getattr(api.QueryProcessor, name))
    for cls in (api.QueryProcessorBasic, api.QueryProcessorAdvanced):
        assert callable(cls.Process) and issubclass(cls, api.QueryProcessor)
    # query_processor.cpp:36-44 returns true and does nothing; so does the default here (no device needed)
    qp = api.QueryProcessor.__new__(api.QueryProcessor)
    assert api.QueryProcessor.local_optimisation is False and qp.BundleAdjust(None, None) is True and qp.lo_samples is None
    for name in ("Process(Query &query) override", "QueryProcessor(parse_bundler &parsebundler", "class QueryLoader"):
        assert name in open(os.path.join(os.path.dirname(__file__), "..", "include", "pose_amd.hpp")).read()
    hpp = open(os.path.join(os.path.dirname(__file__), "..", "include", "pose_amd.hpp")).read()
    for name in ("bool Project(", "TestHypothesis(", "bool BundleAdjust(", "virtual bool Process("):
        assert name in hpp
