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
