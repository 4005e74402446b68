"""The C++ host facade (include/pose_amd.hpp) through its own test driver, tests/cpp/facade_test.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "facade_test")


def _exe():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", ROOT, "tests/cpp/facade_test"])
    return EXE


def test_facade_refuses_to_run_without_gpu():
    """Without a device the facade reports the error; it has no CPU path to fall back to."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([_exe()], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 2 and b"pose_amd_create" in r.stdout


def test_io_readers_and_golden_pose_convention(tmp_path):
    """Lowe .key reader, bundle.out reader and LoadGoldenQuery of include/pose_amd_io.hpp (host only)."""
    exe = os.path.join(ROOT, "tests", "cpp", "io_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "tests/cpp/io_test"])
    r = subprocess.run([exe, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
    assert r.returncode == 0 and b"io ok" in r.stdout, r.stdout.decode()


@pytest.mark.gpu
def test_facade_on_gpu(tmp_path):
    r = subprocess.run([_exe(), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()
    assert b"facade ok" in r.stdout
