"""The elimination's trailing updates are v_fmac_f64_dpp instructions written in inline assembly,
where the compiler's hazard recognizer does not look: a DPP read of a register that a vector ALU
instruction wrote less than two wait states earlier would return the old value.  The operands
come straight from LDS reads, so no such pair should exist; this compiles the library's device
code to assembly and checks every one of them.  Runs without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_dpp_read_after_valu_write(tmp_path):
    asm = tmp_path / "pose_amd.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-fPIC", "-S",
                    "--cuda-device-only", "-Wno-unused-function", "-Wno-unused-command-line-argument", "-o", str(asm),
                    os.path.join(ROOT, "pose_estimation_amd", "csrc", "pose_amd.hip")], check=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazards.py"), str(asm)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    n = int(r.stdout.split()[0])
    assert n >= 800          # both kernels that contain the elimination (per-query and shared launches)
