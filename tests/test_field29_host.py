"""The lazy 29-bit-limb arithmetic the hot kernels run (csrc/field29.hpp, csrc/curve29.hpp) checked on the HOST
against the saturated 8x32 Montgomery arithmetic (csrc/field.hpp, csrc/curve.hpp) — g++ build of
tests/cpp/test_field29.cpp — and its bound analysis (tools/bounds29.py).  Pure CPU."""
import os
import subprocess
import sys

from conftest import ROOT


def test_bounds_checker_passes():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bounds29.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all bounds hold" in r.stdout


def test_field29_and_curve29_against_saturated_arithmetic(tmp_path):
    exe = str(tmp_path / "test_field29")
    src = os.path.join(ROOT, "tests", "cpp", "test_field29.cpp")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:]
