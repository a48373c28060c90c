"""The host side of a matrix's way to the GPU (csrc/csr_host.hpp: coefficient dictionary, sliced layout, every pass on all
host threads) checked on the HOST: tests/cpp/test_csr_host.cpp executes the layout the way the sparse-product kernel walks it
and compares with a direct evaluation of the rows (evaluate_constraint, forks/groth16/src/r1cs_to_qap.rs:16-45).  Pure CPU."""
import os
import subprocess

from conftest import ROOT


def test_sliced_layout_and_dictionary_against_direct_row_evaluation(tmp_path):
    exe = str(tmp_path / "test_csr_host")
    src = os.path.join(ROOT, "tests", "cpp", "test_csr_host.cpp")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:]
    # and on one thread (the ranges collapse to one: the count / scan / fill passes must not depend on the cut)
    r = subprocess.run(["taskset", "-c", "0", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:]
