#!/usr/bin/env python3
"""Dump the three matrices of a synthetic circuit of a BASELINE shape for tests/cpp/test_csr_host.cpp's `bench` mode
(the host side of cg_circuit_load timed without a GPU):  python tools/dump_matrices.py rs256-sd /tmp/m.bin"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crescent_credentials_amd import workloads as wl  # noqa: E402

shape, out = sys.argv[1], sys.argv[2]
l, m, M = wl.SHAPES[shape]
cm, _ = wl.synthetic_circuit(0xC5E5CE47 + 3, l, m, M, 0.9, 3, profile="gates")
with open(out, "wb") as f:
    np.array([m, M, 3], np.uint64).tofile(f)
    for mat in (cm.a, cm.b, cm.c):
        np.array([mat.col.size], np.uint64).tofile(f)
        mat.row_ptr.astype(np.uint64).tofile(f)
        mat.col.tofile(f)
        mat.coeff.tofile(f)
print("wrote", out)
