"""Synthetic workloads for tests and bench.py: seeded satisfiable R1CS instances of the
reference circuits' SHAPE, produced by the host-only generator `synth/synth.cpp` (libcg_synth.so)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .api import ConstraintMatrices, _Csr

_HERE = os.path.dirname(os.path.abspath(__file__))
_SYNTH_PATH = os.path.join(_HERE, "libcg_synth.so")

# shapes of BASELINE.json's configs: name -> (num_inputs ℓ, num_constraints m, num_variables M)
SHAPES = {
    "rs256": (20, 1_480_000, 1_500_000),       # S21, D = 2^21
    "rs256-sd": (26, 1_480_000, 1_500_000),    # S21 with ℓ = 26
    "rs256-sd-large": (26, 3_000_000 - 26 - 20_000, 3_000_000),  # S22 stress variant
    "rs256-db": (28, 1_480_000, 1_500_000),
    "mdl1": (21, 2_980_000, 3_000_000),        # S22, D = 2^22
    "rs256-sd-eighth": (26, 185_000, 187_500),   # D = 2^18: the rs256-sd proportions at an eighth of the size (one-thread CPU sample)
    "tiny": (4, 200, 240),
    "small": (6, 3_000, 3_100),                # D = 2^12
    "medium": (20, 60_000, 61_000),            # D = 2^16
}

_synth = None


def _lib():
    global _synth
    if _synth is None:
        if not os.path.exists(_SYNTH_PATH):
            raise RuntimeError("%s not built; run __graft_entry__.build()" % _SYNTH_PATH)
        L = C.CDLL(_SYNTH_PATH)
        L.cgs_generate.restype = C.c_void_p
        L.cgs_generate.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_uint32]
        L.cgs_generate_gates.restype = C.c_void_p
        L.cgs_generate_gates.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_uint32]
        L.cgs_views.restype = None
        L.cgs_views.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
        L.cgs_free.restype = None
        L.cgs_free.argtypes = [C.c_void_p]
        _synth = L
    return _synth


def synthetic_circuit(seed: int, num_inputs: int, num_constraints: int, num_variables: int,
                      bit_fraction: float = 0.9, lc_terms: int = 3, profile: str = "r1"):
    """-> (ConstraintMatrices, witness uint8[M*32] canonical).  bit_fraction 0.9 = 'circom-like' wires
    (45 % zero / 45 % one / 10 % uniform), 0.0 = all-uniform wires.
    profile "r1": booleanity + short product rows (≈3.4 terms per row over A, B, C; lc_terms = mean terms of a
    product row's side); profile "gates": the circomlib gate mix of synth.cpp's cgs_generate_gates (≈11.5 terms
    per row as the real main_c.r1cs files have; lc_terms is ignored: the bigint rows get `gates_limb_terms` limbs a side)."""
    L = _lib()
    if profile == "gates":
        h = L.cgs_generate_gates(seed, num_inputs, num_constraints, num_variables, bit_fraction, gates_limb_terms(bit_fraction))
    elif profile == "r1":
        h = L.cgs_generate(seed, num_inputs, num_constraints, num_variables, bit_fraction, lc_terms)
    else:
        raise ValueError("profile must be 'r1' or 'gates'")
    if not h:
        raise ValueError("unsupported shape (need num_variables - num_inputs >= num_constraints >= 1)")
    try:
        rp = (C.c_void_p * 3)(); col = (C.c_void_p * 3)(); coeff = (C.c_void_p * 3)()
        nnz = (C.c_uint64 * 3)(); wit = C.c_void_p()
        L.cgs_views(h, rp, col, coeff, nnz, C.byref(wit))
        mats = []
        for k in range(3):
            n = int(nnz[k])
            r = np.ctypeslib.as_array(C.cast(rp[k], C.POINTER(C.c_uint64)), shape=(num_constraints + 1,)).copy()
            if n:
                c = np.ctypeslib.as_array(C.cast(col[k], C.POINTER(C.c_uint32)), shape=(n,)).copy()
                f = np.ctypeslib.as_array(C.cast(coeff[k], C.POINTER(C.c_uint8)), shape=(n * 32,)).copy()
            else:
                c = np.zeros(0, np.uint32); f = np.zeros(0, np.uint8)
            mats.append(_Csr(r, c, f))
        w = np.ctypeslib.as_array(C.cast(wit, C.POINTER(C.c_uint8)), shape=(num_variables * 32,)).copy()
    finally:
        L.cgs_free(h)
    cm = ConstraintMatrices(mats[0], mats[1], mats[2], num_inputs, num_variables - num_inputs, num_constraints)
    return cm, w


TERMS_PER_ROW = 11.5     # 17 M terms over 1.48 M rows (from the 595 MB main_c.r1cs)


def gates_limb_terms(bit_fraction: float) -> int:
    """Limbs per side of the bigint product rows such that the whole instance keeps ≈ TERMS_PER_ROW terms per row
    whatever share of the rows are bit gates (those average ≈ 8.5 terms): 17 at bit_fraction 0.9, the RSA-2048
    limb count of circuit_setup/scripts/prepare_setup.py:42-43."""
    bf = min(max(bit_fraction, 0.0), 0.97)
    per_product_row = (TERMS_PER_ROW - bf * 8.5) / (1.0 - bf)
    return max(2, int(round((per_product_row - 3.0) / 2.1)))


def wire_stats(w: np.ndarray) -> dict:
    """share of zero / one / other wires of a canonical witness (drives the assignment MSMs' digit counts)"""
    W = w.reshape(-1, 32)
    rest_zero = ~W[:, 1:].any(axis=1)
    zero = float((rest_zero & (W[:, 0] == 0)).mean())
    one = float((rest_zero & (W[:, 0] == 1)).mean())
    return {"zero": round(zero, 4), "one": round(one, 4), "other": round(1.0 - zero - one, 4)}


def matrices_to_rows(cm: ConstraintMatrices):
    """CSR -> the Vec<Vec<(coeff, column)>> form the Python oracle takes (small circuits only)."""
    out = []
    for m in (cm.a, cm.b, cm.c):
        rows = []
        for i in range(len(m.row_ptr) - 1):
            lo, hi = int(m.row_ptr[i]), int(m.row_ptr[i + 1])
            rows.append([(int.from_bytes(m.coeff[32 * t:32 * t + 32].tobytes(), "little"), int(m.col[t])) for t in range(lo, hi)])
        out.append(rows)
    return tuple(out)


def witness_to_ints(w: np.ndarray):
    b = w.tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]
