"""Pins the C restatement (oracle/cpu_ref.c — the timed CPU baseline) against the golden vectors made by
the pure-Python oracle.  Pure CPU."""
import hashlib
import random
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import load_golden


@pytest.fixture(scope="module")
def ref():
    import cpu_ref
    cpu_ref.lib()
    return cpu_ref


def _scalars(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()


def _csr(rows):
    rp = np.zeros(len(rows) + 1, np.uint64)
    cols, coefs = [], []
    for i, row in enumerate(rows):
        for c, col in row:
            cols.append(col)
            coefs.append(int(c).to_bytes(32, "little"))
        rp[i + 1] = len(cols)
    return SimpleNamespace(row_ptr=rp, col=np.asarray(cols, np.uint32),
                           coeff=np.frombuffer(b"".join(coefs), np.uint8).copy() if coefs else np.zeros(0, np.uint8))


def _rows(j):
    return tuple([[(int(c, 16), col) for c, col in row] for row in mat] for mat in j)


def _pk(j):
    a = lambda h: np.frombuffer(bytes.fromhex(h), dtype=np.uint8).copy()
    vk = SimpleNamespace(alpha_g1=a(j["alpha_g1"]), beta_g2=a(j["beta_g2"]), delta_g2=a(j["delta_g2"]))
    return SimpleNamespace(vk=vk, beta_g1=a(j["beta_g1"]), delta_g1=a(j["delta_g1"]), a_query=a(j["a_query"]),
                           b_g1_query=a(j["b_g1_query"]), b_g2_query=a(j["b_g2_query"]), h_query=a(j["h_query"]),
                           l_query=a(j["l_query"]))


@pytest.mark.parametrize("threads", [1, 4])
def test_prove_d8_golden(ref, threads):
    g = load_golden("groth16_d8.json")
    mats = [_csr(m) for m in _rows(g["matrices"])]
    w = _scalars([int(x, 16) for x in g["witness"]])
    for case in g["proofs"]:
        got = ref.prove(_pk(g["pk"]), mats, g["num_inputs"], g["num_constraints"], g["num_variables"], w,
                        int(case["r"], 16), int(case["s"], 16), nthreads=threads)
        assert got.hex() == case["proof"]


def test_witness_map_golden(ref):
    for name in ("groth16_d8.json", "groth16_tiny.json"):
        g = load_golden(name)
        mats = [_csr(m) for m in _rows(g["matrices"])]
        w = _scalars([int(x, 16) for x in g["witness"]])
        h = ref.witness_map(mats, g["num_inputs"], g["num_constraints"], g["num_variables"], w, nthreads=2)
        assert hashlib.sha256(h.tobytes()).hexdigest() == g["h_sha256"]


def test_ntt_golden(ref, oracle):
    for c in load_golden("ntt.json")["cases"]:
        n = 1 << c["log_n"]
        if c["seed_values"] is not None:
            v = [int(x, 16) for x in c["seed_values"]]
        else:
            r = random.Random(c["rng_seed"])
            v = [r.randrange(oracle.R) for _ in range(n)]
        d = _scalars(v)
        outs = dict(fft=ref.ntt(d), ifft=ref.ntt(d, inverse=True), coset_fft=ref.ntt(d, coset=True),
                    coset_ifft=ref.ntt(d, inverse=True, coset=True, nthreads=3))
        for k, val in outs.items():
            assert hashlib.sha256(val.tobytes()).hexdigest() == c["outputs_sha256"][k], (c["log_n"], k)


@pytest.mark.parametrize("logn", [1, 5, 6, 7, 9, 12])
def test_ntt_four_step_sizes_vs_oracle(ref, oracle, logn):
    """the C restatement's transform is the four-step arrangement (n = n1·n2, independent row transforms between
    transposes) from 2^6 up: odd exponents (n1 != n2), every thread count that does not divide the rows, all four
    variants, against the plain big-int transform (ark-poly semantics, r1cs_to_qap.rs:179-185,198-199,210)"""
    r = random.Random(77 + logn)
    v = [r.randrange(oracle.R) for _ in range(1 << logn)]
    d = _scalars(v)
    want = dict(fft=oracle.fft(v), ifft=oracle.ifft(v), coset_fft=oracle.coset_fft(v), coset_ifft=oracle.coset_ifft(v))
    for nt in (1, 3, 7):
        got = dict(fft=ref.ntt(d, nthreads=nt), ifft=ref.ntt(d, inverse=True, nthreads=nt), coset_fft=ref.ntt(d, coset=True, nthreads=nt),
                   coset_ifft=ref.ntt(d, inverse=True, coset=True, nthreads=nt))
        for k in want:
            assert got[k].tobytes() == _scalars(want[k]).tobytes(), (logn, nt, k)


def test_external_vectors_through_the_c_restatement(ref, oracle):
    """the external known answers of tests/test_oracle_kats.py (3·G1, 2·G2, the published 2^28-th root of unity) through
    oracle/cpu_ref.c, the timed CPU baseline: the third implementation that must agree with them"""
    from test_oracle_kats import G1_TIMES_3, G2_TIMES_2, FR_ROOT_OF_UNITY_2_28
    g = oracle.g1_packed(oracle.G1_GEN)
    g2 = oracle.g1_packed(oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, 2)))
    want3 = G1_TIMES_3[0].to_bytes(32, "little") + G1_TIMES_3[1].to_bytes(32, "little")
    assert ref.msm_g1(g + g2, _scalars([1, 1])) == want3 and ref.msm_g1(g, _scalars([3])) == want3
    (x0, x1), (y0, y1) = G2_TIMES_2
    want2 = b"".join(v.to_bytes(32, "little") for v in (x0, x1, y0, y1))
    h = oracle.g2_packed(oracle.G2_GEN)
    assert ref.msm_g2(h * 2, _scalars([1, 1])) == want2 and ref.msm_g2(h, _scalars([2])) == want2
    for logn in (2, 9, 12):
        n = 1 << logn
        out = ref.ntt(_scalars([0, 1] + [0] * (n - 2)), nthreads=3).tobytes()
        w = pow(FR_ROOT_OF_UNITY_2_28, 1 << (28 - logn), oracle.R)
        for k in (0, 1, 2, n // 2 + 1, n - 1):
            assert int.from_bytes(out[32 * k:32 * k + 32], "little") == pow(w, k, oracle.R)


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_msm_golden(ref, oracle, idx):
    c = load_golden("msm.json")["cases"][idx]
    ks = [int(x, 16) for x in c["base_dlogs"]]
    t1 = oracle.G1.fixed_base_table(oracle.G1_GEN, 8)
    g1 = oracle.G1.batch_to_affine([oracle.G1.fixed_base_mul(t1, k) for k in ks])
    b1 = b"".join(oracle.g1_packed(p) for p in g1)
    sc = _scalars([int(x, 16) for x in c["scalars"]])
    assert ref.msm_g1(b1, sc, nthreads=1).hex() == c["g1_result"]
    assert ref.msm_g1(b1, sc, nthreads=4).hex() == c["g1_result"]
    if c["n"] <= 33:
        t2 = oracle.G2.fixed_base_table(oracle.G2_GEN, 6)
        g2 = oracle.G2.batch_to_affine([oracle.G2.fixed_base_mul(t2, k) for k in ks])
        assert ref.msm_g2(b"".join(oracle.g2_packed(p) for p in g2), sc, nthreads=2).hex() == c["g2_result"]
