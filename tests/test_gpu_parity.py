"""Parity of the HIP path (through the C ABI) with the oracle and the golden vectors.  Needs an MI355X.

Bit-exact comparisons throughout: this path is integer arithmetic."""
import hashlib
import random
import time

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _rows(j):
    return tuple([[(int(c, 16), col) for c, col in row] for row in mat] for mat in j)


def _scalars(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()


def _ints(buf):
    b = bytes(buf)
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def _sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module", autouse=True)
def _init(cc):
    rc = cc.lib().cg_init(0, None)
    assert rc == 0, cc.lib().cg_last_error()


# ------------------------------------------------------------------------------------------- NTT
@pytest.mark.parametrize("case", load_golden("ntt.json")["cases"], ids=lambda c: "log%d" % c["log_n"])
def test_ntt_golden(cc, oracle, case):
    n = 1 << case["log_n"]
    if case["seed_values"] is not None:
        v = [int(x, 16) for x in case["seed_values"]]
    else:
        r = random.Random(case["rng_seed"])
        v = [r.randrange(oracle.R) for _ in range(n)]
    data = _scalars(v)
    outs = dict(fft=cc.fft_in_place(data), ifft=cc.ifft_in_place(data), coset_fft=cc.fft_in_place(data, coset=True),
                coset_ifft=cc.ifft_in_place(data, coset=True))
    for k, val in outs.items():
        assert _sha(val) == case["outputs_sha256"][k], k


@pytest.mark.parametrize("logn", [4, 9, 11, 12, 13, 16, 17])
def test_ntt_vs_oracle_sizes(cc, oracle, logn):
    """every pass-plan shape: single tile, tile + 1..6-stage strided pass, two strided passes"""
    r = random.Random(logn)
    n = 1 << logn
    v = [r.randrange(oracle.R) for _ in range(n)]
    got = _ints(cc.fft_in_place(_scalars(v)))
    assert got == oracle.fft(v)
    if logn <= 13:
        assert _ints(cc.ifft_in_place(_scalars(v), coset=True)) == oracle.coset_ifft(v)


@pytest.mark.parametrize("logn", [20, 22])
def test_ntt_roundtrip_and_linearity_large(cc, oracle, logn):
    """size-independent properties at BASELINE sizes: ifft(fft(x)) = x, coset too, and linearity"""
    rng = np.random.default_rng(logn)
    n = 1 << logn
    def rand_elems():
        a = rng.integers(0, 256, size=n * 32, dtype=np.uint8)
        a.reshape(n, 32)[:, 31] &= 0x1F          # < 2^253 < r
        return a
    x, y = rand_elems(), rand_elems()
    fx = cc.fft_in_place(x)
    assert np.array_equal(cc.ifft_in_place(fx), x)
    assert np.array_equal(cc.ifft_in_place(cc.fft_in_place(x, coset=True), coset=True), x)
    # spot-check F(x)[k] against the definition for one k via a sparse input: x = e_j -> F[k] = w^{jk}
    e = np.zeros(n * 32, np.uint8); j = 12345 % n; e[32 * j] = 1
    fe = cc.fft_in_place(e)
    w = oracle.root_of_unity(n)
    for k in (0, 1, 777 % n, n - 1):
        assert int.from_bytes(fe[32 * k:32 * k + 32].tobytes(), "little") == pow(w, j * k, oracle.R)
    # linearity on a few coordinates
    xi, yi = _ints(x[:32 * 4]), _ints(y[:32 * 4])
    s = _scalars([(a + b) % oracle.R for a, b in zip(_ints(x), _ints(y))]) if logn <= 20 else None
    if s is not None:
        fs, fy = cc.fft_in_place(s), cc.fft_in_place(y)
        for k in (0, 5, n // 2 + 3):
            g = lambda a: int.from_bytes(a[32 * k:32 * k + 32].tobytes(), "little")
            assert g(fs) == (g(fx) + g(fy)) % oracle.R


# ------------------------------------------------------------------------------------------- MSM
def _msm_case_bases(oracle, c):
    ks = [int(x, 16) for x in c["base_dlogs"]]
    t1 = oracle.G1.fixed_base_table(oracle.G1_GEN, 8)
    t2 = oracle.G2.fixed_base_table(oracle.G2_GEN, 8)
    g1 = oracle.G1.batch_to_affine([oracle.G1.fixed_base_mul(t1, k) for k in ks])
    g2 = oracle.G2.batch_to_affine([oracle.G2.fixed_base_mul(t2, k) for k in ks])
    b1 = b"".join(oracle.g1_packed(p) for p in g1)
    b2 = b"".join(oracle.g2_packed(p) for p in g2)
    assert _sha(b1) == c["g1_bases_sha256"] and _sha(b2) == c["g2_bases_sha256"]
    return b1, b2, _scalars([int(x, 16) for x in c["scalars"]])


@pytest.mark.parametrize("case", load_golden("msm.json")["cases"], ids=lambda c: "n%d" % c["n"])
def test_msm_golden(cc, oracle, case):
    b1, b2, sc = _msm_case_bases(oracle, case)
    for wb in (0, 4, 11):
        assert cc.msm_bigint_g1(b1, sc, window_bits=wb).hex() == case["g1_result"], "G1 c=%d" % wb
    assert cc.msm_bigint_g2(b2, sc).hex() == case["g2_result"]
    assert cc.msm_bigint_g2(b2, sc, window_bits=7).hex() == case["g2_result"]


def test_msm_montgomery_bases_and_truncation(cc, oracle):
    case = load_golden("msm.json")["cases"][2]
    b1, _, sc = _msm_case_bases(oracle, case)
    # arkworks in-memory (Montgomery) coordinates give the same result (zkey.rs:397-431 form)
    pts = [b1[i:i + 32] for i in range(0, len(b1), 32)]
    mont = b"".join(((int.from_bytes(p, "little") << 256) % oracle.Q).to_bytes(32, "little") for p in pts)
    assert cc.msm_bigint_g1(mont, sc, coord_form=1).hex() == case["g1_result"]
    # msm_bigint zips: extra scalars or extra bases are ignored
    n = case["n"]
    g1 = [oracle.g1_unpack(b1[64 * i:64 * i + 64]) for i in range(n)]
    scs = [int(x, 16) for x in case["scalars"]]
    exp = oracle.g1_packed(oracle.G1.to_affine(oracle.G1.msm(g1[:10], scs[:10]))).hex()
    assert cc.msm_bigint_g1(b1[:640], sc).hex() == exp
    assert cc.msm_bigint_g1(b1, sc[:320]).hex() == exp


def test_alt_bn128_doubling_vector_on_the_gpu(cc, oracle):
    """2·G1 of alt_bn128 as EIP-196's ecAdd test data has it (an external vector, tests/test_oracle_kats.py), through the
    fixed-base path and through both MSM entries of the HIP library"""
    x2 = 0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3
    y2 = 0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4
    want = x2.to_bytes(32, "little") + y2.to_bytes(32, "little")
    assert bytes(cc.fixed_base_g1(_scalars([2]))) == want
    g = oracle.g1_packed(oracle.G1_GEN)
    assert cc.msm_bigint_g1(g * 2, _scalars([1, 1])) == want
    assert cc.msm_bigint_g1(g, _scalars([2])) == want
    ctx = cc.MsmContext(g * 2, group=1)
    try:
        assert ctx.run(_scalars([1, 1])) == want
    finally:
        ctx.close()


def test_external_vectors_on_the_gpu(cc, oracle):
    """the other external known answers of tests/test_oracle_kats.py through the HIP library: 3·G1 = G1 + 2·G1 (a mixed
    addition of distinct points, via the MSM and the fixed-base path), 2·G2 (the Fq2 arithmetic of the G2 kernels), and the
    2^28-th root of unity that gnark-crypto publishes and ark-ff derives from generator 5 - seen through the transform of
    a unit impulse, whose k-th output is ω^k"""
    from test_oracle_kats import G1_TIMES_3, G2_TIMES_2, FR_ROOT_OF_UNITY_2_28
    want3 = G1_TIMES_3[0].to_bytes(32, "little") + G1_TIMES_3[1].to_bytes(32, "little")
    g = oracle.g1_packed(oracle.G1_GEN)
    g2 = bytes(cc.fixed_base_g1(_scalars([2])))
    assert bytes(cc.fixed_base_g1(_scalars([3]))) == want3
    assert cc.msm_bigint_g1(g + g2, _scalars([1, 1])) == want3
    assert cc.msm_bigint_g1(g, _scalars([3])) == want3
    (x0, x1), (y0, y1) = G2_TIMES_2
    want2 = b"".join(v.to_bytes(32, "little") for v in (x0, x1, y0, y1))
    h = oracle.g2_packed(oracle.G2_GEN)
    assert bytes(cc.fixed_base_g2(_scalars([2]))) == want2
    assert cc.msm_bigint_g2(h * 2, _scalars([1, 1])) == want2
    assert cc.msm_bigint_g2(h, _scalars([2])) == want2
    for logn in (2, 11, 21):
        n = 1 << logn
        delta = np.zeros(n * 32, np.uint8)
        delta[32] = 1                                    # the polynomial x
        out = bytes(cc.fft_in_place(delta))
        w = pow(FR_ROOT_OF_UNITY_2_28, 1 << (28 - logn), oracle.R)
        for k in (0, 1, 2, 3, n // 2 + 1, n - 1):
            assert int.from_bytes(out[32 * k:32 * k + 32], "little") == pow(w, k, oracle.R), (logn, k)
        # and on the coset: x evaluated at 5·ω^k
        outc = bytes(cc.fft_in_place(delta, coset=True))
        for k in (0, 1, n - 1):
            assert int.from_bytes(outc[32 * k:32 * k + 32], "little") == 5 * pow(w, k, oracle.R) % oracle.R, (logn, k)


def test_points_held_by_the_reference_tree_on_the_gpu(cc, oracle):
    """the G1 / G2 points of forks/halo2curves/src/bn256/curve.rs:307-420 through the HIP MSM: [r - 1]·P = -P for each
    (an identity of the group, no oracle value involved), with the default window and with small and wide ones, canonical and
    Montgomery-form bases; and a sum over all five against the oracle's serial sum"""
    from test_oracle_kats import h2c_points
    g1, g2 = h2c_points(oracle)
    rm1 = _scalars([oracle.R - 1])
    mont = lambda b, n: b"".join((int.from_bytes(b[32 * i:32 * i + 32], "little") * (1 << 256) % oracle.Q).to_bytes(32, "little") for i in range(n))
    for P in g1:
        b = oracle.g1_packed(P)
        want = oracle.g1_packed(oracle.G1.neg_affine(P))
        for wb in (0, 3, 13, 20):
            assert cc.msm_bigint_g1(b, rm1, window_bits=wb) == want, wb
        assert cc.msm_bigint_g1(mont(b, 2), rm1, coord_form=cc.api.CG_FORM_MONTGOMERY) == want
    for P in g2:
        b = oracle.g2_packed(P)
        want = oracle.g2_packed(oracle.G2.neg_affine(P))
        for wb in (0, 3, 13, 20):
            assert cc.msm_bigint_g2(b, rm1, window_bits=wb) == want, wb
        assert cc.msm_bigint_g2(mont(b, 4), rm1, coord_form=cc.api.CG_FORM_MONTGOMERY) == want
    ks = [oracle.R - 2, 1, 2, 0x1234567890ABCDEF << 100, 3]
    assert cc.msm_bigint_g1(b"".join(oracle.g1_packed(P) for P in g1), _scalars(ks)) == \
        oracle.g1_packed(oracle.G1.to_affine(oracle.G1.msm_naive(g1, ks)))
    assert cc.msm_bigint_g2(b"".join(oracle.g2_packed(P) for P in g2), _scalars(ks)) == \
        oracle.g2_packed(oracle.G2.to_affine(oracle.G2.msm_naive(g2, ks)))


def test_endomorphism_known_answer_on_the_gpu(cc, oracle):
    """[lambda]·(x, y) = (beta·x, y) (forks/halo2curves tests/curve.rs:413-427, constants fr.rs:12 / fq.rs:14): a full-width
    scalar multiplication with an answer stated by the reference's tree, through the HIP MSM - one-shot entry point with
    several windows, resident-table context, Montgomery-form bases, and with the point repeated 1000 times under scalars
    that add up to lambda"""
    from test_oracle_kats import endo_constants, h2c_points
    lam, beta = endo_constants()
    g1, _ = h2c_points(oracle)
    for P in [oracle.G1_GEN] + g1:
        want = oracle.g1_packed((beta * P[0] % oracle.Q, P[1]))
        b = oracle.g1_packed(P)
        for wb in (0, 2, 7, 16, 20):
            assert cc.msm_bigint_g1(b, _scalars([lam]), window_bits=wb) == want, wb
        for wb in (0, 22):                                   # resident window tables (the prover's arrangement)
            ctx = cc.MsmContext(b, group=1, window_bits=wb)
            try:
                assert ctx.run(_scalars([lam])) == want, wb
            finally:
                ctx.close()
    P = g1[0]
    rng = random.Random(5)
    parts = [rng.randrange(oracle.R) for _ in range(999)]
    parts.append((lam - sum(parts)) % oracle.R)
    assert cc.msm_bigint_g1(oracle.g1_packed(P) * 1000, _scalars(parts)) == oracle.g1_packed((beta * P[0] % oracle.Q, P[1]))


def test_msm_empty_and_all_zero(cc, oracle):
    assert cc.msm_bigint_g1(b"", b"") == bytes(64)
    g = oracle.g1_packed(oracle.G1_GEN)
    assert cc.msm_bigint_g1(g * 5, bytes(32 * 5)) == bytes(64)                      # all-zero scalars
    assert cc.msm_bigint_g1(bytes(64 * 5), _scalars([3, 4, 5, 6, 7])) == bytes(64)  # all-identity bases
    # P + (-P) = identity through the bucket path: scalars s and r - s on the same base
    s = 0x1234567890ABCDEF
    assert cc.msm_bigint_g1(g * 2, _scalars([s, oracle.R - s])) == bytes(64)
    with pytest.raises(cc.CrescentGpuError):
        cc.msm_bigint_g1(g, _scalars([oracle.R]))                                   # non-canonical scalar


def test_msm_large_closed_form(cc, oracle):
    """n = 2^16 with bases k_i·G made by the GPU setup path would be circular; instead use the
    closed form Σ s_i (k_i G) = (Σ s_i k_i) G with bases from the oracle's fixed-base table."""
    rng = random.Random(99)
    n = 3000
    ks = [rng.randrange(1, oracle.R) for _ in range(n)]
    sc = [rng.choice([0, 1, rng.randrange(oracle.R), rng.randrange(oracle.R)]) for _ in range(n)]
    t1 = oracle.G1.fixed_base_table(oracle.G1_GEN, 8)
    g1 = oracle.G1.batch_to_affine([oracle.G1.fixed_base_mul(t1, k) for k in ks])
    e = sum(k * s for k, s in zip(ks, sc)) % oracle.R
    exp = oracle.g1_packed(oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, e))).hex()
    b1 = b"".join(oracle.g1_packed(p) for p in g1)
    for wb in (0, 13, 16):
        assert cc.msm_bigint_g1(b1, _scalars(sc), window_bits=wb).hex() == exp


@pytest.mark.parametrize("group", [1, 2])
def test_msm_every_window_size(cc, oracle, group):
    """Every window size 2 .. 22 - each instantiation of the per-window digit walk (10 .. 22), the run-time walk below it,
    every shape of the bucket matrix (one row, fewer columns than a reduction chunk, 1024 columns) and both partition
    depths - through the one-shot entry (window-major keys, one bucket matrix per window) and through a resident context
    (shared buckets), against the closed form (Σ s_i k_i)·G.  Scalars: full width, 0, 1, r − 1 and short ones."""
    rng = random.Random(2200 + group)
    n = 700 if group == 1 else 300
    ks = [rng.randrange(1, oracle.R) for _ in range(n)]
    ks[3] = 0                                                               # an identity base
    bases = (cc.fixed_base_g1 if group == 1 else cc.fixed_base_g2)(_scalars(ks))
    one_shot = cc.msm_bigint_g1 if group == 1 else cc.msm_bigint_g2
    curve, gen, packed = (oracle.G1, oracle.G1_GEN, oracle.g1_packed) if group == 1 else (oracle.G2, oracle.G2_GEN, oracle.g2_packed)
    sc = [rng.choice([0, 1, oracle.R - 1, rng.randrange(1 << 20), rng.randrange(oracle.R), rng.randrange(oracle.R)]) for _ in range(n)]
    e = sum(k * s for k, s in zip(ks, sc)) % oracle.R
    exp = packed(curve.to_affine(curve.mul_affine(gen, e)))
    for wb in range(2, 23):
        if wb <= 21:
            assert one_shot(bases, _scalars(sc), window_bits=wb) == exp, ("one-shot", group, wb)
        else:   # window-major keys of 22-bit windows need 25 bits, one more than the two partition levels take: refused, not wrong
            with pytest.raises(cc.CrescentGpuError):
                one_shot(bases, _scalars(sc), window_bits=wb)
        if wb % 3 == 2 or wb >= 20:                                          # resident tables: a subset (each builds 255 / wb rows)
            ctx = cc.MsmContext(bases, group=group, window_bits=wb)
            try:
                assert ctx.run(_scalars(sc)) == exp, ("resident", group, wb)
            finally:
                ctx.close()


# ------------------------------------------------------------------------------------------- resident-operand unit entry points
@pytest.mark.parametrize("logn", [0, 1, 2, 3, 5, 9, 10, 11])
def test_ntt_context_all_modes_vs_oracle(cc, oracle, logn):
    """cg_ntt_load / cg_ntt_run on the 29-bit kernels, every (inverse, coset) mode, sizes below, at and above one LDS tile"""
    rng = random.Random(700 + logn)
    n = 1 << logn
    v = [rng.choice([0, 1, oracle.R - 1, rng.randrange(oracle.R)]) for _ in range(n)]
    ctx = cc.NttContext(logn)
    try:
        for inverse in (False, True):
            for coset in (False, True):
                exp = list(v)
                if not inverse:
                    exp = oracle.coset_fft(exp) if coset else oracle.fft(exp)
                else:
                    exp = oracle.coset_ifft(exp) if coset else oracle.ifft(exp)
                assert _ints(ctx.run(_scalars(v), inverse=inverse, coset=coset)) == exp, (logn, inverse, coset)
    finally:
        ctx.close()


def test_ntt_context_device_data_and_errors(cc, oracle):
    import torch
    logn = 14
    n = 1 << logn
    rng = random.Random(5)
    v = [rng.randrange(oracle.R) for _ in range(n)]
    ctx = cc.NttContext(logn)
    try:
        host = ctx.run(_scalars(v), coset=True)
        d = torch.from_numpy(_scalars(v)).cuda()
        ms = ctx.run_dev(d.data_ptr(), coset=True)
        assert ms > 0
        assert bytes(d.cpu().numpy()) == bytes(host)
        ctx.run_dev(d.data_ptr(), inverse=True, coset=True)          # and back, in place on the device
        assert _ints(d.cpu().numpy()) == v
        bad = list(v)
        bad[77] = oracle.R                                           # not a field element
        with pytest.raises(cc.CrescentGpuError):
            ctx.run(_scalars(bad))
        with pytest.raises(ValueError):
            ctx.run(_scalars(v[:-1]))
    finally:
        ctx.close()
    with pytest.raises(cc.CrescentGpuError):
        cc.NttContext(29)                                            # beyond the two-adicity (PolynomialDegreeTooLarge)


def test_fixed_base_vs_oracle(cc, oracle):
    rng = random.Random(17)
    ks = [0, 1, 2, oracle.R - 1] + [rng.randrange(oracle.R) for _ in range(20)]
    out1 = cc.fixed_base_g1(_scalars(ks))
    out2 = cc.fixed_base_g2(_scalars(ks))
    for i, k in enumerate(ks):
        e1 = oracle.g1_packed(oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, k))) if k else bytes(64)
        e2 = oracle.g2_packed(oracle.G2.to_affine(oracle.G2.mul_affine(oracle.G2_GEN, k))) if k else bytes(128)
        assert out1[64 * i:64 * i + 64] == e1, i
        assert out2[128 * i:128 * i + 128] == e2, i
    assert cc.fixed_base_g1(b"") == b""
    with pytest.raises(cc.CrescentGpuError):
        cc.fixed_base_g1(_scalars([oracle.R]))


@pytest.mark.parametrize("group", [1, 2])
def test_msm_context_matches_one_shot_and_closed_form(cc, oracle, group):
    """resident bases (window tables precomputed) == one-shot msm_bigint == (Σ s_i k_i)·G"""
    import torch
    rng = random.Random(40 + group)
    n = 5000
    ks = [rng.randrange(1, oracle.R) for _ in range(n)]
    ks[10] = 0                                                           # an identity base
    bases = (cc.fixed_base_g1 if group == 1 else cc.fixed_base_g2)(_scalars(ks))
    one_shot = cc.msm_bigint_g1 if group == 1 else cc.msm_bigint_g2
    curve, gen, packed = (oracle.G1, oracle.G1_GEN, oracle.g1_packed) if group == 1 else (oracle.G2, oracle.G2_GEN, oracle.g2_packed)
    for wb in (0, 7):
        ctx = cc.MsmContext(bases, group=group, window_bits=wb)
        try:
            for trial in range(3):
                if trial == 0:
                    sc = [rng.randrange(oracle.R) for _ in range(n)]                          # uniform
                elif trial == 1:
                    sc = [rng.choice([0, 0, 1, 1, 1, rng.randrange(oracle.R)]) for _ in range(n)]    # 0/1-heavy
                else:
                    sc = [rng.randrange(oracle.R) for _ in range(n // 3)]                      # fewer scalars than bases
                e = sum(k * s for k, s in zip(ks, sc)) % oracle.R
                exp = packed(curve.to_affine(curve.mul_affine(gen, e))) if e else bytes(64 * group)
                got, tm = ctx.run(_scalars(sc), timings=True)
                assert got == exp, (group, wb, trial)
                assert tm["msm_g%d_pairs" % group] == len(sc)
                if trial == 0:
                    assert one_shot(bases, _scalars(sc)) == exp
                    d = torch.from_numpy(_scalars(sc)).cuda()
                    assert ctx.run_dev(d.data_ptr(), len(sc)) == exp
            assert ctx.run(b"") == bytes(64 * group)
            with pytest.raises(cc.CrescentGpuError):
                ctx.run(_scalars([oracle.R] + [1] * 9))
        finally:
            ctx.close()


@pytest.mark.parametrize("wb", [3, 8, 14, 0])
def test_signed_accumulation_rare_branches_on_the_gpu(cc, oracle, wb):
    """Round 5's G1 accumulation keeps its running sum on signed limbs with the sign of Y tracked apart (curve29.hpp madd29s).
    Its rare branches - the same point met again (doubling, under either sign of the digit and of the tracked sign), a point
    and its negative (cancellation: an empty run flushed as an identity record), runs that restart after a cancellation - are
    reached here on purpose: ONE base repeated (every addition inside a bucket is a doubling candidate), P and -P alternating,
    and both with scalars that put many entries into the same bucket with mixed digit signs; against the closed form
    (Σ s_i k_i)·G, through resident tables with narrow, medium, wide and default windows, twice per handle."""
    rng = random.Random(505 + wb)
    n = 6001
    k0 = rng.randrange(1, oracle.R)
    bases_sets = {
        "one base": [k0] * n,
        "P and -P": [k0 if i & 1 else oracle.R - k0 for i in range(n)],
        "three bases": [(k0, 2 * k0 % oracle.R, oracle.R - k0)[i % 3] for i in range(n)],
    }
    half = 1 << ((wb or 13) - 1)
    pops = {
        "ones": [1] * n,
        "minus ones": [oracle.R - 1] * n,                                   # every digit pattern negative at the top
        "same digit both signs": [(half - 1) if i & 1 else (half + 1) for i in range(n)],       # |d| equal, sign differs after recoding
        "uniform": [rng.randrange(oracle.R) for _ in range(n)],
        "two values": [5 if i % 3 else oracle.R - 5 for i in range(n)],
    }
    for bname, ks in bases_sets.items():
        bases = cc.fixed_base_g1(_scalars(ks))
        ctx = cc.MsmContext(bases, group=1, window_bits=wb)
        try:
            for pname, sc in pops.items():
                e = sum(k * s_ for k, s_ in zip(ks, sc)) % oracle.R
                exp = oracle.g1_packed(oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, e))) if e else bytes(64)
                arr = _scalars(sc)
                assert ctx.run(arr) == exp, (wb, bname, pname)
                assert ctx.run(arr) == exp, (wb, bname, pname, "second run")
        finally:
            ctx.close()


@pytest.mark.parametrize("group,n", [(1, 200_003), (2, 40_001)])
def test_msm_grouping_and_combine_edge_cases(cc, oracle, group, n):
    """The entry grouping (two-level counting partition) and the wave-level piece combination on the populations that
    stress them, each against the closed form (Σ s_i k_i)·G:  one bucket holding every entry (a single run across all
    segments and all three combine levels), two alternating scalars, every scalar 1, a few non-zero scalars among
    zeros, uniform scalars; n is odd and neither a multiple of the level-1 tile (4096 scalars) nor of the level-2 chunk;
    window sizes that give a one-level key space (c <= 13), a two-level one, and the size-based default."""
    nrng = np.random.default_rng(1000 + group)
    kb = nrng.integers(0, 256, (n, 32), dtype=np.uint8)
    kb[:, 31] &= 0x1F                                                      # < 2^253 < r
    kb[7] = 0                                                               # an identity base
    ks = [int.from_bytes(kb[i].tobytes(), "little") for i in range(n)]
    bases = (cc.fixed_base_g1 if group == 1 else cc.fixed_base_g2)(kb.reshape(-1))
    curve, gen, packed = (oracle.G1, oracle.G1_GEN, oracle.g1_packed) if group == 1 else (oracle.G2, oracle.G2_GEN, oracle.g2_packed)
    rng = random.Random(7 * group)
    big = rng.randrange(oracle.R)
    pops = {
        "one value everywhere": [big] * n,
        "all ones": [1] * n,
        "two values": [big if i & 1 else oracle.R - 1 for i in range(n)],
        "sparse": [rng.randrange(oracle.R) if i % 9973 == 0 else 0 for i in range(n)],
        "short (fewer scalars than bases)": [rng.randrange(oracle.R) for _ in range(4097)],
        "small values": [rng.randrange(256) for _ in range(n)],
    }
    ub = nrng.integers(0, 256, (n, 32), dtype=np.uint8)
    ub[:, 31] &= 0x1F
    pops["uniform"] = None
    for wb in ((9, 16, 0) if group == 1 else (11, 0)):
        ctx = cc.MsmContext(bases, group=group, window_bits=wb)
        try:
            for name, sc in pops.items():
                if sc is None:
                    arr = ub.reshape(-1)
                    vals = [int.from_bytes(ub[i].tobytes(), "little") for i in range(n)]
                else:
                    arr, vals = _scalars(sc), sc
                e = sum(k * s_ for k, s_ in zip(ks, vals)) % oracle.R
                exp = packed(curve.to_affine(curve.mul_affine(gen, e))) if e else bytes(64 * group)
                got, tm = ctx.run(arr, timings=True)
                assert got == exp, (group, wb, name)
                assert got == ctx.run(arr), (group, wb, name, "second run on the same engine")
        finally:
            ctx.close()


@pytest.mark.parametrize("rounds", [1, 2, 4])
def test_batch_affine_rounds_agree(cc, oracle, rounds, monkeypatch):
    """The batch-affine pair rounds (csrc/batchaff.hpp; an experiment that is off by default, profiles/r03_s_batch_affine.txt)
    give the same sums as the XYZZ accumulation, on the populations that reach their slow paths: one base repeated (every
    pair a doubling, round after round), P and -P mixed (cancellations: identity records in the next round), an identity
    base, an odd element count, buckets of one element (split pairs), uniform scalars.  Only in a library built with
    -DCG_WITH_BATCH_AFFINE (CG_HIPCC_EXTRA=-DCG_WITH_BATCH_AFFINE python crescent-credentials_amd/build.py): the shipped
    build does not carry the experiment."""
    if b"batch-affine" not in cc.lib().cg_version():
        pytest.skip("the loaded libcrescent_gpu.so was built without -DCG_WITH_BATCH_AFFINE (the default)")
    monkeypatch.setenv("CG_BA_ROUNDS", str(rounds))       # read when an engine is initialised
    rng = random.Random(900 + rounds)
    n = 3001
    k0 = rng.randrange(1, oracle.R)
    keysets = {
        "distinct": [rng.randrange(1, oracle.R) for _ in range(n)],
        "one base": [k0] * n,
        "P and -P": [k0 if rng.random() < 0.5 else oracle.R - k0 for _ in range(n)],
    }
    keysets["distinct"][5] = 0
    big = rng.randrange(oracle.R)
    for kname, ks in keysets.items():
        bases = cc.fixed_base_g1(_scalars(ks))
        for wb in (5, 11, 0):
            ctx = cc.MsmContext(bases, group=1, window_bits=wb)
            try:
                pops = {
                    "one value": [big] * n,
                    "ones": [1] * n,
                    "uniform": [rng.randrange(oracle.R) for _ in range(n)],
                    "sparse": [rng.randrange(oracle.R) if i % 97 == 0 else 0 for i in range(n)],
                    "short": [rng.randrange(oracle.R) for _ in range(1000)],
                }
                for pname, sc in pops.items():
                    e = sum(k * s_ for k, s_ in zip(ks, sc)) % oracle.R
                    exp = oracle.g1_packed(oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, e))) if e else bytes(64)
                    assert ctx.run(_scalars(sc)) == exp, (rounds, kname, wb, pname)
            finally:
                ctx.close()
    # the one-shot entry point (window index in the key, no window tables) takes the same rounds
    ks = keysets["P and -P"]
    sc = [rng.randrange(oracle.R) for _ in range(n)]
    e = sum(k * s_ for k, s_ in zip(ks, sc)) % oracle.R
    exp = oracle.g1_packed(oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, e))) if e else bytes(64)
    assert cc.msm_bigint_g1(cc.fixed_base_g1(_scalars(ks)), _scalars(sc)) == exp


# ------------------------------------------------------------------------------------------- setup
def _pk_from_json(cc, j):
    a = lambda h: np.frombuffer(bytes.fromhex(h), dtype=np.uint8).copy()
    vk = cc.VerifyingKey(alpha_g1=a(j["alpha_g1"]), beta_g2=a(j["beta_g2"]), gamma_g2=a(j["gamma_g2"]),
                         delta_g1=a(j["delta_g1"]), delta_g2=a(j["delta_g2"]), gamma_abc_g1=a(j["gamma_abc_g1"]))
    return cc.ProvingKey(vk=vk, beta_g1=a(j["beta_g1"]), delta_g1=a(j["delta_g1"]), a_query=a(j["a_query"]),
                         b_g1_query=a(j["b_g1_query"]), b_g2_query=a(j["b_g2_query"]), h_query=a(j["h_query"]),
                         l_query=a(j["l_query"]))


def _pk_digest(pk):
    return dict(alpha_g1=_sha(pk.vk.alpha_g1), beta_g1=_sha(pk.beta_g1), delta_g1=_sha(pk.delta_g1), beta_g2=_sha(pk.vk.beta_g2),
                gamma_g2=_sha(pk.vk.gamma_g2), delta_g2=_sha(pk.vk.delta_g2), gamma_abc_g1=_sha(pk.vk.gamma_abc_g1),
                a_query=_sha(pk.a_query), b_g1_query=_sha(pk.b_g1_query), b_g2_query=_sha(pk.b_g2_query),
                h_query=_sha(pk.h_query), l_query=_sha(pk.l_query))


def _case_matrices(cc, oracle, g):
    if "matrices" in g:
        mats = _rows(g["matrices"])
    else:
        d = g["dummy"]
        mats = oracle.dummy_circuit(int(d["a"], 16), int(d["b"], 16), d["num_variables"], d["num_constraints"], d["num_inputs"])[0]
    return cc.ConstraintMatrices.from_rows(mats[0], mats[1], mats[2], g["num_inputs"], g["num_variables"]), mats


@pytest.mark.parametrize("name", ["groth16_d8.json", "groth16_tiny.json", "groth16_dummy1024.json"])
def test_setup_matches_oracle_pk(cc, oracle, name):
    """cg_setup (generator.rs:50-228 restated on the GPU) reproduces the oracle's proving key byte for byte"""
    g = load_golden(name)
    cm, _ = _case_matrices(cc, oracle, g)
    t = g["trapdoor"]
    pk = cc.generate_parameters_with_qap(cm, int(t["alpha"], 16), int(t["beta"], 16), int(t["delta"], 16), int(t["tau"], 16))
    assert _pk_digest(pk) == g["pk_sha256"]


# ------------------------------------------------------------------------------------------- prove
@pytest.mark.parametrize("name", ["groth16_d8.json", "groth16_tiny.json", "groth16_dummy1024.json"])
def test_prove_golden(cc, oracle, name):
    """create_proof_with_reduction_and_matrices (prover.rs:26-51): 256 proof bytes identical to the oracle's"""
    g = load_golden(name)
    cm, _ = _case_matrices(cc, oracle, g)
    if "pk" in g:
        pk = _pk_from_json(cc, g["pk"])
    else:
        t = g["trapdoor"]
        pk = cc.generate_parameters_with_qap(cm, int(t["alpha"], 16), int(t["beta"], 16), int(t["delta"], 16), int(t["tau"], 16))
    w = _scalars([int(x, 16) for x in g["witness"]])
    # default: h query moved to the coset evaluation basis at load (six transforms per proof); and the reference's own
    # arrangement (coefficient basis, seven transforms)
    for coeff_basis in (False, True):
        prover = cc.Prover(pk, cm, h_coefficient_basis=coeff_basis)
        try:
            assert prover.domain_size == g["domain_size"]
            h = prover.witness_map(w)
            assert _sha(h) == g["h_sha256"]
            for case in g["proofs"]:
                r, s = int(case["r"], 16), int(case["s"], 16)
                assert prover.prove(w, r, s).serialize_uncompressed().hex() == case["proof"], coeff_basis
        finally:
            prover.close()
    try:
        # reference call shape
        p = cc.Groth16.create_proof_with_reduction_and_matrices(pk, r, s, cm, g["num_inputs"], g["num_constraints"], w)
        assert p.data.hex() == case["proof"]
    finally:
        cc.Groth16.clear_cache()


def _pk_from_oracle(cc, oracle, pk):
    g1s = lambda pts: np.frombuffer(b"".join(oracle.g1_packed(p) for p in pts), dtype=np.uint8).copy()
    g2s = lambda pts: np.frombuffer(b"".join(oracle.g2_packed(p) for p in pts), dtype=np.uint8).copy()
    v = pk["vk"]
    vk = cc.VerifyingKey(alpha_g1=g1s([v["alpha_g1"]]), beta_g2=g2s([v["beta_g2"]]), gamma_g2=g2s([v["gamma_g2"]]),
                         delta_g1=g1s([v["delta_g1"]]), delta_g2=g2s([v["delta_g2"]]), gamma_abc_g1=g1s(v["gamma_abc_g1"]))
    return cc.ProvingKey(vk=vk, beta_g1=g1s([pk["beta_g1"]]), delta_g1=g1s([pk["delta_g1"]]), a_query=g1s(pk["a_query"]),
                         b_g1_query=g1s(pk["b_g1_query"]), b_g2_query=g2s(pk["b_g2_query"]), h_query=g1s(pk["h_query"]),
                         l_query=g1s(pk["l_query"]))


@pytest.mark.parametrize("l,m,M", [(1, 1, 1), (1, 1, 2), (3, 5, 8), (3, 6, 9), (4, 3, 4), (2, 0, 5), (1, 7, 3), (5, 11, 16)],
                         ids=lambda v: str(v))
def test_prove_edge_shapes(cc, oracle, l, m, M):
    """Ragged and degenerate shapes against the Python oracle: no witness variables at all (M = l: empty l_query and
    a one-point a/b query), m + l exactly a power of two and one past it, no constraints, more constraints than
    variables; rows that are empty, repeat a column, or carry the coefficients 0, 1 and r - 1; witnesses that do NOT
    satisfy the system (parity is about bytes, not about acceptance), all-zero and all-(r-1) witnesses."""
    rng = random.Random(1000 * l + 10 * m + M)
    def row():
        k = rng.choice([0, 1, 1, 2, 3])
        return [(rng.choice([0, 1, 1, oracle.R - 1, rng.randrange(oracle.R)]), rng.randrange(M)) for _ in range(k)]
    mats = tuple([row() for _ in range(m)] for _ in range(3))
    trap = [rng.randrange(1, oracle.R) for _ in range(4)]
    pk_o, _ = oracle.generate_parameters(mats, l, m, M, *trap)
    cm = cc.ConstraintMatrices.from_rows(mats[0], mats[1], mats[2], l, M)
    pk_gpu = cc.generate_parameters_with_qap(cm, trap[1], trap[2], trap[3], trap[0])
    pk = _pk_from_oracle(cc, oracle, pk_o)
    assert _pk_digest(pk_gpu) == _pk_digest(pk)                       # cg_setup on the same degenerate shape
    prover = cc.Prover(pk, cm, h_coefficient_basis=(M % 2 == 0))
    try:
        for wit in ("random", "zeros", "max"):
            if wit == "random":
                w = [1] + [rng.randrange(oracle.R) for _ in range(M - 1)]
            elif wit == "zeros":
                w = [1] + [0] * (M - 1)
            else:
                w = [1] + [oracle.R - 1] * (M - 1)
            assert _ints(prover.witness_map(_scalars(w))) == oracle.witness_map_from_matrices(mats, l, m, w)
            for r, s in ((0, 0), (rng.randrange(oracle.R), 0), (rng.randrange(oracle.R), rng.randrange(oracle.R))):
                exp = oracle.proof_uncompressed(oracle.create_proof_with_reduction_and_matrices(pk_o, r, s, mats, l, m, w))
                assert prover.prove(_scalars(w), r, s).data == exp, (wit, r != 0, s != 0)
    finally:
        prover.close()


def test_prove_verifies_with_pairing(cc, oracle):
    """acceptance criterion of the reference's own tests (verify == true; verifier.rs:44-77), checked by the
    oracle's pairing on a GPU-made key + GPU-made proof, plus the trapdoor closed form."""
    from crescent_credentials_amd import workloads as wl
    l, m, M = 5, 500, 540
    cm, w = wl.synthetic_circuit(4242, l, m, M, 0.5, 3)
    rng = random.Random(17)
    tau, alpha, beta, delta = (rng.randrange(1, oracle.R) for _ in range(4))
    pk = cc.generate_parameters_with_qap(cm, alpha, beta, delta, tau)
    prover = cc.Prover(pk, cm)
    try:
        r, s = rng.randrange(oracle.R), rng.randrange(oracle.R)
        proof = prover.prove(w, r, s)
        hb = prover.witness_map(w)
    finally:
        prover.close()
    # decode the proof (uncompressed a ‖ b ‖ c, flags in the top bits of each point's last byte)
    def g1(b):
        b = bytearray(b); b[63] &= 0x3F
        return oracle.g1_unpack(bytes(b))
    def g2(b):
        b = bytearray(b); b[127] &= 0x3F
        return oracle.g2_unpack(bytes(b))
    pr = (g1(proof.a), g2(proof.b), g1(proof.c))
    assert oracle.proof_uncompressed(pr) == proof.data            # flags agree with the oracle's serialiser
    vk = dict(alpha_g1=oracle.g1_unpack(bytes(pk.vk.alpha_g1)), beta_g2=oracle.g2_unpack(bytes(pk.vk.beta_g2)),
              gamma_g2=oracle.g2_unpack(bytes(pk.vk.gamma_g2)), delta_g1=oracle.g1_unpack(bytes(pk.vk.delta_g1)),
              delta_g2=oracle.g2_unpack(bytes(pk.vk.delta_g2)),
              gamma_abc_g1=[oracle.g1_unpack(bytes(pk.vk.gamma_abc_g1[64 * i:64 * i + 64])) for i in range(l)])
    wi = wl.witness_to_ints(w)
    assert oracle.verify_proof(vk, pr, wi[1:l])
    bad = list(wi[1:l]); bad[0] ^= 1
    assert not oracle.verify_proof(vk, pr, bad)
    # closed form from the trapdoor
    mats = wl.matrices_to_rows(cm)
    _, qap = None, None
    D = oracle.domain_size_for(m + l)
    u = oracle.evaluate_all_lagrange_coefficients(D, tau)
    a = [0] * M; b = [0] * M; c = [0] * M
    for i in range(l): a[i] = u[m + i]
    for i in range(m):
        for coeff, idx in mats[0][i]: a[idx] = (a[idx] + u[i] * coeff) % oracle.R
        for coeff, idx in mats[1][i]: b[idx] = (b[idx] + u[i] * coeff) % oracle.R
        for coeff, idx in mats[2][i]: c[idx] = (c[idx] + u[i] * coeff) % oracle.R
    dinv = pow(delta, oracle.R - 2, oracle.R)
    qap = dict(a=a, b=b, l=[(beta * a[i] + alpha * b[i] + c[i]) * dinv % oracle.R for i in range(l, M)],
               zt=oracle.evaluate_vanishing_polynomial(D, tau), D=D)
    assert oracle.closed_form_proof(qap, (tau, alpha, beta, delta), r, s, _ints(hb), wi, l) == pr


def test_sharded_partials_assemble_to_same_proof(cc, oracle):
    """range-sharded contexts produce partial sums whose gather + assemble equals the unsharded proof"""
    g = load_golden("groth16_tiny.json")
    cm, _ = _case_matrices(cc, oracle, g)
    t = g["trapdoor"]
    pk = cc.generate_parameters_with_qap(cm, int(t["alpha"], 16), int(t["beta"], 16), int(t["delta"], 16), int(t["tau"], 16))
    w = _scalars([int(x, 16) for x in g["witness"]])
    for nshard in (2, 3):
        shards = [cc.Prover(pk, cm, shard_rank=k, shard_count=nshard) for k in range(nshard)]
        try:
            for case in g["proofs"]:
                r, s = int(case["r"], 16), int(case["s"], 16)
                parts = b"".join(p.prove_partial(w, r) for p in shards)
                assert shards[0].assemble(parts, nshard, r, s).data.hex() == case["proof"]
        finally:
            for p in shards:
                p.close()


def test_b_queries_share_their_grouping_only_when_they_vanish_together(cc, oracle):
    """The G2 MSM of a proof takes over the grouped digit entries of the G1 MSM over the same scalars (b_g1_query and
    b_g2_query are b_i(τ)·G1 and b_i(τ)·G2, generator.rs:162,168: identity at the same indices).  A key from elsewhere is
    not trusted to have that property: with one b_g1 point zeroed and one b_g2 point zeroed at another index the proof
    must still be what the reference computes for THAT key (the C restatement's bytes), and with the honest key too."""
    import copy
    import cpu_ref
    from crescent_credentials_amd import workloads as wl
    l, m, M = 6, 9_000, 9_200
    cm, w = wl.synthetic_circuit(606, l, m, M, 0.6, 3, profile="gates")
    rng = random.Random(606)
    pk = cc.generate_parameters_with_qap(cm, *(rng.randrange(1, oracle.R) for _ in range(4)))
    odd = copy.copy(pk)
    odd.b_g1_query = pk.b_g1_query.copy()
    odd.b_g2_query = pk.b_g2_query.copy()
    nz = [i for i in range(50, M) if pk.b_g1_query[64 * i:64 * i + 64].any() and w[32 * i:32 * i + 32].any()]
    i1, i2 = nz[3], nz[40]
    odd.b_g1_query[64 * i1:64 * i1 + 64] = 0                       # identity in G1 only
    odd.b_g2_query[128 * i2:128 * i2 + 128] = 0                    # identity in G2 only
    for key in (pk, odd):
        prover = cc.Prover(key, cm, proof_slots=2)
        try:
            for r, s in ((rng.randrange(oracle.R), rng.randrange(oracle.R)), (0, 5), (rng.randrange(oracle.R), 0)):
                for _ in range(2):                                  # before and after the one-time window re-tune
                    got = prover.prove(w, r, s).data
                assert got == cpu_ref.prove(key, (cm.a, cm.b, cm.c), l, m, M, w, r, s, nthreads=8), (key is odd, r == 0)
        finally:
            prover.close()


def test_concurrent_proofs_on_one_context_and_across_contexts(cc, oracle):
    """SURVEY 8b threading: `create_client_state` runs concurrently from several tasks (sample/client_helper/src/main.rs:
    177-216), so cg_prove must be re-entrant across contexts and overlap proofs within one.  Twelve host threads drive a
    six-slot context and a two-slot context of another circuit at the same time - the first proofs race with the one-time
    window re-tune, the entry grouping runs with concurrent atomics in another order every time - and every proof must
    be byte-identical to the one a single-slot context makes alone (sums are exact, so order must not matter)."""
    from concurrent.futures import ThreadPoolExecutor
    from crescent_credentials_amd import workloads as wl
    rng = random.Random(31337)
    circuits = []
    for seed, (l, m, M), bits in ((1, (9, 30_000, 31_000), 0.9), (2, (5, 12_000, 12_500), 0.3)):
        cm, w = wl.synthetic_circuit(seed, l, m, M, bits, 3, profile="gates")
        pk = cc.generate_parameters_with_qap(cm, *(rng.randrange(1, oracle.R) for _ in range(4)))
        ws = [w, _scalars([rng.randrange(oracle.R) for _ in range(M)])]           # the witness and an arbitrary assignment
        jobs = [(ws[k % 2], rng.randrange(oracle.R) if k % 5 else 0, rng.randrange(oracle.R)) for k in range(10)]
        alone = cc.Prover(pk, cm)
        try:
            expect = [alone.prove(*j).data for j in jobs]
        finally:
            alone.close()
        circuits.append((pk, cm, jobs, expect))
    provers = [cc.Prover(circuits[0][0], circuits[0][1], proof_slots=6), cc.Prover(circuits[1][0], circuits[1][1], proof_slots=2)]
    try:
        work = [(c, k) for rep in range(4) for c in (0, 1) for k in range(10)]
        rng.shuffle(work)
        with ThreadPoolExecutor(max_workers=12) as ex:
            got = list(ex.map(lambda ck: provers[ck[0]].prove(*circuits[ck[0]][2][ck[1]]).data, work))
        for (c, k), g in zip(work, got):
            assert g == circuits[c][3][k], (c, k)
    finally:
        for p in provers:
            p.close()


@pytest.mark.parametrize("shape,nshard", [("log11", 2), ("log13", 4), ("log14", 8), ("medium", 8), ("log17", 16)])
def test_strided_shards_assemble_to_same_proof(cc, oracle, shape, nshard):
    """A power-of-two shard count gives every shard the coset points j = rank (mod count) of the h MSM (two of its four
    transforms shrink by the shard count, csrc/wmap29.hpp Wm29Strided); the assembled proof must not change: bytes equal
    the unsharded context's (itself pinned to oracle/cpu_ref.c below) for satisfying and arbitrary assignments."""
    from crescent_credentials_amd import workloads as wl
    l, m, M = _CPU_SHAPES[shape]
    cm, w = wl.synthetic_circuit(77, l, m, M, 0.5, 3)
    rng = random.Random(5)
    tau, alpha, beta, delta = (rng.randrange(1, oracle.R) for _ in range(4))
    pk = cc.generate_parameters_with_qap(cm, alpha, beta, delta, tau)
    w_bad = _scalars([rng.randrange(oracle.R) for _ in range(M)])       # does not satisfy the circuit: the identities hold anyway
    whole = cc.Prover(pk, cm)
    shards = [cc.Prover(pk, cm, shard_rank=k, shard_count=nshard) for k in range(nshard)]
    try:
        for wit in (w, w_bad):
            for r, s in ((0, 0), (rng.randrange(oracle.R), rng.randrange(oracle.R))):
                parts = b"".join(p.prove_partial(wit, r) for p in shards)
                assert shards[-1].assemble(parts, nshard, r, s).data == whole.prove(wit, r, s).data
    finally:
        whole.close()
        for p in shards:
            p.close()


def test_context_flags_change_the_arrangement_not_the_bytes(cc, oracle):
    """what used to be environment switches of the shipped library are cg_options flags - CG_FLAG_LATENCY_MODE /
    CG_FLAG_THROUGHPUT_MODE (the arrangement whatever proof_slots says), CG_FLAG_SPIN_WAIT, CG_FLAG_CONTIGUOUS_H_SHARDS.  Every
    arrangement gives the unflagged context's bytes; cg_ctx_get_info reports the forced mode; both mode flags together are
    refused."""
    from concurrent.futures import ThreadPoolExecutor
    from crescent_credentials_amd import workloads as wl
    l, m, M = _CPU_SHAPES["log14"]
    cm, w = wl.synthetic_circuit(78, l, m, M, 0.7, 3, profile="gates")
    rng = random.Random(6)
    pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, oracle.R) for _ in range(4)])
    rs = [(0, 0)] + [(rng.randrange(oracle.R), rng.randrange(oracle.R)) for _ in range(3)]
    whole = cc.Prover(pk, cm)
    want = [whole.prove(w, r, s).data for r, s in rs]
    assert whole.info()["latency_mode"] == 1
    try:
        for kw, lat in ((dict(proof_slots=1, mode="throughput"), 0), (dict(proof_slots=3, mode="latency"), 1),
                        (dict(proof_slots=3, spin_wait=True), 0), (dict(proof_slots=1, mode="throughput", spin_wait=True), 0)):
            p = cc.Prover(pk, cm, **kw)
            try:
                assert p.info()["latency_mode"] == lat, kw
                with ThreadPoolExecutor(max_workers=4) as ex:
                    got = list(ex.map(lambda k: p.prove(w, *rs[k % len(rs)]).data, range(12)))
                assert got == [want[k % len(rs)] for k in range(12)], kw
            finally:
                p.close()
        with pytest.raises(cc.CrescentGpuError) as ei:
            cc.Prover(pk, cm, flags=2 | 4)
        assert ei.value.code == -1 and "exclusive" in str(ei.value)
        # contiguous h ranges for a power-of-two shard count (the default there is the strided arrangement)
        for contig in (False, True):
            shards = [cc.Prover(pk, cm, shard_rank=k, shard_count=4, contiguous_h_shards=contig) for k in range(4)]
            try:
                for (r, s), exp in zip(rs, want):
                    parts = b"".join(q.prove_partial(w, r) for q in shards)
                    assert shards[0].assemble(parts, 4, r, s).data == exp, contig
            finally:
                for q in shards:
                    q.close()
    finally:
        whole.close()


@pytest.mark.parametrize("contig", [False, True], ids=["strided", "contiguous"])
def test_h_scalars_from_one_witness_map_for_all_shards(cc, oracle, contig):
    """SURVEY 8e's other arrangement: the witness map runs ONCE (cg_witness_map_coset on a context that
    has the resources) and every shard proves with its slice of the coset values (cg_prove_partial_q); shards loaded with
    CG_FLAG_H_SCALARS_EXTERNAL hold no witness-map memory and can prove no other way.  Assembled bytes == the unsharded
    context's, for slices arriving in host and in device memory, satisfying and arbitrary assignments, r = 0 included."""
    import torch
    from crescent_credentials_amd import workloads as wl
    l, m, M = _CPU_SHAPES["log14"]
    cm, w = wl.synthetic_circuit(79, l, m, M, 0.6, 3, profile="gates")
    rng = random.Random(8)
    pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, oracle.R) for _ in range(4)])
    w_bad = _scalars([rng.randrange(oracle.R) for _ in range(M)])
    n = 4
    whole = cc.Prover(pk, cm)
    first = cc.Prover(pk, cm, shard_rank=0, shard_count=n, contiguous_h_shards=contig)                 # runs the witness map for everybody
    others = [cc.Prover(pk, cm, shard_rank=k, shard_count=n, contiguous_h_shards=contig, h_scalars_external=True) for k in range(1, n)]
    shards = [first] + others
    try:
        D = whole.domain_size
        slices = [first.h_scalars_slice(k) for k in range(n)]
        assert sum(c for _, c in slices) == D and slices[0][0] == 0
        assert all(p.h_scalars_slice(k) == slices[k] for p in others for k in range(n))
        assert first.info()["slot_transform_bytes"] > 4 * D * 32 > others[0].info()["slot_transform_bytes"]
        for wit in (w, w_bad):
            q = first.witness_map_coset(wit)
            assert q.size == D * 32
            qd = torch.from_numpy(q).cuda()
            wd = torch.from_numpy(np.ascontiguousarray(wit)).cuda()
            first.witness_map_coset(wd.data_ptr(), on_device=True, out_dev=qd.data_ptr())               # device in, device out: the same
            assert bytes(qd.cpu().numpy()) == bytes(q)
            for r, s in ((0, 0), (rng.randrange(oracle.R), rng.randrange(oracle.R))):
                want = whole.prove(wit, r, s).data
                parts_h = b"".join(p.prove_partial_q(wit, q[o * 32:(o + c) * 32], r) for p, (o, c) in zip(shards, slices))
                parts_d = b"".join(p.prove_partial_q(wd.data_ptr(), qd.data_ptr() + o * 32, r, on_device=True, q_on_device=True)
                                   for p, (o, c) in zip(shards, slices))
                assert parts_h == parts_d
                assert others[-1].assemble(parts_h, n, r, s).data == want
                # and the recomputing arrangement on the shard that can: the same partial sums
                assert first.prove_partial(wit, r) == parts_h[:384]
        # what an external shard refuses, and what it checks
        for bad_call in (lambda: others[0].prove_partial(w, 5), lambda: others[0].witness_map_coset(w), lambda: others[0].witness_map(w),
                         lambda: whole.prove_partial_q(w, q[:32], 5)):
            with pytest.raises(cc.CrescentGpuError) as ei:
                bad_call()
            assert ei.value.code == -1
        o1, c1 = slices[1]
        q_bad = q[o1 * 32:(o1 + c1) * 32].copy()
        q_bad[32 * 3:32 * 4] = np.frombuffer(oracle.R.to_bytes(32, "little"), np.uint8)
        with pytest.raises(cc.CrescentGpuError):
            others[0].prove_partial_q(w, q_bad, 5)
        w_nc = np.ascontiguousarray(w).copy()
        w_nc[32 * 7:32 * 8] = np.frombuffer(oracle.R.to_bytes(32, "little"), np.uint8)
        with pytest.raises(cc.CrescentGpuError):
            others[0].prove_partial_q(w_nc, q[o1 * 32:(o1 + c1) * 32], 5)
        assert others[0].prove_partial_q(w, q[o1 * 32:(o1 + c1) * 32], 0) is not None                       # and it still works afterwards
        with pytest.raises(cc.CrescentGpuError):
            cc.Prover(pk, cm, h_scalars_external=True)                                                   # needs a sharded context
        with pytest.raises(cc.CrescentGpuError):
            cc.Prover(pk, cm, shard_rank=0, shard_count=2, h_scalars_external=True, h_coefficient_basis=True)
    finally:
        whole.close()
        for p in shards:
            p.close()


_CPU_SHAPES = {"log11": (4, 1_500, 1_600), "exact12": (6, 4_090, 4_200), "log13": (10, 5_000, 5_100), "log14": (3, 9_000, 16_000),
               "log15": (12, 20_000, 20_500), "medium": (20, 60_000, 61_000), "log17": (8, 100_000, 100_100), "large18": (26, 250_000, 255_000)}


@pytest.mark.parametrize("shape,bit_fraction", [("log11", 0.5), ("exact12", 0.9), ("log13", 0.0), ("log14", 0.9), ("log15", 0.3),
                                                ("medium", 0.9), ("medium", 0.0), ("log17", 0.7), ("large18", 0.9)])
def test_prove_equals_cpu_restatement(cc, oracle, shape, bit_fraction):
    """Full proofs at D = 2^11 .. 2^18 (too large for the Python oracle): byte-identical to oracle/cpu_ref.c, the C
    restatement that tests/test_cpu_ref.py pins to the golden vectors; covers every NTT pass plan up to three passes
    (one LDS pass + strided passes of 1..6 stages), m + l exactly a power of two, M much larger than m, and the
    production window sizes."""
    import cpu_ref
    from crescent_credentials_amd import workloads as wl
    l, m, M = _CPU_SHAPES[shape]
    cm, w = wl.synthetic_circuit(2024, l, m, M, bit_fraction, 3)
    rng = random.Random(11)
    tau, alpha, beta, delta = (rng.randrange(1, oracle.R) for _ in range(4))
    pk = cc.generate_parameters_with_qap(cm, alpha, beta, delta, tau)
    prover = cc.Prover(pk, cm, proof_slots=2, h_coefficient_basis=(shape == "log13"))
    try:
        assert bytes(prover.witness_map(w)) == bytes(cpu_ref.witness_map((cm.a, cm.b, cm.c), l, m, M, w, nthreads=8))
        for r, s in ((0, 0), (rng.randrange(oracle.R), rng.randrange(oracle.R))):
            assert prover.prove(w, r, s).data == cpu_ref.prove(pk, (cm.a, cm.b, cm.c), l, m, M, w, r, s, nthreads=8)
    finally:
        prover.close()


def test_folded_key_equals_reference_arrangement_on_arbitrary_assignments(cc, oracle):
    """The default key layout (h query in the coset evaluation basis, C matrix folded into the l query: four
    transforms) and the reference's arrangement (CG_FLAG_H_COEFFICIENT_BASIS: seven transforms, l query as loaded)
    are different algorithms for the same group elements; they must agree byte for byte on ANY assignment —
    satisfying, random, sparse, all-ones — and any (r, s), and one of the two is pinned to the C restatement."""
    import cpu_ref
    from crescent_credentials_amd import workloads as wl
    l, m, M = 9, 40_000, 52_000
    cm, w_sat = wl.synthetic_circuit(31337, l, m, M, 0.6, 3)
    rng = random.Random(5150)
    trap = [rng.randrange(1, oracle.R) for _ in range(4)]
    pk = cc.generate_parameters_with_qap(cm, *trap)
    folded = cc.Prover(pk, cm, proof_slots=2)
    plain = cc.Prover(pk, cm, proof_slots=2, h_coefficient_basis=True)
    try:
        nprng = np.random.default_rng(99)
        def random_assignment(kind):
            a = nprng.integers(0, 256, (M, 32), dtype=np.uint8)
            a[:, 31] %= 0x30
            if kind == "sparse":
                a[nprng.random(M) < 0.97] = 0
            elif kind == "ones":
                a[:] = 0
                a[:, 0] = 1
            a[0] = 0
            a[0, 0] = 1                                   # the constant-one wire
            return a.reshape(-1).copy()
        cases = [w_sat] + [random_assignment(k) for k in ("random", "random", "sparse", "ones")]
        for i, w in enumerate(cases):
            for r, s in ((0, 0), (rng.randrange(oracle.R), rng.randrange(oracle.R))):
                pf = folded.prove(w, r, s).data
                assert pf == plain.prove(w, r, s).data, (i, r != 0)
            if i in (0, 1):
                assert pf == cpu_ref.prove(pk, (cm.a, cm.b, cm.c), l, m, M, w, r, s, nthreads=8), i
    finally:
        folded.close()
        plain.close()


def test_proofs_in_flight_are_independent(cc, oracle):
    """several host threads on one context (proof_slots) give the same bytes as serial proving"""
    from concurrent.futures import ThreadPoolExecutor
    from crescent_credentials_amd import workloads as wl
    l, m, M = 6, 3_000, 3_100
    cm, w = wl.synthetic_circuit(77, l, m, M, 0.5, 3)
    rng = random.Random(5)
    pk = cc.generate_parameters_with_qap(cm, *(rng.randrange(1, oracle.R) for _ in range(4)))
    rs = [(rng.randrange(oracle.R), rng.randrange(oracle.R)) for _ in range(12)]
    serial = cc.Prover(pk, cm)
    par = cc.Prover(pk, cm, proof_slots=3)
    try:
        expect = [serial.prove(w, r, s).data for r, s in rs]
        with ThreadPoolExecutor(max_workers=4) as ex:
            got = list(ex.map(lambda x: par.prove(w, x[0], x[1]).data, rs))
        assert got == expect
    finally:
        serial.close(); par.close()


def test_a_proof_that_arrives_alone_takes_the_lone_slot(cc, oracle):
    """A throughput context holds two slots more than proof_slots, arranged as a latency context's (five streams): a proof that
    finds fewer than two others in flight runs on one (sample/client_helper/src/main.rs:177-216: one task per credential, so a server's
    context is often between requests), proofs that find others in flight take the one-stream slots, a timed proof always
    does.  The bytes are the oracle's on every route, before and after the one-time re-tune (which re-sizes the lone slot's
    engines with the others); CG_FLAG_NO_LONE_SLOT leaves it out."""
    import cpu_ref
    from concurrent.futures import ThreadPoolExecutor
    from crescent_credentials_amd import workloads as wl
    l, m, M = 12, 40_000, 41_000
    cm, w = wl.synthetic_circuit(4141, l, m, M, 0.9, 3, profile="gates")
    rng = random.Random(41)
    pk = cc.generate_parameters_with_qap(cm, *(rng.randrange(1, oracle.R) for _ in range(4)))
    rs = [(rng.randrange(oracle.R), rng.randrange(oracle.R)) for _ in range(10)] + [(0, 5), (0, 0)]
    want = [cpu_ref.prove(pk, (cm.a, cm.b, cm.c), l, m, M, w, r, s, nthreads=8) for r, s in rs]
    with_lone = cc.Prover(pk, cm, proof_slots=3)
    without = cc.Prover(pk, cm, proof_slots=3, lone_slot=False)
    try:
        i = with_lone.info()
        assert i["lone_slots"] == 2 and i["proof_slots"] == 3 and i["latency_mode"] == 0
        # a latency slot holds a set of entry lists per MSM where a one-stream slot holds one for all five
        assert i["lone_slot_bytes"] > 2 * (i["slot_bytes"] - i["slot_upload_bytes"])
        j = without.info()
        assert j["lone_slots"] == 0 and j["lone_slot_bytes"] == 0 and j["proof_slots"] == 3
        assert i["total_bytes"] == j["total_bytes"] + i["lone_slot_bytes"]
        # one at a time: the first of these is the context's first proof (size-based windows), the second triggers the re-tune
        assert [with_lone.prove(w, r, s).data for r, s in rs] == want
        assert with_lone.info()["tuned"] == 1
        assert [with_lone.prove(w, r, s, timings=True)[0].data for r, s in rs[:4]] == want[:4]      # timed: a one-stream slot
        assert [without.prove(w, r, s).data for r, s in rs] == want
        # callers arriving together: the first two find the context (nearly) empty, the others do not
        for _ in range(3):
            with ThreadPoolExecutor(max_workers=4) as ex:
                assert list(ex.map(lambda x: with_lone.prove(w, x[0], x[1]).data, rs)) == want
            assert with_lone.prove(w, *rs[0]).data == want[0]                                       # alone again
        # the point of it: a proof alone is not slower on the lone slot than on a one-stream slot (at 2^21 it is 1.5x faster)
        def median_ms(p):
            ts = []
            for k in range(12):
                t0 = time.perf_counter()
                p.prove(w, *rs[k % len(rs)])
                ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2] * 1e3
        median_ms(with_lone); median_ms(without)
        a, b = median_ms(with_lone), median_ms(without)
        print("a proof alone at D = 2^16: %.2f ms on the lone slot, %.2f ms on a one-stream slot" % (a, b))
        assert a < 1.15 * b
    finally:
        with_lone.close(); without.close()


@pytest.mark.parametrize("bit_fraction", [0.0, 0.9])
def test_prove_medium_properties(cc, oracle, bit_fraction):
    """D = 2^16: no oracle proof at this size in seconds, so check structure-independent properties:
    determinism, (r, s) re-randomisation consistency A' - A = (r' - r)·delta_g1, and window-size independence."""
    from crescent_credentials_amd import workloads as wl
    l, m, M = 20, 60_000, 61_000
    cm, w = wl.synthetic_circuit(31337, l, m, M, bit_fraction, 3)
    rng = random.Random(3)
    tau, alpha, beta, delta = (rng.randrange(1, oracle.R) for _ in range(4))
    pk = cc.generate_parameters_with_qap(cm, alpha, beta, delta, tau)
    p1 = cc.Prover(pk, cm)
    p2 = cc.Prover(pk, cm, window_bits=9)
    try:
        r, s = rng.randrange(oracle.R), rng.randrange(oracle.R)
        a = p1.prove(w, r, s)
        assert p1.prove(w, r, s).data == a.data
        assert p2.prove(w, r, s).data == a.data
        a0 = p1.prove(w, 0, 0)
        def g1(b):
            b = bytearray(b); b[63] &= 0x3F
            return oracle.g1_unpack(bytes(b))
        # A(r) = A(0) + r·delta_g1  (prover.rs:94-96)
        d1 = oracle.g1_unpack(bytes(pk.delta_g1))
        exp = oracle.G1.to_affine(oracle.G1.add(oracle.G1.to_jac(g1(a0.a)), oracle.G1.mul_affine(d1, r)))
        assert g1(a.a) == exp
        # A, B, C themselves against the trapdoor closed form ([alpha + Σ w_i a_i(tau) + r·delta]·G, ...), with a_i(tau),
        # b_i(tau), c_i(tau) recomputed sparsely on the host (oracle/keycheck.py over oracle/cpu_ref.c: r1cs_to_qap.rs:103-147)
        import cpu_ref
        import keycheck
        trap = (alpha, beta, delta, tau)
        scal = keycheck.check_key(oracle, cpu_ref, pk, cm, l, m, M, trap, stride=1021)
        assert keycheck.closed_form(oracle, cpu_ref, scal, trap, r, s, w, l) == keycheck.decode_proof(oracle, a.data)
        assert keycheck.closed_form(oracle, cpu_ref, scal, trap, 0, 0, w, l) == keycheck.decode_proof(oracle, a0.data)
        assert keycheck.verify(oracle, pk, l, w, a.data)
    finally:
        p1.close(); p2.close()


def test_non_canonical_witness_is_rejected(cc, oracle):
    g = load_golden("groth16_d8.json")
    cm, _ = _case_matrices(cc, oracle, g)
    pk = _pk_from_json(cc, g["pk"])
    w = [int(x, 16) for x in g["witness"]]
    prover = cc.Prover(pk, cm)
    try:
        bad = list(w); bad[3] = oracle.R          # = modulus: not a canonical field element
        with pytest.raises(cc.CrescentGpuError) as ei:
            prover.prove(_scalars(bad), 1, 2)
        assert ei.value.code == -1 and "modulus" in str(ei.value)
        with pytest.raises(cc.CrescentGpuError):
            prover.witness_map(_scalars(bad))
        # the context stays usable
        case = g["proofs"][1]
        assert prover.prove(_scalars(w), int(case["r"], 16), int(case["s"], 16)).data.hex() == case["proof"]
    finally:
        prover.close()


def test_rejected_witnesses_among_proofs_in_flight(cc, oracle):
    """A caller that hands over a non-canonical assignment gets its error; the proofs of the other callers in flight on
    the same context at that moment are untouched (the rejection flag belongs to the proof slot), host and device witnesses
    alike."""
    from concurrent.futures import ThreadPoolExecutor
    from crescent_credentials_amd import workloads as wl
    l, m, M = 6, 3_000, 3_100
    cm, w = wl.synthetic_circuit(78, l, m, M, 0.5, 3)
    rng = random.Random(6)
    pk = cc.generate_parameters_with_qap(cm, *(rng.randrange(1, oracle.R) for _ in range(4)))
    bad = w.copy()
    bad.reshape(-1, 32)[M // 2] = np.frombuffer(oracle.R.to_bytes(32, "little"), dtype=np.uint8)
    jobs = [(k % 3 == 1, rng.randrange(oracle.R), rng.randrange(oracle.R)) for k in range(36)]
    serial = cc.Prover(pk, cm)
    par = cc.Prover(pk, cm, proof_slots=4)
    try:
        expect = [None if is_bad else serial.prove(w, r, s).data for is_bad, r, s in jobs]

        def one(job):
            is_bad, r, s = job
            try:
                return par.prove(bad if is_bad else w, r, s).data
            except cc.CrescentGpuError as e:
                assert e.code == -1 and "modulus" in str(e)
                return None
        with ThreadPoolExecutor(max_workers=6) as ex:
            got = list(ex.map(one, jobs))
        assert got == expect
    finally:
        serial.close(); par.close()



@pytest.mark.parametrize("n,contig,slots", [(4, False, 1), (8, False, 1), (4, True, 1), (4, False, 3)],
                         ids=["4-strided", "8-strided", "4-contiguous", "4-strided-throughput-slots"])
def test_sharded_proof_in_two_calls(cc, oracle, n, contig, slots):
    """cg_prove_partial_q_begin / cg_partial_witness_map_coset / cg_prove_partial_q_finish: every shard opens the proof (its l, a,
    b1, b2 partial sums are queued and run), the first shard's witness map runs ON the open proof's working set - a one-slot
    context has no other - and the h share follows with the slice.  The 384-byte records are the one-call form's, the assembled
    proof the unsharded context's; satisfying and arbitrary assignments, r = 0, slices in host and in device memory, latency
    (one slot, five streams) and throughput (several slots, one stream) contexts; abort gives the slot back."""
    import torch
    from crescent_credentials_amd import workloads as wl
    l, m, M = _CPU_SHAPES["log14"]
    cm, w = wl.synthetic_circuit(83 + n, l, m, M, 0.6, 3, profile="gates")
    rng = random.Random(18 + n)
    pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, oracle.R) for _ in range(4)])
    w_bad = _scalars([rng.randrange(oracle.R) for _ in range(M)])
    whole = cc.Prover(pk, cm)
    first = cc.Prover(pk, cm, shard_rank=0, shard_count=n, contiguous_h_shards=contig, proof_slots=slots)
    others = [cc.Prover(pk, cm, shard_rank=k, shard_count=n, contiguous_h_shards=contig, h_scalars_external=True, proof_slots=slots)
              for k in range(1, n)]
    shards = [first] + others
    try:
        D = whole.domain_size
        slices = [first.h_scalars_slice(k) for k in range(n)]
        for wit in (w, w_bad):
            wd = torch.from_numpy(np.ascontiguousarray(wit)).cuda()
            for r, s in ((0, 0), (rng.randrange(oracle.R), rng.randrange(oracle.R))):
                want = whole.prove(wit, r, s).data
                q_ref = first.witness_map_coset(wit)
                one_call = b"".join(p.prove_partial_q(wit, q_ref[o * 32:(o + c) * 32], r) for p, (o, c) in zip(shards, slices))
                # host memory throughout
                opened = [p.prove_partial_q_begin(wit, r) for p in shards]
                q = opened[0].witness_map_coset()
                assert bytes(q) == bytes(q_ref)
                parts = b"".join(op.finish(q[o * 32:(o + c) * 32]) for op, (o, c) in zip(opened, slices))
                assert parts == one_call and others[-1].assemble(parts, n, r, s).data == want
                # device memory throughout
                qd = torch.empty(D * 32, dtype=torch.uint8, device="cuda")
                opened = [p.prove_partial_q_begin(wd.data_ptr(), r, on_device=True) for p in shards]
                opened[0].witness_map_coset(out_dev=qd.data_ptr())
                parts = b"".join(op.finish(qd.data_ptr() + o * 32, q_on_device=True) for op, (o, c) in zip(opened, slices))
                assert parts == one_call
        # the witness map in two halves (cg_witness_map_coset_half / cg_partial_witness_map_coset_half / cg_prove_partial_q_finish2):
        # q is the product of the two sides mod r, each side is laid out like q, and a shard that multiplies its own two slices
        # on the GPU produces the record it produces from q's slice
        for wit in (w, w_bad):
            r = rng.randrange(oracle.R)
            q_ref = first.witness_map_coset(wit)
            a_side, b_side = first.witness_map_coset_half(wit, 0), first.witness_map_coset_half(wit, 1)
            qi = lambda arr, j: int.from_bytes(arr[32 * j:32 * j + 32].tobytes(), "little")
            for j in (0, 1, 5, D // 2 + 3, D - 1):
                assert qi(a_side, j) * qi(b_side, j) % oracle.R == qi(q_ref, j) and qi(a_side, j) < oracle.R and qi(b_side, j) < oracle.R
            one_call = b"".join(p.prove_partial_q(wit, q_ref[o * 32:(o + c) * 32], r) for p, (o, c) in zip(shards, slices))
            opened = [p.prove_partial_q_begin(wit, r) for p in shards]
            assert bytes(opened[0].witness_map_coset_half(0)) == bytes(a_side) and bytes(opened[0].witness_map_coset_half(1)) == bytes(b_side)
            parts = b"".join(op.finish2(a_side[o * 32:(o + c) * 32], b_side[o * 32:(o + c) * 32]) for op, (o, c) in zip(opened, slices))
            assert parts == one_call
            wd = torch.from_numpy(np.ascontiguousarray(wit)).cuda()
            ad, bd = torch.empty(D * 32, dtype=torch.uint8, device="cuda"), torch.empty(D * 32, dtype=torch.uint8, device="cuda")
            first.witness_map_coset_half(wd.data_ptr(), 0, on_device=True, out_dev=ad.data_ptr())
            opened = [p.prove_partial_q_begin(wd.data_ptr(), r, on_device=True) for p in shards]
            opened[0].witness_map_coset_half(1, out_dev=bd.data_ptr())
            assert bytes(ad.cpu().numpy()) == bytes(a_side) and bytes(bd.cpu().numpy()) == bytes(b_side)
            parts = b"".join(op.finish2(ad.data_ptr() + o * 32, bd.data_ptr() + o * 32, on_device=True) for op, (o, c) in zip(opened, slices))
            assert parts == one_call
        with pytest.raises(cc.CrescentGpuError):
            first.witness_map_coset_half(w, 2)                                        # which is 0 or 1
        o1, c1 = slices[1]
        a_bad = a_side[o1 * 32:(o1 + c1) * 32].copy()
        a_bad[32:64] = np.frombuffer(oracle.R.to_bytes(32, "little"), np.uint8)
        with pytest.raises(cc.CrescentGpuError):                                      # a non-canonical operand of the product is refused
            others[0].prove_partial_q_begin(w_bad, r).finish2(a_bad, b_side[o1 * 32:(o1 + c1) * 32])
        # an open proof holds its slot; abort gives it back; a bad slice fails the finish and gives it back too
        r = 5
        q = first.witness_map_coset(w)
        o1, c1 = slices[1]
        for _ in range(slots + 2):
            others[0].prove_partial_q_begin(w, r).abort()
        q_bad = q[o1 * 32:(o1 + c1) * 32].copy()
        q_bad[32 * 2:32 * 3] = np.frombuffer(oracle.R.to_bytes(32, "little"), np.uint8)
        for _ in range(slots + 1):
            with pytest.raises(cc.CrescentGpuError):
                others[0].prove_partial_q_begin(w, r).finish(q_bad)
        assert others[0].prove_partial_q_begin(w, r).finish(q[o1 * 32:(o1 + c1) * 32]) == others[0].prove_partial_q(w, q[o1 * 32:(o1 + c1) * 32], r)
        with pytest.raises(cc.CrescentGpuError):
            others[0].prove_partial_q_begin(w, r).witness_map_coset()                 # no witness-map resources on an external shard
        with pytest.raises(cc.CrescentGpuError):
            whole.prove_partial_q_begin(w, r)                                         # needs a sharded context
        # ... and the slots are all still there: as many open proofs as slots, at once
        held = [others[0].prove_partial_q_begin(w, r) for _ in range(slots)]
        outs = [h.finish(q[o1 * 32:(o1 + c1) * 32]) for h in held]
        assert len(set(outs)) == 1
    finally:
        whole.close()
        for p in shards:
            p.close()


def test_unequal_shares_of_a_sharded_proof(cc, oracle):
    """cg_options.shard_span: a shard owns [n·lo/10000, n·hi/10000) of every query instead of an equal part, so that the ranks
    that also compute (half of) the witness map carry less of the MSMs.  Four shards of 8 / 17 / 30 / 45 %: the one-call partial
    sums (every shard runs the witness map for its contiguous range of coset points) and the two-halves flow (slices cut by the
    spans, products on the shards) both assemble to the unsharded context's bytes; cg_h_scalars_slice answers for the own shard."""
    from crescent_credentials_amd import workloads as wl
    l, m, M = _CPU_SHAPES["log14"]
    cm, w = wl.synthetic_circuit(97, l, m, M, 0.6, 3, profile="gates")
    rng = random.Random(28)
    pk = cc.generate_parameters_with_qap(cm, *[rng.randrange(1, oracle.R) for _ in range(4)])
    spans = [(0, 800), (800, 2500), (2500, 5500), (5500, 10000)]
    whole = cc.Prover(pk, cm)
    shards = [cc.Prover(pk, cm, shard_rank=k, shard_count=4, shard_span=spans[k], h_scalars_external=(k > 1)) for k in range(4)]
    try:
        D = whole.domain_size
        slices = [(D * lo // 10000, D * hi // 10000 - D * lo // 10000) for lo, hi in spans]
        assert [p.h_scalars_slice(k) for k, p in enumerate(shards)] == slices and sum(c for _, c in slices) == D
        with pytest.raises(cc.CrescentGpuError):
            shards[0].h_scalars_slice(1)                                 # a span context knows its own shard only
        for r, s in ((0, 0), (rng.randrange(oracle.R), rng.randrange(oracle.R))):
            want = whole.prove(w, r, s).data
            # recompute on the two shards that can, the slices of q on the two that cannot
            q = shards[0].witness_map_coset(w)                           # natural order (contiguous shards)
            assert bytes(q) == bytes(shards[1].witness_map_coset(w))
            parts = b"".join(p.prove_partial(w, r) if k < 2 else p.prove_partial_q(w, q[o * 32:(o + c) * 32], r)
                             for k, (p, (o, c)) in enumerate(zip(shards, slices)))
            assert shards[3].assemble(parts, 4, r, s).data == want
            # the two halves, products on the shards
            opened = [p.prove_partial_q_begin(w, r) for p in shards]
            a_side, b_side = opened[0].witness_map_coset_half(0), opened[1].witness_map_coset_half(1)
            parts2 = b"".join(op.finish2(a_side[o * 32:(o + c) * 32], b_side[o * 32:(o + c) * 32]) for op, (o, c) in zip(opened, slices))
            assert parts2 == parts
        for bad in ((5, 5), (7000, 6000), (0, 10001)):
            with pytest.raises(cc.CrescentGpuError):
                cc.Prover(pk, cm, shard_rank=0, shard_count=4, shard_span=bad)
        with pytest.raises(cc.CrescentGpuError):
            cc.Prover(pk, cm, shard_span=(0, 5000))                      # needs a sharded context
    finally:
        whole.close()
        for p in shards:
            p.close()
