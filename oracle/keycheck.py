"""Acceptance checks for a GPU-made proving key and a GPU-made proof AT ANY SIZE.  TEST INFRASTRUCTURE (the checker of tests/,
smoke() and the checker leg of bench.py; the product never imports it).

Everything the full-size tests and the bench prove on comes out of `cg_setup` (the product).  GPU == cpu_ref on the same
key only says the two provers agree; this module says the key is a Groth16 key for the circuit and that the proof is
accepted, with no code shared with `csrc/setup.hip`:

  verify        the reference's one acceptance criterion (forks/groth16/src/test.rs:70-71, creds/src/lib.rs:286-290,
                verifier.rs:44-65) by the Python oracle's pairing; a flipped public input must be refused.
  key_scalars   a_i(tau), b_i(tau), c_i(tau), zt by oracle/cpu_ref.c `ref_qap_at` (r1cs_to_qap.rs:103-147 restated; pinned to
                the Python oracle in tests/test_keycheck_cpu.py), then the scalars of the five queries as generator.rs:118-194
                defines them.
  check_key     (1) every fixed point of the key and every gamma_abc entry == [scalar]·G by the Python oracle's
                double-and-add; (2) a strided sample of each query the same way; (3) EVERY entry of each query through a
                random linear combination: sum rho_i·Q_i (cpu_ref's Pippenger - the C restatement, not the product)
                == [sum rho_i·s_i]·G.  A single wrong entry fails (3) except with probability ~2^-120; identities where
                the scalar is zero are checked exactly (a zero scalar must be the all-zero record and vice versa).
  closed_form   A, B, C of a proof from the trapdoor: A = [alpha + a(tau) + r·delta]G, B = [beta + b(tau) + s·delta]H,
                C = [sum_aux w_i·l_i + (a(tau)·b(tau) - c(tau))/delta + s·A + r·B - r·s·delta]G, for a SATISFYING
                assignment (then h(tau)·zt = a(tau)·b(tau) - c(tau) exactly; no transform, no MSM over the key).
"""

import numpy as np


def g1_of_proof(oracle, b):
    b = bytearray(b); b[63] &= 0x3F
    return oracle.g1_unpack(bytes(b))


def g2_of_proof(oracle, b):
    b = bytearray(b); b[127] &= 0x3F
    return oracle.g2_unpack(bytes(b))


def decode_proof(oracle, data: bytes):
    pr = (g1_of_proof(oracle, data[:64]), g2_of_proof(oracle, data[64:192]), g1_of_proof(oracle, data[192:256]))
    assert oracle.proof_uncompressed(pr) == bytes(data), "flag bits disagree with the oracle's serialiser"
    return pr


def vk_of(oracle, pk, l):
    v = pk.vk
    return dict(alpha_g1=oracle.g1_unpack(bytes(v.alpha_g1)), beta_g2=oracle.g2_unpack(bytes(v.beta_g2)),
                gamma_g2=oracle.g2_unpack(bytes(v.gamma_g2)), delta_g1=oracle.g1_unpack(bytes(v.delta_g1)),
                delta_g2=oracle.g2_unpack(bytes(v.delta_g2)),
                gamma_abc_g1=[oracle.g1_unpack(bytes(v.gamma_abc_g1[64 * i:64 * i + 64])) for i in range(l)])


def _ints(b, idx=None):
    b = np.asarray(b, np.uint8).reshape(-1, 32)
    if idx is not None:
        b = b[idx]
    raw = b.tobytes()
    return [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]


def verify(oracle, pk, l, w, proof_bytes, expect_bad_rejected=True):
    """verifier.rs:44-65 on the proof bytes; True iff accepted (and, when asked, a flipped public input is refused)."""
    pr = decode_proof(oracle, proof_bytes)
    vk = vk_of(oracle, pk, l)
    pub = _ints(np.asarray(w, np.uint8).reshape(-1, 32)[1:l])
    ok = oracle.verify_proof(vk, pr, pub)
    if ok and expect_bad_rejected and l > 1:
        bad = list(pub); bad[0] ^= 1
        ok = not oracle.verify_proof(vk, pr, bad)
    return ok


def key_scalars(oracle, cpu_ref, cm, l, m, M, trap):
    """trap = (alpha, beta, delta, tau).  -> dict of canonical byte arrays: a, b (M x 32), l (M - l), h (D - 1), gabc (l)"""
    alpha, beta, delta, tau = trap
    a, b, c, zt = cpu_ref.qap_at((cm.a, cm.b, cm.c), l, m, M, tau)
    D = oracle.domain_size_for(m + l)
    dinv = pow(delta, oracle.R - 2, oracle.R)
    comb = cpu_ref.fr_combine(a, b, c, beta, alpha, dinv)                      # generator.rs:124-128
    gabc = cpu_ref.fr_combine(a[:32 * l], b[:32 * l], c[:32 * l], beta, alpha, 1)   # :118-122 with gamma = 1 (:28)
    hq = cpu_ref.fr_powers(zt * dinv % oracle.R, tau, D - 1)                   # r1cs_to_qap.rs:215-225, generator.rs:178
    return dict(a=a, b=b, c=c, l=comb[32 * l:], h=hq, gabc=gabc, zt=zt, D=D)


def _g1(oracle, k):
    return oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, k % oracle.R))


def _g2(oracle, k):
    return oracle.G2.to_affine(oracle.G2.mul_affine(oracle.G2_GEN, k % oracle.R))


def check_key(oracle, cpu_ref, pk, cm, l, m, M, trap, scal=None, nthreads=1, stride=4097, seed=0x5EED):
    alpha, beta, delta, tau = trap
    scal = scal or key_scalars(oracle, cpu_ref, cm, l, m, M, trap)
    R = oracle.R
    # (1) fixed points (generator.rs:196-228: vk + beta_g1/delta_g1) and gamma_abc
    assert oracle.g1_unpack(bytes(pk.vk.alpha_g1)) == _g1(oracle, alpha), "alpha_g1"
    assert oracle.g1_unpack(bytes(pk.beta_g1)) == _g1(oracle, beta), "beta_g1"
    assert oracle.g1_unpack(bytes(pk.delta_g1)) == _g1(oracle, delta), "delta_g1"
    assert oracle.g1_unpack(bytes(pk.vk.delta_g1)) == _g1(oracle, delta), "vk.delta_g1"
    assert oracle.g2_unpack(bytes(pk.vk.beta_g2)) == _g2(oracle, beta), "beta_g2"
    assert oracle.g2_unpack(bytes(pk.vk.delta_g2)) == _g2(oracle, delta), "delta_g2"
    assert oracle.g2_unpack(bytes(pk.vk.gamma_g2)) == oracle.G2_GEN, "gamma_g2 (gamma = 1, generator.rs:28)"
    for i, k in enumerate(_ints(scal["gabc"])):
        assert oracle.g1_unpack(bytes(pk.vk.gamma_abc_g1[64 * i:64 * i + 64])) == _g1(oracle, k), ("gamma_abc_g1", i)
    queries = [("a_query", pk.a_query, scal["a"], 64), ("b_g1_query", pk.b_g1_query, scal["b"], 64),
               ("b_g2_query", pk.b_g2_query, scal["b"], 128), ("h_query", pk.h_query, scal["h"], 64),
               ("l_query", pk.l_query, scal["l"], 64)]
    rng = np.random.default_rng(seed)
    for name, q, sc, width in queries:
        n = sc.size // 32
        assert q.size == n * width, (name, "length", q.size // width, n)     # data_structures.rs:101-118 lengths
        if n == 0:
            continue
        Q = np.asarray(q, np.uint8).reshape(n, width)
        S = np.asarray(sc, np.uint8).reshape(n, 32)
        # identities exactly where the scalar is zero (generator.rs:140,162,168: zero columns of the QAP)
        zs, zq = ~S.any(axis=1), ~Q.any(axis=1)
        assert np.array_equal(zs, zq), (name, "identity pattern", int(zs.sum()), int(zq.sum()))
        # (2) strided sample by the Python oracle's scalar multiplication
        idx = sorted(set(list(range(0, n, stride)) + [0, n - 1, n // 2]))
        ks = _ints(S, idx)
        for i, k in zip(idx, ks):
            if width == 64:
                assert oracle.g1_unpack(Q[i].tobytes()) == _g1(oracle, k), (name, i)
            else:
                assert oracle.g2_unpack(Q[i].tobytes()) == _g2(oracle, k), (name, i)
        # (3) all of it: sum rho_i Q_i == [sum rho_i s_i] G, rho_i 120-bit (cpu_ref's MSM is the C restatement)
        rho = np.zeros((n, 32), np.uint8)
        rho[:, :15] = rng.integers(0, 256, size=(n, 15), dtype=np.uint8)
        k = cpu_ref.fr_inner(rho, S)
        if width == 64:
            got = oracle.g1_unpack(cpu_ref.msm_g1(Q, rho, nthreads=nthreads))
            assert got == _g1(oracle, k), (name, "random linear combination")
        else:
            got = oracle.g2_unpack(cpu_ref.msm_g2(Q, rho, nthreads=nthreads))
            assert got == _g2(oracle, k), (name, "random linear combination")
    return scal


def closed_form(oracle, cpu_ref, scal, trap, r, s, w, l):
    """(A, B, C) affine from the trapdoor, for a satisfying assignment w (canonical bytes)."""
    alpha, beta, delta, tau = trap
    R = oracle.R
    w = np.asarray(w, np.uint8).reshape(-1)
    at = cpu_ref.fr_inner(w, scal["a"]); bt = cpu_ref.fr_inner(w, scal["b"]); ct = cpu_ref.fr_inner(w, scal["c"])
    a_s = (alpha + at + r * delta) % R
    b_s = (beta + bt + s * delta) % R
    dinv = pow(delta, R - 2, R)
    c_s = (cpu_ref.fr_inner(w[32 * l:], scal["l"]) + (at * bt - ct) * dinv + s * a_s + r * b_s - r * s % R * delta) % R
    return (_g1(oracle, a_s), _g2(oracle, b_s), _g1(oracle, c_s))
