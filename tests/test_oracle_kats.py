"""Pins the oracle (oracle/bn254_oracle.py) against every known-answer datum the reference's own tests
hold for this path: field/curve encodings, the Fr modulus, the .r1cs parser KAT, and
the Groth16 verification equation as the acceptance criterion."""
import random

from conftest import load_golden

K = load_golden("reference_kats.json")


def test_fq_montgomery_one_bytes(oracle):
    # forks/circom-compat/src/zkey.rs:397-402 (`fq_buf`): LE bytes of R mod q
    assert (oracle.MONT_R % oracle.Q).to_bytes(32, "little") == bytes(K["fq_montgomery_one_le"])


def test_g1_generator_montgomery_bytes(oracle):
    # zkey.rs:408-415 (`g1_buf`) = Mont(1) ‖ Mont(2); generator (1, 2) per zkey.rs:434-439
    x, y = oracle.G1_GEN
    enc = (x * oracle.MONT_R % oracle.Q).to_bytes(32, "little") + (y * oracle.MONT_R % oracle.Q).to_bytes(32, "little")
    assert enc == bytes(K["g1_generator_montgomery_le"])
    assert oracle.G1.is_on_curve(oracle.G1_GEN)


def test_g2_generator_bytes_and_decimals(oracle):
    # zkey.rs:421-431 (`g2_buf`, order x.c0 ‖ x.c1 ‖ y.c0 ‖ y.c1) and zkey.rs:442-460 (decimals)
    d = K["g2_generator_decimal"]
    gen = ((int(d["x_c0"]), int(d["x_c1"])), (int(d["y_c0"]), int(d["y_c1"])))
    assert gen == oracle.G2_GEN
    enc = b"".join((c * oracle.MONT_R % oracle.Q).to_bytes(32, "little") for c in (gen[0][0], gen[0][1], gen[1][0], gen[1][1]))
    assert enc == bytes(K["g2_generator_montgomery_le"])
    assert oracle.G2.is_on_curve(oracle.G2_GEN)
    assert oracle.G2.to_affine(oracle.G2.mul_affine(oracle.G2_GEN, oracle.R)) is None


def test_fr_modulus(oracle):
    # r1cs_reader.rs:183 and witness_calculator.rs:464-467
    assert oracle.R.to_bytes(32, "little").hex() == K["fr_modulus_le_hex"]
    assert "%064x" % oracle.R == K["fr_modulus_hex"]
    assert oracle.FR_MODULUS_LE.hex() == K["fr_modulus_le_hex"]


def test_root_of_unity(oracle):
    w = oracle.FR_ROOT_2_28
    assert pow(w, 1 << 28, oracle.R) == 1 and pow(w, 1 << 27, oracle.R) != 1
    assert w == 0x2A3C09F0A58A7E8500E0A7EB8EF62ABC402D111E41112ED49BD61B6E725B19F0
    assert oracle.root_of_unity(1 << 21) == 0x1DED8980AE2BDD1A4222150E8598FC8C58F50577CA5A5CE3B2C87885FCD0B523


def test_r1cs_sample(oracle):
    # r1cs_reader.rs:264-345
    e = K["r1cs_sample_expected"]
    f = oracle.parse_r1cs(bytes.fromhex(K["r1cs_sample_hex"]))
    h = f["header"]
    assert f["version"] == e["version"] and h["field_size"] == e["field_size"]
    assert (h["n_wires"], h["n_pub_out"], h["n_pub_in"], h["n_prv_in"]) == (e["n_wires"], e["n_pub_out"], e["n_pub_in"], e["n_prv_in"])
    assert h["n_labels"] == e["n_labels"] and h["n_constraints"] == e["n_constraints"]
    c = f["constraints"]
    assert len(c) == e["n_constraints_len"] and len(c[0][0]) == e["c0_a_len"]
    assert c[0][0][0] == (e["c0_a0_wire"], e["c0_a0_coeff"])
    assert c[2][1][0] == (e["c2_b0_wire"], e["c2_b0_coeff"])
    assert len(c[1][2]) == e["c1_c_len"]
    assert len(f["wire_mapping"]) == e["wire_mapping_len"] and f["wire_mapping"][1] == e["wire_mapping_1"]
    mats, l, m, M = oracle.r1cs_to_matrices(f)
    assert (l, m, M) == (1 + 2 + 1, 3, 7)       # r1cs_reader.rs:28-30


def test_r1cs_rejections(oracle):
    import pytest
    good = bytearray(bytes.fromhex(K["r1cs_sample_hex"]))
    bad = bytearray(good); bad[0] ^= 1
    with pytest.raises(ValueError, match="magic"):
        oracle.parse_r1cs(bytes(bad))
    bad = bytearray(good); bad[4] = 2
    with pytest.raises(ValueError, match="version"):
        oracle.parse_r1cs(bytes(bad))
    bad = bytearray(good); bad[28] ^= 1          # first byte of the prime
    with pytest.raises(ValueError, match="bn256"):
        oracle.parse_r1cs(bytes(bad))


def test_alt_bn128_doubling_vector(oracle):
    """An EXTERNAL known answer (not from the reference tree, which pins no group arithmetic): 2·G1 on alt_bn128 / BN254 as
    every EIP-196 ecAdd test suite carries it.  Pins the oracle's G1 addition law and base field to the public curve."""
    x2 = 0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3
    y2 = 0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4
    assert oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, 2)) == (x2, y2)
    assert (y2 * y2 - x2 * x2 * x2 - 3) % oracle.Q == 0
    # and through the MSM path: 1·G + 1·G
    assert oracle.G1.to_affine(oracle.G1.msm([oracle.G1_GEN, oracle.G1_GEN], [1, 1])) == (x2, y2)


# Further EXTERNAL known answers, from the published test data and constants of other BN254 / alt_bn128 implementations
# (none of them in the reference tree, which pins no arithmetic): py_ecc's bn128 curve tests and the EIP-196 / EIP-197
# precompile suites carry 3·G1 and 2·G2; gnark-crypto (ecc/bn254/fr/fft, `rootOfUnity`, multiplicative generator 5) and
# ark-ff's derivation from `GENERATOR = 5` give the 2^28-th root of unity of Fr that ark-poly's radix-2 domains are built on.
G1_TIMES_3 = (3353031288059533942658390886683067124040920775575537747144343083137631628272,
              19321533766552368860946552437480515441416830039777911637913418824951667761761)
G2_TIMES_2 = ((18029695676650738226693292988307914797657423701064905010927197838374790804409,
               14583779054894525174450323658765874724019480979794335525732096752006891875705),
              (2140229616977736810657479771656733941598412651537078903776637920509952744750,
               11474861747383700316476719153975578001603231366361248090558603872215261634898))
FR_ROOT_OF_UNITY_2_28 = 19103219067921713944291392827692070036145651957329286315305642004821462161904


def test_external_group_vectors(oracle):
    """3·G1 = G1 + 2·G1 (a mixed addition of DISTINCT points, which the doubling vector does not exercise) and 2·G2 (the
    Fq2 arithmetic: u^2 = -1, component order c0 then c1, twist coefficient 3/(9+u))"""
    g, g2 = oracle.G1_GEN, oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, 2))
    assert oracle.G1.to_affine(oracle.G1.mul_affine(g, 3)) == G1_TIMES_3
    assert oracle.G1.to_affine(oracle.G1.add_affine(oracle.G1.to_jac(g2), g)) == G1_TIMES_3
    assert oracle.G1.to_affine(oracle.G1.msm([g, g2], [1, 1])) == G1_TIMES_3
    assert oracle.G2.to_affine(oracle.G2.mul_affine(oracle.G2_GEN, 2)) == G2_TIMES_2
    assert oracle.G2.to_affine(oracle.G2.msm([oracle.G2_GEN, oracle.G2_GEN], [1, 1])) == G2_TIMES_2
    (x0, x1), (y0, y1) = G2_TIMES_2                     # on the twist y^2 = x^3 + 3/(9+u)
    q = oracle.Q
    f2mul = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)
    x3 = f2mul(f2mul((x0, x1), (x0, x1)), (x0, x1))
    inv9u = (9 * pow(82, -1, q) % q, -pow(82, -1, q) % q)          # 1/(9+u) = (9-u)/82
    b = f2mul((3, 0), inv9u)
    y2 = f2mul((y0, y1), (y0, y1))
    assert y2 == ((x3[0] + b[0]) % q, (x3[1] + b[1]) % q)


def test_external_root_of_unity(oracle):
    """the [ark-mem] constants of SURVEY Appendix A, pinned from outside: Fr's multiplicative generator 5 and its 2^28-th
    root of unity - the point set of every evaluation domain (r1cs_to_qap.rs:156,179-185), hence of h"""
    w28 = pow(5, (oracle.R - 1) >> 28, oracle.R)
    assert w28 == FR_ROOT_OF_UNITY_2_28
    assert oracle.FR_GENERATOR == 5
    for logn in (1, 3, 10, 21, 28):
        assert oracle.root_of_unity(1 << logn) == pow(FR_ROOT_OF_UNITY_2_28, 1 << (28 - logn), oracle.R)


def test_pairing_bilinear(oracle):
    from bn254_oracle import _f12_one, _f12_pow
    e1 = oracle.pairing(oracle.G1_GEN, oracle.G2_GEN)
    assert e1 != _f12_one() and _f12_pow(e1, oracle.R) == _f12_one()
    a, b = 0x1234567, 0x7654321
    pa = oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, a))
    qb = oracle.G2.to_affine(oracle.G2.mul_affine(oracle.G2_GEN, b))
    assert oracle.pairing(pa, qb) == _f12_pow(e1, a * b)


def test_serialisation_flags(oracle):
    # SWFlags [ark-mem]: bit 7 <=> y > -y; bit 6 <=> infinity with zero coordinates
    P = oracle.G1_GEN
    enc = oracle.g1_uncompressed(P)
    assert enc[63] & 0x80 == 0 and enc[:32] == (1).to_bytes(32, "little")         # y = 2 < q - 2
    encn = oracle.g1_uncompressed(oracle.G1.neg_affine(P))
    assert encn[63] & 0x80 == 0x80
    assert oracle.g1_uncompressed(None) == bytes(63) + b"\x40"
    assert oracle.g2_uncompressed(None) == bytes(127) + b"\x40"
    q = oracle.G2_GEN
    f1, f2 = oracle.g2_uncompressed(q)[127] & 0x80, oracle.g2_uncompressed(oracle.G2.neg_affine(q))[127] & 0x80
    assert {f1, f2} == {0, 0x80}
    assert len(oracle.proof_compressed((P, q, P))) == 128 and len(oracle.proof_uncompressed((P, q, P))) == 256


def test_msm_matches_serial_sum(oracle):
    # forks/halo2curves/src/msm.rs:601-636 pattern: Pippenger vs the naive sum
    rng = random.Random(5)
    pts = [oracle.G1.to_affine(oracle.G1.mul_affine(oracle.G1_GEN, rng.randrange(1, oracle.R))) for _ in range(20)] + [None]
    sc = [rng.randrange(oracle.R) for _ in range(19)] + [0, 1]
    assert oracle.G1.to_affine(oracle.G1.msm(pts, sc)) == oracle.G1.to_affine(oracle.G1.msm_naive(pts, sc))


def h2c_points(oracle):
    """the BN254 points the reference's tree holds (forks/halo2curves/src/bn256/curve.rs:307-420)"""
    k = K["halo2curves_bn256_hash_to_curve_points"]
    g1 = [(int(x, 16), int(y, 16)) for x, y in k["g1"]]
    g2 = [((int(x0, 16), int(x1, 16)), (int(y0, 16), int(y1, 16))) for x0, x1, y0, y1 in k["g2"]]
    return g1, g2


def test_points_held_by_the_reference_tree(oracle):
    """Five G1 and five G2 points of BN254 from the reference's own tests (the in-tree halo2curves fork, taken by it from
    gnark-crypto): each is on the oracle's curve - y² = x³ + 3 and the twist y² = x³ + 3/(9 + u) with Fq2 = (c0, c1) in the
    order the prover's encodings use - and [r - 1]·P = -P by the oracle's group law (doublings and additions along 254 bits
    of an independent point, over Fq and over Fq2: a wrong formula does not land on -P), and through the C restatement's
    MSM; sums of several of them agree between the two."""
    import cpu_ref
    g1, g2 = h2c_points(oracle)
    rm1 = (oracle.R - 1).to_bytes(32, "little")
    for P in g1:
        assert oracle.G1.is_on_curve(P)
        assert oracle.G1.to_affine(oracle.G1.mul_affine(P, oracle.R - 1)) == oracle.G1.neg_affine(P)
        assert cpu_ref.msm_g1(oracle.g1_packed(P), rm1) == oracle.g1_packed(oracle.G1.neg_affine(P))
    for P in g2:
        assert oracle.G2.is_on_curve(P)
        assert oracle.G2.to_affine(oracle.G2.mul_affine(P, oracle.R - 1)) == oracle.G2.neg_affine(P)
        assert cpu_ref.msm_g2(oracle.g2_packed(P), rm1) == oracle.g2_packed(oracle.G2.neg_affine(P))
    ks = [oracle.R - 2, 1, 2, 0x1234567890ABCDEF << 100, 3]
    sc = b"".join(k.to_bytes(32, "little") for k in ks)
    want1 = oracle.g1_packed(oracle.G1.to_affine(oracle.G1.msm_naive(g1, ks)))
    want2 = oracle.g2_packed(oracle.G2.to_affine(oracle.G2.msm_naive(g2, ks)))
    assert cpu_ref.msm_g1(b"".join(oracle.g1_packed(P) for P in g1), sc) == want1
    assert cpu_ref.msm_g2(b"".join(oracle.g2_packed(P) for P in g2), sc) == want2


def test_fq2_constants_held_by_the_reference_tree(oracle):
    """The in-tree halo2curves fork spells out xi^((q^k - 1)/6), k = 1..3, and xi^((q - 1)/2) for xi = 9 + u as Montgomery limbs
    (forks/halo2curves/src/bn256/fq12.rs:40-100, engine.rs:164-177).  The oracle's Fq2 multiplication and squaring - the
    arithmetic under its G2 group law and its pairing - must reach them through 250 to 760 steps each."""
    k = K["halo2curves_bn256_fq2_constants"]
    Q, F = oracle.Q, oracle.Fq2Ops
    rinv = pow(1 << 256, Q - 2, Q)
    canon = lambda limbs: sum(int(x, 16) << (64 * i) for i, x in enumerate(limbs)) * rinv % Q
    val = lambda name: (canon(k[name][0]), canon(k[name][1]))

    def fpow(a, e):
        r = (1, 0)
        while e:
            if e & 1:
                r = F.mul(r, a)
            a = F.sqr(a)
            e >>= 1
        return r
    xi = (9, 1)
    assert fpow(xi, (Q - 1) // 6) == val("xi_pow_q1_minus_1_over_6")
    assert fpow(xi, (Q * Q - 1) // 6) == val("xi_pow_q2_minus_1_over_6")
    assert fpow(xi, (Q ** 3 - 1) // 6) == val("xi_pow_q3_minus_1_over_6")
    assert fpow(xi, (Q - 1) // 2) == val("xi_pow_q1_minus_1_over_2")
    # and the twist the G2 points live on is y^2 = x^3 + 3/xi with this xi (forks/halo2curves/src/bn256/curve.rs: G2_B)
    assert F.mul(oracle.G2.b, xi) == (3, 0)


def endo_constants():
    k = K["halo2curves_bn256_endomorphism"]
    return int(k["fr_zeta_lambda"], 16), int(k["fq_zeta_beta"], 16)


def test_endomorphism_known_answer_from_the_reference_tree(oracle):
    """forks/halo2curves tests/curve.rs:413-427 asserts g * Fr::ZETA == g.endo() = (x * Fq::ZETA, y): a 254-bit scalar
    multiplication on BN254 G1 whose answer the reference's tree states.  Through the oracle's double-and-add, its Pippenger,
    and the C restatement's MSM, for the generator and for the five G1 points the tree holds."""
    import cpu_ref
    lam, beta = endo_constants()
    assert pow(lam, 3, oracle.R) == 1 and lam != 1 and pow(beta, 3, oracle.Q) == 1 and beta != 1
    g1, _ = h2c_points(oracle)
    for P in [oracle.G1_GEN] + g1:
        want = (beta * P[0] % oracle.Q, P[1])
        assert oracle.G1.to_affine(oracle.G1.mul_affine(P, lam)) == want
        assert oracle.G1.to_affine(oracle.G1.msm([P], [lam])) == want
        assert cpu_ref.msm_g1(oracle.g1_packed(P), lam.to_bytes(32, "little")) == oracle.g1_packed(want)


def test_bn_parameter_and_miller_loop_digits_held_by_the_reference_tree(oracle):
    """forks/halo2curves/src/bn256/mod.rs:17-24: u and the signed digits of 6u + 2.  Both moduli are polynomials in u, and the
    digit string is the one oracle/ark_files.py walks when it prepares a G2 point (marked [ark-mem] there: now also pinned
    to the tree)."""
    import ark_files
    k = K["halo2curves_bn256_parameter"]
    u = int(k["bn_x"])
    assert 36 * u ** 4 + 36 * u ** 3 + 24 * u ** 2 + 6 * u + 1 == oracle.Q
    assert 36 * u ** 4 + 36 * u ** 3 + 18 * u ** 2 + 6 * u + 1 == oracle.R
    assert sum(d << i for i, d in enumerate(k["six_u_plus_2_naf"])) == 6 * u + 2
    assert ark_files.BN_X == u and list(ark_files.ATE_LOOP_COUNT) == k["six_u_plus_2_naf"]
    assert len(k["six_u_plus_2_naf"]) == 65
