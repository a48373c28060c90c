"""TEST INFRASTRUCTURE (never imported by the product): the file formats either side of the prove step, written and
read independently of the product's C++ (csrc/serialize.hip, csrc/r1cs.hip), plus the prepared-pairing verification
the reference's caller runs on a fresh proof.

Restates, with citations into /root/reference:
  * forks/groth16/src/verifier.rs:13-20   prepare_verifying_key
  * forks/groth16/src/verifier.rs:44-65   verify_proof_with_prepared_inputs (one multi-Miller loop against
                                          pvk.gamma_g2_neg_pc / pvk.delta_g2_neg_pc, compared with alpha_g1_beta_g2)
  * forks/groth16/src/data_structures.rs:62-71   PreparedVerifyingKey field order
  * creds/src/lib.rs:58-63                ProverParams = groth16_params ‖ groth16_pvk ‖ config_str
  * creds/src/groth16rand.rs:23-35        ClientState field order
  * forks/circom-compat/src/circom/r1cs_reader.rs:54-256   the iden3 .r1cs container (writer side here)
  * creds/src/structs.rs:26-68            io_locations.sym rows

PARITY STATUS: "parity unpinned".  `G2Prepared`, the Miller loop over prepared coefficients and the final
exponentiation live in ark-ec 0.4 `models/bn` (third-party, not in the tree, no Cargo.lock); they are restated here
from the published algorithm [ark-mem] and anchored on what can be checked without arkworks:
  - ATE_LOOP_COUNT sums to 6u + 2 and the hard-part exponent is a multiple of (q^4 - q^2 + 1)/r;
  - the pairing computed through the prepared coefficients is bilinear and non-degenerate, and its final exponentiation
    is bn254_oracle's plain (q^12-1)/r power raised to that multiple (tests/test_file_formats.py), so it accepts exactly
    the proofs bn254_oracle.verify_proof accepts;
  - the total size of a ClientState with these structures reproduces the 39 KB the reference's README states for its
    pre-generated rs256 client_state.bin (creds/test-vectors/README.md:5-10).
"""
import struct

import bn254_oracle as o

Q, R = o.Q, o.R
F2 = o.Fq2Ops
BN_X = 4965661367192848881
# ark-bn254 Config::ATE_LOOP_COUNT [ark-mem]: 6x + 2 in signed binary digits, least significant first
ATE_LOOP_COUNT = [0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0, 1, 1, 1, 0, 0, -1,
                  0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 1, 0, 0, -1, 0, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, 1, 1]
assert sum(b << i for i, b in enumerate(ATE_LOOP_COUNT)) == 6 * BN_X + 2

XI = (9, 1)


def _fq2_pow(a, e):
    r = F2.one
    for bit in bin(e)[2:]:
        r = F2.sqr(r)
        if bit == "1":
            r = F2.mul(r, a)
    return r


TWIST_MUL_BY_Q_X = _fq2_pow(XI, (Q - 1) // 3)
TWIST_MUL_BY_Q_Y = _fq2_pow(XI, (Q - 1) // 2)
TWO_INV = pow(2, Q - 2, Q)


def _conj(a):
    return (a[0], (-a[1]) % Q)


def _mul_by_char(P):
    """ark-ec bn::g2::mul_by_char: the q-power Frobenius carried to the twist"""
    return (F2.mul(_conj(P[0]), TWIST_MUL_BY_Q_X), F2.mul(_conj(P[1]), TWIST_MUL_BY_Q_Y))


def g2_prepare(Qaff):
    """ark-ec bn::G2Prepared::from(G2Affine) [ark-mem]: line coefficients of the optimal-ate Miller loop in homogeneous
    projective coordinates (Costello-Lange-Naehrig doubling / addition steps), D-type twist ordering."""
    if Qaff is None:
        return {"ell_coeffs": [], "infinity": True}
    rx, ry, rz = Qaff[0], Qaff[1], F2.one
    coeffs = []

    def double():
        nonlocal rx, ry, rz
        a = F2.muli(F2.mul(rx, ry), TWO_INV)
        b = F2.sqr(ry)
        c = F2.sqr(rz)
        e = F2.mul(o.B2, F2.add(F2.add(c, c), c))
        f = F2.add(F2.add(e, e), e)
        g = F2.muli(F2.add(b, f), TWO_INV)
        h = F2.sub(F2.sqr(F2.add(ry, rz)), F2.add(b, c))
        i = F2.sub(e, b)
        j = F2.sqr(rx)
        e_sq = F2.sqr(e)
        rx = F2.mul(a, F2.sub(b, f))
        ry = F2.sub(F2.sqr(g), F2.add(F2.add(e_sq, e_sq), e_sq))
        rz = F2.mul(b, h)
        coeffs.append((F2.neg(h), F2.add(F2.add(j, j), j), i))

    def add(P):
        nonlocal rx, ry, rz
        theta = F2.sub(ry, F2.mul(P[1], rz))
        lam = F2.sub(rx, F2.mul(P[0], rz))
        c = F2.sqr(theta)
        d = F2.sqr(lam)
        e = F2.mul(lam, d)
        f = F2.mul(rz, c)
        g = F2.mul(rx, d)
        h = F2.sub(F2.add(e, f), F2.add(g, g))
        rx = F2.mul(lam, h)
        ry = F2.sub(F2.mul(theta, F2.sub(g, h)), F2.mul(e, ry))
        rz = F2.mul(rz, e)
        j = F2.sub(F2.mul(theta, P[0]), F2.mul(lam, P[1]))
        coeffs.append((lam, F2.neg(theta), j))

    negq = (Qaff[0], F2.neg(Qaff[1]))
    for bit in list(reversed(ATE_LOOP_COUNT))[1:]:
        double()
        if bit == 1:
            add(Qaff)
        elif bit == -1:
            add(negq)
    q1 = _mul_by_char(Qaff)
    q2 = _mul_by_char(q1)
    q2 = (q2[0], F2.neg(q2[1]))
    add(q1)
    add(q2)
    return {"ell_coeffs": coeffs, "infinity": False}


# ---- Fq12 as the tower Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - xi), carried on bn254_oracle's flat basis -----------
# (w^6 = xi = 9 + u): the Fq2 coefficient of v^i w^j sits at powers w^k and w^(k+6), k = 2i + j
def _tower_to_flat(coeffs):
    """coeffs[(i, j)] -> Fq2; returns the 12 flat coefficients"""
    out = [0] * 12
    for (i, j), a in coeffs.items():
        k = 2 * i + j
        out[k] = (out[k] + a[0] - 9 * a[1]) % Q
        out[k + 6] = (out[k + 6] + a[1]) % Q
    return out


def _flat_to_tower(f):
    out = {}
    for i in range(3):
        for j in range(2):
            k = 2 * i + j
            y = f[k + 6] % Q
            out[(i, j)] = ((f[k] + 9 * y) % Q, y)
    return out


def fq12_bytes(f) -> bytes:
    """ark-serialize of Fp12 [ark-mem]: c0 ‖ c1 (Fq6 each: c0 ‖ c1 ‖ c2 in v; Fq2 each: c0 ‖ c1), 384 bytes"""
    t = _flat_to_tower(f)
    return b"".join(o.fe_bytes(t[(i, j)][0]) + o.fe_bytes(t[(i, j)][1]) for j in range(2) for i in range(3))


def fq12_from_bytes(b: bytes):
    vals = [int.from_bytes(b[32 * k:32 * k + 32], "little") for k in range(12)]
    t = {}
    n = 0
    for j in range(2):
        for i in range(3):
            t[(i, j)] = (vals[n], vals[n + 1])
            n += 2
    return _tower_to_flat(t)


def _ell(f, coeff, P):
    """ark-ec bn::Bn::ell, D-type twist: f *= (c0·y_P) + (c1·x_P)·w + c2·v·w   (mul_by_034)"""
    c0, c1, c2 = coeff
    line = _tower_to_flat({(0, 0): F2.muli(c0, P[1]), (0, 1): F2.muli(c1, P[0]), (1, 1): c2})
    return o._f12_mul(f, line)


def multi_miller_loop(pairs):
    """ark-ec bn::Bn::multi_miller_loop over (G1 affine, prepared G2) pairs [ark-mem]; X_IS_NEGATIVE = false for BN254"""
    live = [(P, iter(pq["ell_coeffs"])) for P, pq in pairs if P is not None and not pq["infinity"]]
    f = o._f12_one()
    n = len(ATE_LOOP_COUNT)
    for i in range(n - 1, 0, -1):
        if i != n - 1:
            f = o._f12_mul(f, f)
        for P, it in live:
            f = _ell(f, next(it), P)
        if ATE_LOOP_COUNT[i - 1] != 0:
            for P, it in live:
                f = _ell(f, next(it), P)
    for _ in range(2):
        for P, it in live:
            f = _ell(f, next(it), P)
    for _, it in live:
        assert next(it, None) is None
    return f


# hard part of ark-ec's BN final exponentiation (Fuentes-Castaneda, Knapp, Rodriguez-Henriquez): the exponent it realises
_Z = BN_X
HARD_EXP = (Q ** 3 * (12 * _Z ** 3 + 6 * _Z ** 2 + 4 * _Z - 1) + Q ** 2 * (12 * _Z ** 3 + 6 * _Z ** 2 + 6 * _Z)
            + Q * (12 * _Z ** 3 + 6 * _Z ** 2 + 4 * _Z) + 12 * _Z ** 3 + 12 * _Z ** 2 + 6 * _Z + 1)
_PHI12_OVER_R = (Q ** 4 - Q ** 2 + 1) // R
assert (Q ** 4 - Q ** 2 + 1) % R == 0 and HARD_EXP % _PHI12_OVER_R == 0
ARK_PAIRING_POWER = HARD_EXP // _PHI12_OVER_R        # final_exponentiation(f) = f^((q^12-1)/r) ^ ARK_PAIRING_POWER  (= 2z(6z^2 + 3z + 1))
assert ARK_PAIRING_POWER == 2 * _Z * (6 * _Z ** 2 + 3 * _Z + 1)


def final_exponentiation(f):
    easy = o._f12_pow(f, (Q ** 6 - 1) * (Q ** 2 + 1))
    return o._f12_pow(easy, HARD_EXP)


def pairing(P, Qaff):
    """E::pairing(P, Q).0"""
    return final_exponentiation(multi_miller_loop([(P, g2_prepare(Qaff))]))


# ---- verifier.rs ------------------------------------------------------------------------------------------------------
def prepare_verifying_key(vk):
    """forks/groth16/src/verifier.rs:13-20"""
    neg = lambda P: None if P is None else (P[0], F2.neg(P[1]))
    return {"vk": vk, "alpha_g1_beta_g2": pairing(vk["alpha_g1"], vk["beta_g2"]),
            "gamma_g2_neg_pc": g2_prepare(neg(vk["gamma_g2"])), "delta_g2_neg_pc": g2_prepare(neg(vk["delta_g2"]))}


def verify_proof_with_prepared_inputs(pvk, proof, prepared_inputs_affine) -> bool:
    """forks/groth16/src/verifier.rs:44-65"""
    a, b, c = proof
    qap = multi_miller_loop([(a, g2_prepare(b)), (prepared_inputs_affine, pvk["gamma_g2_neg_pc"]), (c, pvk["delta_g2_neg_pc"])])
    return final_exponentiation(qap) == pvk["alpha_g1_beta_g2"]


def verify_with_processed_vk(pvk, public_inputs, proof) -> bool:
    """`Groth16::verify_with_processed_vk` as create_client_state calls it (creds/src/lib.rs:288-289):
    prepare_inputs (verifier.rs:25-39) then the prepared check"""
    ic = o.G1.to_affine(o.prepare_inputs(pvk["vk"], public_inputs))
    return verify_proof_with_prepared_inputs(pvk, proof, ic)


# ---- ark-serialize (uncompressed) of the structures around the proof --------------------------------------------------
def _fq2_bytes(a) -> bytes:
    return o.fe_bytes(a[0]) + o.fe_bytes(a[1])


def g2_prepared_bytes(pq) -> bytes:
    items = [_fq2_bytes(c0) + _fq2_bytes(c1) + _fq2_bytes(c2) for c0, c1, c2 in pq["ell_coeffs"]]
    return struct.pack("<Q", len(items)) + b"".join(items) + (b"\x01" if pq["infinity"] else b"\x00")


def pvk_bytes(pvk) -> bytes:
    """PreparedVerifyingKey, data_structures.rs:62-71"""
    return (o.vk_uncompressed(pvk["vk"]) + fq12_bytes(pvk["alpha_g1_beta_g2"]) + g2_prepared_bytes(pvk["gamma_g2_neg_pc"])
            + g2_prepared_bytes(pvk["delta_g2_neg_pc"]))


class _Rd:
    def __init__(self, b):
        self.b, self.o = bytes(b), 0

    def take(self, n):
        if self.o + n > len(self.b):
            raise ValueError("unexpected end")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def fq(self):
        return int.from_bytes(self.take(32), "little")

    def fq2(self):
        return (self.fq(), self.fq())

    def g1(self):
        b = bytearray(self.take(64))
        inf = b[63] & 0x40
        b[63] &= 0x3F
        return None if inf else o.g1_unpack(bytes(b))

    def g2(self):
        b = bytearray(self.take(128))
        inf = b[127] & 0x40
        b[127] &= 0x3F
        return None if inf else o.g2_unpack(bytes(b))

    def string(self):
        return self.take(self.u64()).decode("utf-8")

    def vk(self):
        vk = dict(alpha_g1=self.g1(), beta_g2=self.g2(), gamma_g2=self.g2(), delta_g1=self.g1(), delta_g2=self.g2())
        vk["gamma_abc_g1"] = [self.g1() for _ in range(self.u64())]
        return vk

    def g2_prepared(self):
        n = self.u64()
        coeffs = [(self.fq2(), self.fq2(), self.fq2()) for _ in range(n)]
        return {"ell_coeffs": coeffs, "infinity": self.take(1) == b"\x01"}

    def pvk(self):
        vk = self.vk()
        return {"vk": vk, "alpha_g1_beta_g2": fq12_from_bytes(self.take(384)), "gamma_g2_neg_pc": self.g2_prepared(),
                "delta_g2_neg_pc": self.g2_prepared()}


def pvk_from_bytes(b):
    r = _Rd(b)
    pvk = r.pvk()
    if r.o != len(r.b):
        raise ValueError("trailing bytes")
    return pvk


def _string(s: str) -> bytes:
    e = s.encode("utf-8")
    return struct.pack("<Q", len(e)) + e


def prover_params_bytes(pk, pvk, config_str: str) -> bytes:
    """ProverParams (creds/src/lib.rs:58-63) as run_zksetup writes it (creds/src/lib.rs:245-248); pk is the oracle's
    dict form"""
    return o.pk_uncompressed(pk) + pvk_bytes(pvk) + _string(config_str)


def client_state_bytes(inputs, aux, proof, vk, pvk, config_str, credtype="jwt") -> bytes:
    """a ClientState fresh out of ClientState::new (creds/src/groth16rand.rs:60-80): no randomness, no openings"""
    out = struct.pack("<Q", len(inputs)) + b"".join(o.fe_bytes(x) for x in inputs)
    out += b"\x00" if aux is None else b"\x01" + _string(aux)
    out += o.proof_uncompressed(proof) + o.vk_uncompressed(vk) + pvk_bytes(pvk)
    out += b"\x00"                                  # input_com_randomness: None
    out += struct.pack("<Q", 0)                     # committed_input_openings: empty
    return out + _string(credtype) + _string(config_str)


def client_state_from_bytes(b):
    r = _Rd(b)
    cs = {"inputs": [r.fq() for _ in range(r.u64())]}
    cs["aux"] = r.string() if r.take(1) == b"\x01" else None
    cs["proof"] = (r.g1(), r.g2(), r.g1())
    cs["vk"] = r.vk()
    cs["pvk"] = r.pvk()
    cs["input_com_randomness"] = r.fq() if r.take(1) == b"\x01" else None
    n = r.u64()
    cs["committed_input_openings"] = []
    for _ in range(n):
        bases = [r.g1() for _ in range(r.u64())]
        cs["committed_input_openings"].append({"bases": bases, "m": r.fq(), "r": r.fq(), "c": r.g1()})
    cs["credtype"] = r.string()
    cs["config_str"] = r.string()
    if r.o != len(r.b):
        raise ValueError("trailing bytes")
    return cs


# ---- iden3 .r1cs writer (the container r1cs_reader.rs:54-256 parses; section 3 directly after section 2, :125) --------
def r1cs_file_bytes(matrices, n_wires: int, n_pub_out: int, n_pub_in: int, n_prv_in: int, n_labels=None) -> bytes:
    """matrices = (A, B, C), each a list of rows of (coeff, wire).  Sections in the order header(1), constraints(2),
    wire map(3), as circom emits them."""
    A, B, Cm = matrices
    m = len(A)
    if n_labels is None:
        n_labels = n_wires
    header = (struct.pack("<I", 32) + o.FR_MODULUS_LE + struct.pack("<IIII", n_wires, n_pub_out, n_pub_in, n_prv_in)
              + struct.pack("<Q", n_labels) + struct.pack("<I", m))
    cons = bytearray()
    for i in range(m):
        for mat in (A, B, Cm):
            row = mat[i]
            cons += struct.pack("<I", len(row))
            for coeff, wire in row:
                cons += struct.pack("<I", wire) + o.fe_bytes(coeff % R)
    wmap = b"".join(struct.pack("<Q", i) for i in range(n_wires))
    out = b"r1cs" + struct.pack("<II", 1, 3)
    for typ, payload in ((1, header), (2, bytes(cons)), (3, wmap)):
        out += struct.pack("<IQ", typ, len(payload)) + payload
    return out


def io_locations_sym(names_to_wire: dict) -> str:
    """io_locations.sym rows `name,location` (creds/src/structs.rs:41-68)"""
    return "".join("%s,%d\n" % (k, v) for k, v in names_to_wire.items())
