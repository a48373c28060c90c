"""Pure-Python big-int ORACLE for the Groth16 prove path of Crescent (BN254).

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything under ``oracle/``.  The product path (``crescent-credentials_amd/``) never
does; it fails loudly when its HIP library is missing.

PARITY STATUS: **parity unpinned** for proof bytes.  The reference
(`/root/reference`, Rust on un-vendored crates.io arkworks ^0.4, no Cargo.lock,
no rustc in this image) can be neither compiled nor imported, and none of its
tests pin proof bytes (they assert `verify == true` only:
forks/groth16/src/test.rs:70-71, creds/src/lib.rs:288-290,
creds/src/rangeproof.rs:511).  What the reference DOES pin, and what this oracle
is checked against in tests/test_oracle_kats.py:
  * Fq Montgomery one / G1 generator / G2 generator bytes and decimals
    (forks/circom-compat/src/zkey.rs:397-460),
  * the Fr modulus bytes (forks/circom-compat/src/circom/r1cs_reader.rs:183),
  * the .r1cs parser KAT (forks/circom-compat/src/circom/r1cs_reader.rs:264-345),
  * acceptance by the Groth16 verification equation
    (forks/groth16/src/verifier.rs:13-77).
Everything the path computes (MSM, polynomial division) is mathematically
single-valued, so bit-exactness against arkworks reduces to identical constants,
identical inputs and identical serialisation; the serialisation flag rules are
from memory of ark-serialize 0.4 ("[ark-mem]") and isolated in `_sw_flags`.

Each function cites the reference file:line it restates.
"""
from __future__ import annotations

import hashlib
import struct
from typing import List, Optional, Sequence, Tuple

# ----------------------------------------------------------------------------
# Constants (pinned in-tree at
# forks/halo2curves/src/bn256/fq.rs:12, fr.rs:10, r1cs_reader.rs:183)
# ----------------------------------------------------------------------------
Q = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # base field
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # scalar field
FR_GENERATOR = 5          # ark-bn254 Fr::GENERATOR [ark-mem]; coset offset r1cs_to_qap.rs:182
FR_TWO_ADICITY = 28
FR_ROOT_2_28 = pow(FR_GENERATOR, (R - 1) >> FR_TWO_ADICITY, R)
MONT_R = 1 << 256

G1_GEN = (1, 2)
# zkey.rs:442-460 (decimal coordinates of the G2 generator)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)
B1 = 3
# b' = 3/(9+u)
def _fq2_inv_raw(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % Q, Q - 2, Q)
    return (a[0] * n % Q, (-a[1]) * n % Q)
_xi_inv = _fq2_inv_raw((9, 1))
B2 = (3 * _xi_inv[0] % Q, 3 * _xi_inv[1] % Q)


# ----------------------------------------------------------------------------
# Field "ops tables": the curve code below is generic over these
# ----------------------------------------------------------------------------
class FqOps:
    zero = 0
    one = 1
    @staticmethod
    def add(a, b): return (a + b) % Q
    @staticmethod
    def sub(a, b): return (a - b) % Q
    @staticmethod
    def neg(a): return (-a) % Q
    @staticmethod
    def mul(a, b): return a * b % Q
    @staticmethod
    def sqr(a): return a * a % Q
    @staticmethod
    def inv(a): return pow(a, Q - 2, Q)
    @staticmethod
    def is_zero(a): return a == 0
    @staticmethod
    def muli(a, k): return a * k % Q


class Fq2Ops:
    """Fq2 = Fq[u]/(u^2+1)  (forks/halo2curves/src/bn256/fq.rs:29-31)."""
    zero = (0, 0)
    one = (1, 0)
    @staticmethod
    def add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
    @staticmethod
    def sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
    @staticmethod
    def neg(a): return ((-a[0]) % Q, (-a[1]) % Q)
    @staticmethod
    def mul(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
    @staticmethod
    def sqr(a):
        return ((a[0] + a[1]) * (a[0] - a[1]) % Q, 2 * a[0] * a[1] % Q)
    @staticmethod
    def inv(a): return _fq2_inv_raw(a)
    @staticmethod
    def is_zero(a): return a[0] == 0 and a[1] == 0
    @staticmethod
    def muli(a, k): return (a[0] * k % Q, a[1] * k % Q)


# ----------------------------------------------------------------------------
# Short-Weierstrass a=0 curve arithmetic, Jacobian (X,Y,Z), None-free:
# infinity is Z == 0.  Affine points are (x, y) tuples or None for infinity.
# ----------------------------------------------------------------------------
class Curve:
    def __init__(self, F, b, gen):
        self.F = F
        self.b = b
        self.gen = gen

    def is_on_curve(self, P) -> bool:
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.sqr(y) == F.add(F.mul(F.sqr(x), x), self.b)

    def to_jac(self, P):
        F = self.F
        if P is None:
            return (F.one, F.one, F.zero)
        return (P[0], P[1], F.one)

    def jac_infinity(self):
        F = self.F
        return (F.one, F.one, F.zero)

    def to_affine(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def dbl(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return J
        A = F.sqr(X)
        Bv = F.sqr(Y)
        C = F.sqr(Bv)
        D = F.muli(F.sub(F.sub(F.sqr(F.add(X, Bv)), A), C), 2)
        E = F.muli(A, 3)
        Fv = F.sqr(E)
        X3 = F.sub(Fv, F.muli(D, 2))
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), F.muli(C, 8))
        Z3 = F.muli(F.mul(Y, Z), 2)
        return (X3, Y3, Z3)

    def add(self, J1, J2):
        F = self.F
        X1, Y1, Z1 = J1
        X2, Y2, Z2 = J2
        if F.is_zero(Z1):
            return J2
        if F.is_zero(Z2):
            return J1
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(F.mul(Y1, Z2), Z2Z2)
        S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
        if U1 == U2:
            if S1 == S2:
                return self.dbl(J1)
            return self.jac_infinity()
        H = F.sub(U2, U1)
        I = F.sqr(F.muli(H, 2))
        Jv = F.mul(H, I)
        r = F.muli(F.sub(S2, S1), 2)
        V = F.mul(U1, I)
        X3 = F.sub(F.sub(F.sqr(r), Jv), F.muli(V, 2))
        Y3 = F.sub(F.mul(r, F.sub(V, X3)), F.muli(F.mul(S1, Jv), 2))
        Z3 = F.mul(F.sub(F.sub(F.sqr(F.add(Z1, Z2)), Z1Z1), Z2Z2), H)
        return (X3, Y3, Z3)

    def add_affine(self, J, P):
        if P is None:
            return J
        return self.add(J, self.to_jac(P))

    def neg(self, J):
        return (J[0], self.F.neg(J[1]), J[2])

    def neg_affine(self, P):
        if P is None:
            return None
        return (P[0], self.F.neg(P[1]))

    def mul(self, J, k: int):
        """double-and-add, MSB first; k is a non-negative integer (NOT reduced)."""
        acc = self.jac_infinity()
        if k == 0:
            return acc
        for bit in bin(k)[2:]:
            acc = self.dbl(acc)
            if bit == "1":
                acc = self.add(acc, J)
        return acc

    def mul_affine(self, P, k: int):
        return self.mul(self.to_jac(P), k)

    def batch_to_affine(self, Js):
        """Montgomery-trick normalisation (generator.rs:210-214 `normalize_batch`)."""
        F = self.F
        prods = []
        acc = F.one
        for J in Js:
            if not F.is_zero(J[2]):
                acc = F.mul(acc, J[2])
            prods.append(acc)
        inv = F.inv(acc) if not F.is_zero(acc) else F.zero
        out = [None] * len(Js)
        for i in range(len(Js) - 1, -1, -1):
            J = Js[i]
            if F.is_zero(J[2]):
                continue
            prev = prods[i - 1] if i > 0 else F.one
            zi = F.mul(inv, prev)
            inv = F.mul(inv, J[2])
            zi2 = F.sqr(zi)
            out[i] = (F.mul(J[0], zi2), F.mul(J[1], F.mul(zi2, zi)))
        return out

    def fixed_base_table(self, P, window: int, bits: int = 254):
        """table[j][d] = d * 2^(window*j) * P as affine points (FixedBase precedent,
        generator.rs:133-194 uses ark-ec FixedBase; this is just a fast equivalent)."""
        nwin = (bits + window - 1) // window
        rows = []
        base = self.to_jac(P)
        for _ in range(nwin):
            row = [self.jac_infinity()]
            acc = self.jac_infinity()
            for _d in range(1, 1 << window):
                acc = self.add(acc, base)
                row.append(acc)
            rows.append(row)
            for _ in range(window):
                base = self.dbl(base)
        flat = [J for row in rows for J in row]
        aff = self.batch_to_affine(flat)
        n = 1 << window
        return [aff[j * n:(j + 1) * n] for j in range(nwin)], window

    def fixed_base_mul(self, table, k: int):
        rows, window = table
        acc = self.jac_infinity()
        mask = (1 << window) - 1
        j = 0
        while k:
            d = k & mask
            if d:
                acc = self.add_affine(acc, rows[j][d])
            k >>= window
            j += 1
        return acc

    def msm_naive(self, bases, scalars):
        """Σ s_i P_i by independent double-and-add (the 'serial sum' side of
        forks/halo2curves/src/msm.rs:601-636's cross-check pattern)."""
        acc = self.jac_infinity()
        for P, s in zip(bases, scalars):  # zip truncates like msm_bigint [ark-mem]
            if P is None or s == 0:
                continue
            acc = self.add(acc, self.mul_affine(P, s))
        return acc

    def msm(self, bases, scalars, c: Optional[int] = None):
        """Pippenger, unsigned windows (ark-ec VariableBaseMSM::msm_bigint, call sites
        forks/groth16/src/prover.rs:66,74,266).  Result is the unique group element
        Σ s_i P_i, so the window choice is irrelevant to parity."""
        n = min(len(bases), len(scalars))
        if n == 0:
            return self.jac_infinity()
        if c is None:
            c = 3 if n < 32 else max(3, n.bit_length() * 69 // 100 + 2)
        nwin = (254 + c - 1) // c
        mask = (1 << c) - 1
        total = self.jac_infinity()
        for w in range(nwin - 1, -1, -1):
            for _ in range(c):
                total = self.dbl(total)
            buckets = [None] * (1 << c)
            sh = w * c
            for i in range(n):
                P = bases[i]
                if P is None:
                    continue
                d = (scalars[i] >> sh) & mask
                if d:
                    b = buckets[d]
                    buckets[d] = self.add_affine(b, P) if b is not None else self.to_jac(P)
            run = self.jac_infinity()
            acc = self.jac_infinity()
            for d in range(mask, 0, -1):
                if buckets[d] is not None:
                    run = self.add(run, buckets[d])
                acc = self.add(acc, run)
            total = self.add(total, acc)
        return total


G1 = Curve(FqOps, B1, G1_GEN)
G2 = Curve(Fq2Ops, B2, G2_GEN)


# ----------------------------------------------------------------------------
# Pairing: plain ate pairing a(Q,P) = f_{t-1,Q}(P)^((q^12-1)/r) computed with
# Fq12 = Fq[w]/(w^12 - 18 w^6 + 82) polynomial arithmetic.  Any non-degenerate
# bilinear pairing on G1 x G2 accepts exactly the same proofs as the optimal-ate
# pairing the reference uses (verifier.rs:44-65), because both are fixed non-zero
# powers of each other on the order-r groups.
# ----------------------------------------------------------------------------
_BN_U = 4965661367192848881
_ATE_LOOP = 6 * _BN_U * _BN_U   # t - 1

def _f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    # reduce with w^12 = 18 w^6 - 82
    for k in range(22, 11, -1):
        v = t[k]
        if v:
            t[k - 6] += 18 * v
            t[k - 12] -= 82 * v
    return [x % Q for x in t[:12]]

def _f12_one():
    return [1] + [0] * 11

def _f12_pow(a, e):
    res = _f12_one()
    for bit in bin(e)[2:]:
        res = _f12_mul(res, res)
        if bit == "1":
            res = _f12_mul(res, a)
    return res

def _poly_deg(p):
    d = len(p) - 1
    while d >= 0 and p[d] == 0:
        d -= 1
    return d

def _f12_inv(a):
    """extended Euclid over Fq[w] against the modulus polynomial."""
    lm, hm = [1] + [0] * 12, [0] * 13
    low = list(a) + [0]
    high = [82, 0, 0, 0, 0, 0, (-18) % Q, 0, 0, 0, 0, 0, 1]
    while _poly_deg(low) > 0:
        dl, dh = _poly_deg(low), _poly_deg(high)
        # r = high / low (polynomial division rounding)
        r = [0] * 13
        temp = list(high)
        inv_lead = pow(low[dl], Q - 2, Q)
        for i in range(dh - dl, -1, -1):
            coef = temp[dl + i] * inv_lead % Q
            r[i] = coef
            if coef:
                for c in range(dl + 1):
                    temp[c + i] = (temp[c + i] - coef * low[c]) % Q
        nm = list(hm)
        new = list(high)
        for i in range(13):
            if lm[i] or low[i]:
                for j in range(13 - i):
                    if r[j]:
                        nm[i + j] = (nm[i + j] - lm[i] * r[j]) % Q
                        new[i + j] = (new[i + j] - low[i] * r[j]) % Q
        lm, low, hm, high = nm, new, lm, low
    inv0 = pow(low[0], Q - 2, Q)
    return [x * inv0 % Q for x in lm[:12]]

def _f12_from_fq(a):
    return [a % Q] + [0] * 11

def _f12_from_fq2(a):
    # a0 + a1*u, u = w^6 - 9
    out = [0] * 12
    out[0] = (a[0] - 9 * a[1]) % Q
    out[6] = a[1] % Q
    return out

def _f12_add(a, b): return [(x + y) % Q for x, y in zip(a, b)]
def _f12_sub(a, b): return [(x - y) % Q for x, y in zip(a, b)]

def _untwist(Qaff):
    """E'(Fq2) -> E(Fq12): (x', y') -> (x' w^2, y' w^3)   (D-type twist, w^6 = 9+u)."""
    x = _f12_from_fq2(Qaff[0])
    y = _f12_from_fq2(Qaff[1])
    w2 = [0] * 12; w2[2] = 1
    w3 = [0] * 12; w3[3] = 1
    return (_f12_mul(x, w2), _f12_mul(y, w3))

def _line(P1, P2, T):
    """value at T of the line through P1,P2 (all on E(Fq12), affine)."""
    x1, y1 = P1; x2, y2 = P2; xt, yt = T
    if x1 != x2:
        m = _f12_mul(_f12_sub(y2, y1), _f12_inv(_f12_sub(x2, x1)))
        return _f12_sub(_f12_mul(m, _f12_sub(xt, x1)), _f12_sub(yt, y1))
    if y1 == y2:
        three_x2 = _f12_mul(_f12_from_fq(3), _f12_mul(x1, x1))
        m = _f12_mul(three_x2, _f12_inv(_f12_add(y1, y1)))
        return _f12_sub(_f12_mul(m, _f12_sub(xt, x1)), _f12_sub(yt, y1))
    return _f12_sub(xt, x1)

def _e12_double(P):
    x, y = P
    m = _f12_mul(_f12_mul(_f12_from_fq(3), _f12_mul(x, x)), _f12_inv(_f12_add(y, y)))
    nx = _f12_sub(_f12_mul(m, m), _f12_add(x, x))
    ny = _f12_sub(_f12_mul(m, _f12_sub(x, nx)), y)
    return (nx, ny)

def _e12_add(P1, P2):
    x1, y1 = P1; x2, y2 = P2
    if x1 == x2:
        if y1 == y2:
            return _e12_double(P1)
        return None
    m = _f12_mul(_f12_sub(y2, y1), _f12_inv(_f12_sub(x2, x1)))
    nx = _f12_sub(_f12_sub(_f12_mul(m, m), x1), x2)
    ny = _f12_sub(_f12_mul(m, _f12_sub(x1, nx)), y1)
    return (nx, ny)

def miller_loop(Qaff, Paff):
    """f_{t-1,Q}(P) in Fq12; 1 if either input is the identity."""
    if Qaff is None or Paff is None:
        return _f12_one()
    Q12 = _untwist(Qaff)
    P12 = (_f12_from_fq(Paff[0]), _f12_from_fq(Paff[1]))
    T = Q12
    f = _f12_one()
    for bit in bin(_ATE_LOOP)[3:]:
        f = _f12_mul(_f12_mul(f, f), _line(T, T, P12))
        T = _e12_double(T)
        if bit == "1":
            f = _f12_mul(f, _line(T, Q12, P12))
            T = _e12_add(T, Q12)
    return f

_FINAL_EXP = (Q ** 12 - 1) // R

def final_exponentiation(f):
    return _f12_pow(f, _FINAL_EXP)

def pairing(Paff, Qaff):
    return final_exponentiation(miller_loop(Qaff, Paff))

def pairing_product_is_one(pairs) -> bool:
    """Π e(P_i, Q_i) == 1 with one shared final exponentiation."""
    f = _f12_one()
    for Paff, Qaff in pairs:
        f = _f12_mul(f, miller_loop(Qaff, Paff))
    return final_exponentiation(f) == _f12_one()


# ----------------------------------------------------------------------------
# Radix-2 domain over Fr (ark-poly Radix2EvaluationDomain; call sites
# forks/groth16/src/r1cs_to_qap.rs:156,179-187,198-202,210)
# ----------------------------------------------------------------------------
def domain_size_for(n: int) -> int:
    """D::new(n): smallest power of two >= n (r1cs_to_qap.rs:156-158)."""
    d = 1
    while d < n:
        d <<= 1
    if d.bit_length() - 1 > FR_TWO_ADICITY:
        raise ValueError("PolynomialDegreeTooLarge")
    return d

def root_of_unity(D: int) -> int:
    logd = D.bit_length() - 1
    assert 1 << logd == D
    return pow(FR_ROOT_2_28, 1 << (FR_TWO_ADICITY - logd), R)

def _bitrev_permute(a):
    n = len(a)
    logn = n.bit_length() - 1
    for i in range(n):
        j = int(bin(i)[2:].zfill(logn)[::-1], 2) if logn else 0
        if i < j:
            a[i], a[j] = a[j], a[i]

def ntt(a: List[int], omega: int) -> List[int]:
    """in-order radix-2 NTT: out[k] = Σ_j a[j] ω^{jk}."""
    a = list(a)
    n = len(a)
    _bitrev_permute(a)
    length = 2
    while length <= n:
        wl = pow(omega, n // length, R)
        half = length >> 1
        tw = [1] * half
        for i in range(1, half):
            tw[i] = tw[i - 1] * wl % R
        for start in range(0, n, length):
            for k in range(half):
                u = a[start + k]
                v = a[start + k + half] * tw[k] % R
                a[start + k] = (u + v) % R
                a[start + k + half] = (u - v) % R
        length <<= 1
    return a

def fft(a, D=None):
    D = D or len(a)
    return ntt(list(a) + [0] * (D - len(a)), root_of_unity(D))

def ifft(a, D=None):
    D = D or len(a)
    inv_d = pow(D, R - 2, R)
    w_inv = pow(root_of_unity(D), R - 2, R)
    return [x * inv_d % R for x in ntt(list(a) + [0] * (D - len(a)), w_inv)]

def coset_fft(a, g=FR_GENERATOR):
    """coset_domain.fft_in_place: scale coeff i by g^i, then fft (r1cs_to_qap.rs:182-185)."""
    out, p = [], 1
    for x in a:
        out.append(x * p % R)
        p = p * g % R
    return fft(out)

def coset_ifft(a, g=FR_GENERATOR):
    """coset_domain.ifft_in_place: ifft, then scale coeff i by g^-i (r1cs_to_qap.rs:210)."""
    c = ifft(a)
    gi = pow(g, R - 2, R)
    out, p = [], 1
    for x in c:
        out.append(x * p % R)
        p = p * gi % R
    return out

def evaluate_vanishing_polynomial(D: int, t: int) -> int:
    return (pow(t, D, R) - 1) % R

def evaluate_all_lagrange_coefficients(D: int, t: int) -> List[int]:
    """u_i = L_i(t) over the size-D subgroup (r1cs_to_qap.rs:119 via ark-poly [ark-mem])."""
    w = root_of_unity(D)
    zt = evaluate_vanishing_polynomial(D, t)
    if zt == 0:  # t in the domain
        out = [0] * D
        p = 1
        for i in range(D):
            if p == t % R:
                out[i] = 1
            p = p * w % R
        return out
    zd = zt * pow(D, R - 2, R) % R
    out = []
    p = 1
    for _ in range(D):
        out.append(zd * p % R * pow((t - p) % R, R - 2, R) % R)
        p = p * w % R
    return out


# ----------------------------------------------------------------------------
# R1CS -> QAP witness map (forks/groth16/src/r1cs_to_qap.rs:16-45,150-213)
# matrices: (A, B, C), each a list of rows, each row a list of (coeff, column)
# ----------------------------------------------------------------------------
def evaluate_constraint(terms, assignment) -> int:
    """r1cs_to_qap.rs:16-45 (the coeff.is_one() shortcut is value-neutral)."""
    s = 0
    for coeff, idx in terms:
        s += assignment[idx] * coeff
    return s % R

def witness_map_from_matrices(matrices, num_inputs, num_constraints, full_assignment):
    """LibsnarkReduction::witness_map_from_matrices (r1cs_to_qap.rs:150-213) -> h (len D)."""
    A, B, C = matrices
    D = domain_size_for(num_constraints + num_inputs)
    a = [0] * D
    b = [0] * D
    for i in range(num_constraints):                       # :164-171
        a[i] = evaluate_constraint(A[i], full_assignment)
        b[i] = evaluate_constraint(B[i], full_assignment)
    for i in range(num_inputs):                            # :173-177
        a[num_constraints + i] = full_assignment[i] % R
    a = coset_fft(ifft(a))                                 # :179-185
    b = coset_fft(ifft(b))
    ab = [x * y % R for x, y in zip(a, b)]                 # :187
    c = [0] * D
    for i in range(num_constraints):                       # :191-196
        c[i] = evaluate_constraint(C[i], full_assignment)
    c = coset_fft(ifft(c))                                 # :198-199
    vinv = pow(evaluate_vanishing_polynomial(D, FR_GENERATOR), R - 2, R)   # :201-204
    ab = [(x - y) * vinv % R for x, y in zip(ab, c)]       # :205-208
    return coset_ifft(ab)                                  # :210


# ----------------------------------------------------------------------------
# Groth16 objects (forks/groth16/src/data_structures.rs)
# pk dict keys mirror ProvingKey/VerifyingKey field names.
# ----------------------------------------------------------------------------
def generate_parameters(matrices, num_inputs, num_constraints, num_variables,
                        tau, alpha, beta, delta, gamma=1):
    """generate_parameters_with_qap (forks/groth16/src/generator.rs:50-228) with the
    toxic waste given explicitly (the fork fixes gamma = 1, :28, and the standard
    generators, :34-35).  `num_variables` = M (instance + witness, incl. the constant)."""
    A, B, C = matrices
    l = num_inputs
    m = num_constraints
    M = num_variables
    D = domain_size_for(m + l)                                   # :92-93
    zt = evaluate_vanishing_polynomial(D, tau)
    u = evaluate_all_lagrange_coefficients(D, tau)               # r1cs_to_qap.rs:119
    a = [0] * M; b = [0] * M; c = [0] * M                        # r1cs_to_qap.rs:124-126
    for i in range(l):                                           # :128-133
        a[i] = u[m + i]
    for i in range(m):                                           # :135-145
        ui = u[i]
        for coeff, idx in A[i]:
            a[idx] = (a[idx] + ui * coeff) % R
        for coeff, idx in B[i]:
            b[idx] = (b[idx] + ui * coeff) % R
        for coeff, idx in C[i]:
            c[idx] = (c[idx] + ui * coeff) % R
    ginv = pow(gamma, R - 2, R)
    dinv = pow(delta, R - 2, R)
    gamma_abc = [(beta * a[i] + alpha * b[i] + c[i]) * ginv % R for i in range(l)]       # :118-122
    lq = [(beta * a[i] + alpha * b[i] + c[i]) * dinv % R for i in range(l, M)]           # :124-128
    hq = []                                                                              # r1cs_to_qap.rs:215-225
    p = zt * dinv % R
    for _ in range(D - 1):                                                               # m_raw - 1, generator.rs:178
        hq.append(p)
        p = p * tau % R

    t1 = G1.fixed_base_table(G1_GEN, 8)
    t2 = G2.fixed_base_table(G2_GEN, 6)
    g1m = lambda ks: G1.batch_to_affine([G1.fixed_base_mul(t1, k % R) for k in ks])
    g2m = lambda ks: G2.batch_to_affine([G2.fixed_base_mul(t2, k % R) for k in ks])
    vk = dict(
        alpha_g1=g1m([alpha])[0], beta_g2=g2m([beta])[0], gamma_g2=g2m([gamma])[0],
        delta_g1=g1m([delta])[0], delta_g2=g2m([delta])[0], gamma_abc_g1=g1m(gamma_abc),
    )
    pk = dict(
        vk=vk, beta_g1=g1m([beta])[0], delta_g1=vk["delta_g1"],
        a_query=g1m(a), b_g1_query=g1m(b), b_g2_query=g2m(b), h_query=g1m(hq), l_query=g1m(lq),
    )
    qap = dict(a=a, b=b, c=c, zt=zt, D=D, l=lq, gamma_abc=gamma_abc)
    return pk, qap


def calculate_coeff(curve, initial, query, vk_param, assignment):
    """prover.rs:256-274: initial + query[0] + MSM(query[1..], assignment) + vk_param."""
    acc = curve.msm(query[1:], assignment)
    res = curve.add_affine(initial, query[0])
    res = curve.add(res, acc)
    return curve.add_affine(res, vk_param)


def create_proof_with_assignment(pk, r, s, h, input_assignment, aux_assignment):
    """prover.rs:54-136.  Returns Proof as (a, b, c) affine."""
    h_acc = G1.msm(pk["h_query"], h)                                      # :63-66
    l_aux_acc = G1.msm(pk["l_query"], aux_assignment)                     # :70-74
    delta_g1 = G1.to_jac(pk["delta_g1"])
    r_s_delta_g1 = G1.mul(G1.mul(delta_g1, r), s)                         # :76-80
    assignment = list(input_assignment) + list(aux_assignment)            # :84-89
    r_g1 = G1.mul(delta_g1, r)                                            # :94
    g_a = calculate_coeff(G1, r_g1, pk["a_query"], pk["vk"]["alpha_g1"], assignment)   # :96
    s_g_a = G1.mul(g_a, s)                                                # :98
    if r != 0:                                                            # :102-112
        s_g1 = G1.mul(delta_g1, s)
        g1_b = calculate_coeff(G1, s_g1, pk["b_g1_query"], pk["beta_g1"], assignment)
    else:
        g1_b = G1.jac_infinity()
    s_g2 = G2.mul(G2.to_jac(pk["vk"]["delta_g2"]), s)                     # :116
    g2_b = calculate_coeff(G2, s_g2, pk["b_g2_query"], pk["vk"]["beta_g2"], assignment)  # :117
    r_g1_b = G1.mul(g1_b, r)                                              # :118
    g_c = s_g_a                                                           # :123-128
    g_c = G1.add(g_c, r_g1_b)
    g_c = G1.add(g_c, G1.neg(r_s_delta_g1))
    g_c = G1.add(g_c, l_aux_acc)
    g_c = G1.add(g_c, h_acc)
    return (G1.to_affine(g_a), G2.to_affine(g2_b), G1.to_affine(g_c))      # :131-135


def create_proof_with_reduction_and_matrices(pk, r, s, matrices, num_inputs,
                                             num_constraints, full_assignment):
    """prover.rs:26-51."""
    h = witness_map_from_matrices(matrices, num_inputs, num_constraints, full_assignment)
    return create_proof_with_assignment(
        pk, r % R, s % R, h, full_assignment[1:num_inputs], full_assignment[num_inputs:])


def closed_form_proof(qap, trapdoor, r, s, h, full_assignment, num_inputs):
    """Trapdoor closed form: shares no MSM/NTT-with-bases code with
    the path.  A = [α + Σ w_i a_i(τ) + rδ]G, B = [β + Σ w_i b_i(τ) + sδ]H,
    C = [Σ_aux w_i l_i + h(τ) zt/δ + s·A + r·B − rsδ]G."""
    tau, alpha, beta, delta = trapdoor
    w = full_assignment
    a_s = (alpha + sum(x * y for x, y in zip(w, qap["a"])) + r * delta) % R
    b_s = (beta + sum(x * y for x, y in zip(w, qap["b"])) + s * delta) % R
    dinv = pow(delta, R - 2, R)
    h_tau = 0
    for coef in reversed(h[:qap["D"] - 1]):
        h_tau = (h_tau * tau + coef) % R
    c_s = (sum(x * y for x, y in zip(w[num_inputs:], qap["l"]))
           + h_tau * qap["zt"] % R * dinv + s * a_s + r * b_s - r * s % R * delta) % R
    if r == 0:   # prover.rs:102-112 skips B-in-G1 when r == 0 (the r·B term vanishes anyway)
        pass
    return (G1.to_affine(G1.mul_affine(G1_GEN, a_s)),
            G2.to_affine(G2.mul_affine(G2_GEN, b_s)),
            G1.to_affine(G1.mul_affine(G1_GEN, c_s)))


def prepare_inputs(vk, public_inputs):
    """verifier.rs:25-39."""
    if len(public_inputs) + 1 != len(vk["gamma_abc_g1"]):
        raise ValueError("MalformedVerifyingKey")
    g_ic = G1.to_jac(vk["gamma_abc_g1"][0])
    for x, Bp in zip(public_inputs, vk["gamma_abc_g1"][1:]):
        g_ic = G1.add(g_ic, G1.mul_affine(Bp, x % R))
    return g_ic


def verify_proof(vk, proof, public_inputs) -> bool:
    """verifier.rs:44-77: e(A,B) == e(α,β)·e(IC,γ)·e(C,δ), as one pairing product."""
    a, b, c = proof
    ic = G1.to_affine(prepare_inputs(vk, public_inputs))
    return pairing_product_is_one([
        (a, b),
        (G1.neg_affine(vk["alpha_g1"]), vk["beta_g2"]),
        (G1.neg_affine(ic), vk["gamma_g2"]),
        (G1.neg_affine(c), vk["delta_g2"]),
    ])


# ----------------------------------------------------------------------------
# ark-serialize (uncompressed / compressed) encoders  [ark-mem] for flag rules:
# SURVEY.md Appendix B; struct field order data_structures.rs:7-14,31-44,101-118
# ----------------------------------------------------------------------------
def fe_bytes(x: int) -> bytes:
    return int(x).to_bytes(32, "little")

def _fq2_gt(a, b) -> bool:
    """QuadExtField ordering: compare c1 first, then c0 [ark-mem]."""
    return (a[1], a[0]) > (b[1], b[0])

def _sw_flags(y, neg_y, is_g2: bool) -> int:
    """bit7: y is the larger of {y,-y} ('negative'); bit6 reserved for infinity."""
    if is_g2:
        return 0x80 if _fq2_gt(y, neg_y) else 0
    return 0x80 if y > neg_y else 0

def g1_uncompressed(P) -> bytes:
    if P is None:
        out = bytearray(64); out[63] |= 0x40; return bytes(out)
    out = bytearray(fe_bytes(P[0]) + fe_bytes(P[1]))
    out[63] |= _sw_flags(P[1], (-P[1]) % Q, False)
    return bytes(out)

def g2_uncompressed(P) -> bytes:
    if P is None:
        out = bytearray(128); out[127] |= 0x40; return bytes(out)
    (x0, x1), (y0, y1) = P
    out = bytearray(fe_bytes(x0) + fe_bytes(x1) + fe_bytes(y0) + fe_bytes(y1))
    out[127] |= _sw_flags(P[1], Fq2Ops.neg(P[1]), True)
    return bytes(out)

def g1_compressed(P) -> bytes:
    if P is None:
        out = bytearray(32); out[31] |= 0x40; return bytes(out)
    out = bytearray(fe_bytes(P[0]))
    out[31] |= _sw_flags(P[1], (-P[1]) % Q, False)
    return bytes(out)

def g2_compressed(P) -> bytes:
    if P is None:
        out = bytearray(64); out[63] |= 0x40; return bytes(out)
    out = bytearray(fe_bytes(P[0][0]) + fe_bytes(P[0][1]))
    out[63] |= _sw_flags(P[1], Fq2Ops.neg(P[1]), True)
    return bytes(out)

def proof_uncompressed(proof) -> bytes:
    """Proof = a ‖ b ‖ c (data_structures.rs:7-14) -> 256 bytes."""
    a, b, c = proof
    return g1_uncompressed(a) + g2_uncompressed(b) + g1_uncompressed(c)

def proof_compressed(proof) -> bytes:
    a, b, c = proof
    return g1_compressed(a) + g2_compressed(b) + g1_compressed(c)

def _vec(items: Sequence[bytes]) -> bytes:
    return struct.pack("<Q", len(items)) + b"".join(items)

def vk_uncompressed(vk) -> bytes:
    """VerifyingKey field order data_structures.rs:31-44 (fork adds delta_g1 :38-39)."""
    return (g1_uncompressed(vk["alpha_g1"]) + g2_uncompressed(vk["beta_g2"])
            + g2_uncompressed(vk["gamma_g2"]) + g1_uncompressed(vk["delta_g1"])
            + g2_uncompressed(vk["delta_g2"]) + _vec([g1_uncompressed(p) for p in vk["gamma_abc_g1"]]))

def pk_uncompressed(pk) -> bytes:
    """ProvingKey field order data_structures.rs:101-118."""
    return (vk_uncompressed(pk["vk"]) + g1_uncompressed(pk["beta_g1"]) + g1_uncompressed(pk["delta_g1"])
            + _vec([g1_uncompressed(p) for p in pk["a_query"]])
            + _vec([g1_uncompressed(p) for p in pk["b_g1_query"]])
            + _vec([g2_uncompressed(p) for p in pk["b_g2_query"]])
            + _vec([g1_uncompressed(p) for p in pk["h_query"]])
            + _vec([g1_uncompressed(p) for p in pk["l_query"]]))

# -- packed "C-ABI" arrays (include/crescent_gpu.h): G1 = x‖y canonical LE 64 B, G2 = x.c0‖x.c1‖y.c0‖y.c1
#    128 B, identity = all-zero (no flag bits).
def g1_packed(P) -> bytes:
    return bytes(64) if P is None else fe_bytes(P[0]) + fe_bytes(P[1])

def g2_packed(P) -> bytes:
    if P is None:
        return bytes(128)
    return fe_bytes(P[0][0]) + fe_bytes(P[0][1]) + fe_bytes(P[1][0]) + fe_bytes(P[1][1])

def g1_unpack(b: bytes):
    x = int.from_bytes(b[:32], "little"); y = int.from_bytes(b[32:64], "little")
    return None if x == 0 and y == 0 else (x, y)

def g2_unpack(b: bytes):
    v = [int.from_bytes(b[i * 32:(i + 1) * 32], "little") for i in range(4)]
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))

def jac_packed_g1(J) -> bytes:
    return fe_bytes(J[0]) + fe_bytes(J[1]) + fe_bytes(J[2])


# ----------------------------------------------------------------------------
# .r1cs (iden3 binary) parser: forks/circom-compat/src/circom/r1cs_reader.rs:54-256
# ----------------------------------------------------------------------------
FR_MODULUS_LE = bytes.fromhex("010000f093f5e1439170b97948e833285d588181b64550b829a031e1724e6430")

def parse_r1cs(data: bytes):
    if data[:4] != b"\x72\x31\x63\x73":                            # :57-62
        raise ValueError("Invalid magic number")
    version, nsec = struct.unpack_from("<II", data, 4)             # :64-72
    if version != 1:
        raise ValueError("Unsupported version")
    off = 12
    sec_off, sec_size = {}, {}
    for _ in range(nsec):                                          # :80-87
        ty, sz = struct.unpack_from("<IQ", data, off)
        off += 12
        sec_off[ty] = off
        sec_size[ty] = sz
        off += sz
    for ty in (1, 2, 3):
        if ty not in sec_off:
            raise ValueError("No section offset for type %d found" % ty)
    # header :162-202
    o = sec_off[1]
    (field_size,) = struct.unpack_from("<I", data, o)
    if field_size != 32:
        raise ValueError("This parser only supports 32-byte fields")
    if sec_size[1] != 32 + field_size:
        raise ValueError("Invalid header section size")
    prime = data[o + 4:o + 36]
    if prime != FR_MODULUS_LE:
        raise ValueError("This parser only supports bn256")
    n_wires, n_pub_out, n_pub_in, n_prv_in, n_labels, n_constraints = struct.unpack_from("<IIIIQI", data, o + 36)
    header = dict(field_size=field_size, prime=prime, n_wires=n_wires, n_pub_out=n_pub_out,
                  n_pub_in=n_pub_in, n_prv_in=n_prv_in, n_labels=n_labels, n_constraints=n_constraints)
    # constraints :205-236 (fork quirk :125: buffer length = off(sec3) - off(sec2))
    o = sec_off[2]
    end = sec_off[3] if sec_off[3] > sec_off[2] else sec_off[2] + sec_size[2]
    constraints = []
    for _ in range(n_constraints):
        row = []
        for _blk in range(3):
            (n,) = struct.unpack_from("<I", data, o); o += 4
            terms = []
            for _t in range(n):
                (wire,) = struct.unpack_from("<I", data, o); o += 4
                coeff = int.from_bytes(data[o:o + 32], "little"); o += 32
                if coeff >= R:
                    raise ValueError("non-canonical field element")
                terms.append((wire, coeff))
            row.append(terms)
        constraints.append(tuple(row))
    if o > end:
        raise ValueError("constraint section overrun")
    # wire map :238-256
    if sec_size[3] != n_wires * 8:
        raise ValueError("Invalid map section size")
    wire_mapping = list(struct.unpack_from("<%dQ" % n_wires, data, sec_off[3]))
    if wire_mapping[0] != 0:
        raise ValueError("Wire 0 should always be mapped to 0")
    return dict(version=version, header=header, constraints=constraints, wire_mapping=wire_mapping)

def r1cs_to_matrices(parsed):
    """R1CS -> (A,B,C) row lists of (coeff, column) with column = wire id
    (r1cs_reader.rs:26-38; circuit.rs:61-67; wire_mapping disabled builder.rs:63-64)."""
    h = parsed["header"]
    num_inputs = 1 + h["n_pub_in"] + h["n_pub_out"]
    A, B, C = [], [], []
    for (a, b, c) in parsed["constraints"]:
        A.append([(coef, w) for (w, coef) in a])
        B.append([(coef, w) for (w, coef) in b])
        C.append([(coef, w) for (w, coef) in c])
    return (A, B, C), num_inputs, h["n_constraints"], h["n_wires"]


# ----------------------------------------------------------------------------
# DummyCircuit of creds/src/rangeproof.rs:442-487 as matrices + assignment
# ----------------------------------------------------------------------------
def dummy_circuit(a_val: int, b_val: int, num_variables: int, num_constraints: int, num_inputs: int):
    """Instance vars: [1, c=a*b, a, a, ... (num_inputs-1 times)] -> ℓ = num_inputs + 1.
    Witness vars: [a, b, a, a, ...] (num_variables - num_inputs total).  All but the last
    constraint are a*b=c; the last is 0*0=0 (rangeproof.rs:476-482)."""
    l = num_inputs + 1
    n_wit = 2 + (num_variables - num_inputs - 2)
    inst = [1, a_val * b_val % R] + [a_val % R] * (num_inputs - 1)
    wit = [a_val % R, b_val % R] + [a_val % R] * (n_wit - 2)
    A = [[(1, l + 0)] for _ in range(num_constraints - 1)] + [[]]
    B = [[(1, l + 1)] for _ in range(num_constraints - 1)] + [[]]
    C = [[(1, 1)] for _ in range(num_constraints - 1)] + [[]]
    return (A, B, C), l, num_constraints, l + n_wit, inst + wit


def sha256_hex(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()
