#!/usr/bin/env python3
"""Bound checker for the lazy 29-bit-limb formulas of csrc/curve29.hpp.

Every field value is tracked as (V, L): V = upper bound of the value in units of the modulus N,
L = upper bound of limbs 0..7 in units of 2^29.  The script replays the exact operation sequence of
madd / add / dbl for Fq and Fq2 and asserts every precondition (64-bit column sums, non-negative
limb-wise subtraction, representability < 2^261) and that the accumulator invariant (X, Y, ZZ, ZZZ
bounds) is closed under all three operations.  Run: python tools/bounds29.py
"""
import math

RN = 169.28          # 2^261 / N  (both BN254 moduli)
CAP = 160.0          # values must stay below 2^261 / N with margin


class B:
    def __init__(self, V, L, norm=None):
        self.V, self.L = V, L
        assert V < CAP, "value bound %.1f exceeds representable range" % V
        assert L * 2 ** 29 < 2 ** 32 - 16, "limb overflow L=%.2f" % L

    def __repr__(self):
        return "B(V=%.2f, L=%.2f)" % (self.V, self.L)


def col_ok(pairs, nprod):
    """pairs: list of (La, Lb) limb bounds (units 2^29) of the nprod-per-column product groups."""
    tot = 0.0
    for la, lb in pairs:
        tot += 9 * (la * 2 ** 29) * (lb * 2 ** 29)
    tot += 9 * (2 ** 29) ** 2 + 2 ** 36      # m*N products + carry
    assert tot < 2 ** 64, "column accumulator overflow: 2^%.2f" % math.log2(tot)


class Fq:
    name = "Fq"

    @staticmethod
    def mul(a, b):
        col_ok([(a.L, b.L)], 9)
        return B(a.V * b.V / RN + 1.0, 1.0)

    @staticmethod
    def sqr(a):
        # doubled copy: limbs 2 a.L against a.L, ~half the products
        col_ok([(2 * a.L, a.L)], 5)
        return B(a.V * a.V / RN + 1.0, 1.0)

    sqr_loose = None   # = sqr, set below

    @staticmethod
    def mul_sub(a, b, c, d, K):
        """a*b - c*d as one dual-product reduction (a b + (K N - c) d)/R'  (curve29.hpp mul_sub, Fq)"""
        assert c.L <= 1.0 + 1e-9, "mul_sub: negated operand must be normalised"
        assert c.V <= K - 1 + 1e-9, "mul_sub: value %.2f needs K >= %d" % (c.V, math.ceil(c.V + 1))
        cneg = B(float(K), 2.0)
        col_ok([(a.L, b.L), (cneg.L, d.L)], 18)
        return B((a.V * b.V + cneg.V * d.V) / RN + 1.0, 1.0)

    @staticmethod
    def add(a, b):
        return B(a.V + b.V, a.L + b.L)

    @staticmethod
    def dbl(a):
        return B(2 * a.V, 2 * a.L)

    @staticmethod
    def sub(a, b, K, T):
        assert b.L <= T + 1e-9, "sub<%d,%d>: subtrahend limbs %.2f exceed boost" % (K, T, b.L)
        assert b.V <= K - 1 + 1e-9, "sub<%d,%d>: subtrahend value %.2f needs K >= %.0f" % (K, T, b.V, math.ceil(b.V + 1))
        return B(a.V + K, a.L + T + 1)

    @staticmethod
    def norm(a):
        return B(a.V, 1.0)


Fq.sqr_loose = Fq.sqr


class Fq2:
    """components tracked with a common bound.  mul2(x0,y0,x1,y1) = (x0 y0 + x1 y1)/R' is one dual-product
    Montgomery reduction (18 products per column):
       mul(a, b): c0 = mul2(a0, b0, a1, NEGK N - b1),  c1 = mul2(a0, b1, a1, b0)      (b = the smaller operand)
       sqr(a):    c0 = mul2(a0, a0, a1, KS N - a1),     c1 = mul(2 a0, a1)"""
    name = "Fq2"
    NEGK = 12
    KS = 17

    @staticmethod
    def mul(a, b):
        if a.V < b.V:
            a, b = b, a
        assert b.L <= 1.0 + 1e-9, "Fq2.mul: negated operand must be normalised"
        assert b.V <= Fq2.NEGK - 1 + 1e-9, "Fq2.mul: operand value %.2f needs NEGK >= %d" % (b.V, math.ceil(b.V + 1))
        bneg = B(float(Fq2.NEGK), 2.0)
        col_ok([(a.L, b.L), (a.L, bneg.L)], 18)
        c0 = (a.V * b.V + a.V * bneg.V) / RN + 1.0
        c1 = 2 * a.V * b.V / RN + 1.0
        return B(max(c0, c1), 1.0)

    @staticmethod
    def sqr(a):
        assert a.L <= 1.0 + 1e-9
        assert a.V <= Fq2.KS - 1 + 1e-9, "Fq2.sqr: value %.2f needs KS >= %d" % (a.V, math.ceil(a.V + 1))
        aneg = B(float(Fq2.KS), 2.0)
        col_ok([(a.L, a.L), (a.L, aneg.L)], 18)
        c0 = (a.V * a.V + a.V * aneg.V) / RN + 1.0
        col_ok([(2 * a.L, a.L)], 9)
        c1 = 2 * a.V * a.V / RN + 1.0
        return B(max(c0, c1), 1.0)

    @staticmethod
    def sqr_loose(a):
        """c0 = (a0 + a1)(a0 + KS N − a1) as ONE product (field29.hpp sqr_loose): a larger value, fewer instructions"""
        assert a.L <= 1.0 + 1e-9
        assert a.V <= Fq2.KS - 1 + 1e-9, "Fq2.sqr_loose: value %.2f needs KS >= %d" % (a.V, math.ceil(a.V + 1))
        s = B(2 * a.V, 2 * a.L)                       # a0 + a1
        t = B(a.V + Fq2.KS, a.L + 2.0)                # a0 + KS·N − a1 (sub<KS, 1>)
        col_ok([(s.L, t.L)], 9)
        c0 = s.V * t.V / RN + 1.0
        col_ok([(2 * a.L, a.L)], 9)
        c1 = 2 * a.V * a.V / RN + 1.0
        return B(max(c0, c1), 1.0)

    @staticmethod
    def mul_sub(a, b, c, d, KY):
        """a·b − c·d, each component one reduction over four products (curve29.hpp mul_sub, Fq2)"""
        for x in (a, b, c, d):
            assert x.L <= 1.0 + 1e-9, "Fq2.mul_sub: operands must be normalised"
        assert b.V <= Fq2.NEGK - 1 + 1e-9 and c.V <= KY - 1 + 1e-9
        bneg, cneg = B(float(Fq2.NEGK), 2.0), B(float(KY), 2.0)
        col_ok([(a.L, b.L), (a.L, bneg.L), (cneg.L, d.L), (c.L, d.L)], 36)
        c0 = (a.V * b.V + a.V * bneg.V + cneg.V * d.V + c.V * d.V) / RN + 1.0
        c1 = (2 * a.V * b.V + 2 * cneg.V * d.V) / RN + 1.0
        return B(max(c0, c1), 1.0)

    add = Fq.add
    dbl = Fq.dbl
    sub = Fq.sub
    norm = Fq.norm


# ---- the formulas of curve29.hpp, operation by operation ----------------------------------------------


def madd(F, inv, k):
    X1, Y1, ZZ1, ZZZ1 = (B(inv["x"], 1.0), B(inv["y"], 1.0), B(inv["z"], 1.0), B(inv["z"], 1.0))
    X2 = B(1.0, 1.0)
    Y2 = F.sub(B(0.0, 0.0), B(1.0, 1.0), 2, 1)             # table y, possibly negated: 2N - y ...
    if F is not Fq:
        Y2 = F.norm(Y2)                                    # ... left lazy over Fq (load_table_point_lazy_y), normalised over Fq2
    U2 = F.mul(X2, ZZ1)
    S2 = F.mul(Y2, ZZZ1)
    P = F.norm(F.sub(U2, X1, k["KX"], 1))
    R = F.norm(F.sub(S2, Y1, k["KY"], 1))
    PP = F.sqr_loose(P)
    PPP = F.mul(P, PP)
    Q = F.mul(X1, PP)
    RR = F.sqr(R)
    t = F.sub(RR, PPP, k["K1"], 1)
    X3 = F.norm(F.sub(t, F.dbl(Q), k["K2"], 2))
    d = F.sub(Q, X3, k["KX"], 1)
    if F is Fq:
        Y3 = Fq.mul_sub(d, R, Y1, PPP, k["KY"])                # d stays lazy (for_mul_sub): limbs up to 3·2^29
    else:
        Y3 = Fq2.mul_sub(F.norm(d), R, Y1, PPP, k["KY"])
    ZZ3 = F.mul(ZZ1, PP)
    ZZZ3 = F.mul(ZZZ1, PPP)
    return X3, Y3, ZZ3, ZZZ3


def add(F, inv, k):
    X1 = X2 = B(inv["x"], 1.0)
    Y1 = Y2 = B(inv["y"], 1.0)
    Z = B(inv["z"], 1.0)
    U1 = F.mul(X1, Z); U2 = F.mul(X2, Z)
    S1 = F.mul(Y1, Z); S2 = F.mul(Y2, Z)
    P = F.norm(F.sub(U2, U1, k["K1"], 1))
    R = F.norm(F.sub(S2, S1, k["K1"], 1))
    PP = F.sqr_loose(P)
    PPP = F.mul(P, PP)
    Q = F.mul(U1, PP)
    RR = F.sqr(R)
    t = F.sub(RR, PPP, k["K1"], 1)
    X3 = F.norm(F.sub(t, F.dbl(Q), k["K2"], 2))
    d = F.sub(Q, X3, k["KX"], 1)
    if F is Fq:
        Y3 = Fq.mul_sub(d, R, S1, PPP, k["KY"])                # as in madd: one dual-product reduction, d lazy
    else:
        Y3 = Fq2.mul_sub(F.norm(d), R, S1, PPP, k["KY"])
    ZZ3 = F.mul(F.mul(Z, Z), PP)
    ZZZ3 = F.mul(F.mul(Z, Z), PPP)
    return X3, Y3, ZZ3, ZZZ3


def dbl(F, inv, k, affine=False):
    X1, Y1 = B(inv["x"], 1.0), B(inv["y"], 1.0)
    if affine:
        X1 = Y1 = B(1.0, 1.0)
    Z = B(inv["z"], 1.0)
    U = F.norm(F.dbl(Y1))
    V = F.sqr(U)
    W = F.mul(U, V)
    S = F.mul(X1, V)
    X2 = F.sqr(X1)
    M = F.norm(F.add(F.dbl(X2), X2))
    X3 = F.norm(F.sub(F.sqr(M), F.dbl(S), k["K2"], 2))
    d = F.norm(F.sub(S, X3, k["KX"], 1))
    Y3 = F.norm(F.sub(F.mul(M, d), F.mul(W, Y1), k["K1"], 1))
    if affine:
        return X3, Y3, V, W
    return X3, Y3, F.mul(V, Z), F.mul(W, Z)


def check(F, inv, k):
    for name, fn in (("madd", lambda: madd(F, inv, k)), ("add", lambda: add(F, inv, k)), ("dbl", lambda: dbl(F, inv, k)),
                     ("dbl_affine", lambda: dbl(F, inv, k, True))):
        X3, Y3, ZZ3, ZZZ3 = fn()
        ok = X3.V <= inv["x"] and Y3.V <= inv["y"] and ZZ3.V <= inv["z"] and ZZZ3.V <= inv["z"]
        print("%-4s %-10s X3 %.2f (<=%g)  Y3 %.2f (<=%g)  ZZ3 %.3f ZZZ3 %.3f (<=%g)  %s" %
              (F.name, name, X3.V, inv["x"], Y3.V, inv["y"], ZZ3.V, ZZZ3.V, inv["z"], "ok" if ok else "VIOLATED"))
        assert ok
    assert k["KX"] >= inv["x"] + 1 and k["KY"] >= inv["y"] + 1


# ---- the SIGNED mixed addition of the G1 bucket accumulation (curve29.hpp madd29s, field29.hpp S29) ------------------------
# Values are tracked as intervals (lo, hi) in units of N.  Every operand of a product is s-normalised (limbs 0..7 in
# [0, 2^29), limb 8 signed and small) or a limb-wise difference of two s-normalised values (|limbs| < 2^29), so the column
# bound does not depend on the values - it is asserted once per product shape - and what has to be proved is that the
# intervals are closed under the step, that the test `is_zero_mod(ZZ3)` sees a value in [0, 2N), and that acc_to_stored's
# offsets land inside the stored invariant.
class I:
    def __init__(self, lo, hi):
        assert lo <= hi
        self.lo, self.hi = lo, hi
        assert max(abs(lo), abs(hi)) < CAP / 2, "signed value out of the representable range"
        # top limb = value / 2^232 must stay far inside an int32: |v| < 84 N  <=>  |l8| < 2^28
        assert max(abs(lo), abs(hi)) < 84.0

    def __repr__(self):
        return "I(%.3f, %.3f)" % (self.lo, self.hi)


def s_col_ok(nprod, lim_a=1.0, lim_b=1.0, fused=0):
    """nprod products per column term group (9 each), operand limb magnitudes in units of 2^29, `fused` extra 2^31 terms"""
    tot = nprod * 9 * (lim_a * 2 ** 29) * (lim_b * 2 ** 29) + 9 * (2 ** 29) ** 2 + fused * 2 ** 31 + 2 ** 36
    assert tot < 2 ** 63, "signed column accumulator overflow: 2^%.2f" % math.log2(tot)


def s_prod(a, b):
    c = [a.lo * b.lo, a.lo * b.hi, a.hi * b.lo, a.hi * b.hi]
    return min(c) / RN, max(c) / RN


def s_mul(a, b, sub=()):
    """a·b/R' + [0, N) - Σ k·x for (k, x) in sub (fused)"""
    s_col_ok(1, fused=len(sub))
    lo, hi = s_prod(a, b)
    hi += 1.0
    for k, x in sub:                         # subtract k·x, k > 0; a signed multiplier (+-1) is given as k·x with x symmetric
        lo -= k * x.hi
        hi -= k * x.lo
    return I(lo, hi)


def s_sqr(a, sub=()):
    s_col_ok(1, lim_a=2.0, fused=len(sub))   # the doubled copy
    m = max(abs(a.lo), abs(a.hi))
    lo, hi = 0.0, m * m / RN + 1.0
    for k, x in sub:
        lo -= k * x.hi
        hi -= k * x.lo
    return I(lo, hi)


def s_mul2(a, b, c, d):
    s_col_ok(2)
    l1, h1 = s_prod(a, b)
    l2, h2 = s_prod(c, d)
    return I(l1 + l2, h1 + h2 + 1.0)


def signed_madd(inv):
    X, SY, ZZ, ZZZ = I(*inv["x"]), I(*inv["sy"]), I(*inv["zz"]), I(*inv["zzz"])
    px = py = I(0.0, 1.0)                                    # table coordinates: canonical
    sym = lambda v: I(-max(abs(v.lo), abs(v.hi)), max(abs(v.lo), abs(v.hi)))
    P = s_mul(ZZ, px, [(1, X)])
    PP = s_sqr(P)
    ZZ3 = s_mul(ZZ, PP)
    assert ZZ3.lo >= 0.0 and ZZ3.hi < 2.0, "is_zero_mod(ZZ3) needs 0 <= ZZ3 < 2N"
    Q = s_mul(X, PP)
    PPP = s_mul(P, PP)
    Rs = s_mul(ZZZ, py, [(1, sym(SY))])                      # - u·sy with u = +-1
    ZZZ3 = s_mul(ZZZ, PPP)
    X3 = s_sqr(Rs, [(1, PPP), (2, Q)])
    d = sym(I(X3.lo - Q.hi, X3.hi - Q.lo))                   # u·(X3 - Q), limb-wise: |limbs| < 2^29
    SY3 = s_mul2(Rs, d, SY, PPP)
    return dict(x=X3, sy=SY3, zz=ZZ3, zzz=ZZZ3), dict(P=P, PP=PP, Q=Q, PPP=PPP, Rs=Rs, d=d)


def check_signed():
    # the running accumulator's intervals; the widest start is right after a run begins (x, sy = canonical table coordinates,
    # zz = zzz = R' mod N) or after the rare doubling (dbl_affine29's outputs made canonical: below N)
    inv = dict(x=(-3.5, 1.2), sy=(-1.2, 1.2), zz=(0.0, 1.05), zzz=(-0.01, 1.01))
    out, mid = signed_madd(inv)
    for k in ("x", "sy", "zz", "zzz"):
        ok = inv[k][0] <= out[k].lo and out[k].hi <= inv[k][1]
        print("G1s  madd29s    %-3s in (%.3f, %.3f)  within (%g, %g)  %s" % (k, out[k].lo, out[k].hi, inv[k][0], inv[k][1], "ok" if ok else "VIOLATED"))
        assert ok
    print("G1s  intermediates:", ", ".join("%s %r" % kv for kv in mid.items()))
    # acc_to_stored: x + 4N, t·sy + 2N, zz, zzz + N inside the stored invariant (X < 13 N, Y < 8 N, ZZ, ZZZ < 3 N), all >= 0
    assert inv["x"][0] + 4 >= 0 and inv["x"][1] + 4 < 13
    assert -max(map(abs, inv["sy"])) + 2 >= 0 and max(map(abs, inv["sy"])) + 2 < 8
    assert inv["zz"][0] >= 0 and inv["zz"][1] < 3
    assert inv["zzz"][0] + 1 >= 0 and inv["zzz"][1] + 1 < 3


if __name__ == "__main__":
    check_signed()
    # invariant of stored accumulators (values in units of N) and the K constants of curve29.hpp
    INV = dict(x=13.0, y=8.0, z=3.0)
    KC = dict(KX=14, KY=9, K1=4, K2=6)
    check(Fq, INV, KC)
    check(Fq2, INV, KC)
    print("all bounds hold; constants:", INV, KC)
