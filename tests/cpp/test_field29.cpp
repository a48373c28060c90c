// Host-side check of the lazy 29-bit-limb arithmetic (csrc/field29.hpp, csrc/curve29.hpp) against the
// saturated 8x32 Montgomery arithmetic (csrc/field.hpp, csrc/curve.hpp), which the GPU parity tests pin to the
// oracle.  Plain g++; exits non-zero on the first mismatch.  Also asserts the bounds tools/bounds29.py proves.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#include "../../crescent-credentials_amd/csrc/curve29.hpp"

using namespace cg;

static uint64_t rng_s = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }

template <class P>
static Fp<P> rand_canon() {
    Fp<P> a;
    for (;;) {
        for (int i = 0; i < 8; ++i) a.l[i] = (uint32_t)rnd();
        a.l[7] &= 0x3fffffffu;
        bool lt = false;
        for (int i = 7; i >= 0; --i) { if (a.l[i] < P::N[i]) { lt = true; break; } if (a.l[i] > P::N[i]) break; }
        if (lt) return a;
    }
}
#define CHECK(c, msg) do { if (!(c)) { printf("FAIL %s (line %d)\n", msg, __LINE__); exit(1); } } while (0)

// value of a lazy element as canonical bytes of x (leaves Montgomery form)
template <class P29>
static Fp<typename P29::P256> val(const F29<P29>& a) { return to_canonical_bytes(a); }

template <class P29>
static void check_norm_bound(const F29<P29>& a, const char* what) {
    for (int i = 0; i < 8; ++i) CHECK(a.l[i] <= M29, what);
}

template <class P29>
static void test_field(const char* name) {
    typedef typename P29::P256 P;
    for (int it = 0; it < 20000; ++it) {
        Fp<P> a = rand_canon<P>(), b = rand_canon<P>();
        if (it == 0) { a = Fp<P>::zero(); }
        if (it == 1) { for (int i = 0; i < 8; ++i) a.l[i] = P::N[i]; a.l[0] -= 1; b = a; }   // N-1
        F29<P29> x = from_canonical_bytes<P29>(a), y = from_canonical_bytes<P29>(b);
        check_norm_bound(x, "from_canonical normalised");
        Fp<P> am = to_mont(a), bm = to_mont(b);
        CHECK(val(x) == a, "roundtrip");
        CHECK(val(mul(x, y)) == from_mont(mul(am, bm)), "mul");
        CHECK(val(sqr(x)) == from_mont(sqr(am)), "sqr");
        CHECK(val(add(x, y)) == from_mont(add(am, bm)), "add");
        CHECK(val(sub<2, 1>(x, y)) == from_mont(sub(am, bm)), "sub");
        // lazy chain: (x + y)^2 * (x - y + 3N) with un-normalised operands at their limits
        F29<P29> s = add(x, y), d = normalize(sub<3, 1>(x, y));
        Fp<P> sm = add(am, bm), dm = sub(am, bm);
        CHECK(val(mul(sqr(s), d)) == from_mont(mul(sqr(sm), dm)), "lazy chain");
        // doubled operand in sqr at the 2^30 limit
        CHECK(val(sqr(dbl(x))) == from_mont(sqr(dbl(am))), "sqr of doubled");
        // from_mont256 path
        CHECK(val(from_mont256<P29>(am)) == a, "from_mont256");
        // pack / unpack
        uint32_t w[8];
        F29<P29> c = canonical(mul(x, y));
        check_norm_bound(c, "canonical normalised");
        pack29(c, w);
        F29<P29> c2 = unpack29<P29>(w);
        for (int i = 0; i < 9; ++i) CHECK(c.l[i] == c2.l[i], "pack/unpack");
        // is_zero_mod
        CHECK(is_zero_mod(mul(x, F29<P29>::zero())) , "zero product");
        CHECK(is_zero_mod(canonical(sub<2, 1>(x, x))), "x - x");
        if (!a.is_zero()) CHECK(!is_zero_mod(canonical(x)), "nonzero");
    }
    printf("%s field ok\n", name);
}

static Fq2 rand_fq2() { return {to_mont(rand_canon<FqP>()), to_mont(rand_canon<FqP>())}; }
static Fq2_29 to29(const Fq2& a) { return {from_mont256<Fq29P>(a.c0), from_mont256<Fq29P>(a.c1)}; }
static Fq29 to29(const Fq& a) { return from_mont256<Fq29P>(a); }
static bool same(const Fq2_29& a, const Fq2& b) { return val(a.c0) == from_mont(b.c0) && val(a.c1) == from_mont(b.c1); }
static bool same(const Fq29& a, const Fq& b) { return val(a) == from_mont(b); }

static void test_fq2() {
    for (int it = 0; it < 20000; ++it) {
        Fq2 a = rand_fq2(), b = rand_fq2();
        Fq2_29 x = to29(a), y = to29(b);
        CHECK(same(mul(x, y), mul(a, b)), "fq2 mul");
        CHECK(same(sqr(x), sqr(a)), "fq2 sqr");
        // operands at the documented limits: a loose (limbs < 2^30), b normalised with value ~10N
        Fq2_29 xl = add(x, y);
        Fq2_29 yb = normalize(sub<9, 1>(y, x));       // value < 10N
        CHECK(same(mul(xl, yb), mul(add(a, b), sub(b, a))), "fq2 mul at limits");
        Fq2_29 xs = normalize(sub<14, 1>(x, y));      // value < 15N
        CHECK(same(sqr(xs), sqr(sub(a, b))), "fq2 sqr at limits");
    }
    printf("Fq2 ok\n");
}

// ---- curve ------------------------------------------------------------------------------------------
static Fq fq_from_decimal(const char* s) {
    Fq acc = Fq::zero(), ten = Fq::zero();
    ten.l[0] = 10; ten = to_mont(ten);
    for (const char* p = s; *p; ++p) { Fq d = Fq::zero(); d.l[0] = (uint32_t)(*p - '0'); acc = add(mul(acc, ten), to_mont(d)); }
    return acc;
}
template <class F> struct Gen;
template <> struct Gen<Fq> { static Affine<Fq> g() { return {Fq::one(), add(Fq::one(), Fq::one())}; } };
template <> struct Gen<Fq2> {
    static Affine<Fq2> g() {
        return {{fq_from_decimal("10857046999023057135944570762232829481370756359578518086990519993285655852781"),
                 fq_from_decimal("11559732032986387107991004021392285783925812861821192530917403151452391805634")},
                {fq_from_decimal("8495653923123431417604973247489272438418190587263600148770280649306958101930"),
                 fq_from_decimal("4082367875863433681332203403145435568316851327593401208105741076214120093531")}};
    }
};
static Affine29<Fq29> to_aff29(const Affine<Fq>& p) { return {canonical(to29(p.x)), canonical(to29(p.y))}; }
static Affine29<Fq2_29> to_aff29(const Affine<Fq2>& p) { return {canonical(to29(p.x)), canonical(to29(p.y))}; }

template <class F29T, class F>
static bool acc_equals(const XYZZ29<F29T>& a, bool inf, const XYZZ<F>& ref) {
    if (inf || ref.is_inf()) return inf && ref.is_inf();
    // compare affine: x = X/ZZ, y = Y/ZZZ  <=>  X * ref.zz == ref.x * ZZ etc.  (cross-multiplied in the reference field)
    // convert the lazy accumulator to the saturated field through canonical bytes
    auto conv = [](const auto& f) { return f; };
    (void)conv;
    return true;
}

template <class T> static T conv_back(const Fq29& a, const Fq*) { return to_mont(val(a)); }
static Fq back(const Fq29& a) { return to_mont(val(a)); }
static Fq2 back(const Fq2_29& a) { return {to_mont(val(a.c0)), to_mont(val(a.c1))}; }

template <class F, class F29T>
static void expect_same_point(const XYZZ29<F29T>& a, bool inf, XYZZ<F> ref, const char* what) {
    if (ref.is_inf() || inf) { CHECK(ref.is_inf() && inf, what); return; }
    XYZZ<F> b{back(a.x), back(a.y), back(a.zz), back(a.zzz)};
    Affine<F> pa = to_affine(b), pr = to_affine(ref);
    CHECK(pa.x == pr.x && pa.y == pr.y, what);
}

static void check_inv(const Fq29& f, double vmax, const char* what) {
    for (int i = 0; i < 8; ++i) CHECK(f.l[i] <= M29, what);
    // value bound: limb 8 holds bits >= 232; N's limb 8 is 0x30644e -> value/N ~ l[8] / 0x30644e
    CHECK((double)f.l[8] / (double)0x30644e <= vmax + 0.01, what);
}
static void check_inv(const Fq2_29& f, double vmax, const char* what) { check_inv(f.c0, vmax, what); check_inv(f.c1, vmax, what); }

template <class F, class F29T>
static void test_curve(const char* name) {
    Affine<F> g = Gen<F>::g();
    // a few hundred distinct points k_i * G
    const int NP = 300;
    static Affine<F> pts[NP];
    static Affine29<F29T> pts29[NP];
    XYZZ<F> run = XYZZ<F>::from_affine(g);
    for (int i = 0; i < NP; ++i) {
        uint32_t k[8] = {(uint32_t)rnd(), (uint32_t)rnd(), 0, 0, 0, 0, 0, 0};
        run = scalar_mul(run, k);
        if (run.is_inf()) run = XYZZ<F>::from_affine(g);
        pts[i] = to_affine(run);
        pts29[i] = to_aff29(pts[i]);
    }
    // long accumulation chain with the special cases mixed in; invariant checked at every step
    XYZZ<F> ref = XYZZ<F>::inf();
    XYZZ29<F29T> acc{};
    bool inf = true;
    for (int it = 0; it < 6000; ++it) {
        int i = (int)(rnd() % NP);
        int kind = (int)(rnd() % 64);
        Affine<F> p = pts[i];
        Affine29<F29T> p29 = pts29[i];
        if (kind == 0 && !ref.is_inf()) {            // add the current sum itself (forces the doubling branch)
            Affine<F> cur = to_affine(ref);
            p = cur; p29 = to_aff29(cur);
        } else if (kind == 1 && !ref.is_inf()) {     // add the negative of the current sum -> identity
            Affine<F> cur = to_affine(ref);
            p = neg(cur); p29 = to_aff29(p);
        } else if (kind == 2) {                      // negated table point via the lazy negation used by the kernels
            p = neg(p);
            p29.y = normalize(sub<2, 1>(F29T::zero(), p29.y));
        }
        madd(ref, p);
        madd29(acc, inf, p29);
        if (!inf) {
            check_inv(acc.x, 13, "X invariant"); check_inv(acc.y, 8, "Y invariant");
            check_inv(acc.zz, 3, "ZZ invariant"); check_inv(acc.zzz, 3, "ZZZ invariant");
        }
        if ((it & 15) == 0 || kind < 3) expect_same_point(acc, inf, ref, "madd chain");
    }
    // XYZZ + XYZZ, including equal and opposite operands, and doubling
    for (int it = 0; it < 1500; ++it) {
        XYZZ<F> r1 = XYZZ<F>::inf(), r2 = XYZZ<F>::inf();
        XYZZ29<F29T> a1{}, a2{};
        bool i1 = true, i2 = true;
        int n1 = (int)(rnd() % 4), n2 = (int)(rnd() % 4);
        int same_pts = (rnd() % 8) == 0, opposite = (rnd() % 8) == 1;
        int idx[4] = {(int)(rnd() % NP), (int)(rnd() % NP), (int)(rnd() % NP), (int)(rnd() % NP)};
        for (int k = 0; k < n1; ++k) { madd(r1, pts[idx[k]]); madd29(a1, i1, pts29[idx[k]]); }
        for (int k = 0; k < n2; ++k) {
            int j = (same_pts || opposite) ? idx[k] : (int)(rnd() % NP);
            Affine<F> p = pts[j]; Affine29<F29T> p29 = pts29[j];
            if (opposite) { p = neg(p); p29.y = normalize(sub<2, 1>(F29T::zero(), p29.y)); }
            madd(r2, p); madd29(a2, i2, p29);
        }
        if (same_pts || opposite) { /* make the operand multisets equal */ if (n1 != n2) continue; }
        add(r1, r2);
        add29(a1, i1, a2, i2);
        expect_same_point(a1, i1, r1, "add29");
        if (!i1) {
            check_inv(a1.x, 13, "X invariant (add)"); check_inv(a1.y, 8, "Y invariant (add)");
            XYZZ29<F29T> d = dbl29(a1);
            expect_same_point(d, false, dbl(r1), "dbl29");
            check_inv(d.x, 13, "X invariant (dbl)"); check_inv(d.y, 8, "Y invariant (dbl)");
        }
    }
    printf("%s curve ok\n", name);
}

// ---- the signed-limb mixed addition of the G1 bucket accumulation (curve29.hpp madd29s) ----------------------------------
static void check_s_normalised(const Fq29s& f, double lo, double hi, const char* what) {
    for (int i = 0; i < 8; ++i) CHECK(f.l[i] >= 0 && (uint32_t)f.l[i] <= M29, what);
    const double v = (double)f.l[8] / (double)0x30644e;          // value / N, to a hundredth
    CHECK(v >= lo - 0.02 && v <= hi + 0.02, what);
}
static void test_signed_field() {
    const int32_t neg1 = -1, neg2 = -2;
    for (int it = 0; it < 20000; ++it) {
        Fq a = rand_canon<FqP>(), b = rand_canon<FqP>(), c = rand_canon<FqP>(), e = rand_canon<FqP>();
        Fq29 x = from_canonical_bytes<Fq29P>(a), y = from_canonical_bytes<Fq29P>(b), z = from_canonical_bytes<Fq29P>(c), w = from_canonical_bytes<Fq29P>(e);
        Fq am = to_mont(a), bm = to_mont(b), cm = to_mont(c), em = to_mont(e);
        Fq29s xs = Fq29s::from_unsigned(x), ys = Fq29s::from_unsigned(y), zs = Fq29s::from_unsigned(z), ws = Fq29s::from_unsigned(w);
        CHECK(val(to_unsigned<0>(mul_s(xs, ys))) == from_mont(mul(am, bm)), "mul_s");
        CHECK(val(to_unsigned<0>(sqr_s(xs))) == from_mont(sqr(am)), "sqr_s");
        // fused difference (may be negative): x·y - z, and with a per-lane sign
        Fq29s dneg = mul_s(xs, ys, FuseMul1<Fq29P>{zs, neg1});
        check_s_normalised(dneg, -1.0, 1.0, "fused difference s-normalised");
        CHECK(val(to_unsigned<2>(dneg)) == from_mont(sub(mul(am, bm), cm)), "mul_s fused -z");
        Fq29s dpos = mul_s(xs, ys, FuseMul1<Fq29P>{zs, 1});
        CHECK(val(to_unsigned<0>(dpos)) == from_mont(add(mul(am, bm), cm)), "mul_s fused +z");
        // a NEGATIVE multiplicand: (x·y - z)·w and (x·y - z)²
        CHECK(val(to_unsigned<2>(mul_s(dneg, ws))) == from_mont(mul(sub(mul(am, bm), cm), em)), "mul_s with a negative operand");
        CHECK(val(to_unsigned<0>(sqr_s(dneg))) == from_mont(sqr(sub(mul(am, bm), cm))), "sqr_s of a negative value");
        // x² - z - 2w fused
        Fq t2 = sub(sub(sqr(am), cm), add(em, em));
        Fq29s f2 = sqr_s(xs, FuseMul2<Fq29P>{zs, neg1, ws, neg2});
        check_s_normalised(f2, -3.0, 1.0, "fused square s-normalised");
        CHECK(val(to_unsigned<4>(f2)) == from_mont(t2), "sqr_s fused -z -2w");
        // dual product with a limb-wise (lazy, signed) difference as one operand
        Fq29s lazy;
        for (int i = 0; i < 9; ++i) lazy.l[i] = zs.l[i] - ws.l[i];
        CHECK(val(to_unsigned<2>(mul2_s(xs, lazy, ys, dneg))) == from_mont(add(mul(am, sub(cm, em)), mul(bm, sub(mul(am, bm), cm)))), "mul2_s");
    }
    printf("signed Fq ok\n");
}
static void test_signed_madd() {
    const int32_t neg1 = -1, neg2 = -2;
    Affine<Fq> g = Gen<Fq>::g();
    const int NP = 300;
    static Affine<Fq> pts[NP];
    static Affine29<Fq29> pts29[NP];
    XYZZ<Fq> run = XYZZ<Fq>::from_affine(g);
    for (int i = 0; i < NP; ++i) {
        uint32_t k[8] = {(uint32_t)rnd(), (uint32_t)rnd(), 0, 0, 0, 0, 0, 0};
        run = scalar_mul(run, k);
        if (run.is_inf()) run = XYZZ<Fq>::from_affine(g);
        pts[i] = to_affine(run);
        pts29[i] = to_aff29(pts[i]);
    }
    XYZZ<Fq> ref = XYZZ<Fq>::inf();
    G1AccS acc{};
    bool inf = true;
    double xlo = 0, xhi = 0, ylo = 0, yhi = 0, zlo = 0, zhi = 0;
    for (int it = 0; it < 40000; ++it) {
        int i = (int)(rnd() % NP);
        int kind = (int)(rnd() % 64);
        Affine<Fq> p = pts[i];
        Affine29<Fq29> p29 = pts29[i];
        int32_t sigma = (rnd() & 1) ? -1 : 1;            // signed digits: half of the entries carry a negated point
        if (kind == 0 && !ref.is_inf()) {                // the current sum itself (doubling branch), under either sign convention
            Affine<Fq> cur = to_affine(ref);
            if (sigma < 0) { p29 = to_aff29(neg(cur)); } else { p29 = to_aff29(cur); }
            p = cur;
        } else if (kind == 1 && !ref.is_inf()) {         // the negative of the current sum -> identity
            Affine<Fq> cur = to_affine(ref);
            if (sigma < 0) { p29 = to_aff29(cur); } else { p29 = to_aff29(neg(cur)); }
            p = neg(cur);
        } else if (sigma < 0) {
            p = neg(p);
        }
        madd(ref, p);
        madd29s(acc, inf, p29, sigma, neg1, neg2);
        if (!inf) {
            check_s_normalised(acc.x, -3.5, 1.2, "signed X");        // the intervals tools/bounds29.py proves closed
            check_s_normalised(acc.sy, -1.2, 1.2, "signed sY");
            check_s_normalised(acc.zz, 0.0, 1.05, "signed ZZ");
            check_s_normalised(acc.zzz, -0.01, 1.01, "signed ZZZ");
            CHECK(acc.t == 1 || acc.t == -1, "sign");
            const double N8 = (double)0x30644e;
            xlo = std::min(xlo, acc.x.l[8] / N8); xhi = std::max(xhi, acc.x.l[8] / N8);
            ylo = std::min(ylo, acc.sy.l[8] / N8); yhi = std::max(yhi, acc.sy.l[8] / N8);
            zlo = std::min(zlo, acc.zzz.l[8] / N8); zhi = std::max(zhi, acc.zzz.l[8] / N8);
            if ((it & 7) == 0 || kind < 2) {
                XYZZ29<Fq29> st = acc_to_stored(acc);
                {   // the record the accumulation writes and what its readers make of it (load_acc's conversion)
                    uint32_t w[36], ref_w[36];
                    pack_signed_record(acc, w);
                    CHECK((w[26] & 0x80000000u) != 0, "signed record mark");
                    signed_record_to_stored(w);
                    store_limbs(st.x, ref_w); store_limbs(st.y, ref_w + 9); store_limbs(st.zz, ref_w + 18); store_limbs(st.zzz, ref_w + 27);
                    CHECK(memcmp(w, ref_w, sizeof(w)) == 0, "signed record -> stored form");
                    CHECK((ref_w[26] & 0xc0000000u) == 0, "a stored record never carries the mark");
                }
                check_inv(st.x, 13, "stored X"); check_inv(st.y, 8, "stored Y"); check_inv(st.zz, 3, "stored ZZ"); check_inv(st.zzz, 3, "stored ZZZ");
                expect_same_point(st, false, ref, "signed madd chain");
                // and the stored form feeds the unsigned formulas (the combine / reduction kernels)
                if ((it & 63) == 0) {
                    XYZZ29<Fq29> two = st;
                    bool i2 = false;
                    add29(two, i2, st, false);
                    expect_same_point(two, i2, dbl(ref), "stored form through add29");
                }
            }
        } else {
            CHECK(ref.is_inf(), "signed madd chain: identity");
        }
    }
    printf("signed G1 madd ok (observed X in (%.2f, %.2f) N, sY in (%.2f, %.2f) N, ZZZ in (%.2f, %.2f) N)\n", xlo, xhi, ylo, yhi, zlo, zhi);
}

int main() {
    test_signed_field();
    test_signed_madd();
    test_field<Fq29P>("Fq");
    test_field<Fr29P>("Fr");
    test_fq2();
    test_curve<Fq, Fq29>("G1");
    test_curve<Fq2, Fq2_29>("G2");
    printf("ALL OK\n");
    return 0;
}
