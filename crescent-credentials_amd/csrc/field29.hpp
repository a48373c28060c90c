// BN254 field arithmetic on UNSATURATED 29-bit limbs — the representation the hot kernels compute in.
//
// Why: on gfx950 `v_mad_u64_u32` (32x32+64 -> 64) issues at the same rate as a plain 32-bit add
// (profiles/r01_valu_issue_rates.txt), so the cost of a modular product is its instruction count, and
// in a saturated 8 x 32-bit Montgomery product three quarters of the instructions are carry handling
// and register shuffling.  With nine 29-bit limbs (261 bits) every partial product is < 2^58..2^60,
// a whole column of up to 18 products accumulates in ONE 64-bit register pair by chained mads with no
// carry instruction at all, and additions/subtractions are nine independent 32-bit adds.
//
//   value(a) = Σ a.l[i] * 2^(29 i),   Montgomery factor R' = 2^261  (R'/N ≈ 169)
//
// Elements are LAZY: limbs may exceed 29 bits and values may exceed N; each operation states what it
// needs and what it returns.  "normalised" = limbs 0..7 < 2^29 (limb 8 carries the rest of the value).
//   mul(a,b)  needs  (max limb a)·(max limb b) <= 2^60;  returns normalised, value < A·B/R' + N
//   add(a,b)  limb-wise, no carry
//   sub<K,T>(a,b) = a + K·N - b limb-wise; needs b's limbs 0..7 <= T·(2^29-1) and value(b) < (K-1)·N
//   normalize(a)  carry-propagates (value unchanged)
// The 32 bytes that reach HBM for a table point are the canonical value x·R' mod N packed as eight
// u32; accumulators travel as nine u32 per coordinate.
#pragma once
#include "field.hpp"

namespace cg {

static constexpr uint32_t M29 = (1u << 29) - 1u;

struct Fq29P {
    static constexpr uint32_t N[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                                      0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t NINV = 0x04866389u;  // -N^-1 mod 2^29
    static constexpr uint32_t ONE[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u,
                                        0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};  // R' mod N
    static constexpr uint32_t R2[9] = {0x059bac10u, 0x0d1503a3u, 0x018016b8u, 0x10ab0ca8u, 0x02632639u,
                                       0x02c0169fu, 0x169bfd53u, 0x11869d4cu, 0x002a11a6u};   // R'^2 mod N
    static constexpr uint32_t C256[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u,
                                         0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};  // R'^2 / 2^256 mod N
    typedef FqP P256;
};
struct Fr29P {
    static constexpr uint32_t N[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                                      0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t NINV = 0x0fffffffu;
    static constexpr uint32_t ONE[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu,
                                        0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    static constexpr uint32_t R2[9] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu,
                                       0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};
    static constexpr uint32_t C256[9] = {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau,
                                         0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
    typedef FrP P256;
};

template <class P>
struct F29 {
    uint32_t l[9];
    CG_HD static F29 zero() {
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = 0;
        return r;
    }
    CG_HD static F29 one() {
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = P::ONE[i];
        return r;
    }
    CG_HD static F29 from_limbs(const uint32_t (&c)[9]) {
        F29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = c[i];
        return r;
    }
    CG_HD bool all_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) o |= l[i];
        return o == 0;
    }
};

// limbs of K*N with every limb 0..7 boosted by T*2^29 (borrowed from the limb above), so that
// a + KN<K,T> - b never goes negative in any limb when b's limbs are <= T*(2^29-1).
template <class P, int K, int T>
struct KN {
    struct Arr { uint32_t v[9]; };
    static constexpr Arr make() {
        Arr a{};
        uint64_t carry = 0;
        for (int i = 0; i < 9; ++i) {
            uint64_t x = (uint64_t)P::N[i] * (uint64_t)K + carry;
            if (i < 8) { a.v[i] = (uint32_t)(x & M29); carry = x >> 29; }
            else a.v[i] = (uint32_t)x;
        }
        a.v[0] += (uint32_t)T << 29;
        for (int i = 1; i < 8; ++i) a.v[i] += ((uint32_t)T << 29) - (uint32_t)T;
        a.v[8] -= (uint32_t)T;
        return a;
    }
    static constexpr Arr value = make();
};

template <class P>
CG_HD F29<P> add(const F29<P>& a, const F29<P>& b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <class P>
CG_HD F29<P> dbl(const F29<P>& a) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] << 1;
    return r;
}
template <int K, int T, class P>
CG_HD F29<P> sub(const F29<P>& a, const F29<P>& b) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + KN<P, K, T>::value.v[i] - b.l[i];
    return r;
}
template <class P>
CG_HD F29<P> normalize(const F29<P>& a) {
    F29<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t t = a.l[i] + c;
        r.l[i] = t & M29;
        c = t >> 29;
    }
    r.l[8] = a.l[8] + c;
    return r;
}

// CG_PIN(c): makes the running column sum an opaque value at that point (an llvm.annotation call, which the code
// generator lowers to its argument: no instruction, no register constraint).  Without it LLVM's reassociation sorts the
// terms of a column by rank, which puts the shifted carry of the previous column LAST: every column then starts a fresh
// accumulator at zero and pays one 64-bit addition (v_lshl_add_u64) to bring the carry in - 21 of the 228 instructions
// of a product.  Pinned, the carry is the addend of the column's first v_mad_u64_u32 and a product is 207 instructions.
// (An empty inline asm pins just as well but makes the hazard recogniser put a wait state after every pin.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CG_NO_PIN)   // CG_NO_PIN: A/B aid
#define CG_PIN(c) c = __builtin_annotation(c, "pin")
#else
#define CG_PIN(c) ((void)0)
#endif

// Montgomery product, product scanning with the reduction interleaved.  One 64-bit accumulator.
// The cores below are always inlined; `mul`/`sqr`/`mul2` wrap them either inline or as real calls (see the end
// of this block).
template <class P>
CG_HD F29<P> mul_core(const F29<P>& a, const F29<P>& b) {
    uint64_t c = 0;
    uint32_t m[9];
    F29<P> r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) { c += (uint64_t)a.l[i] * b.l[k - i]; CG_PIN(c); }
#pragma unroll
        for (int i = 0; i < k; ++i) { c += (uint64_t)m[i] * P::N[k - i]; CG_PIN(c); }
        m[k] = ((uint32_t)c * P::NINV) & M29;
        c += (uint64_t)m[k] * P::N[0];
        c >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) {
            c += (uint64_t)a.l[i] * b.l[k - i]; CG_PIN(c);
            c += (uint64_t)m[i] * P::N[k - i]; CG_PIN(c);
        }
        r.l[k - 9] = (uint32_t)c & M29;
        c >>= 29;
    }
    r.l[8] = (uint32_t)c;
    return r;
}
// Square: cross products once against a doubled copy.  Needs limbs(a) <= 2^30.
template <class P>
CG_HD F29<P> sqr_core(const F29<P>& a) {
    uint64_t c = 0;
    uint32_t m[9], d[9];
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = a.l[i] << 1;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; 2 * i < k; ++i) { c += (uint64_t)d[i] * a.l[k - i]; CG_PIN(c); }
        if ((k & 1) == 0) { c += (uint64_t)a.l[k / 2] * a.l[k / 2]; CG_PIN(c); }
#pragma unroll
        for (int i = 0; i < k; ++i) { c += (uint64_t)m[i] * P::N[k - i]; CG_PIN(c); }
        m[k] = ((uint32_t)c * P::NINV) & M29;
        c += (uint64_t)m[k] * P::N[0];
        c >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; ++i) { c += (uint64_t)d[i] * a.l[k - i]; CG_PIN(c); }
        if ((k & 1) == 0) { c += (uint64_t)a.l[k / 2] * a.l[k / 2]; CG_PIN(c); }
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) { c += (uint64_t)m[i] * P::N[k - i]; CG_PIN(c); }
        r.l[k - 9] = (uint32_t)c & M29;
        c >>= 29;
    }
    r.l[8] = (uint32_t)c;
    return r;
}

// The products are always inlined into their callers (220 / 189 instructions).  Out-of-line forms were measured: a
// call moves 18-36 argument registers and makes the caller keep its live values in callee-saved registers, which cost
// more VGPRs AND more time than the inlined code (G2 accumulation: 218 vs 191 VGPRs, 3.55 vs 2.80 ms at 2^20).
template <class P>
CG_HD F29<P> mul(const F29<P>& a, const F29<P>& b) { return mul_core(a, b); }
template <class P>
CG_HD F29<P> sqr(const F29<P>& a) { return sqr_core(a); }

// a - N if a >= N else a; a normalised with value < 2N.  Result normalised, value < N.
template <class P>
CG_HD F29<P> cond_sub_n(const F29<P>& a) {
    F29<P> t;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t d = a.l[i] - P::N[i] - borrow;
        borrow = d >> 31;
        t.l[i] = (i < 8) ? (d & M29) : d;
    }
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = borrow ? a.l[i] : t.l[i];
    return r;
}

// canonical representative (< N, normalised) of a lazy element (value < 64 N, limbs <= 2^30), same Montgomery form
template <class P>
CG_HD F29<P> canonical(const F29<P>& a) {
    return cond_sub_n(mul(a, F29<P>::one()));   // a·R'/R' = a (mod N), value < a/169 + N < 2N
}
// is a ≡ 0 (mod N)?  a must be a mul()/sqr() output or otherwise normalised with value < 2N.
template <class P>
CG_HD bool is_zero_mod(const F29<P>& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        z |= a.l[i];
        e |= a.l[i] ^ P::N[i];
    }
    return z == 0 || e == 0;
}

// one-limb filter in front of is_zero_mod (same precondition): false for all but 2^-28 of the non-zero values, so the
// hot loops pay two compares instead of twenty-seven logic operations per test
template <class P>
CG_HD bool maybe_zero_mod(const F29<P>& a) { return a.l[0] == 0u || a.l[0] == P::N[0]; }

// 8 x u32 (a 256-bit little-endian integer < 2^256) <-> nine 29-bit limbs
template <class P>
CG_HD F29<P> unpack29(const uint32_t w[8]) {
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t v = w[wi];
        if (wi + 1 < 8) v |= (uint64_t)w[wi + 1] << 32;
        r.l[i] = (uint32_t)(v >> sh) & (i < 8 ? M29 : 0xffffffffu);
    }
    return r;
}
// a must be normalised with value < 2^256
template <class P>
CG_HD void pack29(const F29<P>& a, uint32_t w[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // bits [32 j, 32 j + 32)
        const int lo = (32 * j) / 29, sh = 32 * j - 29 * lo;
        uint64_t v = (uint64_t)a.l[lo] >> sh;
        v |= (uint64_t)a.l[lo + 1] << (29 - sh);
        if (lo + 2 < 9 && 58 - sh < 32) v |= (uint64_t)a.l[lo + 2] << (58 - sh);
        w[j] = (uint32_t)v;
    }
}

// conversions with the 8 x 32 Montgomery (R = 2^256) world
template <class P>
CG_HD F29<P> from_canonical_bytes(const Fp<typename P::P256>& c) {   // plain integer x -> x·R'
    return canonical(mul(unpack29<P>(c.l), F29<P>::from_limbs(P::R2)));
}
template <class P>
CG_HD F29<P> from_mont256(const Fp<typename P::P256>& c) {           // x·2^256 -> x·R'
    return canonical(mul(unpack29<P>(c.l), F29<P>::from_limbs(P::C256)));
}
template <class P>
CG_HD Fp<typename P::P256> to_canonical_bytes(const F29<P>& a) {     // x·R' -> plain integer x
    F29<P> one = F29<P>::zero();
    one.l[0] = 1;
    F29<P> t = cond_sub_n(mul(a, one));
    Fp<typename P::P256> r;
    pack29(t, r.l);
    return r;
}

using Fq29 = F29<Fq29P>;
using Fr29 = F29<Fr29P>;

// ---- SIGNED 29-bit limbs: the G1 bucket accumulation's inner arithmetic (curve29.hpp madd29s) ----------------------------
// value(a) = Σ a.l[i]·2^(29 i) with SIGNED limbs; "s-normalised" = limbs 0..7 in [0, 2^29), limb 8 signed (it carries the
// sign of the value).  Products are chains of v_mad_i64_i32 into one signed 64-bit column sum, carries are arithmetic
// shifts, and the Montgomery reduction is the same interleaved one (m_k from the low 29 bits of the column, which two's
// complement keeps right).  What the signed form buys (round 5; the round-4 verdict's "signed limbs"):
//   * a - b needs no multiple of N added to stay representable: limb-wise it is ONE subtraction per limb, not two;
//   * better, a difference that follows a product is FUSED into the product's high columns - "c -= x_j" is one
//     v_mad_i64_i32 (x_j, -1, c) on the column that is about to give limb j - so `normalize(sub<K,T>(mul(a, b), x))`
//     (18 + 25 instructions after the product) becomes 9, the result leaves the product s-normalised, and a per-lane sign
//     (u = +-1 in a VGPR) as the multiplier makes a CONDITIONAL negation of x free.
// The mixed addition drops from ~192 to ~54 instructions of additive work around its 1629 + 430 of products (-6 %).
// Column bound: |Σ| < 2^63.  With every multiplicand limb below 2^29 in magnitude (2^30 for the doubled copy of a square)
// a single product's column holds 9 products + 9 reduction terms + one fused term + the carry: 18·2^58 + 2^31 + 2^35 <
// 2^62.2; the dual product 27·2^58 < 2^62.8.  tools/bounds29.py replays the value bounds (they are small: |X| < 4.5 N,
// |sY| < 1.2 N, 0 <= ZZ < 1.05 N, |ZZZ| < 1.01 N).
template <class P>
struct S29 {
    int32_t l[9];
    CG_HD static S29 from_unsigned(const F29<P>& a) {      // a normalised, top limb below 2^31
        S29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = (int32_t)a.l[i];
        return r;
    }
    CG_HD static S29 one() { return from_unsigned(F29<P>::one()); }
};
// hooks of the fused forms: called as hook(j, c) on the column sum that is about to give limb j (j = 8: the top limb)
struct NoFuse { CG_HD void operator()(int, int64_t&) const {} };
template <class P>
struct FuseMul1 {            // c += s·x_j   (s: an opaque -1, or a per-lane +-1)
    const S29<P>& x; int32_t s;
    CG_HD void operator()(int j, int64_t& c) const { c += (int64_t)x.l[j] * (int64_t)s; }
};
template <class P>
struct FuseMul2 {            // c += s1·x_j + s2·y_j
    const S29<P>& x; int32_t s1; const S29<P>& y; int32_t s2;
    CG_HD void operator()(int j, int64_t& c) const { c += (int64_t)x.l[j] * (int64_t)s1; CG_PIN(c); c += (int64_t)y.l[j] * (int64_t)s2; }
};
#define CG_SN(i) ((int64_t)(int32_t)P::N[i])
// (a·b)/R' + fused terms; a, b: limbs below 2^29 in magnitude (see the column bound above).  Result s-normalised.
template <class P, class Fuse>
CG_HD S29<P> mul_s(const S29<P>& a, const S29<P>& b, Fuse fuse) {
    int64_t c = 0;
    int32_t m[9];
    S29<P> r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) { c += (int64_t)a.l[i] * (int64_t)b.l[k - i]; CG_PIN(c); }
#pragma unroll
        for (int i = 0; i < k; ++i) { c += (int64_t)m[i] * CG_SN(k - i); CG_PIN(c); }
        m[k] = (int32_t)(((uint32_t)c * P::NINV) & M29);
        c += (int64_t)m[k] * CG_SN(0);
        c >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) {
            c += (int64_t)a.l[i] * (int64_t)b.l[k - i]; CG_PIN(c);
            c += (int64_t)m[i] * CG_SN(k - i); CG_PIN(c);
        }
        fuse(k - 9, c); CG_PIN(c);
        r.l[k - 9] = (int32_t)((uint32_t)c & M29);
        c >>= 29;
    }
    fuse(8, c);
    r.l[8] = (int32_t)c;
    return r;
}
template <class P>
CG_HD S29<P> mul_s(const S29<P>& a, const S29<P>& b) { return mul_s(a, b, NoFuse{}); }
// a²/R' + fused terms
template <class P, class Fuse>
CG_HD S29<P> sqr_s(const S29<P>& a, Fuse fuse) {
    int64_t c = 0;
    int32_t m[9], d[9];
    S29<P> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = a.l[i] * 2;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; 2 * i < k; ++i) { c += (int64_t)d[i] * (int64_t)a.l[k - i]; CG_PIN(c); }
        if ((k & 1) == 0) { c += (int64_t)a.l[k / 2] * (int64_t)a.l[k / 2]; CG_PIN(c); }
#pragma unroll
        for (int i = 0; i < k; ++i) { c += (int64_t)m[i] * CG_SN(k - i); CG_PIN(c); }
        m[k] = (int32_t)(((uint32_t)c * P::NINV) & M29);
        c += (int64_t)m[k] * CG_SN(0);
        c >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; ++i) { c += (int64_t)d[i] * (int64_t)a.l[k - i]; CG_PIN(c); }
        if ((k & 1) == 0) { c += (int64_t)a.l[k / 2] * (int64_t)a.l[k / 2]; CG_PIN(c); }
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) { c += (int64_t)m[i] * CG_SN(k - i); CG_PIN(c); }
        fuse(k - 9, c); CG_PIN(c);
        r.l[k - 9] = (int32_t)((uint32_t)c & M29);
        c >>= 29;
    }
    fuse(8, c);
    r.l[8] = (int32_t)c;
    return r;
}
template <class P>
CG_HD S29<P> sqr_s(const S29<P>& a) { return sqr_s(a, NoFuse{}); }
// (x0·y0 + x1·y1)/R'; limbs below 2^29 in magnitude (x0 may be a limb-wise difference of two s-normalised values)
template <class P>
CG_HD S29<P> mul2_s(const S29<P>& x0, const S29<P>& y0, const S29<P>& x1, const S29<P>& y1) {
    int64_t c = 0;
    int32_t m[9];
    S29<P> r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            c += (int64_t)x0.l[i] * (int64_t)y0.l[k - i]; CG_PIN(c);
            c += (int64_t)x1.l[i] * (int64_t)y1.l[k - i]; CG_PIN(c);
        }
#pragma unroll
        for (int i = 0; i < k; ++i) { c += (int64_t)m[i] * CG_SN(k - i); CG_PIN(c); }
        m[k] = (int32_t)(((uint32_t)c * P::NINV) & M29);
        c += (int64_t)m[k] * CG_SN(0);
        c >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) {
            c += (int64_t)x0.l[i] * (int64_t)y0.l[k - i]; CG_PIN(c);
            c += (int64_t)x1.l[i] * (int64_t)y1.l[k - i]; CG_PIN(c);
            c += (int64_t)m[i] * CG_SN(k - i); CG_PIN(c);
        }
        r.l[k - 9] = (int32_t)((uint32_t)c & M29);
        c >>= 29;
    }
    r.l[8] = (int32_t)c;
    return r;
}
#undef CG_SN
// value + K·N, carried: an s-normalised (or limb-wise lazy) value above -K·N -> the unsigned normalised form
template <int K, class P>
CG_HD F29<P> to_unsigned(const S29<P>& a) {
    F29<P> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int32_t t = a.l[i] + (int32_t)KN<P, K, 0>::value.v[i] + c;
        r.l[i] = (uint32_t)t & M29;
        c = t >> 29;
    }
    r.l[8] = (uint32_t)(a.l[8] + (int32_t)KN<P, K, 0>::value.v[8] + c);
    return r;
}
// is a ≡ 0 (mod N)?  a s-normalised with 0 <= value < 2N (a product of two non-negative values)
template <class P>
CG_HD bool is_zero_mod(const S29<P>& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        z |= (uint32_t)a.l[i];
        e |= (uint32_t)a.l[i] ^ P::N[i];
    }
    return z == 0 || e == 0;
}
using Fq29s = S29<Fq29P>;

// ---- Fq2 = Fq[u]/(u^2+1) on 29-bit limbs ------------------------------------------------------------
// A product is two DUAL-PRODUCT Montgomery reductions, (x0 y0 + x1 y1)/R' with 18 partial products per
// column in the same 64-bit accumulator: c0 = a0 b0 + a1 (K N - b1), c1 = a0 b1 + a1 b0.  That is the
// multiply count of Karatsuba (3 x 162 = 2 x 243) with no subtraction of reduced values, so results stay
// small.  Bounds are verified by tools/bounds29.py.
struct Fq2_29 {
    Fq29 c0, c1;
    CG_HD static Fq2_29 zero() { return {Fq29::zero(), Fq29::zero()}; }
    CG_HD static Fq2_29 one() { return {Fq29::one(), Fq29::zero()}; }
    CG_HD bool all_zero() const { return c0.all_zero() && c1.all_zero(); }
};
static constexpr int FQ2_NEGK = 12;   // mul(a, b): value(b) < 11 N, b normalised
static constexpr int FQ2_KS = 17;     // sqr(a):    value(a) < 16 N, a normalised

CG_HD Fq2_29 add(const Fq2_29& a, const Fq2_29& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
CG_HD Fq2_29 dbl(const Fq2_29& a) { return {dbl(a.c0), dbl(a.c1)}; }
template <int K, int T>
CG_HD Fq2_29 sub(const Fq2_29& a, const Fq2_29& b) { return {sub<K, T>(a.c0, b.c0), sub<K, T>(a.c1, b.c1)}; }
CG_HD Fq2_29 normalize(const Fq2_29& a) { return {normalize(a.c0), normalize(a.c1)}; }

// (x0 y0 + x1 y1) / R' mod N; needs 9·(Lx0·Ly0 + Lx1·Ly1) + 9·2^58 < 2^64
CG_HD Fq29 mul2_core(const Fq29& x0, const Fq29& y0, const Fq29& x1, const Fq29& y1) {
    typedef Fq29P P;
    uint64_t c = 0;
    uint32_t m[9];
    Fq29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            c += (uint64_t)x0.l[i] * y0.l[k - i]; CG_PIN(c);
            c += (uint64_t)x1.l[i] * y1.l[k - i]; CG_PIN(c);
        }
#pragma unroll
        for (int i = 0; i < k; ++i) { c += (uint64_t)m[i] * P::N[k - i]; CG_PIN(c); }
        m[k] = ((uint32_t)c * P::NINV) & M29;
        c += (uint64_t)m[k] * P::N[0];
        c >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) {
            c += (uint64_t)x0.l[i] * y0.l[k - i]; CG_PIN(c);
            c += (uint64_t)x1.l[i] * y1.l[k - i]; CG_PIN(c);
            c += (uint64_t)m[i] * P::N[k - i]; CG_PIN(c);
        }
        r.l[k - 9] = (uint32_t)c & M29;
        c >>= 29;
    }
    r.l[8] = (uint32_t)c;
    return r;
}
CG_HD Fq29 mul2(const Fq29& x0, const Fq29& y0, const Fq29& x1, const Fq29& y1) { return mul2_core(x0, y0, x1, y1); }
// (x0 y0 + x1 y1 + x2 y2 + x3 y3) / R' mod N: four products per column term, one reduction.  Needs
// 9·Σ Lxi·Lyi + 9·2^58 + carry < 2^64, i.e. Σ (limb bounds in units of 2^29) <= 6 (tools/bounds29.py).
CG_HD Fq29 mul4_core(const Fq29& x0, const Fq29& y0, const Fq29& x1, const Fq29& y1, const Fq29& x2, const Fq29& y2, const Fq29& x3,
                     const Fq29& y3) {
    typedef Fq29P P;
    uint64_t c = 0;
    uint32_t m[9];
    Fq29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            c += (uint64_t)x0.l[i] * y0.l[k - i]; CG_PIN(c);
            c += (uint64_t)x1.l[i] * y1.l[k - i]; CG_PIN(c);
            c += (uint64_t)x2.l[i] * y2.l[k - i]; CG_PIN(c);
            c += (uint64_t)x3.l[i] * y3.l[k - i]; CG_PIN(c);
        }
#pragma unroll
        for (int i = 0; i < k; ++i) { c += (uint64_t)m[i] * P::N[k - i]; CG_PIN(c); }
        m[k] = ((uint32_t)c * P::NINV) & M29;
        c += (uint64_t)m[k] * P::N[0];
        c >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) {
            c += (uint64_t)x0.l[i] * y0.l[k - i]; CG_PIN(c);
            c += (uint64_t)x1.l[i] * y1.l[k - i]; CG_PIN(c);
            c += (uint64_t)x2.l[i] * y2.l[k - i]; CG_PIN(c);
            c += (uint64_t)x3.l[i] * y3.l[k - i]; CG_PIN(c);
            c += (uint64_t)m[i] * P::N[k - i]; CG_PIN(c);
        }
        r.l[k - 9] = (uint32_t)c & M29;
        c >>= 29;
    }
    r.l[8] = (uint32_t)c;
    return r;
}
// b must be the operand that is normalised with value < (FQ2_NEGK-1) N; a may have limbs < 2^30.
CG_HD Fq2_29 mul(const Fq2_29& a, const Fq2_29& b) {
    Fq29 nb1 = sub<FQ2_NEGK, 1>(Fq29::zero(), b.c1);          // K N - b1, limbs < 2^30
    return {mul2(a.c0, b.c0, a.c1, nb1), mul2(a.c0, b.c1, a.c1, b.c0)};
}
// a normalised, value < (FQ2_KS-1) N:  c0 = a0^2 - a1^2,  c1 = 2 a0 a1
CG_HD Fq2_29 sqr(const Fq2_29& a) {
    Fq29 na1 = sub<FQ2_KS, 1>(Fq29::zero(), a.c1);
    return {mul2(a.c0, a.c0, a.c1, na1), mul(dbl(a.c0), a.c1)};
}
// The same value from two SINGLE products, c0 = (a0 + a1)·(a0 + KS·N - a1) (limbs 2·2^29 against 3·2^29: 54 of the 64
// units of column headroom): 80 instructions fewer, but c0 comes out up to 2·V·(V + KS)/169 + 1 instead of
// V·(V + KS)/169 + 1 - for squares that only feed further products (PP in the mixed addition), not sums.
CG_HD Fq2_29 sqr_loose(const Fq2_29& a) {
    return {mul(add(a.c0, a.c1), sub<FQ2_KS, 1>(a.c0, a.c1)), mul(dbl(a.c0), a.c1)};
}
template <class P>
CG_HD F29<P> sqr_loose(const F29<P>& a) { return sqr_core(a); }
CG_HD Fq2_29 canonical(const Fq2_29& a) { return {canonical(a.c0), canonical(a.c1)}; }
CG_HD bool is_zero_mod(const Fq2_29& a) { return is_zero_mod(a.c0) && is_zero_mod(a.c1); }
CG_HD bool maybe_zero_mod(const Fq2_29& a) { return maybe_zero_mod(a.c0) && maybe_zero_mod(a.c1); }

}  // namespace cg
