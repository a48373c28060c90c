// XYZZ group arithmetic over the lazy 29-bit-limb fields (field29.hpp): the formulas the MSM hot
// kernels run.  Same EFD formulas as curve.hpp (madd-2008-s, add-2008-s, dbl-2008-s-1), but every
// subtraction names the multiple of N it adds and every place a carry propagation is needed is
// explicit.  The bounds (values in units of N, limbs in units of 2^29) are machine-checked by
// tools/bounds29.py for BOTH coordinate fields with these constants:
//
//     stored accumulator invariant:  X < 13 N,  Y < 8 N,  ZZ, ZZZ < 3 N,  all normalised
//     KX = 14, KY = 9, K1 = 4, K2 = 6        (and FQ2_NEGK = 12, FQ2_KS = 17 inside Fq2)
//
// Identity is tracked by an explicit flag in registers and stored as ZZ = all-zero limbs.
#pragma once
#include "curve.hpp"
#include "field29.hpp"

namespace cg {

static constexpr int KX = 14, KY = 9, K1 = 4, K2 = 6;

template <class F>
struct Affine29 {   // table point: coordinates canonical (< N), normalised; never the identity
    F x, y;
};
template <class F>
struct XYZZ29 {
    F x, y, zz, zzz;
};

// number of u32 words of a packed table point / of a stored accumulator
template <class F> struct Words29;
template <> struct Words29<Fq29> { static constexpr int AFF = 16, ACC = 36, NF = 1; };
template <> struct Words29<Fq2_29> { static constexpr int AFF = 32, ACC = 72, NF = 2; };

// ---- memory forms ------------------------------------------------------------------------------------
CG_HD Fq29 load_packed(const uint32_t* w) { return unpack29<Fq29P>(w); }
CG_HD void load_coord(Fq29& f, const uint32_t* w) { f = unpack29<Fq29P>(w); }
CG_HD void load_coord(Fq2_29& f, const uint32_t* w) { f.c0 = unpack29<Fq29P>(w); f.c1 = unpack29<Fq29P>(w + 8); }
CG_HD void load_limbs(Fq29& f, const uint32_t* w) {
#pragma unroll
    for (int i = 0; i < 9; ++i) f.l[i] = w[i];
}
CG_HD void load_limbs(Fq2_29& f, const uint32_t* w) { load_limbs(f.c0, w); load_limbs(f.c1, w + 9); }
CG_HD void store_limbs(const Fq29& f, uint32_t* w) {
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = f.l[i];
}
CG_HD void store_limbs(const Fq2_29& f, uint32_t* w) { store_limbs(f.c0, w); store_limbs(f.c1, w + 9); }

#if defined(__HIPCC__)
// table point i of `table` (AFF words each), y negated when `negate`
template <class F>
CG_HD Affine29<F> load_table_point(const uint32_t* __restrict__ table, uint32_t idx, bool negate) {
    constexpr int AFF = Words29<F>::AFF;
    const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)idx * AFF);
    uint32_t w[AFF];
#pragma unroll
    for (int i = 0; i < AFF / 4; ++i) {
        uint4 v = p[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    Affine29<F> a;
    load_coord(a.x, w);
    load_coord(a.y, w + AFF / 2);
    F ny = normalize(sub<2, 1>(F::zero(), a.y));   // 2N - y
    if (negate) a.y = ny;
    return a;
}

// The bucket accumulation's form: a negated y is left as 2N − y limb-wise (limbs below 2^30, value below 2N).  Its only
// uses in madd29 are one product against a normalised operand (2^30·2^29 stays inside the column bound) and the
// start of a run, which normalises it; the eight-step carry chain per entry is saved.  Fq only: an Fq2 product wants
// its smaller operand normalised (field29.hpp).
CG_HD Affine29<Fq29> load_table_point_lazy_y(const uint32_t* __restrict__ table, uint32_t idx, bool negate) {
    constexpr int AFF = Words29<Fq29>::AFF;
    const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)idx * AFF);
    uint32_t w[AFF];
#pragma unroll
    for (int i = 0; i < AFF / 4; ++i) {
        uint4 v = p[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    Affine29<Fq29> a;
    load_coord(a.x, w);
    load_coord(a.y, w + AFF / 2);
    Fq29 ny = sub<2, 1>(Fq29::zero(), a.y);
    if (negate) a.y = ny;
    return a;
}

// the signed accumulation's form: both coordinates as stored (the digit's sign is applied inside madd29s, as a multiplier)
CG_HD Affine29<Fq29> load_table_point_plain(const uint32_t* __restrict__ table, uint32_t idx) {
    constexpr int AFF = Words29<Fq29>::AFF;
    const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)idx * AFF);
    uint32_t w[AFF];
#pragma unroll
    for (int i = 0; i < AFF / 4; ++i) {
        uint4 v = p[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    Affine29<Fq29> a;
    load_coord(a.x, w);
    load_coord(a.y, w + AFF / 2);
    return a;
}

// the same in two steps, so that the (random, HBM-latency) read can be issued an iteration early
template <class F>
struct RawPoint29 {
    uint4 q[Words29<F>::AFF / 4];
};
template <class F>
CG_HD RawPoint29<F> load_raw_point(const uint32_t* __restrict__ table, uint32_t idx) {
    constexpr int AFF = Words29<F>::AFF;
    const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)idx * AFF);
    RawPoint29<F> r;
#pragma unroll
    for (int i = 0; i < AFF / 4; ++i) r.q[i] = p[i];
    return r;
}
template <class F>
CG_HD Affine29<F> unpack_point(const RawPoint29<F>& r, bool negate) {
    constexpr int AFF = Words29<F>::AFF;
    uint32_t w[AFF];
#pragma unroll
    for (int i = 0; i < AFF / 4; ++i) {
        w[4 * i] = r.q[i].x; w[4 * i + 1] = r.q[i].y; w[4 * i + 2] = r.q[i].z; w[4 * i + 3] = r.q[i].w;
    }
    Affine29<F> a;
    load_coord(a.x, w);
    load_coord(a.y, w + AFF / 2);
    F ny = normalize(sub<2, 1>(F::zero(), a.y));
    if (negate) a.y = ny;
    return a;
}

// accumulator <-> memory (ACC words; identity = zz all zero)
template <class F>
CG_HD void store_acc(uint32_t* __restrict__ dst, const XYZZ29<F>& a, bool inf) {
    constexpr int ACC = Words29<F>::ACC;
    uint4* p = reinterpret_cast<uint4*>(dst);
    if (inf) {       // a branch of its own (rare), not a select on every word of the common case
#pragma unroll
        for (int i = 0; i < ACC / 4; ++i) p[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    uint32_t w[ACC];
    store_limbs(a.x, w);
    store_limbs(a.y, w + ACC / 4);
    store_limbs(a.zz, w + ACC / 2);
    store_limbs(a.zzz, w + 3 * ACC / 4);
#pragma unroll
    for (int i = 0; i < ACC / 4; ++i) p[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// A G1 record may be in the SIGNED form the bucket accumulation leaves (store_acc_signed below): bit 31 of the top ZZ word
// set, bit 30 = the sign of Y.  It is brought to the stored invariant here, by whoever reads it: the conversion (~160
// instructions) is paid once per record by kernels whose additions cost 3000, instead of inside the accumulation's loop,
// where a flush runs for one or two lanes of a wave in 71 % of the iterations and every instruction of it costs the wave.
CG_HD void signed_record_to_stored(uint32_t* w);          // defined after G1AccS
template <class F>
CG_HD bool load_acc(const uint32_t* __restrict__ src, XYZZ29<F>& a) {   // returns inf
    constexpr int ACC = Words29<F>::ACC;
    uint32_t w[ACC];
    const uint4* p = reinterpret_cast<const uint4*>(src);
#pragma unroll
    for (int i = 0; i < ACC / 4; ++i) {
        uint4 v = p[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    if constexpr (Words29<F>::NF == 1) {
        if (w[ACC / 2 + 8] & 0x80000000u) signed_record_to_stored(w);
    }
    load_limbs(a.x, w);
    load_limbs(a.y, w + ACC / 4);
    load_limbs(a.zz, w + ACC / 2);
    load_limbs(a.zzz, w + 3 * ACC / 4);
    return a.zz.all_zero();
}
#endif  // __HIPCC__

// ---- group law -------------------------------------------------------------------------------------
// 2·(affine p)  -> XYZZ           (mdbl-2008-s-1)
template <class F>
CG_HD XYZZ29<F> dbl_affine29(const Affine29<F>& p) {
    F U = normalize(dbl(p.y));
    F V = sqr(U);
    F W = mul(U, V);
    F S = mul(p.x, V);
    F X2 = sqr(p.x);
    F M = normalize(add(dbl(X2), X2));
    F X3 = normalize(sub<K2, 2>(sqr(M), dbl(S)));
    F d = normalize(sub<KX, 1>(S, X3));
    F Y3 = normalize(sub<K1, 1>(mul(d, M), mul(p.y, W)));
    return {X3, Y3, V, W};
}
// 2·a (a not the identity)        (dbl-2008-s-1, curve a = 0)
template <class F>
CG_HD XYZZ29<F> dbl29(const XYZZ29<F>& a) {
    F U = normalize(dbl(a.y));
    F V = sqr(U);
    F W = mul(U, V);
    F S = mul(a.x, V);
    F X2 = sqr(a.x);
    F M = normalize(add(dbl(X2), X2));
    F X3 = normalize(sub<K2, 2>(sqr(M), dbl(S)));
    F d = normalize(sub<KX, 1>(S, X3));
    F Y3 = normalize(sub<K1, 1>(mul(d, M), mul(a.y, W)));
    return {X3, Y3, mul(a.zz, V), mul(a.zzz, W)};
}

// a·b − c·d.  Over Fq one dual-product reduction, (a·b + (KY·N − c)·d)/R' (c normalised, value < (KY−1)·N — a stored Y):
// one Montgomery reduction and one subtraction of reduced values fewer than two products, and a result below 2.2 N.
CG_HD Fq29 mul_sub(const Fq29& a, const Fq29& b, const Fq29& c, const Fq29& d) {
    return mul2_core(a, b, sub<KY, 1>(Fq29::zero(), c), d);
}
// Over Fq2 each component is ONE reduction over four products:
//   c0 = a0 b0 − a1 b1 − c0 d0 + c1 d1,   c1 = a0 b1 + a1 b0 − c0 d1 − c1 d0
// with the subtracted factors negated limb-wise first (b: normalised, value < (FQ2_NEGK−1)·N; c: a stored Y, normalised,
// value < (KY−1)·N).  All of a, b, c, d normalised: the four limb-bound products sum to 1 + 2 + 2 + 1 = 6 units.
CG_HD Fq2_29 mul_sub(const Fq2_29& a, const Fq2_29& b, const Fq2_29& c, const Fq2_29& d) {
    const Fq29 nb1 = sub<FQ2_NEGK, 1>(Fq29::zero(), b.c1);
    const Fq29 nc0 = sub<KY, 1>(Fq29::zero(), c.c0), nc1 = sub<KY, 1>(Fq29::zero(), c.c1);
    return {mul4_core(a.c0, b.c0, a.c1, nb1, nc0, d.c0, c.c1, d.c1), mul4_core(a.c0, b.c1, a.c1, b.c0, nc0, d.c1, nc1, d.c0)};
}
// the first operand of mul_sub as it may be handed over: Fq's dual product takes limbs up to 3·2^29 against a
// normalised partner (9·(3 + 2)·2^58 stays below 2^64, tools/bounds29.py), so the carry chain is skipped; an Fq2
// product wants normalised operands
CG_HD Fq29 for_mul_sub(const Fq29& a) { return a; }
CG_HD Fq2_29 for_mul_sub(const Fq2_29& a) { return normalize(a); }

// acc += p (p affine, never the identity)          (madd-2008-s)
// Statement order keeps at most seven field values live (an Fq2 value is 18 VGPRs).
// In every Fq2 product the SECOND operand is the one that is normalised with the smaller bound.
template <class F>
CG_HD void madd29(XYZZ29<F>& acc, bool& inf, const Affine29<F>& p) {
    if (inf) {
        acc.x = p.x; acc.y = normalize(p.y); acc.zz = F::one(); acc.zzz = F::one();   // p.y may be a lazy 2N − y
        inf = false;
        return;
    }
    F P = normalize(sub<KX, 1>(mul(acc.zz, p.x), acc.x));       // U2 - X1
    F PP = sqr_loose(P);
    F ZZ3 = mul(acc.zz, PP);
    if (is_zero_mod(ZZ3)) {                      // P ≡ 0: same x.  Rare (repeated base / s and r-s).
        F R0 = normalize(sub<KY, 1>(mul(acc.zzz, p.y), acc.y));
        if (is_zero_mod(canonical(R0))) acc = dbl_affine29(p);
        else inf = true;
        return;
    }
    F Q = mul(acc.x, PP);
    F PPP = mul(P, PP);
    F R = normalize(sub<KY, 1>(mul(acc.zzz, p.y), acc.y));      // S2 - Y1
    F ZZZ3 = mul(acc.zzz, PPP);
    F X3 = normalize(sub<K2, 2>(sub<K1, 1>(sqr(R), PPP), dbl(Q)));
    F d = for_mul_sub(sub<KX, 1>(Q, X3));
    acc.y = mul_sub(d, R, acc.y, PPP);                          // R·(Q − X3) − Y1·PPP
    acc.x = X3;
    acc.zz = ZZ3;
    acc.zzz = ZZZ3;
}

// ---- the same mixed addition on SIGNED limbs (field29.hpp S29): the G1 bucket accumulation's inner loop -----------------
// The running accumulator of a lane stays in this form from one entry to the next and is converted to the stored
// (unsigned) invariant only when a run is flushed (acc_to_stored).  It carries
//     x, sy, zz, zzz  s-normalised,   t = +-1   with   Y = t·sy,
// and every subtraction of madd-2008-s is either fused into the product that precedes it or a limb-wise difference:
//     u   = sigma·t                                  (sigma = -1 for a negated table point)
//     P   = zz·px - x                                 fused                                   U2 - X1
//     R*  = zzz·py - u·sy                             fused, the per-lane u as the multiplier  = sigma·(S2 - Y1)
//     X3  = R*² - PPP - 2Q                            fused
//     sy3 = R*·(u·(X3 - Q)) + sy·PPP,  t3 = -t        Y3 = R·(Q - X3) - Y1·PPP = -t·sy3: the sign flips instead of a negation
// Bounds (tools/bounds29.py, "signed accumulator"): 0 <= zz < 1.05 N, -0.01 N < zzz < 1.01 N, -3.5 N < x < 1.2 N, |sy| < 1.2 N;
// P in (-1.3 N, 4.6 N), |R*| < 2.3 N, PP < 1.2 N, |PPP| < 1.1 N, |Q| < 1.1 N; every multiplicand limb is below 2^29 in
// magnitude except the limb-wise difference X3 - Q (below 2^29 as well: both are s-normalised).
struct G1AccS {
    Fq29s x, sy, zz, zzz;
    int32_t t;
};
// neg1, neg2: the constants -1 and -2 held OPAQUE (a kernel keeps them in SGPRs the optimiser cannot see through), so that
// "c -= x" stays the one v_mad_i64_i32 it is written as instead of becoming a two-instruction 64-bit subtraction
CG_HD void madd29s(G1AccS& acc, bool& inf, const Affine29<Fq29>& p, int32_t sigma, int32_t neg1, int32_t neg2) {
    typedef Fq29P P;
    const Fq29s px = Fq29s::from_unsigned(p.x), py = Fq29s::from_unsigned(p.y);
    if (inf) {
        acc.x = px; acc.sy = py; acc.zz = Fq29s::one(); acc.zzz = Fq29s::one(); acc.t = sigma;
        inf = false;
        return;
    }
    const int32_t u = sigma * acc.t;
    const Fq29s Pd = mul_s(acc.zz, px, FuseMul1<P>{acc.x, neg1});
    const Fq29s PP = sqr_s(Pd);
    const Fq29s ZZ3 = mul_s(acc.zz, PP);
    if (is_zero_mod(ZZ3)) {                      // P ≡ 0: same x.  Rare (repeated base / s and r-s).
        const Fq29s R0 = mul_s(acc.zzz, py, FuseMul1<P>{acc.sy, -u});
        if (is_zero_mod(canonical(to_unsigned<4>(R0)))) {       // same point: double it (the unsigned formulas; rare)
            Affine29<Fq29> q = p;
            if (sigma < 0) q.y = normalize(sub<2, 1>(Fq29::zero(), p.y));
            const XYZZ29<Fq29> d2 = dbl_affine29(q);
            // made canonical (below N): inside the signed accumulator's intervals whatever the unsigned formulas' bounds are
            acc.x = Fq29s::from_unsigned(canonical(d2.x)); acc.sy = Fq29s::from_unsigned(canonical(d2.y));
            acc.zz = Fq29s::from_unsigned(canonical(d2.zz)); acc.zzz = Fq29s::from_unsigned(canonical(d2.zzz)); acc.t = 1;
        } else {
            inf = true;
        }
        return;
    }
    const Fq29s Q = mul_s(acc.x, PP);
    const Fq29s PPP = mul_s(Pd, PP);
    const Fq29s Rs = mul_s(acc.zzz, py, FuseMul1<P>{acc.sy, -u});
    const Fq29s ZZZ3 = mul_s(acc.zzz, PPP);
    const Fq29s X3 = sqr_s(Rs, FuseMul2<P>{PPP, neg1, Q, neg2});
    // d = u·(X3 - Q) limb-wise.  With um = 0 / -1 for u = +1 / -1:  u·(a - b) = (a ^ um) + (b ^ ~um) + 1  - two v_xad_u32 per
    // limb (a 32-bit multiplication by u would be a quarter-rate v_mul_lo)
    const uint32_t um = (uint32_t)(u >> 31), un = ~um;
    Fq29s d;
#pragma unroll
    for (int i = 0; i < 9; ++i) d.l[i] = (int32_t)(((uint32_t)X3.l[i] ^ um) + (((uint32_t)Q.l[i] ^ un) + 1u));
    acc.sy = mul2_s(Rs, d, acc.sy, PPP);
    acc.t = -acc.t;
    acc.x = X3;
    acc.zz = ZZ3;
    acc.zzz = ZZZ3;
}
// the signed running accumulator -> the stored invariant (X < 13 N, Y < 8 N, ZZ, ZZZ < 3 N, normalised, non-negative)
CG_HD XYZZ29<Fq29> acc_to_stored(const G1AccS& a) {
    XYZZ29<Fq29> r;
    r.x = to_unsigned<4>(a.x);                       // (-3.5 N, 1.2 N) + 4 N
    Fq29s y;
    const int32_t tm = a.t >> 31;                    // 0 / -1
#pragma unroll
    for (int i = 0; i < 9; ++i) y.l[i] = (a.sy.l[i] ^ tm) - tm;      // t·sy limb-wise
    r.y = to_unsigned<2>(y);                         // (-1.2 N, 1.2 N) + 2 N
    r.zz = to_unsigned<0>(a.zz);                     // non-negative as it is
    r.zzz = to_unsigned<1>(a.zzz);                   // (-0.01 N, 1.01 N) + N: a product's negative side is a hundredth of N
    return r;
}

// The accumulation's record: the signed accumulator as it is (36 words: x, sy, zz, zzz), the sign of Y in bit 30 and the
// "signed record" mark in bit 31 of the top ZZ word (zz is non-negative and below 1.05 N: its top word is below 2^23).
CG_HD void pack_signed_record(const G1AccS& a, uint32_t* w) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        w[i] = (uint32_t)a.x.l[i]; w[9 + i] = (uint32_t)a.sy.l[i]; w[18 + i] = (uint32_t)a.zz.l[i]; w[27 + i] = (uint32_t)a.zzz.l[i];
    }
    w[26] |= 0x80000000u | (((uint32_t)a.t >> 31) << 30);
}
#if defined(__HIPCC__)
// The flush of the accumulation's loop: it runs for one or two lanes of a wave in 71 % of the iterations (equal segments
// over runs of ~52 entries), so the WAVE pays every vector-ALU instruction of it 0.71 times per entry.  The record leaves as
// nine 16-byte stores.  (Round 5 measured two other ways out - 36 four-byte stores straight from the registers, and through a
// 144-byte LDS slot per wave; neither won, profiles/r05_d_flush_variants.txt, and round 6 took them out of the sources.)
CG_HD void store_acc_signed(uint32_t* __restrict__ dst, const G1AccS& a, bool inf) {
    uint4* p = reinterpret_cast<uint4*>(dst);
    if (inf) {
#pragma unroll
        for (int i = 0; i < 9; ++i) p[i] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    uint32_t w[36];
    pack_signed_record(a, w);
#pragma unroll
    for (int i = 0; i < 9; ++i) p[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
#endif
CG_HD void signed_record_to_stored(uint32_t* w) {
    G1AccS a;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        a.x.l[i] = (int32_t)w[i]; a.sy.l[i] = (int32_t)w[9 + i]; a.zz.l[i] = (int32_t)w[18 + i]; a.zzz.l[i] = (int32_t)w[27 + i];
    }
    a.t = (w[26] & 0x40000000u) ? -1 : 1;
    a.zz.l[8] = (int32_t)(w[26] & 0x3fffffffu);
    const XYZZ29<Fq29> r = acc_to_stored(a);
    store_limbs(r.x, w);
    store_limbs(r.y, w + 9);
    store_limbs(r.zz, w + 18);
    store_limbs(r.zzz, w + 27);
}

// acc += q (both XYZZ under the stored invariant)   (add-2008-s)
template <class F>
CG_HD void add29(XYZZ29<F>& acc, bool& inf, const XYZZ29<F>& q, bool qinf) {
    if (qinf) return;
    if (inf) {
        acc = q;
        inf = false;
        return;
    }
    F U1 = mul(acc.x, q.zz);
    F P = normalize(sub<K1, 1>(mul(q.x, acc.zz), U1));          // U2 - U1
    F PP = sqr_loose(P);
    F ZZ3 = mul(mul(acc.zz, q.zz), PP);
    if (is_zero_mod(ZZ3)) {
        F R0 = normalize(sub<K1, 1>(mul(q.y, acc.zzz), mul(acc.y, q.zzz)));
        if (is_zero_mod(canonical(R0))) acc = dbl29(acc);
        else inf = true;
        return;
    }
    F Q = mul(U1, PP);
    F PPP = mul(P, PP);
    F S1 = mul(acc.y, q.zzz);
    F R = normalize(sub<K1, 1>(mul(q.y, acc.zzz), S1));         // S2 - S1
    F ZZZ3 = mul(mul(acc.zzz, q.zzz), PPP);
    F X3 = normalize(sub<K2, 2>(sub<K1, 1>(sqr(R), PPP), dbl(Q)));
    F d = for_mul_sub(sub<KX, 1>(Q, X3));
    acc.y = mul_sub(d, R, S1, PPP);                             // R·(Q − X3) − S1·PPP (S1 a product: normalised, below 2N)
    acc.x = X3;
    acc.zz = ZZ3;
    acc.zzz = ZZZ3;
}

// ---- inversion / affine normalisation (load-time and result paths only) -----------------------------------
// a^(N-2), a normalised with value < 16 N
template <class P>
CG_HD F29<P> inv29(const F29<P>& a) {
    typedef typename P::P256 P8;
    F29<P> r = F29<P>::one();
    bool started = false;
    for (int i = 7; i >= 0; --i) {
        uint32_t e = P8::N[i] - (i == 0 ? 2u : 0u);   // N - 2: no borrow for either modulus
        for (int b = 31; b >= 0; --b) {
            if (started) r = sqr(r);
            if ((e >> b) & 1u) { r = started ? mul(r, a) : a; started = true; }
        }
    }
    return r;
}
CG_HD Fq2_29 inv29(const Fq2_29& a) {
    Fq29 n = inv29(normalize(add(sqr(a.c0), sqr(a.c1))));
    return {mul(a.c0, n), normalize(sub<3, 1>(Fq29::zero(), mul(a.c1, n)))};
}
CG_HD void store_packed_coord(const Fq29& c, uint32_t* w) { pack29(canonical(c), w); }
CG_HD void store_packed_coord(const Fq2_29& c, uint32_t* w) { pack29(canonical(c.c0), w); pack29(canonical(c.c1), w + 8); }
// XYZZ (not the identity) -> packed affine table point
template <class F>
CG_HD void store_table_point_from_xyzz(const XYZZ29<F>& a, uint32_t* w) {
    constexpr int AFF = Words29<F>::AFF;
    F t = inv29(mul(a.zz, a.zzz));
    F izz = mul(a.zzz, t);
    F izzz = mul(a.zz, t);
    store_packed_coord(mul(a.x, izz), w);
    store_packed_coord(mul(a.y, izzz), w + AFF / 2);
}

using G1Affine29 = Affine29<Fq29>;
using G2Affine29 = Affine29<Fq2_29>;
using G1XYZZ29 = XYZZ29<Fq29>;
using G2XYZZ29 = XYZZ29<Fq2_29>;

// which 29-bit field carries which saturated field
template <class F> struct To29;
template <> struct To29<Fq> { typedef Fq29 type; };
template <> struct To29<Fq2> { typedef Fq2_29 type; };

// ---- conversions with the saturated (R = 2^256) forms used by the host and the load-time kernels ---------
// Montgomery(2^256) affine point -> packed table point (canonical x·R' mod N)
CG_HD void pack_table_coord(const Fq& c, uint32_t* w) {
    Fq29 t = from_mont256<Fq29P>(c);
    pack29(t, w);
}
CG_HD void pack_table_point(const Affine<Fq>& p, uint32_t* w) { pack_table_coord(p.x, w); pack_table_coord(p.y, w + 8); }
CG_HD void pack_table_point(const Affine<Fq2>& p, uint32_t* w) {
    pack_table_coord(p.x.c0, w); pack_table_coord(p.x.c1, w + 8);
    pack_table_coord(p.y.c0, w + 16); pack_table_coord(p.y.c1, w + 24);
}

}  // namespace cg
