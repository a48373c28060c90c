// G2 bucket accumulation with every Fq2 value SPLIT OVER A LANE PAIR: lanes 2i and 2i+1 of a wave work on one mixed
// addition together, the even lane holding the c0 component of every Fq2 value and the odd lane the c1 component.
//
// Why.  With one lane per addition an Fq2 XYZZ accumulator is 72 VGPRs and a mixed addition's temporaries push the
// kernel to ~180 registers even with the accumulator parked in LDS (k_accum_affine_g2: two waves per SIMD, 72 KB of LDS
// per block, 46 % of its instruction floor stand-alone - two waves cannot hide the dependent-issue latency of the
// multiply-add chains).  Split over a pair, a lane carries nine limbs per value exactly as in the G1 kernel: the
// accumulator lives in registers, no LDS, and the kernel fits three to four waves per SIMD.
//
// How.  Fq2 = Fq[u]/(u^2 + 1): (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u.  Both components are
// DUAL products over Fq (field29.hpp mul2), so the two lanes run the SAME instruction stream on different operands:
//     even lane:  c0 = a0·b0 + a1·(K N - b1)          odd lane:  c1 = a1·b0 + a0·b1
// i.e. mul2(mine_a, y0, other_a, y1) with (y0, y1) = (mine_b, -other_b) on the even lane and (other_b, mine_b) on the
// odd one.  "other" comes from the partner lane through a DPP move (quad_perm [1,0,3,2]: full rate, no LDS), the
// operand choice is a bit-select on a lane mask.  Per Fq2 product that is one dual product per lane (what the one-lane
// form spends per COMPONENT) plus 18 moves, 18 selects and 9 subtractions: ~+10 % instructions per addition, bought back
// by the occupancy.  Sums, differences, doublings and carry propagations act on the lane's own nine limbs.
// The formulas, their statement order and every bound (tools/bounds29.py, curve29.hpp) are those of the one-lane form:
// the same products are formed from the same operands, only on another lane.
#pragma once
#include "curve29.hpp"

namespace cg {
#if defined(__HIPCC__)

// the partner lane's copy of a nine-limb value (both lanes of a pair must be active: they share all control flow)
__device__ __forceinline__ Fq29 pr_swap(const Fq29& a) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.l[i], 0xB1, 0xF, 0xF, true);
    return r;
}
__device__ __forceinline__ uint32_t pr_swap(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }
// odd = all ones on the odd lane of a pair, zero on the even lane
__device__ __forceinline__ Fq29 pr_sel(uint32_t odd, const Fq29& if_odd, const Fq29& if_even) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (if_odd.l[i] & odd) | (if_even.l[i] & ~odd);
    return r;
}

// this lane's component of a·b.  b: normalised, value < (K - 1)·N (K = FQ2_NEGK for a general product, FQ2_KS for the
// tight square a·a); pb = pr_swap(b), handed in because several products share a second operand.
template <int K>
__device__ __forceinline__ Fq29 pr_mul(const Fq29& a, const Fq29& b, const Fq29& pb, uint32_t odd) {
    const Fq29 pa = pr_swap(a);
    const Fq29 npb = sub<K, 1>(Fq29::zero(), pb);
    return mul2(a, pr_sel(odd, pb, b), pa, pr_sel(odd, b, npb));
}
// this lane's component of a^2 from ONE single product per lane: c0 = (a0 + a1)(a0 + KS·N - a1), c1 = (2 a0)·a1
// (field29.hpp sqr_loose: for squares that only feed further products).  a normalised, value < (FQ2_KS - 1)·N.
__device__ __forceinline__ Fq29 pr_sqr_loose(const Fq29& a, uint32_t odd) {
    const Fq29 pa = pr_swap(a);
    return mul(pr_sel(odd, dbl(pa), add(a, pa)), pr_sel(odd, a, sub<FQ2_KS, 1>(a, pa)));
}
// this lane's component of a·b - c·d, one reduction over four products (curve29.hpp mul_sub over Fq2):
//   c0 = a0 b0 - a1 b1 - c0 d0 + c1 d1,   c1 = a0 b1 + a1 b0 - c0 d1 - c1 d0
// b normalised with value < (FQ2_NEGK - 1)·N, c a stored Y (normalised, value < (KY - 1)·N), a and d normalised.
__device__ __forceinline__ Fq29 pr_mul_sub(const Fq29& a, const Fq29& b, const Fq29& c, const Fq29& d, uint32_t odd) {
    const Fq29 pa = pr_swap(a), pb = pr_swap(b), pc = pr_swap(c), pd = pr_swap(d);
    const Fq29 npb = sub<FQ2_NEGK, 1>(Fq29::zero(), pb);
    const Fq29 nc = sub<KY, 1>(Fq29::zero(), c), npc = sub<KY, 1>(Fq29::zero(), pc);
    //            even lane                    odd lane
    // term 1:    a0 ·  b0                     a1 ·  b0
    // term 2:    a1 · (-b1)                   a0 ·  b1
    // term 3:  (-c0)·  d0                   (-c1)·  d0
    // term 4:    c1 ·  d1                   (-c0)·  d1
    return mul4_core(a, pr_sel(odd, pb, b), pa, pr_sel(odd, b, npb), nc, pr_sel(odd, pd, d), pr_sel(odd, npc, pc), pr_sel(odd, d, pd));
}

struct PairAcc {      // this lane's components of an XYZZ accumulator over Fq2
    Fq29 x, y, zz, zzz;
};
// is the Fq2 value whose component this lane holds ≡ 0?  (a: a product output, or otherwise normalised with value < 2N)
__device__ __forceinline__ bool pr_is_zero(const Fq29& a) {
    const uint32_t mine = is_zero_mod(a) ? 1u : 0u;
    return (mine & pr_swap(mine)) != 0u;
}

// 2·(px, py) -> XYZZ        (mdbl-2008-s-1, statement for statement curve29.hpp dbl_affine29 over Fq2)
__device__ __forceinline__ PairAcc pr_dbl_affine(const Fq29& px, const Fq29& py, uint32_t odd) {
    const Fq29 U = normalize(dbl(py));
    const Fq29 V = pr_mul<FQ2_KS>(U, U, pr_swap(U), odd);
    const Fq29 pV = pr_swap(V);
    const Fq29 W = pr_mul<FQ2_NEGK>(U, V, pV, odd);
    const Fq29 S = pr_mul<FQ2_NEGK>(px, V, pV, odd);
    const Fq29 X2 = pr_mul<FQ2_KS>(px, px, pr_swap(px), odd);
    const Fq29 M = normalize(add(dbl(X2), X2));
    const Fq29 X3 = normalize(sub<K2, 2>(pr_mul<FQ2_KS>(M, M, pr_swap(M), odd), dbl(S)));
    const Fq29 d = normalize(sub<KX, 1>(S, X3));
    const Fq29 Y3 = normalize(sub<K1, 1>(pr_mul<FQ2_NEGK>(d, M, pr_swap(M), odd), pr_mul<FQ2_NEGK>(py, W, pr_swap(W), odd)));
    return {X3, Y3, V, W};
}

// this lane's component (half = 0: c0, 1: c1) of coordinate `which` (0 = x, 1 = y) of table point idx; y negated when
// `negate`.  The two coordinates are fetched where the formula first needs them (y late: nine registers fewer across the
// first four products; the second fetch hits the cache line the first one brought in).
__device__ __forceinline__ Fq29 pr_load_coord(const uint32_t* __restrict__ table, uint32_t idx, uint32_t half, int which, bool negate) {
    const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)idx * 32 + which * 16 + half * 8);
    const uint4 q0 = p[0], q1 = p[1];
    const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    Fq29 c = unpack29<Fq29P>(w);
    if (which == 1) {
        const Fq29 nc = normalize(sub<2, 1>(Fq29::zero(), c));
        if (negate) c = nc;
    }
    return c;
}
// acc += table point idx (its y negated when `negate`)   (madd-2008-s in the statement order of curve29.hpp madd29)
__device__ __forceinline__ void pr_madd(PairAcc& acc, bool& inf, const uint32_t* __restrict__ table, uint32_t idx, uint32_t half, bool negate,
                                        uint32_t odd) {
    const Fq29 px = pr_load_coord(table, idx, half, 0, false);
    if (inf) {
        const Fq29 one = pr_sel(odd, Fq29::zero(), Fq29::one());       // 1 + 0·u
        acc.x = px; acc.y = pr_load_coord(table, idx, half, 1, negate); acc.zz = one; acc.zzz = one;
        inf = false;
        return;
    }
    const Fq29 P = normalize(sub<KX, 1>(pr_mul<FQ2_NEGK>(acc.zz, px, pr_swap(px), odd), acc.x));      // U2 - X1
    const Fq29 PP = pr_sqr_loose(P, odd);
    const Fq29 pPP = pr_swap(PP);
    const Fq29 ZZ3 = pr_mul<FQ2_NEGK>(acc.zz, PP, pPP, odd);
    {   // same x: doubling or cancellation (rare); one compare per lane filters it out
        const uint32_t maybe = maybe_zero_mod(ZZ3) ? 1u : 0u;
        if ((maybe & pr_swap(maybe)) != 0u && pr_is_zero(ZZ3)) {
            const Fq29 py = pr_load_coord(table, idx, half, 1, negate);
            const Fq29 R0 = normalize(sub<KY, 1>(pr_mul<FQ2_NEGK>(acc.zzz, py, pr_swap(py), odd), acc.y));
            if (pr_is_zero(canonical(R0))) acc = pr_dbl_affine(px, py, odd);
            else inf = true;
            return;
        }
    }
    const Fq29 Q = pr_mul<FQ2_NEGK>(acc.x, PP, pPP, odd);
    const Fq29 PPP = pr_mul<FQ2_NEGK>(P, PP, pPP, odd);
    const Fq29 py = pr_load_coord(table, idx, half, 1, negate);
    const Fq29 R = normalize(sub<KY, 1>(pr_mul<FQ2_NEGK>(acc.zzz, py, pr_swap(py), odd), acc.y));      // S2 - Y1
    const Fq29 ZZZ3 = pr_mul<FQ2_NEGK>(acc.zzz, PPP, pr_swap(PPP), odd);
    const Fq29 X3 = normalize(sub<K2, 2>(sub<K1, 1>(pr_mul<FQ2_KS>(R, R, pr_swap(R), odd), PPP), dbl(Q)));
    const Fq29 d = normalize(sub<KX, 1>(Q, X3));
    acc.y = pr_mul_sub(d, R, acc.y, PPP, odd);                            // R·(Q - X3) - Y1·PPP
    acc.x = X3;
    acc.zz = ZZ3;
    acc.zzz = ZZZ3;
}

// the stored form of an accumulator is the one-lane kernels' (ACC = 72 words: x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0
// zzz.c1, nine limbs each; identity = all zero): every lane writes its four nine-limb components
__device__ __forceinline__ void pr_store_acc(uint32_t* __restrict__ dst, const PairAcc& a, bool inf, uint32_t half) {
    uint32_t* d = dst + half * 9;
    const uint32_t keep = inf ? 0u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        d[i] = a.x.l[i] & keep;
        d[18 + i] = a.y.l[i] & keep;
        d[36 + i] = a.zz.l[i] & keep;
        d[54 + i] = a.zzz.l[i] & keep;
    }
}

#endif  // __HIPCC__
}  // namespace cg
