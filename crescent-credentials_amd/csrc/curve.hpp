// Short-Weierstrass (a = 0) group arithmetic for BN254 G1 (over Fq) and G2 (over Fq2), generic over
// the coordinate field.  Accumulators use extended-Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): a mixed add costs 8M + 2S against 7M + 4S for plain Jacobian
// and keeps no Z to square, which is what the bucket accumulation of
// ark-ec's VariableBaseMSM::msm_bigint (call sites forks/groth16/src/prover.rs:66,74,266) spends
// nearly all of its time in.  The sums these produce are the same group elements as arkworks'
// projective results; only the final affine normalisation (prover.rs:131-135) is observable.
#pragma once
#include "field.hpp"

namespace cg {

// Affine point; the all-zero encoding is the point at infinity ((0,0) is on neither curve).
template <class F>
struct alignas(16) Affine {
    F x, y;
    CG_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    CG_HD static Affine inf() { return {F::zero(), F::zero()}; }
};

template <class F>
struct alignas(16) XYZZ {
    F x, y, zz, zzz;
    CG_HD bool is_inf() const { return zz.is_zero(); }
    CG_HD static XYZZ inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
    CG_HD static XYZZ from_affine(const Affine<F>& p) {
        if (p.is_inf()) return inf();
        return {p.x, p.y, F::one(), F::one()};
    }
};

template <class F>
CG_HD Affine<F> neg(const Affine<F>& p) {
    return {p.x, neg(p.y)};  // neg(0) == 0 keeps the infinity encoding
}
template <class F>
CG_HD XYZZ<F> neg(const XYZZ<F>& p) {
    return {p.x, neg(p.y), p.zz, p.zzz};
}

// 2 * (affine p), p != inf           (EFD mdbl-2008-s-1)
template <class F>
CG_HD XYZZ<F> dbl_affine(const Affine<F>& p) {
    F U = dbl(p.y);
    F V = sqr(U);
    F W = mul(U, V);
    F S = mul(p.x, V);
    F X2 = sqr(p.x);
    F M = add(dbl(X2), X2);
    F X3 = sub(sqr(M), dbl(S));
    F Y3 = sub(mul(M, sub(S, X3)), mul(W, p.y));
    return {X3, Y3, V, W};
}

// 2 * p                                (EFD dbl-2008-s-1, a = 0)
template <class F>
CG_HD XYZZ<F> dbl(const XYZZ<F>& p) {
    if (p.is_inf()) return p;
    F U = dbl(p.y);
    F V = sqr(U);
    F W = mul(U, V);
    F S = mul(p.x, V);
    F X2 = sqr(p.x);
    F M = add(dbl(X2), X2);
    F X3 = sub(sqr(M), dbl(S));
    F Y3 = sub(mul(M, sub(S, X3)), mul(W, p.y));
    return {X3, Y3, mul(V, p.zz), mul(W, p.zzz)};
}

// acc += p (affine, may be inf)         (EFD madd-2008-s)
template <class F>
CG_HD void madd(XYZZ<F>& acc, const Affine<F>& p) {
    if (p.is_inf()) return;
    if (acc.is_inf()) {
        acc = {p.x, p.y, F::one(), F::one()};
        return;
    }
    F U2 = mul(p.x, acc.zz);
    F S2 = mul(p.y, acc.zzz);
    F P = sub(U2, acc.x);
    F R = sub(S2, acc.y);
    if (P.is_zero()) {
        if (R.is_zero()) acc = dbl_affine(p);
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = sqr(P);
    F PPP = mul(P, PP);
    F Q = mul(acc.x, PP);
    F X3 = sub(sub(sqr(R), PPP), dbl(Q));
    F Y3 = sub(mul(R, sub(Q, X3)), mul(acc.y, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = mul(acc.zz, PP);
    acc.zzz = mul(acc.zzz, PPP);
}

// acc += q (XYZZ)                       (EFD add-2008-s)
template <class F>
CG_HD void add(XYZZ<F>& acc, const XYZZ<F>& q) {
    if (q.is_inf()) return;
    if (acc.is_inf()) {
        acc = q;
        return;
    }
    F U1 = mul(acc.x, q.zz);
    F U2 = mul(q.x, acc.zz);
    F S1 = mul(acc.y, q.zzz);
    F S2 = mul(q.y, acc.zzz);
    F P = sub(U2, U1);
    F R = sub(S2, S1);
    if (P.is_zero()) {
        if (R.is_zero()) acc = dbl(acc);
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = sqr(P);
    F PPP = mul(P, PP);
    F Q = mul(U1, PP);
    F X3 = sub(sub(sqr(R), PPP), dbl(Q));
    F Y3 = sub(mul(R, sub(Q, X3)), mul(S1, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = mul(mul(acc.zz, q.zz), PP);
    acc.zzz = mul(mul(acc.zzz, q.zzz), PPP);
}

template <class F>
CG_HD Affine<F> to_affine(const XYZZ<F>& p) {
    if (p.is_inf()) return Affine<F>::inf();
    // 1/zzz, then 1/zz = zzz^-2 * zz^2 ... cheaper: one inversion of zz*zzz
    F t = inv(mul(p.zz, p.zzz));
    F izz = mul(t, p.zzz);
    F izzz = mul(t, p.zz);
    return {mul(p.x, izz), mul(p.y, izzz)};
}

// k * p for a little-endian 256-bit integer k (8 x u32), MSB-first double-and-add.
template <class F>
CG_HD XYZZ<F> scalar_mul(const XYZZ<F>& p, const uint32_t k[8]) {
    XYZZ<F> acc = XYZZ<F>::inf();
    bool started = false;
    for (int i = 7; i >= 0; --i)
        for (int b = 31; b >= 0; --b) {
            if (started) acc = dbl(acc);
            if ((k[i] >> b) & 1u) {
                add(acc, p);
                started = true;
            }
        }
    return acc;
}

// k * p for a small integer k
template <class F>
CG_HD XYZZ<F> scalar_mul_u32(const XYZZ<F>& p, uint32_t k) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int b = 31; b >= 0; --b) {
        acc = dbl(acc);
        if ((k >> b) & 1u) add(acc, p);
    }
    return acc;
}

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;

// curve constants in Montgomery form are produced at run time by the host (b = 3; b' = 3/(9+u)).

}  // namespace cg
