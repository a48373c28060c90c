// BN254 prime-field arithmetic for gfx950 (and the host side of the same library).
//
// Representation: Montgomery form, R = 2^256, eight 32-bit limbs little-endian.  The 32 bytes are
// identical to arkworks' in-memory `Fp256<MontBackend<_,4>>` (four u64 LE limbs), so a Rust shim
// can hand its field elements over without conversion (forks/circom-compat/src/zkey.rs:397-402
// pins R mod q in exactly this byte form).
//
// CDNA4 has no 64x64 multiplier; the widest integer multiply is v_mad_u64_u32
// (32x32 + 64 -> 64).  Every product below is written as `(u64)a * b + c` on 32-bit operands so
// hipcc emits exactly that instruction; carries ride in the upper half of the 64-bit accumulator.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#if defined(__HIP_DEVICE_COMPILE__)
#define CG_HD __host__ __device__ __forceinline__
#else
#define CG_HD __host__ __device__ inline   /* host pass: let the compiler outline the big bodies */
#endif
/* The Montgomery product is a real function call on the device: a mixed addition is ten of them and
 * a G2 addition over forty; inlined they would be 50-200 KB of straight-line code against a 64 KB
 * instruction cache shared by two CUs. */
#define CG_MUL_FN __host__ __device__ __attribute__((noinline))
#else
#define CG_HD inline
#define CG_MUL_FN inline
#endif

namespace cg {

// ---- field parameter packs ---------------------------------------------------------------------
// Fq: base field of BN254 (forks/halo2curves/src/bn256/fq.rs:12)
struct FqP {
    static constexpr uint32_t N[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t NINV = 0xe4866389u;  // -q^-1 mod 2^32
    static constexpr uint64_t NINV64 = 0x87d20782e4866389ull;  // -q^-1 mod 2^64 (host path)
    // R mod q  (Montgomery one; KAT zkey.rs:397-402)
    static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                        0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    // R^2 mod q
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                       0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};
// Fr: scalar field of BN254 (forks/halo2curves/src/bn256/fr.rs:10; LE bytes r1cs_reader.rs:183)
struct FrP {
    static constexpr uint32_t N[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t NINV = 0xefffffffu;  // -r^-1 mod 2^32
    static constexpr uint64_t NINV64 = 0xc2e1f593efffffffull;  // -r^-1 mod 2^64 (host path)
    static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                        0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

template <class P>
struct alignas(16) Fp {
    uint32_t l[8];

    CG_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = 0;
        return r;
    }
    CG_HD static Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = P::ONE[i];
        return r;
    }
    CG_HD static Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = P::R2[i];
        return r;
    }
    CG_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= l[i];
        return o == 0;
    }
    CG_HD bool operator==(const Fp& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    CG_HD bool operator!=(const Fp& b) const { return !(*this == b); }
};

// r = a - N if a >= N else a     (a < 2N < 2^256)
template <class P>
CG_HD void reduce_once(uint32_t a[8]) {
    uint32_t t[8];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)a[i] - P::N[i] - br;
        t[i] = (uint32_t)d;
        br = (d >> 32) & 1u;
    }
    // br == 1  <=>  a < N : keep a
    uint32_t keep = (uint32_t)0 - (uint32_t)br;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (a[i] & keep) | (t[i] & ~keep);
}

template <class P>
CG_HD Fp<P> add(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)a.l[i] + b.l[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    reduce_once<P>(r.l);  // a+b < 2N < 2^255: no carry out of limb 7
    return r;
}

template <class P>
CG_HD Fp<P> sub(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)a.l[i] - b.l[i] - br;
        r.l[i] = (uint32_t)d;
        br = (d >> 32) & 1u;
    }
    uint32_t m = (uint32_t)0 - (uint32_t)br;  // borrow -> add N back
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)r.l[i] + (P::N[i] & m);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

template <class P>
CG_HD Fp<P> neg(const Fp<P>& a) {
    Fp<P> r;
    uint64_t br = 0;
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) nz |= a.l[i];
    uint32_t m = nz ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)(P::N[i] & m) - a.l[i] - br;
        r.l[i] = (uint32_t)d;
        br = (d >> 32) & 1u;
    }
    return r;
}

template <class P>
CG_HD Fp<P> dbl(const Fp<P>& a) {
    Fp<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r.l[i] = (a.l[i] << 1) | c;
        c = a.l[i] >> 31;
    }
    reduce_once<P>(r.l);
    return r;
}

// Montgomery product a*b*R^-1 mod N, CIOS with the two inner passes fused.
// Invariant: the running value stays < 2N < 2^255, so nine words suffice.
template <class P>
CG_MUL_FN Fp<P> mul(const Fp<P> a, const Fp<P> b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    // host: same CIOS on four 64-bit limbs (the CPU has a 64x64 multiplier)
    typedef unsigned __int128 u128;
    uint64_t A[4], B[4], N[4], T[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        A[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
        B[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
        N[i] = (uint64_t)P::N[2 * i] | ((uint64_t)P::N[2 * i + 1] << 32);
    }
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)A[j] * B[i] + T[j];
            T[j] = (uint64_t)c;
            c >>= 64;
        }
        c += T[4];
        T[4] = (uint64_t)c;
        uint64_t m = T[0] * P::NINV64;
        c = (u128)m * N[0] + T[0];
        c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)m * N[j] + T[j];
            T[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += T[4];
        T[3] = (uint64_t)c;
        T[4] = (uint64_t)(c >> 64);
    }
    Fp<P> rh;
    for (int i = 0; i < 4; ++i) {
        rh.l[2 * i] = (uint32_t)T[i];
        rh.l[2 * i + 1] = (uint32_t)(T[i] >> 32);
    }
    reduce_once<P>(rh.l);
    return rh;
#else
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t bi = b.l[i];
        // t += a * b[i]
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c += (uint64_t)a.l[j] * bi + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;  // < 2^32 because t < 2N + (2^32-1)N < 2^32 * 2^256 / 4
        // t = (t + m*N) / 2^32
        const uint32_t m = t[0] * P::NINV;
        c = (uint64_t)m * P::N[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            c += (uint64_t)m * P::N[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = (uint32_t)(c >> 32);
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = t[i];
    reduce_once<P>(r.l);
    return r;
#endif
}

template <class P>
CG_HD Fp<P> sqr(const Fp<P>& a) {
    return mul(a, a);
}

template <class P>
CG_HD Fp<P> to_mont(const Fp<P>& a) {
    return mul(a, Fp<P>::r2());
}
template <class P>
CG_HD Fp<P> from_mont(const Fp<P>& a) {
    Fp<P> o = Fp<P>::zero();
    o.l[0] = 1;
    return mul(a, o);
}

// a^e for a 256-bit little-endian exponent (host-side setup work and the rare device inversion)
template <class P>
CG_HD Fp<P> pow_limbs(const Fp<P>& a, const uint32_t e[8]) {
    Fp<P> r = Fp<P>::one();
    for (int i = 7; i >= 0; --i)
        for (int b = 31; b >= 0; --b) {
            r = sqr(r);
            if ((e[i] >> b) & 1u) r = mul(r, a);
        }
    return r;
}

// a^-1 by Fermat (a != 0); N - 2 never borrows past limb 0 for either modulus.
template <class P>
CG_HD Fp<P> inv(const Fp<P>& a) {
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = P::N[i];
    e[0] -= 2u;
    return pow_limbs(a, e);
}

template <class P>
CG_HD Fp<P> select(bool c, const Fp<P>& a, const Fp<P>& b) {  // c ? a : b
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}

using Fq = Fp<FqP>;
using Fr = Fp<FrP>;

// ---- Fq2 = Fq[u]/(u^2+1)  (forks/halo2curves/src/bn256/fq.rs:29-31) ------------------------------
struct alignas(16) Fq2 {
    Fq c0, c1;
    CG_HD static Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
    CG_HD static Fq2 one() { return {Fq::one(), Fq::zero()}; }
    CG_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    CG_HD bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
    CG_HD bool operator!=(const Fq2& b) const { return !(*this == b); }
};
CG_HD Fq2 add(const Fq2& a, const Fq2& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
CG_HD Fq2 sub(const Fq2& a, const Fq2& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
CG_HD Fq2 neg(const Fq2& a) { return {neg(a.c0), neg(a.c1)}; }
CG_HD Fq2 dbl(const Fq2& a) { return {dbl(a.c0), dbl(a.c1)}; }
CG_HD Fq2 mul(const Fq2& a, const Fq2& b) {  // Karatsuba: 3 Fq products
    Fq v0 = mul(a.c0, b.c0);
    Fq v1 = mul(a.c1, b.c1);
    Fq s = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    return {sub(v0, v1), sub(sub(s, v0), v1)};
}
CG_HD Fq2 sqr(const Fq2& a) {  // (c0+c1)(c0-c1), 2 c0 c1
    Fq t = mul(a.c0, a.c1);
    return {mul(add(a.c0, a.c1), sub(a.c0, a.c1)), dbl(t)};
}
CG_HD Fq2 inv(const Fq2& a) {
    Fq n = inv(add(sqr(a.c0), sqr(a.c1)));
    return {mul(a.c0, n), neg(mul(a.c1, n))};
}
CG_HD Fq2 select(bool c, const Fq2& a, const Fq2& b) {
    return {select(c, a.c0, b.c0), select(c, a.c1, b.c1)};
}

}  // namespace cg
