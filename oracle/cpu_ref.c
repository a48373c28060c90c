/*
 * cpu_ref.c — CPU restatement (plain C, gcc) of the reference's Groth16 prove path, BN254.
 *
 * THIS FILE IS TEST INFRASTRUCTURE: the checker for tests/ and the `cpu_baseline` leg of bench.py.
 * The product (crescent-credentials_amd/) never links, loads or calls it.
 *
 * PARITY STATUS: "parity unpinned" for proof bytes — the reference (Rust over un-vendored crates.io
 * arkworks ^0.4, no Cargo.lock, no rustc here) can be neither built nor run, and its tests pin no
 * proof bytes (forks/groth16/src/test.rs:70-71, creds/src/lib.rs:288-290).  This file is pinned
 * instead (tests/test_cpu_ref.py) against the pure-Python oracle's golden vectors, which are in turn
 * pinned against the reference's encoding KATs and accepted by its verification equation.
 *
 * It follows the ALGORITHMS arkworks uses on this path, so that it can stand in as the timed
 * "arkworks-equivalent" baseline (never labelled "arkworks"):
 *   - prove:        forks/groth16/src/prover.rs:26-136,256-274
 *   - witness map:  forks/groth16/src/r1cs_to_qap.rs:16-45,150-213
 *   - MSM:          ark-ec 0.4 VariableBaseMSM::msm_bigint [ark-mem]: window c = 3 if n < 32 else
 *                   ln(n)+2 with ln(n) ~ log2(n)*69/100, unsigned c-bit digits, 2^c - 1 Jacobian
 *                   buckets per window filled by mixed additions, zero scalars skipped and unit
 *                   scalars added directly in window 0, running-sum bucket reduction, ONE task per
 *                   window (that is all the parallelism arkworks' MSM has), windows folded with c
 *                   doublings each.
 *   - NTT:          radix-2 in-order transforms, ω = 5^((r-1)/2^k), coset offset g = 5
 *                   (call sites r1cs_to_qap.rs:179-185,198-199,210), loops split across threads.
 *   - field:        4 x 64-bit-limb Montgomery (R = 2^256), the representation arkworks uses.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;            /* field element, Montgomery form */
typedef struct { const uint64_t n[4]; uint64_t ninv; fe one; fe r2; } field_t;

static const field_t FQ = {
    {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
    0x87d20782e4866389ull,
    {{0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full}},
    {{0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full}}};
static const field_t FR = {
    {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
    0xc2e1f593efffffffull,
    {{0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full}},
    {{0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull}}};

static inline int fe_is_zero(const fe* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe* a, const fe* b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int ge_mod(const uint64_t a[4], const uint64_t n[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > n[i]) return 1;
        if (a[i] < n[i]) return 0;
    }
    return 1;
}
static inline void sub_mod_raw(uint64_t a[4], const uint64_t n[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - n[i] - (uint64_t)br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static inline void fe_add(const field_t* F, fe* r, const fe* a, const fe* b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a->l[i] + b->l[i];
        r->l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (ge_mod(r->l, F->n)) sub_mod_raw(r->l, F->n);
}
static inline void fe_sub(const field_t* F, fe* r, const fe* a, const fe* b) {
    u128 br = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br;
        t[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)t[i] + F->n[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(r->l, t, 32);
}
static inline void fe_neg(const field_t* F, fe* r, const fe* a) {
    fe z = {{0, 0, 0, 0}};
    fe_sub(F, r, &z, a);
}
static inline void fe_dbl(const field_t* F, fe* r, const fe* a) { fe_add(F, r, a, a); }
/* Montgomery multiplication: coarsely integrated operand scanning in the "no-carry" form ark-ff generates for a modulus
 * whose top bit is clear (both BN254 moduli are below 2^254): the running value never needs a fifth limb, so every
 * step is one 64x64 -> 128 product plus two additions that cannot overflow 128 bits.  Always inlined: the field
 * constants fold in and the sixteen + sixteen products unroll into mulx / adc chains (-mbmi2 -madx; arkworks' `asm`
 * feature, creds/Cargo.toml:12-15, selects the same instructions). */
#define CG_MM_STEP(I)                                                                     \
    do {                                                                                  \
        u128 p = (u128)a->l[0] * b->l[I] + t0;                                            \
        uint64_t lo = (uint64_t)p, A = (uint64_t)(p >> 64);                               \
        const uint64_t m = lo * F->ninv;                                                  \
        p = (u128)m * F->n[0] + lo;                                                       \
        uint64_t C = (uint64_t)(p >> 64);                                                 \
        p = (u128)a->l[1] * b->l[I] + t1 + A; lo = (uint64_t)p; A = (uint64_t)(p >> 64);  \
        p = (u128)m * F->n[1] + lo + C; t0 = (uint64_t)p; C = (uint64_t)(p >> 64);        \
        p = (u128)a->l[2] * b->l[I] + t2 + A; lo = (uint64_t)p; A = (uint64_t)(p >> 64);  \
        p = (u128)m * F->n[2] + lo + C; t1 = (uint64_t)p; C = (uint64_t)(p >> 64);        \
        p = (u128)a->l[3] * b->l[I] + t3 + A; lo = (uint64_t)p; A = (uint64_t)(p >> 64);  \
        p = (u128)m * F->n[3] + lo + C; t2 = (uint64_t)p; C = (uint64_t)(p >> 64);        \
        t3 = C + A;                                                                       \
    } while (0)
static inline __attribute__((always_inline)) void fe_mul(const field_t* F, fe* r, const fe* a, const fe* b) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    CG_MM_STEP(0); CG_MM_STEP(1); CG_MM_STEP(2); CG_MM_STEP(3);
    uint64_t t[4] = {t0, t1, t2, t3};
    if (ge_mod(t, F->n)) sub_mod_raw(t, F->n);
    memcpy(r->l, t, 32);
}
#undef CG_MM_STEP
static inline void fe_sqr(const field_t* F, fe* r, const fe* a) { fe_mul(F, r, a, a); }
static void fe_pow(const field_t* F, fe* r, const fe* a, const uint64_t e[4]) {
    fe acc = F->one;
    for (int i = 3; i >= 0; --i)
        for (int b = 63; b >= 0; --b) {
            fe_sqr(F, &acc, &acc);
            if ((e[i] >> b) & 1) fe_mul(F, &acc, &acc, a);
        }
    *r = acc;
}
static void fe_inv(const field_t* F, fe* r, const fe* a) {
    uint64_t e[4] = {F->n[0] - 2, F->n[1], F->n[2], F->n[3]};
    fe_pow(F, r, a, e);
}
static void fe_from_canonical(const field_t* F, fe* r, const uint8_t b[32]) {
    fe t;
    memcpy(t.l, b, 32);
    fe_mul(F, r, &t, &F->r2);
}
static void fe_to_canonical(const field_t* F, uint8_t b[32], const fe* a) {
    fe one = {{1, 0, 0, 0}}, t;
    fe_mul(F, &t, a, &one);
    memcpy(b, t.l, 32);
}
static void fe_from_u64(const field_t* F, fe* r, uint64_t v) {
    fe t = {{v, 0, 0, 0}};
    fe_mul(F, r, &t, &F->r2);
}

/* ---- Fq2 = Fq[u]/(u^2+1) --------------------------------------------------------------------- */
typedef struct { fe c0, c1; } fe2;
static inline int fe2_is_zero(const fe2* a) { return fe_is_zero(&a->c0) && fe_is_zero(&a->c1); }
static inline void fe2_add(fe2* r, const fe2* a, const fe2* b) { fe_add(&FQ, &r->c0, &a->c0, &b->c0); fe_add(&FQ, &r->c1, &a->c1, &b->c1); }
static inline void fe2_sub(fe2* r, const fe2* a, const fe2* b) { fe_sub(&FQ, &r->c0, &a->c0, &b->c0); fe_sub(&FQ, &r->c1, &a->c1, &b->c1); }
static inline void fe2_neg(fe2* r, const fe2* a) { fe_neg(&FQ, &r->c0, &a->c0); fe_neg(&FQ, &r->c1, &a->c1); }
static void fe2_mul(fe2* r, const fe2* a, const fe2* b) {
    fe v0, v1, s, t, u;
    fe_mul(&FQ, &v0, &a->c0, &b->c0);
    fe_mul(&FQ, &v1, &a->c1, &b->c1);
    fe_add(&FQ, &s, &a->c0, &a->c1);
    fe_add(&FQ, &t, &b->c0, &b->c1);
    fe_mul(&FQ, &u, &s, &t);
    fe_sub(&FQ, &u, &u, &v0);
    fe_sub(&FQ, &r->c1, &u, &v1);
    fe_sub(&FQ, &r->c0, &v0, &v1);
}
static void fe2_sqr(fe2* r, const fe2* a) { fe2 t = *a; fe2_mul(r, &t, &t); }
static void fe2_inv(fe2* r, const fe2* a) {
    fe n0, n1, n;
    fe_sqr(&FQ, &n0, &a->c0);
    fe_sqr(&FQ, &n1, &a->c1);
    fe_add(&FQ, &n, &n0, &n1);
    fe_inv(&FQ, &n, &n);
    fe_mul(&FQ, &r->c0, &a->c0, &n);
    fe_mul(&FQ, &n0, &a->c1, &n);
    fe_neg(&FQ, &r->c1, &n0);
}

/* ---- Jacobian group arithmetic, generated for both coordinate fields ----------------------------
 * arkworks' short_weierstrass::Projective is Jacobian; formulas: dbl-2009-l, add-2007-bl, madd-2007-bl. */
#define DEFINE_GROUP(SUF, T, ADD, SUB, MUL, SQR, ISZ, INV, NEG)                                            \
    typedef struct { T x, y; int inf; } aff##SUF;                                                         \
    typedef struct { T x, y, z; } jac##SUF; /* z == 0 <=> infinity */                                      \
    static void jac##SUF##_set_inf(jac##SUF* p) { memset(p, 0, sizeof(*p)); }                             \
    static int jac##SUF##_is_inf(const jac##SUF* p) { return ISZ(&p->z); }                                \
    static void jac##SUF##_dbl(jac##SUF* r, const jac##SUF* p) {                                          \
        if (ISZ(&p->z)) { *r = *p; return; }                                                              \
        T a, b, c, d, e, f, t, x3, y3, z3;                                                                 \
        SQR(&a, &p->x); SQR(&b, &p->y); SQR(&c, &b);                                                       \
        ADD(&t, &p->x, &b); SQR(&t, &t); SUB(&t, &t, &a); SUB(&t, &t, &c); ADD(&d, &t, &t);                \
        ADD(&e, &a, &a); ADD(&e, &e, &a); SQR(&f, &e);                                                     \
        SUB(&x3, &f, &d); SUB(&x3, &x3, &d);                                                               \
        SUB(&t, &d, &x3); MUL(&y3, &e, &t); ADD(&c, &c, &c); ADD(&c, &c, &c); ADD(&c, &c, &c); SUB(&y3, &y3, &c); \
        MUL(&z3, &p->y, &p->z); ADD(&z3, &z3, &z3);                                                        \
        r->x = x3; r->y = y3; r->z = z3;                                                                   \
    }                                                                                                      \
    static void jac##SUF##_add(jac##SUF* r, const jac##SUF* p, const jac##SUF* q) {                       \
        if (ISZ(&p->z)) { *r = *q; return; }                                                              \
        if (ISZ(&q->z)) { *r = *p; return; }                                                              \
        T z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;                                        \
        SQR(&z1z1, &p->z); SQR(&z2z2, &q->z);                                                              \
        MUL(&u1, &p->x, &z2z2); MUL(&u2, &q->x, &z1z1);                                                    \
        MUL(&s1, &p->y, &q->z); MUL(&s1, &s1, &z2z2);                                                      \
        MUL(&s2, &q->y, &p->z); MUL(&s2, &s2, &z1z1);                                                      \
        SUB(&h, &u2, &u1); SUB(&rr, &s2, &s1);                                                             \
        if (ISZ(&h)) { if (ISZ(&rr)) jac##SUF##_dbl(r, p); else jac##SUF##_set_inf(r); return; }           \
        ADD(&i, &h, &h); SQR(&i, &i); MUL(&j, &h, &i); ADD(&rr, &rr, &rr); MUL(&v, &u1, &i);                \
        SQR(&x3, &rr); SUB(&x3, &x3, &j); SUB(&x3, &x3, &v); SUB(&x3, &x3, &v);                            \
        SUB(&t, &v, &x3); MUL(&y3, &rr, &t); MUL(&t, &s1, &j); ADD(&t, &t, &t); SUB(&y3, &y3, &t);          \
        ADD(&z3, &p->z, &q->z); SQR(&z3, &z3); SUB(&z3, &z3, &z1z1); SUB(&z3, &z3, &z2z2); MUL(&z3, &z3, &h); \
        r->x = x3; r->y = y3; r->z = z3;                                                                   \
    }                                                                                                      \
    static void jac##SUF##_madd(jac##SUF* r, const jac##SUF* p, const aff##SUF* q, const T* one) {        \
        if (q->inf) { *r = *p; return; }                                                                  \
        if (ISZ(&p->z)) { r->x = q->x; r->y = q->y; r->z = *one; return; }                                 \
        T z1z1, u2, s2, h, hh, i, j, rr, v, t, x3, y3, z3;                                                  \
        SQR(&z1z1, &p->z); MUL(&u2, &q->x, &z1z1); MUL(&s2, &q->y, &p->z); MUL(&s2, &s2, &z1z1);            \
        SUB(&h, &u2, &p->x); SUB(&rr, &s2, &p->y);                                                         \
        if (ISZ(&h)) { if (ISZ(&rr)) jac##SUF##_dbl(r, p); else jac##SUF##_set_inf(r); return; }           \
        SQR(&hh, &h); ADD(&i, &hh, &hh); ADD(&i, &i, &i); MUL(&j, &h, &i); ADD(&rr, &rr, &rr); MUL(&v, &p->x, &i); \
        SQR(&x3, &rr); SUB(&x3, &x3, &j); SUB(&x3, &x3, &v); SUB(&x3, &x3, &v);                            \
        SUB(&t, &v, &x3); MUL(&y3, &rr, &t); MUL(&t, &p->y, &j); ADD(&t, &t, &t); SUB(&y3, &y3, &t);        \
        ADD(&z3, &p->z, &h); SQR(&z3, &z3); SUB(&z3, &z3, &z1z1); SUB(&z3, &z3, &hh);                       \
        r->x = x3; r->y = y3; r->z = z3;                                                                   \
    }                                                                                                      \
    static void jac##SUF##_neg(jac##SUF* r, const jac##SUF* p) { *r = *p; NEG(&r->y, &p->y); }             \
    static void jac##SUF##_to_affine(aff##SUF* r, const jac##SUF* p) {                                    \
        if (ISZ(&p->z)) { memset(r, 0, sizeof(*r)); r->inf = 1; return; }                                  \
        T zi, zi2, zi3;                                                                                    \
        INV(&zi, &p->z); SQR(&zi2, &zi); MUL(&zi3, &zi2, &zi);                                             \
        MUL(&r->x, &p->x, &zi2); MUL(&r->y, &p->y, &zi3); r->inf = 0;                                      \
    }                                                                                                      \
    /* k * p, k a 256-bit little-endian integer, MSB-first double-and-add (mul_bigint) */                  \
    static void jac##SUF##_mul(jac##SUF* r, const jac##SUF* p, const uint64_t k[4]) {                     \
        jac##SUF acc; jac##SUF##_set_inf(&acc);                                                            \
        for (int i = 3; i >= 0; --i) for (int b = 63; b >= 0; --b) {                                       \
            jac##SUF##_dbl(&acc, &acc);                                                                    \
            if ((k[i] >> b) & 1) jac##SUF##_add(&acc, &acc, p);                                            \
        }                                                                                                  \
        *r = acc;                                                                                          \
    }

static inline void q_add(fe* r, const fe* a, const fe* b) { fe_add(&FQ, r, a, b); }
static inline void q_sub(fe* r, const fe* a, const fe* b) { fe_sub(&FQ, r, a, b); }
static inline void q_mul(fe* r, const fe* a, const fe* b) { fe_mul(&FQ, r, a, b); }
static inline void q_sqr(fe* r, const fe* a) { fe_sqr(&FQ, r, a); }
static inline void q_inv(fe* r, const fe* a) { fe_inv(&FQ, r, a); }
static inline void q_neg(fe* r, const fe* a) { fe_neg(&FQ, r, a); }
DEFINE_GROUP(1, fe, q_add, q_sub, q_mul, q_sqr, fe_is_zero, q_inv, q_neg)
DEFINE_GROUP(2, fe2, fe2_add, fe2_sub, fe2_mul, fe2_sqr, fe2_is_zero, fe2_inv, fe2_neg)

static const fe2* fe2_one(void) {
    static fe2 o;
    o.c0 = FQ.one;
    memset(&o.c1, 0, sizeof(fe));
    return &o;
}

/* packed canonical bytes <-> affine (identity = all zero, include/crescent_gpu.h conventions) */
static int all_zero(const uint8_t* b, int n) { for (int i = 0; i < n; ++i) if (b[i]) return 0; return 1; }
static void aff1_load(aff1* p, const uint8_t b[64]) {
    if (all_zero(b, 64)) { memset(p, 0, sizeof(*p)); p->inf = 1; return; }
    fe_from_canonical(&FQ, &p->x, b); fe_from_canonical(&FQ, &p->y, b + 32); p->inf = 0;
}
static void aff2_load(aff2* p, const uint8_t b[128]) {
    if (all_zero(b, 128)) { memset(p, 0, sizeof(*p)); p->inf = 1; return; }
    fe_from_canonical(&FQ, &p->x.c0, b); fe_from_canonical(&FQ, &p->x.c1, b + 32);
    fe_from_canonical(&FQ, &p->y.c0, b + 64); fe_from_canonical(&FQ, &p->y.c1, b + 96); p->inf = 0;
}
static void aff1_store(uint8_t b[64], const aff1* p) {
    if (p->inf) { memset(b, 0, 64); return; }
    fe_to_canonical(&FQ, b, &p->x); fe_to_canonical(&FQ, b + 32, &p->y);
}
static void aff2_store(uint8_t b[128], const aff2* p) {
    if (p->inf) { memset(b, 0, 128); return; }
    fe_to_canonical(&FQ, b, &p->x.c0); fe_to_canonical(&FQ, b + 32, &p->x.c1);
    fe_to_canonical(&FQ, b + 64, &p->y.c0); fe_to_canonical(&FQ, b + 96, &p->y.c1);
}

/* ---- Pippenger, as ark-ec 0.4 msm_bigint [ark-mem] ------------------------------------------- */
static int ln_without_floats(uint64_t a) {  /* log2(a) * 69 / 100 */
    int lg = 0;
    while ((1ull << (lg + 1)) <= a && lg < 62) ++lg;
    return lg * 69 / 100;
}
static inline int big_is_zero(const uint64_t s[4]) { return (s[0] | s[1] | s[2] | s[3]) == 0; }
static inline int big_is_one(const uint64_t s[4]) { return s[0] == 1 && (s[1] | s[2] | s[3]) == 0; }
static inline uint64_t big_window(const uint64_t s[4], int start, int c) {
    int w = start >> 6, sh = start & 63;
    uint64_t v = s[w] >> sh;
    if (sh && w + 1 < 4) v |= s[w + 1] << (64 - sh);
    return v & ((1ull << c) - 1);
}

#define DEFINE_MSM(SUF, ONEPTR)                                                                            \
    static void msm##SUF(jac##SUF* out, const aff##SUF* bases, const uint64_t (*scalars)[4], uint64_t n, int nthreads) { \
        jac##SUF##_set_inf(out);                                                                           \
        if (n == 0) return;                                                                                \
        const int c = n < 32 ? 3 : ln_without_floats(n) + 2;                                               \
        const int num_bits = 254;                                                                          \
        const int nwin = (num_bits + c - 1) / c;                                                           \
        jac##SUF* sums = (jac##SUF*)malloc(sizeof(jac##SUF) * nwin);                                       \
        _Pragma("omp parallel for schedule(dynamic, 1) num_threads(nthreads)")                             \
        for (int w = 0; w < nwin; ++w) {                                                                   \
            const int start = w * c;                                                                       \
            const uint64_t nb = (1ull << c) - 1;                                                           \
            jac##SUF* buckets = (jac##SUF*)calloc(nb, sizeof(jac##SUF));                                    \
            jac##SUF res; jac##SUF##_set_inf(&res);                                                        \
            for (uint64_t i = 0; i < n; ++i) {                                                             \
                const uint64_t* s = scalars[i];                                                            \
                if (big_is_zero(s)) continue;                                                              \
                if (big_is_one(s)) { if (start == 0) jac##SUF##_madd(&res, &res, &bases[i], ONEPTR); continue; } \
                uint64_t d = big_window(s, start, c);                                                      \
                if (d) jac##SUF##_madd(&buckets[d - 1], &buckets[d - 1], &bases[i], ONEPTR);               \
            }                                                                                              \
            jac##SUF run; jac##SUF##_set_inf(&run);                                                        \
            for (uint64_t b = nb; b-- > 0;) {                                                              \
                jac##SUF##_add(&run, &run, &buckets[b]);                                                   \
                jac##SUF##_add(&res, &res, &run);                                                          \
            }                                                                                              \
            sums[w] = res;                                                                                 \
            free(buckets);                                                                                 \
        }                                                                                                  \
        jac##SUF total; jac##SUF##_set_inf(&total);                                                        \
        for (int w = nwin - 1; w >= 1; --w) {                                                              \
            jac##SUF##_add(&total, &total, &sums[w]);                                                      \
            for (int k = 0; k < c; ++k) jac##SUF##_dbl(&total, &total);                                    \
        }                                                                                                  \
        jac##SUF##_add(out, &total, &sums[0]);                                                             \
        free(sums);                                                                                        \
    }
DEFINE_MSM(1, &FQ.one)
DEFINE_MSM(2, fe2_one())

/* ---- radix-2 NTT over Fr (ark-poly Radix2EvaluationDomain semantics) --------------------------- */
static void fr_root_of_unity(fe* w, int logn) {
    /* 5^((r-1)/2^28) squared down to order 2^logn */
    uint64_t e[4], nm1[4] = {FR.n[0] - 1, FR.n[1], FR.n[2], FR.n[3]};
    for (int i = 0; i < 4; ++i) e[i] = (nm1[i] >> 28) | (i + 1 < 4 ? nm1[i + 1] << 36 : 0);
    fe g;
    fe_from_u64(&FR, &g, 5);
    fe_pow(&FR, w, &g, e);
    for (int i = 28; i > logn; --i) fe_sqr(&FR, w, w);
}
static uint64_t bitrev(uint64_t x, int bits) {
    uint64_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
/* in place, natural order in and out: x[k] <- Σ_j x[j] root^{jk} over m = 2^logm points that fit the cache: the textbook
 * loop (bit reversal, then log m decimation-in-time stages); tw[i] = root^i for i < m/2.  One thread. */
static void ntt_small(fe* x, int logm, const fe* tw) {
    const uint64_t m = 1ull << logm;
    for (uint64_t i = 0; i < m; ++i) {
        uint64_t j = bitrev(i, logm);
        if (i < j) { fe t = x[i]; x[i] = x[j]; x[j] = t; }
    }
    for (int s = 1; s <= logm; ++s) {
        const uint64_t len = 1ull << s, half = len >> 1, step = m / len;
        for (uint64_t i0 = 0; i0 < m; i0 += len)
            for (uint64_t k = 0; k < half; ++k) {
                fe v, u = x[i0 + k];
                fe_mul(&FR, &v, &x[i0 + k + half], &tw[k * step]);
                fe_add(&FR, &x[i0 + k], &u, &v);
                fe_sub(&FR, &x[i0 + k + half], &u, &v);
            }
    }
}
static void powers_table(fe* tw, uint64_t count, const fe* root) {
    fe p = FR.one;
    for (uint64_t i = 0; i < count; ++i) { tw[i] = p; fe_mul(&FR, &p, &p, root); }
}
/* dst (cols x rows) = transpose of src (rows x cols), tile by tile */
static void transpose(fe* dst, const fe* src, uint64_t rows, uint64_t cols, int nthreads) {
    const uint64_t B = 16;
    _Pragma("omp parallel for collapse(2) schedule(static) num_threads(nthreads)")
    for (uint64_t r0 = 0; r0 < rows; r0 += B)
        for (uint64_t c0 = 0; c0 < cols; c0 += B) {
            const uint64_t r1 = r0 + B < rows ? r0 + B : rows, c1 = c0 + B < cols ? c0 + B : cols;
            for (uint64_t r = r0; r < r1; ++r)
                for (uint64_t c = c0; c < c1; ++c) dst[c * rows + r] = src[r * cols + c];
        }
}
/* in place, natural order in and out: a[k] <- Σ_j a[j] w^{jk}.
 * The same transform as ark-poly's radix-2 domain computes, arranged so that every thread works on its own data between
 * a handful of joins (the four-step form: n = n1·n2, j = j1·n2 + j2, k = k1 + n1·k2):
 *     A[k1 + n1·k2] = Σ_{j2} w^{j2·k1} · ( Σ_{j1} a[j1·n2 + j2] · (w^{n2})^{j1·k1} ) · (w^{n1})^{j2·k2}
 *   1. transpose to n2 x n1;  2. n2 independent in-cache transforms of size n1, each element then times w^{j2·k1};
 *   3. transpose back to n1 x n2;  4. n1 independent in-cache transforms of size n2;  5. transpose into natural order.
 * Every row transform is a statically assigned, independent task: the butterfly stages themselves need no barrier (round
 * 2's stage-by-stage sweeps joined all threads log n - 12 times per transform).  The arithmetic is exact, so the values
 * are those of any other radix-2 evaluation order. */
static void ntt_inplace(fe* a, int logn, const fe* w, int nthreads) {
    const uint64_t n = 1ull << logn;
    if (n == 1) return;
    if (logn < 6) {                                    /* too small to split */
        fe* tw = (fe*)malloc(sizeof(fe) * (n / 2));
        powers_table(tw, n / 2, w);
        ntt_small(a, logn, tw);
        free(tw);
        return;
    }
    const int log1 = logn / 2, log2 = logn - log1;
    const uint64_t n1 = 1ull << log1, n2 = 1ull << log2;
    fe w1 = *w, w2 = *w;                               /* w1 = w^{n2} (order n1), w2 = w^{n1} (order n2) */
    for (int i = 0; i < log2; ++i) fe_sqr(&FR, &w1, &w1);
    for (int i = 0; i < log1; ++i) fe_sqr(&FR, &w2, &w2);
    fe* tw1 = (fe*)malloc(sizeof(fe) * (n1 / 2));
    fe* tw2 = (fe*)malloc(sizeof(fe) * (n2 / 2));
    powers_table(tw1, n1 / 2, &w1);
    powers_table(tw2, n2 / 2, &w2);
    fe* t = (fe*)malloc(sizeof(fe) * n);
    transpose(t, a, n1, n2, nthreads);                 /* t[j2][j1] */
    _Pragma("omp parallel num_threads(nthreads)")
    {
        const int id = omp_get_thread_num(), T = omp_get_num_threads();
        const uint64_t lo = n2 * id / T, hi = n2 * (id + 1) / T;
        if (lo < hi) {
            uint64_t e[4] = {lo, 0, 0, 0};
            fe rho; fe_pow(&FR, &rho, w, e);           /* w^{j2} for the first row of this thread */
            for (uint64_t j2 = lo; j2 < hi; ++j2) {
                fe* x = t + j2 * n1;
                ntt_small(x, log1, tw1);
                fe p = rho;                            /* x[k1] *= w^{j2·k1} */
                for (uint64_t k1 = 1; k1 < n1; ++k1) { fe_mul(&FR, &x[k1], &x[k1], &p); fe_mul(&FR, &p, &p, &rho); }
                fe_mul(&FR, &rho, &rho, w);
            }
        }
    }
    transpose(a, t, n2, n1, nthreads);                 /* a[k1][j2] */
    _Pragma("omp parallel for schedule(static) num_threads(nthreads)")
    for (uint64_t k1 = 0; k1 < n1; ++k1) ntt_small(a + k1 * n2, log2, tw2);
    transpose(t, a, n1, n2, nthreads);                 /* t[k2][k1] = A[k1 + n1·k2] */
    _Pragma("omp parallel for schedule(static) num_threads(nthreads)")
    for (uint64_t i = 0; i < n; i += 4096) memcpy(a + i, t + i, sizeof(fe) * (n - i < 4096 ? n - i : 4096));
    free(t); free(tw1); free(tw2);
}
typedef struct { int logn; uint64_t n; fe w, winv, ninv, g, ginv; } domain_t;
static void domain_init(domain_t* d, int logn) {
    d->logn = logn; d->n = 1ull << logn;
    fr_root_of_unity(&d->w, logn);
    fe_inv(&FR, &d->winv, &d->w);
    fe nn; fe_from_u64(&FR, &nn, d->n); fe_inv(&FR, &d->ninv, &nn);
    fe_from_u64(&FR, &d->g, 5);                       /* F::GENERATOR (r1cs_to_qap.rs:182) */
    fe_inv(&FR, &d->ginv, &d->g);
}
static void scale_powers(fe* a, uint64_t n, const fe* base, const fe* scale0, int nthreads) {
    /* a[i] *= scale0 * base^i, split in chunks with an independent power start per chunk */
    _Pragma("omp parallel num_threads(nthreads)")
    {
        int t = omp_get_thread_num(), T = omp_get_num_threads();
        uint64_t lo = n * t / T, hi = n * (t + 1) / T;
        uint64_t e[4] = {lo, 0, 0, 0};
        fe p; fe_pow(&FR, &p, base, e); fe_mul(&FR, &p, &p, scale0);
        for (uint64_t i = lo; i < hi; ++i) { fe_mul(&FR, &a[i], &a[i], &p); fe_mul(&FR, &p, &p, base); }
    }
}
static void dom_fft(const domain_t* d, fe* a, int coset, int nt) {
    if (coset) scale_powers(a, d->n, &d->g, &FR.one, nt);      /* coset_domain.fft_in_place */
    ntt_inplace(a, d->logn, &d->w, nt);
}
static void dom_ifft(const domain_t* d, fe* a, int coset, int nt) {
    ntt_inplace(a, d->logn, &d->winv, nt);
    fe one = FR.one;
    scale_powers(a, d->n, coset ? &d->ginv : &one, &d->ninv, nt);  /* x 1/n, and g^-i for the coset */
}

/* ---- witness map (r1cs_to_qap.rs:16-45,150-213) ----------------------------------------------- */
typedef struct { const uint64_t* row_ptr; const uint32_t* col; const uint8_t* coeff; uint64_t nnz; } csr_t;

static void spmv(const csr_t* m, uint64_t rows, const fe* w, fe* out, int nt) {
    _Pragma("omp parallel for schedule(static, 1024) num_threads(nt)")
    for (uint64_t i = 0; i < rows; ++i) {
        fe acc = {{0, 0, 0, 0}};
        for (uint64_t t = m->row_ptr[i]; t < m->row_ptr[i + 1]; ++t) {
            const uint8_t* cb = m->coeff + 32 * t;
            fe v = w[m->col[t]];
            int is_one = cb[0] == 1 && all_zero(cb + 1, 31);          /* coeff.is_one() shortcut :31-35 */
            if (!is_one) { fe c; fe_from_canonical(&FR, &c, cb); fe_mul(&FR, &v, &v, &c); }
            fe_add(&FR, &acc, &acc, &v);
        }
        out[i] = acc;
    }
}
static int ilog2_ceil(uint64_t n) { int l = 0; while ((1ull << l) < n) ++l; return l; }

/* h (Montgomery) of length D; returns D or 0 when the domain is too large */
static uint64_t witness_map(const csr_t abc[3], uint64_t l, uint64_t m, const fe* w, fe** h_out, int nt) {
    int logd = ilog2_ceil(m + l);
    if (logd > 28) return 0;                                       /* PolynomialDegreeTooLarge :156-157 */
    domain_t d; domain_init(&d, logd);
    uint64_t D = d.n;
    fe* a = (fe*)calloc(D, sizeof(fe)); fe* b = (fe*)calloc(D, sizeof(fe)); fe* c = (fe*)calloc(D, sizeof(fe));
    spmv(&abc[0], m, w, a, nt); spmv(&abc[1], m, w, b, nt);        /* :164-171 */
    memcpy(a + m, w, l * sizeof(fe));                              /* :173-177 */
    dom_ifft(&d, a, 0, nt); dom_ifft(&d, b, 0, nt);                /* :179-180 */
    dom_fft(&d, a, 1, nt); dom_fft(&d, b, 1, nt);                  /* :182-185 */
    _Pragma("omp parallel for num_threads(nt)")
    for (uint64_t i = 0; i < D; ++i) fe_mul(&FR, &a[i], &a[i], &b[i]);   /* :187 */
    spmv(&abc[2], m, w, c, nt);                                    /* :191-196 */
    dom_ifft(&d, c, 0, nt); dom_fft(&d, c, 1, nt);                 /* :198-199 */
    fe gn, vinv; uint64_t e[4] = {D, 0, 0, 0};
    fe_pow(&FR, &gn, &d.g, e); fe_sub(&FR, &gn, &gn, &FR.one); fe_inv(&FR, &vinv, &gn);   /* :201-204 */
    _Pragma("omp parallel for num_threads(nt)")
    for (uint64_t i = 0; i < D; ++i) { fe_sub(&FR, &a[i], &a[i], &c[i]); fe_mul(&FR, &a[i], &a[i], &vinv); }   /* :205-208 */
    dom_ifft(&d, a, 1, nt);                                        /* :210 */
    free(b); free(c);
    *h_out = a;
    return D;
}

/* ---- ark-serialize uncompressed encodings [ark-mem], isolated ---------------------------------- */
static int canon_gt(const uint8_t a[32], const uint8_t b[32]) {
    for (int i = 31; i >= 0; --i) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
    return 0;
}
static void ser_g1(uint8_t out[64], const aff1* p) {
    if (p->inf) { memset(out, 0, 64); out[63] |= 0x40; return; }
    aff1_store(out, p);
    fe ny; uint8_t nb[32];
    fe_neg(&FQ, &ny, &p->y); fe_to_canonical(&FQ, nb, &ny);
    if (canon_gt(out + 32, nb)) out[63] |= 0x80;                   /* y > -y */
}
static void ser_g2(uint8_t out[128], const aff2* p) {
    if (p->inf) { memset(out, 0, 128); out[127] |= 0x40; return; }
    aff2_store(out, p);
    fe2 ny; uint8_t n0[32], n1[32];
    fe2_neg(&ny, &p->y); fe_to_canonical(&FQ, n0, &ny.c0); fe_to_canonical(&FQ, n1, &ny.c1);
    int gt = canon_gt(out + 96, n1) || (!memcmp(out + 96, n1, 32) && canon_gt(out + 64, n0));   /* c1 first, then c0 */
    if (gt) out[127] |= 0x80;
}

/* ---- public entry points (ctypes) -------------------------------------------------------------- */
typedef struct {
    const uint8_t *alpha_g1, *beta_g1, *delta_g1, *beta_g2, *delta_g2;
    const uint8_t *a_query, *b_g1_query, *b_g2_query, *h_query, *l_query;   /* canonical packed */
    uint64_t a_len, b_g1_len, b_g2_len, h_len, l_len;
} ref_pk;
typedef struct { double witness_map_s, msm_h_s, msm_l_s, msm_a_s, msm_b1_s, msm_b2_s, load_s, total_s; } ref_timings;

static void load_scalars(uint64_t (*dst)[4], const uint8_t* src, uint64_t n) { memcpy(dst, src, n * 32); }

static void load_g1s(aff1* dst, const uint8_t* src, uint64_t n, int nt) {
    _Pragma("omp parallel for num_threads(nt)")
    for (uint64_t i = 0; i < n; ++i) aff1_load(&dst[i], src + 64 * i);
}
static void load_g2s(aff2* dst, const uint8_t* src, uint64_t n, int nt) {
    _Pragma("omp parallel for num_threads(nt)")
    for (uint64_t i = 0; i < n; ++i) aff2_load(&dst[i], src + 128 * i);
}

int ref_num_procs(void) { return omp_get_num_procs(); }

/* Σ s_i P_i over G1 / G2; result packed canonical affine */
int ref_msm_g1(const uint8_t* bases, const uint8_t* scalars, uint64_t n, uint8_t out[64], int nthreads) {
    aff1* b = (aff1*)malloc(sizeof(aff1) * (n ? n : 1));
    load_g1s(b, bases, n, nthreads);
    jac1 r; msm1(&r, b, (const uint64_t(*)[4])scalars, n, nthreads);
    aff1 a; jac1_to_affine(&a, &r); aff1_store(out, &a);
    free(b);
    return 0;
}
int ref_msm_g2(const uint8_t* bases, const uint8_t* scalars, uint64_t n, uint8_t out[128], int nthreads) {
    aff2* b = (aff2*)malloc(sizeof(aff2) * (n ? n : 1));
    load_g2s(b, bases, n, nthreads);
    jac2 r; msm2(&r, b, (const uint64_t(*)[4])scalars, n, nthreads);
    aff2 a; jac2_to_affine(&a, &r); aff2_store(out, &a);
    free(b);
    return 0;
}
/* natural-order in-place transform on canonical data */
int ref_ntt(uint8_t* data, int logn, int inverse, int coset, int nthreads) {
    if (logn > 28) return -5;
    domain_t d; domain_init(&d, logn);
    fe* a = (fe*)malloc(sizeof(fe) * d.n);
    for (uint64_t i = 0; i < d.n; ++i) fe_from_canonical(&FR, &a[i], data + 32 * i);
    if (inverse) dom_ifft(&d, a, coset, nthreads); else dom_fft(&d, a, coset, nthreads);
    for (uint64_t i = 0; i < d.n; ++i) fe_to_canonical(&FR, data + 32 * i, &a[i]);
    free(a);
    return 0;
}
int ref_witness_map(const csr_t abc[3], uint64_t l, uint64_t m, uint64_t M, const uint8_t* w_bytes, uint8_t* h_out, int nthreads) {
    fe* w = (fe*)malloc(sizeof(fe) * M);
    _Pragma("omp parallel for num_threads(nthreads)")
    for (uint64_t i = 0; i < M; ++i) fe_from_canonical(&FR, &w[i], w_bytes + 32 * i);
    fe* h; uint64_t D = witness_map(abc, l, m, w, &h, nthreads);
    free(w);
    if (!D) return -5;
    _Pragma("omp parallel for num_threads(nthreads)")
    for (uint64_t i = 0; i < D; ++i) fe_to_canonical(&FR, h_out + 32 * i, &h[i]);
    free(h);
    return 0;
}

/* ---- the QAP at one point: what the KEY CHECK of the full-size tests needs (oracle/keycheck.py) -----------------
 * LibsnarkReduction::instance_map_with_evaluation (r1cs_to_qap.rs:103-147) at t: u_j = L_j(t) over the size-D subgroup
 * (ark-poly evaluate_all_lagrange_coefficients [ark-mem]: L_j(t) = (t^D - 1)/D * w^j / (t - w^j), t outside the domain),
 * a_i(t) = sum_j u_j A[j][i] (+ u_{m+i} for the instance wires, :128-133), b_i(t), c_i(t), zt = t^D - 1.
 * Column accumulation in row order, one thread (it is a checker: a few seconds at 16 M terms).
 * a_out/b_out/c_out: M x 32 canonical; u_out (nullable): D x 32.  Returns -5 when t lies in the domain. */
int ref_qap_at(const csr_t abc[3], uint64_t l, uint64_t m, uint64_t M, const uint8_t t_b[32], uint8_t* a_out, uint8_t* b_out,
               uint8_t* c_out, uint8_t zt_out[32], uint8_t* u_out) {
    int logd = ilog2_ceil(m + l);
    if (logd > 28) return -5;
    uint64_t D = 1ull << logd;
    fe t, w, zt, zd, nn, ninv;
    fe_from_canonical(&FR, &t, t_b);
    fr_root_of_unity(&w, logd);
    uint64_t e[4] = {D, 0, 0, 0};
    fe_pow(&FR, &zt, &t, e); fe_sub(&FR, &zt, &zt, &FR.one);       /* evaluate_vanishing_polynomial */
    if (fe_is_zero(&zt)) return -5;
    fe_from_u64(&FR, &nn, D); fe_inv(&FR, &ninv, &nn); fe_mul(&FR, &zd, &zt, &ninv);
    fe* u = (fe*)malloc(sizeof(fe) * D);      /* first t - w^j, then its inverse by one shared inversion, then L_j(t) */
    fe* pre = (fe*)malloc(sizeof(fe) * D);
    fe p = FR.one, run = FR.one;
    for (uint64_t j = 0; j < D; ++j) { fe_sub(&FR, &u[j], &t, &p); pre[j] = run; fe_mul(&FR, &run, &run, &u[j]); fe_mul(&FR, &p, &p, &w); }
    fe inv; fe_inv(&FR, &inv, &run);
    for (uint64_t j = D; j-- > 0;) { fe d = u[j]; fe_mul(&FR, &u[j], &inv, &pre[j]); fe_mul(&FR, &inv, &inv, &d); }
    p = zd;                                                         /* zd * w^j */
    for (uint64_t j = 0; j < D; ++j) { fe_mul(&FR, &u[j], &u[j], &p); fe_mul(&FR, &p, &p, &w); }
    free(pre);
    fe* acc[3]; uint8_t* outs[3] = {a_out, b_out, c_out};
    for (int k = 0; k < 3; ++k) acc[k] = (fe*)calloc(M, sizeof(fe));
    for (uint64_t i = 0; i < l; ++i) acc[0][i] = u[m + i];          /* :128-133 */
    for (int k = 0; k < 3; ++k)                                     /* :135-145 */
        for (uint64_t j = 0; j < m; ++j)
            for (uint64_t q = abc[k].row_ptr[j]; q < abc[k].row_ptr[j + 1]; ++q) {
                fe c, v; fe_from_canonical(&FR, &c, abc[k].coeff + 32 * q);
                fe_mul(&FR, &v, &u[j], &c);
                fe* dst = &acc[k][abc[k].col[q]];
                fe_add(&FR, dst, dst, &v);
            }
    for (int k = 0; k < 3; ++k) {
        for (uint64_t i = 0; i < M; ++i) fe_to_canonical(&FR, outs[k] + 32 * i, &acc[k][i]);
        free(acc[k]);
    }
    if (u_out) for (uint64_t j = 0; j < D; ++j) fe_to_canonical(&FR, u_out + 32 * j, &u[j]);
    fe_to_canonical(&FR, zt_out, &zt);
    free(u);
    return 0;
}
/* sum_i a_i * b_i over Fr, canonical in and out */
int ref_fr_inner(const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t out[32]) {
    fe acc = {{0, 0, 0, 0}};
    for (uint64_t i = 0; i < n; ++i) {
        fe x, y; fe_from_canonical(&FR, &x, a + 32 * i); fe_from_canonical(&FR, &y, b + 32 * i);
        fe_mul(&FR, &x, &x, &y); fe_add(&FR, &acc, &acc, &x);
    }
    fe_to_canonical(&FR, out, &acc);
    return 0;
}
/* out_i = (x * a_i + y * b_i + c_i) * z over Fr (the l_query / gamma_abc scalars of generator.rs:118-128), and
 * out_i = s * t^i (h_query scalars, r1cs_to_qap.rs:215-225) */
int ref_fr_combine(const uint8_t* a, const uint8_t* b, const uint8_t* c, uint64_t n, const uint8_t x_b[32], const uint8_t y_b[32],
                   const uint8_t z_b[32], uint8_t* out) {
    fe x, y, z; fe_from_canonical(&FR, &x, x_b); fe_from_canonical(&FR, &y, y_b); fe_from_canonical(&FR, &z, z_b);
    for (uint64_t i = 0; i < n; ++i) {
        fe p, q, r; fe_from_canonical(&FR, &p, a + 32 * i); fe_from_canonical(&FR, &q, b + 32 * i); fe_from_canonical(&FR, &r, c + 32 * i);
        fe_mul(&FR, &p, &p, &x); fe_mul(&FR, &q, &q, &y); fe_add(&FR, &p, &p, &q); fe_add(&FR, &p, &p, &r); fe_mul(&FR, &p, &p, &z);
        fe_to_canonical(&FR, out + 32 * i, &p);
    }
    return 0;
}
int ref_fr_powers(const uint8_t s_b[32], const uint8_t t_b[32], uint64_t n, uint8_t* out) {
    fe p, t; fe_from_canonical(&FR, &p, s_b); fe_from_canonical(&FR, &t, t_b);
    for (uint64_t i = 0; i < n; ++i) { fe_to_canonical(&FR, out + 32 * i, &p); fe_mul(&FR, &p, &p, &t); }
    return 0;
}

/* prover.rs:256-274 */
static void calculate_coeff1(jac1* res, const jac1* initial, const aff1* query, uint64_t qlen, const aff1* vk_param,
                             const uint64_t (*assignment)[4], uint64_t alen, int nt) {
    jac1 acc; uint64_t n = qlen - 1 < alen ? qlen - 1 : alen;
    msm1(&acc, query + 1, assignment, n, nt);
    jac1_madd(res, initial, &query[0], &FQ.one);
    jac1_add(res, res, &acc);
    jac1_madd(res, res, vk_param, &FQ.one);
}
static void calculate_coeff2(jac2* res, const jac2* initial, const aff2* query, uint64_t qlen, const aff2* vk_param,
                             const uint64_t (*assignment)[4], uint64_t alen, int nt) {
    jac2 acc; uint64_t n = qlen - 1 < alen ? qlen - 1 : alen;
    msm2(&acc, query + 1, assignment, n, nt);
    jac2_madd(res, initial, &query[0], fe2_one());
    jac2_add(res, res, &acc);
    jac2_madd(res, res, vk_param, fe2_one());
}

/* create_proof_with_reduction_and_matrices (prover.rs:26-51) -> 256-byte uncompressed proof.
 * The key arrives as packed canonical arrays; converting it to Montgomery affine points is the
 * analogue of deserialising prover_params.bin (creds/src/lib.rs:268) and is timed separately (load_s). */
int ref_prove(const ref_pk* pk, const csr_t abc[3], uint64_t l, uint64_t m, uint64_t M, const uint8_t* w_bytes,
              const uint8_t r_b[32], const uint8_t s_b[32], uint8_t proof_out[256], int nthreads, ref_timings* tm) {
    double t_start = omp_get_wtime();
    aff1 *aq = (aff1*)malloc(sizeof(aff1) * pk->a_len), *b1q = (aff1*)malloc(sizeof(aff1) * pk->b_g1_len);
    aff1 *hq = (aff1*)malloc(sizeof(aff1) * pk->h_len), *lq = (aff1*)malloc(sizeof(aff1) * (pk->l_len ? pk->l_len : 1));
    aff2* b2q = (aff2*)malloc(sizeof(aff2) * pk->b_g2_len);
    load_g1s(aq, pk->a_query, pk->a_len, nthreads); load_g1s(b1q, pk->b_g1_query, pk->b_g1_len, nthreads);
    load_g1s(hq, pk->h_query, pk->h_len, nthreads); load_g1s(lq, pk->l_query, pk->l_len, nthreads);
    load_g2s(b2q, pk->b_g2_query, pk->b_g2_len, nthreads);
    aff1 alpha_g1, beta_g1, delta_g1; aff2 beta_g2, delta_g2;
    aff1_load(&alpha_g1, pk->alpha_g1); aff1_load(&beta_g1, pk->beta_g1); aff1_load(&delta_g1, pk->delta_g1);
    aff2_load(&beta_g2, pk->beta_g2); aff2_load(&delta_g2, pk->delta_g2);
    fe* w = (fe*)malloc(sizeof(fe) * M);
    _Pragma("omp parallel for num_threads(nthreads)")
    for (uint64_t i = 0; i < M; ++i) fe_from_canonical(&FR, &w[i], w_bytes + 32 * i);
    double t_loaded = omp_get_wtime();

    /* witness map (prover.rs:37-43) */
    fe* h; uint64_t D = witness_map(abc, l, m, w, &h, nthreads);
    if (!D) return -5;
    double t_wm = omp_get_wtime();

    /* into_bigint (prover.rs:63-65,70-72,84-89) */
    uint64_t (*h_big)[4] = (uint64_t(*)[4])malloc(32 * D);
    _Pragma("omp parallel for num_threads(nthreads)")
    for (uint64_t i = 0; i < D; ++i) fe_to_canonical(&FR, (uint8_t*)h_big[i], &h[i]);
    const uint64_t (*w_big)[4] = (const uint64_t(*)[4])w_bytes;   /* canonical scalars as given */
    uint64_t r[4], s[4];
    memcpy(r, r_b, 32); memcpy(s, s_b, 32);

    jac1 h_acc, l_aux_acc;
    msm1(&h_acc, hq, (const uint64_t(*)[4])h_big, pk->h_len < D ? pk->h_len : D, nthreads);   /* :66 (zip truncates) */
    double t_h = omp_get_wtime();
    msm1(&l_aux_acc, lq, w_big + l, pk->l_len < M - l ? pk->l_len : M - l, nthreads);           /* :74 */
    double t_l = omp_get_wtime();

    jac1 d1 = {delta_g1.x, delta_g1.y, FQ.one}, r_g1, rs_delta;
    jac1_mul(&r_g1, &d1, r);                                       /* :76-80, :94 */
    jac1_mul(&rs_delta, &r_g1, s);
    jac1 g_a, s_g_a;
    calculate_coeff1(&g_a, &r_g1, aq, pk->a_len, &alpha_g1, w_big + 1, M - 1, nthreads);        /* :96 */
    jac1_mul(&s_g_a, &g_a, s);                                     /* :98 */
    double t_a = omp_get_wtime();
    jac1 g1_b; jac1_set_inf(&g1_b);
    if (!big_is_zero(r)) {                                         /* :102-112 */
        jac1 s_g1; jac1_mul(&s_g1, &d1, s);
        calculate_coeff1(&g1_b, &s_g1, b1q, pk->b_g1_len, &beta_g1, w_big + 1, M - 1, nthreads);
    }
    double t_b1 = omp_get_wtime();
    jac2 d2 = {delta_g2.x, delta_g2.y, *fe2_one()}, s_g2, g2_b;
    jac2_mul(&s_g2, &d2, s);                                       /* :116 */
    calculate_coeff2(&g2_b, &s_g2, b2q, pk->b_g2_len, &beta_g2, w_big + 1, M - 1, nthreads);    /* :117 */
    jac1 r_g1_b; jac1_mul(&r_g1_b, &g1_b, r);                      /* :118 */
    double t_b2 = omp_get_wtime();
    jac1 g_c = s_g_a, neg;                                         /* :123-128 */
    jac1_add(&g_c, &g_c, &r_g1_b);
    jac1_neg(&neg, &rs_delta); jac1_add(&g_c, &g_c, &neg);
    jac1_add(&g_c, &g_c, &l_aux_acc);
    jac1_add(&g_c, &g_c, &h_acc);
    aff1 pa, pc; aff2 pb;                                          /* :131-135 */
    jac1_to_affine(&pa, &g_a); jac2_to_affine(&pb, &g2_b); jac1_to_affine(&pc, &g_c);
    ser_g1(proof_out, &pa); ser_g2(proof_out + 64, &pb); ser_g1(proof_out + 192, &pc);
    double t_end = omp_get_wtime();
    if (tm) {
        tm->load_s = t_loaded - t_start; tm->witness_map_s = t_wm - t_loaded; tm->msm_h_s = t_h - t_wm; tm->msm_l_s = t_l - t_h;
        tm->msm_a_s = t_a - t_l; tm->msm_b1_s = t_b1 - t_a; tm->msm_b2_s = t_b2 - t_b1; tm->total_s = t_end - t_loaded;
    }
    free(aq); free(b1q); free(hq); free(lq); free(b2q); free(w); free(h); free(h_big);
    return 0;
}

/* ================================================================================================
 * The files either side of the prove step, at full size (TEST INFRASTRUCTURE, as everything here).
 *
 * oracle/ark_files.py writes and reads the same formats in pure Python for the small fixtures; a
 * 0.6 GB main_c.r1cs / prover_params.bin (creds/test-vectors/README.md:5-10) needs a compiled writer
 * and, for the CPU side of bench.py's `cold_start` record, a compiled reader.  Written from the format
 * descriptions, independently of the product's parsers (csrc/r1cs.hip, csrc/serialize.hip):
 *   - iden3 .r1cs container: forks/circom-compat/src/circom/r1cs_reader.rs:54-148 (sections), :162-202
 *     (header), :205-236 (constraints), :238-256 (wire map); section 3 directly after section 2 (:125)
 *   - ark-serialize uncompressed ProvingKey: forks/groth16/src/data_structures.rs:31-44,101-118,
 *     Vec<T> = u64 LE length + items, SWFlags in the two top bits of a point's last byte [ark-mem]
 * ================================================================================================ */
static const uint8_t FR_MODULUS_BYTES[32] = { /* r1cs_reader.rs:183 */
    0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
    0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

static void put32(uint8_t** p, uint32_t v) { memcpy(*p, &v, 4); *p += 4; }
static void put64(uint8_t** p, uint64_t v) { memcpy(*p, &v, 8); *p += 8; }

uint64_t ref_r1cs_file_size(const csr_t abc[3], uint64_t m, uint64_t n_wires) {
    uint64_t cons = 12 * m + 36 * (abc[0].nnz + abc[1].nnz + abc[2].nnz);
    return 4 + 4 + 4 + 3 * 12 + 64 + cons + 8 * n_wires;
}
/* sections in circom's order: header (1), constraints (2), wire map (3) */
int ref_r1cs_write(const csr_t abc[3], uint64_t m, uint32_t n_wires, uint32_t n_pub_out, uint32_t n_pub_in, uint32_t n_prv_in,
                   uint8_t* out, uint64_t cap) {
    if (cap < ref_r1cs_file_size(abc, m, n_wires)) return -1;
    uint8_t* p = out;
    memcpy(p, "r1cs", 4); p += 4;
    put32(&p, 1); put32(&p, 3);
    put32(&p, 1); put64(&p, 64);                                   /* header section: 4 + 32 + 4*4 + 8 + 4 */
    put32(&p, 32); memcpy(p, FR_MODULUS_BYTES, 32); p += 32;
    put32(&p, n_wires); put32(&p, n_pub_out); put32(&p, n_pub_in); put32(&p, n_prv_in);
    put64(&p, n_wires); put32(&p, (uint32_t)m);
    put32(&p, 2); put64(&p, 12 * m + 36 * (abc[0].nnz + abc[1].nnz + abc[2].nnz));
    for (uint64_t i = 0; i < m; ++i)
        for (int k = 0; k < 3; ++k) {
            const uint64_t b = abc[k].row_ptr[i], e = abc[k].row_ptr[i + 1];
            put32(&p, (uint32_t)(e - b));
            for (uint64_t t = b; t < e; ++t) { put32(&p, abc[k].col[t]); memcpy(p, abc[k].coeff + 32 * t, 32); p += 32; }
        }
    put32(&p, 3); put64(&p, 8 * (uint64_t)n_wires);
    for (uint64_t i = 0; i < n_wires; ++i) put64(&p, i);
    return (uint64_t)(p - out) == ref_r1cs_file_size(abc, m, n_wires) ? 0 : -2;
}

/* Sequential reader, as `R1CSFile::new` reads: one pass over the constraint section appending to growing arrays
 * (read_constraints, r1cs_reader.rs:205-236, pushes every term onto a Vec).  Two calls: sizes, then fill. */
typedef struct { uint32_t n_wires, n_pub_out, n_pub_in, n_prv_in, n_constraints; uint64_t nnz[3]; uint64_t cons_off; } ref_r1cs_info;
int ref_r1cs_scan(const uint8_t* data, uint64_t len, ref_r1cs_info* info) {
    if (len < 12 || memcmp(data, "r1cs", 4)) return -1;
    uint32_t ver, nsec; memcpy(&ver, data + 4, 4); memcpy(&nsec, data + 8, 4);
    if (ver != 1) return -2;
    uint64_t off = 12, hdr = 0, cons = 0, cons_size = 0;
    for (uint32_t s = 0; s < nsec; ++s) {
        if (off + 12 > len) return -3;
        uint32_t ty; uint64_t sz; memcpy(&ty, data + off, 4); memcpy(&sz, data + off + 4, 8);
        off += 12;
        if (ty == 1) hdr = off;
        if (ty == 2) { cons = off; cons_size = sz; }
        if (sz > len - off) return -3;
        off += sz;
    }
    if (!hdr || !cons) return -4;
    uint32_t fs; memcpy(&fs, data + hdr, 4);
    if (fs != 32 || memcmp(data + hdr + 4, FR_MODULUS_BYTES, 32)) return -5;
    memcpy(&info->n_wires, data + hdr + 36, 4); memcpy(&info->n_pub_out, data + hdr + 40, 4);
    memcpy(&info->n_pub_in, data + hdr + 44, 4); memcpy(&info->n_prv_in, data + hdr + 48, 4);
    memcpy(&info->n_constraints, data + hdr + 60, 4);
    info->cons_off = cons;
    info->nnz[0] = info->nnz[1] = info->nnz[2] = 0;
    uint64_t p = cons, end = cons + cons_size;
    for (uint32_t i = 0; i < info->n_constraints; ++i)
        for (int k = 0; k < 3; ++k) {
            if (p + 4 > end) return -6;
            uint32_t n; memcpy(&n, data + p, 4);
            p += 4 + 36ull * n;
            if (p > end) return -6;
            info->nnz[k] += n;
        }
    return 0;
}
int ref_r1cs_fill(const uint8_t* data, const ref_r1cs_info* info, uint64_t* row_ptr[3], uint32_t* col[3], uint8_t* coeff[3]) {
    uint64_t p = info->cons_off, t[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) row_ptr[k][0] = 0;
    for (uint32_t i = 0; i < info->n_constraints; ++i)
        for (int k = 0; k < 3; ++k) {
            uint32_t n; memcpy(&n, data + p, 4); p += 4;
            for (uint32_t j = 0; j < n; ++j, p += 36) {
                memcpy(&col[k][t[k]], data + p, 4);
                memcpy(coeff[k] + 32 * t[k], data + p + 4, 32);
                uint64_t c[4]; memcpy(c, data + p + 4, 32);
                if (ge_mod(c, FR.n)) return -7;                     /* deserialize rejects a non-canonical element */
                ++t[k];
            }
            row_ptr[k][i + 1] = t[k];
        }
    return 0;
}

/* ark-serialize SWFlags of a packed canonical point, from its bytes: y > -y decided by computing q - y */
static void neg_canonical_q(uint8_t out[32], const uint8_t y[32]) {
    uint64_t a[4], r[4]; memcpy(a, y, 32);
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)FQ.n[i] - a[i] - (uint64_t)br; r[i] = (uint64_t)d; br = (d >> 64) & 1; }
    memcpy(out, r, 32);
}
static int bytes_zero(const uint8_t* b, int n) { for (int i = 0; i < n; ++i) if (b[i]) return 0; return 1; }
static uint8_t flags_g1(const uint8_t pt[64]) {
    if (bytes_zero(pt, 64)) return 0x40;
    uint8_t ny[32]; neg_canonical_q(ny, pt + 32);
    return canon_gt(pt + 32, ny) ? 0x80 : 0;
}
static uint8_t flags_g2(const uint8_t pt[128]) {
    if (bytes_zero(pt, 128)) return 0x40;
    uint8_t n0[32], n1[32];
    if (bytes_zero(pt + 96, 32)) memset(n1, 0, 32); else neg_canonical_q(n1, pt + 96);      /* -0 = 0 */
    if (bytes_zero(pt + 64, 32)) memset(n0, 0, 32); else neg_canonical_q(n0, pt + 64);
    int gt = canon_gt(pt + 96, n1) || (!memcmp(pt + 96, n1, 32) && canon_gt(pt + 64, n0));   /* c1 first, then c0 */
    return gt ? 0x80 : 0;
}
static void put_g1(uint8_t** p, const uint8_t* pt) { memcpy(*p, pt, 64); (*p)[63] |= flags_g1(pt); *p += 64; }
static void put_g2(uint8_t** p, const uint8_t* pt) { memcpy(*p, pt, 128); (*p)[127] |= flags_g2(pt); *p += 128; }
static void put_g1s(uint8_t** p, const uint8_t* pts, uint64_t n, int nt) {
    put64(p, n);
    uint8_t* base = *p;
    _Pragma("omp parallel for num_threads(nt)")
    for (uint64_t i = 0; i < n; ++i) { uint8_t* q = base + 64 * i; put_g1(&q, pts + 64 * i); }
    *p += 64 * n;
}
static void put_g2s(uint8_t** p, const uint8_t* pts, uint64_t n, int nt) {
    put64(p, n);
    uint8_t* base = *p;
    _Pragma("omp parallel for num_threads(nt)")
    for (uint64_t i = 0; i < n; ++i) { uint8_t* q = base + 128 * i; put_g2(&q, pts + 128 * i); }
    *p += 128 * n;
}
uint64_t ref_pk_file_size(const ref_pk* pk, uint64_t n_abc) {
    return 64 + 128 + 128 + 64 + 128 + 8 + 64 * n_abc + 64 + 64 + 8 + 64 * pk->a_len + 8 + 64 * pk->b_g1_len + 8 + 128 * pk->b_g2_len +
           8 + 64 * pk->h_len + 8 + 64 * pk->l_len;
}
/* ProvingKey { vk { alpha_g1, beta_g2, gamma_g2, delta_g1, delta_g2, gamma_abc_g1 }, beta_g1, delta_g1, a_query, b_g1_query,
 * b_g2_query, h_query, l_query } (data_structures.rs:31-44: the fork's vk carries delta_g1; :101-118) */
int ref_pk_write(const ref_pk* pk, const uint8_t* gamma_g2, const uint8_t* gamma_abc, uint64_t n_abc, uint8_t* out, uint64_t cap, int nt) {
    if (cap < ref_pk_file_size(pk, n_abc)) return -1;
    uint8_t* p = out;
    put_g1(&p, pk->alpha_g1); put_g2(&p, pk->beta_g2); put_g2(&p, gamma_g2); put_g1(&p, pk->delta_g1); put_g2(&p, pk->delta_g2);
    put_g1s(&p, gamma_abc, n_abc, 1);
    put_g1(&p, pk->beta_g1); put_g1(&p, pk->delta_g1);
    put_g1s(&p, pk->a_query, pk->a_len, nt); put_g1s(&p, pk->b_g1_query, pk->b_g1_len, nt); put_g2s(&p, pk->b_g2_query, pk->b_g2_len, nt);
    put_g1s(&p, pk->h_query, pk->h_len, nt); put_g1s(&p, pk->l_query, pk->l_len, nt);
    return (uint64_t)(p - out) == ref_pk_file_size(pk, n_abc) ? 0 : -2;
}
/* reader: offsets and lengths of the five queries inside a serialized ProvingKey, then a flag-stripping copy
 * (deserialize_uncompressed_unchecked: no curve checks, creds/src/utils.rs:186) */
typedef struct { uint64_t n_abc, off_abc, off_beta_g1, off_a, n_a, off_b1, n_b1, off_b2, n_b2, off_h, n_h, off_l, n_l, end; } ref_pk_layout;
int ref_pk_scan(const uint8_t* data, uint64_t len, ref_pk_layout* L) {
    uint64_t off = 64 + 128 + 128 + 64 + 128;
    if (off + 8 > len) return -1;
    memcpy(&L->n_abc, data + off, 8); off += 8; L->off_abc = off;
    if (L->n_abc > (len - off) / 64) return -1;
    off += 64 * L->n_abc;
    L->off_beta_g1 = off; off += 128;
    uint64_t* fields[5][2] = {{&L->off_a, &L->n_a}, {&L->off_b1, &L->n_b1}, {&L->off_b2, &L->n_b2}, {&L->off_h, &L->n_h}, {&L->off_l, &L->n_l}};
    for (int q = 0; q < 5; ++q) {
        const uint64_t sz = q == 2 ? 128 : 64;
        if (off + 8 > len) return -1;
        memcpy(fields[q][1], data + off, 8); off += 8;
        *fields[q][0] = off;
        if (*fields[q][1] > (len - off) / sz) return -1;
        off += sz * *fields[q][1];
    }
    L->end = off;
    return 0;
}
void ref_points_strip(const uint8_t* src, uint64_t n, uint64_t sz, uint8_t* dst, int nt) {
    _Pragma("omp parallel for num_threads(nt)")
    for (uint64_t i = 0; i < n; ++i) {
        uint8_t* d = dst + sz * i;
        memcpy(d, src + sz * i, sz);
        const uint8_t f = d[sz - 1] & 0xC0;
        d[sz - 1] &= 0x3F;
        if (f & 0x40) memset(d, 0, sz);
    }
}
