// Groth16 trusted setup from explicit toxic waste, on the GPU.
//
// Restates `generate_parameters_with_qap` (forks/groth16/src/generator.rs:50-228) and
// `LibsnarkReduction::instance_map_with_evaluation` / `h_query_scalars`
// (forks/groth16/src/r1cs_to_qap.rs:106-148,215-225) with gamma = 1 and the standard generators, as
// the fork fixes them (generator.rs:28,34-35).  The reference walks the constraints and scatters
// u_i * coeff into a/b/c (r1cs_to_qap.rs:135-145); here the matrices are transposed once on the host
// and a_j(tau) = (A^T u)_j is a gather-only sparse product.  The six FixedBase::msm calls
// (generator.rs:140,162,168,174,185,194) become one fixed-base kernel over an 8-bit window table.
#include <memory>

#include "msm.hpp"
#include "ntt.hpp"

namespace cg {
int translate_current_exception();
}
using namespace cg;

// The G2 generator is parsed from its decimal strings at start-up instead of trusting hand-copied limbs.
static Fq fq_from_decimal(const char* s) {
    // acc = acc*10 + digit, in Montgomery form
    Fq acc = Fq::zero();
    Fq ten = Fq::zero();
    ten.l[0] = 10;
    ten = to_mont(ten);
    for (const char* p = s; *p; ++p) {
        Fq d = Fq::zero();
        d.l[0] = (uint32_t)(*p - '0');
        acc = add(mul(acc, ten), to_mont(d));
    }
    return acc;
}
static G1Affine g1_generator() {
    Fq x = Fq::one();
    Fq y = add(Fq::one(), Fq::one());
    return {x, y};
}
static G2Affine g2_generator() {
    return {{fq_from_decimal("10857046999023057135944570762232829481370756359578518086990519993285655852781"),
             fq_from_decimal("11559732032986387107991004021392285783925812861821192530917403151452391805634")},
            {fq_from_decimal("8495653923123431417604973247489272438418190587263600148770280649306958101930"),
             fq_from_decimal("4082367875863433681332203403145435568316851327593401208105741076214120093531")}};
}

static constexpr int FB_NWIN = 32;   // 32 windows of 8 bits

// table[j*256 + d] = d * 2^(8j) * gen
template <class F>
__global__ void __launch_bounds__(256) k_fb_table(Affine<F> gen, Affine<F>* __restrict__ table) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= FB_NWIN * 256) return;
    uint32_t j = t >> 8, d = t & 255u;
    if (d == 0) {
        table[t] = Affine<F>::inf();
        return;
    }
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        // k = d << (8 j): place the byte
        int bitpos = (int)j * 8;
        if ((bitpos >> 5) == w) k[w] = d << (bitpos & 31);
    }
    XYZZ<F> p = scalar_mul(XYZZ<F>::from_affine(gen), k);
    table[t] = to_affine(p);
}

template <class F> struct CoordsOf;
template <> struct CoordsOf<Fq> { static __device__ void to_canonical(Affine<Fq>& p) { p.x = from_mont(p.x); p.y = from_mont(p.y); } };
template <> struct CoordsOf<Fq2> {
    static __device__ void to_canonical(Affine<Fq2>& p) {
        p.x.c0 = from_mont(p.x.c0); p.x.c1 = from_mont(p.x.c1);
        p.y.c0 = from_mont(p.y.c0); p.y.c1 = from_mont(p.y.c1);
    }
};

// out[i] = scalars[i] * gen  (scalars in Montgomery form), written as canonical packed affine
template <class F>
__global__ void __launch_bounds__(256) k_fixed_base(const Affine<F>* __restrict__ table, const Fr* __restrict__ scalars,
                                                    uint64_t n, Affine<F>* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = from_mont(scalars[i]);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int j = 0; j < FB_NWIN; ++j) {
        uint32_t d = 0;   // byte j of the scalar, selected without a dynamic register index
#pragma unroll
        for (int w = 0; w < 8; ++w) d = ((j >> 2) == w) ? s.l[w] : d;
        d = (d >> ((j & 3) * 8)) & 255u;
        if (d) madd(acc, table[j * 256 + d]);
    }
    Affine<F> p = to_affine(acc);
    CoordsOf<F>::to_canonical(p);
    out[i] = p;
}

// u_i = (zt / D) * w_i / (tau - w_i),  w_i = omega^i   (evaluate_all_lagrange_coefficients [ark-mem])
__global__ void __launch_bounds__(256) k_lagrange(const Fr* __restrict__ wpow, Fr tau, Fr zt_over_d, uint64_t n, Fr* __restrict__ u) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr w = wpow[i];
    Fr den = sub(tau, w);
    u[i] = mul(mul(zt_over_d, w), inv(den));
}

// a[i] += u[m + i], i < l   (r1cs_to_qap.rs:128-133)
__global__ void k_add_inputs(Fr* a, const Fr* u, uint64_t m, uint64_t l) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < l) a[i] = add(a[i], u[m + i]);
}

// out[i] = (beta a_i + alpha b_i + c_i) * (i < l ? gamma_inv : delta_inv)   (generator.rs:118-128)
__global__ void __launch_bounds__(256) k_abc_combine(const Fr* a, const Fr* b, const Fr* c, Fr alpha, Fr beta, Fr ginv, Fr dinv,
                                                     uint64_t l, uint64_t M, Fr* out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    Fr t = add(add(mul(beta, a[i]), mul(alpha, b[i])), c[i]);
    out[i] = mul(t, i < l ? ginv : dinv);
}

template <class F>
static void fixed_base_to_host(const Affine<F>* table, const Fr* scalars_dev, uint64_t n, uint8_t* host_out, hipStream_t st) {
    if (!n) return;
    DevBuf<Affine<F>> out(n);
    k_fixed_base<F><<<ceil_div(n, 256), 256, 0, st>>>(table, scalars_dev, n, out.p);
    CG_KERNEL_CHECK();
    CG_HIP(hipMemcpyAsync(host_out, out.p, n * sizeof(Affine<F>), hipMemcpyDeviceToHost, st));
    CG_HIP(hipStreamSynchronize(st));
}

static Fr fr_import_canonical(const uint8_t* b, const char* what) {
    Fr a = fp_from_bytes<Fr>(b);
    if (!fp_is_canonical(a)) throw HipError(CG_ERR_INVALID_ARGUMENT, std::string(what) + " not canonical");
    return to_mont(a);
}

extern "C" int cg_setup(const cg_csr abc[3], uint64_t num_inputs, uint64_t num_constraints, uint64_t num_variables,
                        const uint8_t tau_b[32], const uint8_t alpha_b[32], const uint8_t beta_b[32], const uint8_t delta_b[32],
                        uint8_t* a_query, uint8_t* b_g1_query, uint8_t* b_g2_query, uint8_t* h_query, uint8_t* l_query,
                        uint8_t* gamma_abc_g1, uint8_t vk_points[576]) {
    if (!abc || !tau_b || !alpha_b || !beta_b || !delta_b || !a_query || !b_g1_query || !b_g2_query || !h_query || !l_query ||
        !gamma_abc_g1 || !vk_points)
        return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    const uint64_t l = num_inputs, m = num_constraints, M = num_variables;
    if (l == 0 || l > M) return fail(CG_ERR_INVALID_ARGUMENT, "need 1 <= num_inputs <= num_variables");
    const int logD = ilog2_ceil(m + l);
    if (logD > 28) return fail(CG_ERR_POLY_DEGREE_TOO_LARGE, "domain too large");
    const uint64_t D = 1ull << logD;
    try {
        hipStream_t st = nullptr;
        Fr tau = fr_import_canonical(tau_b, "tau"), alpha = fr_import_canonical(alpha_b, "alpha");
        Fr beta = fr_import_canonical(beta_b, "beta"), delta = fr_import_canonical(delta_b, "delta");
        if (delta.is_zero()) throw HipError(CG_ERR_INVALID_ARGUMENT, "delta must be invertible (SynthesisError::UnexpectedIdentity)");
        Fr zt = sub(fr_pow_u64(tau, D), Fr::one());                       // evaluate_vanishing_polynomial (r1cs_to_qap.rs:115)
        if (zt.is_zero()) throw HipError(CG_ERR_INVALID_ARGUMENT, "tau lies in the evaluation domain");
        Fr dinv = inv(delta);
        Fr ginv = Fr::one();                                              // gamma = 1 (generator.rs:28)
        // Lagrange coefficients u_i = L_i(tau)
        DevBuf<Fr> u(D), wpow(D);
        fr_pow_table(wpow.p, fr_root_of_unity(logD), Fr::one(), D, false, logD, st);
        k_lagrange<<<ceil_div(D, 256), 256, 0, st>>>(wpow.p, tau, mul(zt, inv(fr_from_u64(D))), D, u.p);
        CG_KERNEL_CHECK();
        wpow.release();
        // a, b, c = A^T u, B^T u, C^T u   (r1cs_to_qap.rs:124-145)
        DevBuf<Fr> qa(M), qb(M), qc(M);
        {
            Fr* q[3] = {qa.p, qb.p, qc.p};
            for (int k = 0; k < 3; ++k) {
                HostCsc t;
                csr_transpose(abc[k], m, M, t);
                DevCsr d;
                d.upload(t.view, M, m, st, false);
                spmv(d, u.p, q[k], st);
                CG_HIP(hipStreamSynchronize(st));
            }
        }
        k_add_inputs<<<ceil_div(l, 256), 256, 0, st>>>(qa.p, u.p, m, l);
        CG_KERNEL_CHECK();
        // gamma_abc (i < l) and l-query scalars (i >= l)
        DevBuf<Fr> comb(M);
        k_abc_combine<<<ceil_div(M, 256), 256, 0, st>>>(qa.p, qb.p, qc.p, alpha, beta, ginv, dinv, l, M, comb.p);
        CG_KERNEL_CHECK();
        // h-query scalars zt/delta * tau^i, i < D-1   (r1cs_to_qap.rs:215-225, generator.rs:174-179)
        DevBuf<Fr> hs(D);
        fr_pow_table(hs.p, tau, mul(zt, dinv), D - 1, false, 0, st);
        // fixed-base tables
        DevBuf<G1Affine> t1(FB_NWIN * 256);
        DevBuf<G2Affine> t2(FB_NWIN * 256);
        k_fb_table<Fq><<<FB_NWIN, 256, 0, st>>>(g1_generator(), t1.p);
        CG_KERNEL_CHECK();
        k_fb_table<Fq2><<<FB_NWIN, 256, 0, st>>>(g2_generator(), t2.p);
        CG_KERNEL_CHECK();
        fixed_base_to_host<Fq>(t1.p, qa.p, M, a_query, st);               // generator.rs:162
        fixed_base_to_host<Fq>(t1.p, qb.p, M, b_g1_query, st);            // :168
        fixed_base_to_host<Fq2>(t2.p, qb.p, M, b_g2_query, st);           // :140
        fixed_base_to_host<Fq>(t1.p, hs.p, D - 1, h_query, st);           // :174-179
        fixed_base_to_host<Fq>(t1.p, comb.p + l, M - l, l_query, st);     // :185
        fixed_base_to_host<Fq>(t1.p, comb.p, l, gamma_abc_g1, st);        // :194
        // single points: alpha_g1 ‖ beta_g1 ‖ delta_g1 ‖ beta_g2 ‖ gamma_g2 ‖ delta_g2   (:150-154,191)
        Fr singles_h[4] = {alpha, beta, delta, ginv /* gamma = 1 */};
        DevBuf<Fr> singles(4);
        CG_HIP(hipMemcpyAsync(singles.p, singles_h, sizeof(singles_h), hipMemcpyHostToDevice, st));
        uint8_t g1s[3 * 64], g2s[3 * 128];
        fixed_base_to_host<Fq>(t1.p, singles.p, 3, g1s, st);
        fixed_base_to_host<Fq2>(t2.p, singles.p + 1, 3, g2s, st);        // beta, delta, gamma(=1)
        memcpy(vk_points, g1s, 192);
        memcpy(vk_points + 192, g2s, 128);            // beta_g2
        memcpy(vk_points + 320, g2s + 256, 128);      // gamma_g2
        memcpy(vk_points + 448, g2s + 128, 128);      // delta_g2
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}

// out[i] = scalars[i] * G (canonical affine), G the standard generator of G1 / G2.
template <class F>
static int fixed_base_export(const Affine<F>& gen, const uint8_t* scalars, uint64_t n, uint8_t* out) {
    if (n == 0) return CG_OK;
    if (!scalars || !out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    try {
        hipStream_t st = nullptr;
        std::vector<Fr> host(n);
        for (uint64_t i = 0; i < n; ++i) {
            Fr a = fp_from_bytes<Fr>(scalars + 32 * i);
            if (!fp_is_canonical(a)) return fail(CG_ERR_INVALID_ARGUMENT, "scalar %llu not canonical", (unsigned long long)i);
            host[i] = to_mont(a);
        }
        DevBuf<Fr> sc(n);
        CG_HIP(hipMemcpyAsync(sc.p, host.data(), n * sizeof(Fr), hipMemcpyHostToDevice, st));
        DevBuf<Affine<F>> table(FB_NWIN * 256);
        k_fb_table<F><<<FB_NWIN, 256, 0, st>>>(gen, table.p);
        CG_KERNEL_CHECK();
        fixed_base_to_host<F>(table.p, sc.p, n, out, st);
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}
extern "C" int cg_fixed_base_g1(const uint8_t* scalars, uint64_t n, uint8_t* out) { return fixed_base_export<Fq>(g1_generator(), scalars, n, out); }
extern "C" int cg_fixed_base_g2(const uint8_t* scalars, uint64_t n, uint8_t* out) { return fixed_base_export<Fq2>(g2_generator(), scalars, n, out); }
