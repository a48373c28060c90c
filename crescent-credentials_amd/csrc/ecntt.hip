// The h query moved into the evaluation basis of the coset: an inverse DFT over G1 points, once per key.
// See msm.hpp (ec_transform_h_bases) for the identity; the reference computes the same group element as
// MSM(h_query, coset_ifft(q)) (forks/groth16/src/prover.rs:63-66 after r1cs_to_qap.rs:210).
//
// Layout: n = 2^logn accumulators (XYZZ, 144 B) in one HBM buffer; a decimation-in-time pass per stage, one
// butterfly per lane, in place (a butterfly owns its two slots).  A butterfly multiplies its second operand by
// the 254-bit twiddle with a plain double-and-add over the same lazy 29-bit-limb group law the MSM kernels use
// (csrc/curve29.hpp), so the whole transform is ≈ n/2·log n·380 group operations: ≈1 s at n = 2^21, paid at load.
#include "msm.hpp"
#include "ntt.hpp"

namespace cg {

typedef Fq29 FB;   // coordinate field of G1 on 29-bit limbs
static constexpr int ACC1 = Words29<FB>::ACC;
static constexpr int AFF1 = Words29<FB>::AFF;

__device__ __forceinline__ uint32_t brev_bits(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// k·V for a canonical 256-bit scalar k (V in stored-accumulator form, not the identity)
__device__ void smul_xyzz(XYZZ29<FB>& acc, bool& inf, const XYZZ29<FB>& v, const Fr& k) {
    inf = true;
    bool started = false;
    for (int w = 7; w >= 0; --w) {
        const uint32_t kw = k.l[w];
        if (!started && kw == 0) continue;
        for (int b = 31; b >= 0; --b) {
            if (started && !inf) acc = dbl29(acc);
            if ((kw >> b) & 1u) {
                add29(acc, inf, v, false);
                started = true;
            }
        }
    }
}

// slot[rev(i)] = scale_i · H_i      (identity for i >= n_in or an identity base)
__global__ void __launch_bounds__(256) k_ec_load_scale(const uint32_t* __restrict__ row0, const uint8_t* __restrict__ valid,
                                                       uint64_t n_in, uint64_t n, int logn, const Fr* __restrict__ scale_mont,
                                                       uint32_t* __restrict__ slots) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ29<FB> acc;
    bool inf = true;
    if (i < n_in && valid[i]) {
        const Affine29<FB> p = load_table_point<FB>(row0, (uint32_t)i, false);
        const Fr k = from_mont(scale_mont[i]);
        bool started = false;
        for (int w = 7; w >= 0; --w) {
            const uint32_t kw = k.l[w];
            if (!started && kw == 0) continue;
            for (int b = 31; b >= 0; --b) {
                if (started && !inf) acc = dbl29(acc);
                if ((kw >> b) & 1u) {
                    madd29(acc, inf, p);
                    started = true;
                }
            }
        }
    }
    store_acc(slots + (uint64_t)brev_bits((uint32_t)i, logn) * ACC1, acc, inf);
}

// stage q of the DIT transform with root w = ω^-1:  (U, V) <- (U + w^e·V, U − w^e·V),  e = k·n / 2^(q+1)
__global__ void __launch_bounds__(256) k_ec_stage(uint32_t* __restrict__ slots, uint64_t n, int logn, int q,
                                                  const Fr* __restrict__ tw_inv_mont) {
    uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (n >> 1)) return;
    const uint64_t half = 1ull << q;
    const uint64_t k = b & (half - 1);
    const uint64_t pos = ((b >> q) << (q + 1)) | k;
    XYZZ29<FB> u, v;
    const bool uinf = load_acc(slots + pos * ACC1, u);
    const bool vinf = load_acc(slots + (pos + half) * ACC1, v);
    if (vinf) {                      // w^e·V = identity: both outputs are U
        store_acc(slots + (pos + half) * ACC1, u, uinf);
        return;
    }
    XYZZ29<FB> t = v;
    bool tinf = false;
    if (k != 0) smul_xyzz(t, tinf, v, from_mont(tw_inv_mont[k << (logn - 1 - q)]));
    XYZZ29<FB> nt = t;
    nt.y = normalize(sub<KY, 1>(FB::zero(), t.y));          // −T  (t.y < 8N <= KY·N)
    XYZZ29<FB> s = u, d = u;
    bool sinf = uinf, dinf = uinf;
    add29(s, sinf, t, tinf);
    add29(d, dinf, nt, tinf);
    store_acc(slots + pos * ACC1, s, sinf);
    store_acc(slots + (pos + half) * ACC1, d, dinf);
}

// accumulators -> packed affine table points + validity
__global__ void __launch_bounds__(256) k_ec_store(const uint32_t* __restrict__ slots, uint64_t n, uint32_t* __restrict__ row0,
                                                  uint8_t* __restrict__ valid) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    XYZZ29<FB> a;
    const bool inf = load_acc(slots + j * ACC1, a);
    uint32_t w[AFF1];
    if (inf) {
#pragma unroll
        for (int k = 0; k < AFF1; ++k) w[k] = 0;
    } else {
        store_table_point_from_xyzz(a, w);
    }
    uint4* dst = reinterpret_cast<uint4*>(row0 + j * AFF1);
#pragma unroll
    for (int k = 0; k < AFF1 / 4; ++k) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    valid[j] = inf ? 0 : 1;
}

void ec_inverse_dft(const uint32_t* row0_in, const uint8_t* valid_in, uint64_t n_in, int logn, const Fr& scale_base,
                    const Fr& scale_mult, uint32_t* row0_out, uint8_t* valid_out, hipStream_t st) {
    const uint64_t n = 1ull << logn;
    if (n_in > n) throw HipError(CG_ERR_INVALID_ARGUMENT, "h query longer than the domain");
    const Fr w = fr_root_of_unity(logn);
    DevBuf<Fr> scale(n), tw(n > 1 ? n / 2 : 1);
    fr_pow_table(scale.p, scale_base, scale_mult, n, false, logn, st);                // mult · base^i
    fr_pow_table(tw.p, inv(w), Fr::one(), n > 1 ? n / 2 : 1, false, logn, st);        // ω^-e
    DevBuf<uint32_t> slots(n * ACC1);
    k_ec_load_scale<<<ceil_div(n, 256), 256, 0, st>>>(row0_in, valid_in, n_in, n, logn, scale.p, slots.p);
    CG_KERNEL_CHECK();
    for (int q = 0; q < logn; ++q) {
        k_ec_stage<<<ceil_div(n >> 1, 256), 256, 0, st>>>(slots.p, n, logn, q, tw.p);
        CG_KERNEL_CHECK();
    }
    k_ec_store<<<ceil_div(n, 256), 256, 0, st>>>(slots.p, n, row0_out, valid_out);
    CG_KERNEL_CHECK();
    CG_HIP(hipStreamSynchronize(st));     // the temporaries are released on return
}

void ec_transform_h_bases(const uint32_t* row0_in, const uint8_t* valid_in, uint64_t n_in, int logn, uint32_t* row0_out,
                          uint8_t* valid_out, hipStream_t st) {
    const Fr g = fr_from_u64(5);                               // F::GENERATOR (r1cs_to_qap.rs:182,202)
    ec_inverse_dft(row0_in, valid_in, n_in, logn, inv(g), inv(fr_from_u64(1ull << logn)), row0_out, valid_out, st);   // g^-i / n
}

// ---- the C matrix folded into the l query ----------------------------------------------------------------------------
// term t of C^T (wire k = key[t], constraint j = col[t], coefficient c):  c · G'_j
__global__ void __launch_bounds__(256) k_ec_terms(const uint32_t* __restrict__ col, const uint32_t* __restrict__ cidx,
                                                  const Fr* __restrict__ dict_mont, uint64_t nnz, const uint32_t* __restrict__ g_row0,
                                                  const uint8_t* __restrict__ g_valid, uint32_t* __restrict__ pts) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz) return;
    const uint32_t j = col[t];
    XYZZ29<FB> acc;
    bool inf = true;
    if (g_valid[j]) {
        const Affine29<FB> p = load_table_point<FB>(g_row0, j, false);
        const uint32_t ci = cidx[t];
        if (ci == 0) {                       // the literal one
            madd29(acc, inf, p);
        } else {
            const Fr k = from_mont(dict_mont[ci]);
            bool started = false;
            for (int w = 7; w >= 0; --w) {
                const uint32_t kw = k.l[w];
                if (!started && kw == 0) continue;
                for (int b = 31; b >= 0; --b) {
                    if (started && !inf) acc = dbl29(acc);
                    if ((kw >> b) & 1u) {
                        madd29(acc, inf, p);
                        started = true;
                    }
                }
            }
        }
    }
    store_acc(pts + t * ACC1, acc, inf);
}

// new l base k = P_k (+ l_query[k − num_inputs] for a witness wire), as a packed table point
__global__ void __launch_bounds__(256) k_ec_fold_l(const uint32_t* __restrict__ p_sums, const uint32_t* __restrict__ l_row0,
                                                   const uint8_t* __restrict__ l_valid, uint64_t num_inputs, uint64_t M,
                                                   uint32_t* __restrict__ row0, uint8_t* __restrict__ valid) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    XYZZ29<FB> acc;
    bool inf = load_acc(p_sums + k * ACC1, acc);
    if (k >= num_inputs && l_valid[k - num_inputs]) madd29(acc, inf, load_table_point<FB>(l_row0, (uint32_t)(k - num_inputs), false));
    uint32_t w[AFF1];
    if (inf) {
#pragma unroll
        for (int i = 0; i < AFF1; ++i) w[i] = 0;
    } else {
        store_table_point_from_xyzz(acc, w);
    }
    uint4* dst = reinterpret_cast<uint4*>(row0 + k * AFF1);
#pragma unroll
    for (int i = 0; i < AFF1 / 4; ++i) dst[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
    valid[k] = inf ? 0 : 1;
}

void ec_fold_c_into_l(const uint32_t* h_row0, const uint8_t* h_valid, uint64_t n_h, int logn, const Fr& vanishing_inv,
                      const cg_csr& c_matrix, uint64_t num_constraints, uint64_t num_inputs, uint64_t M,
                      const uint32_t* l_row0, const uint8_t* l_valid, uint32_t* row0_out, uint8_t* valid_out, hipStream_t st) {
    HostCsc t;
    if (c_matrix.nnz && num_constraints) csr_transpose(c_matrix, num_constraints, M, t);
    else { t.ptr.assign(M + 1, 0); t.view = cg_csr{t.ptr.data(), nullptr, nullptr, 0}; }
    ec_fold_ct_into_l(h_row0, h_valid, n_h, logn, vanishing_inv, t, num_constraints, num_inputs, M, l_row0, l_valid, row0_out, valid_out, st);
}

// the same with the transpose of C already made (a staged load makes it while the caller's arrays are still there and
// folds later, on its worker thread)
void ec_fold_ct_into_l(const uint32_t* h_row0, const uint8_t* h_valid, uint64_t n_h, int logn, const Fr& vanishing_inv,
                       const HostCsc& t, uint64_t num_constraints, uint64_t num_inputs, uint64_t M,
                       const uint32_t* l_row0, const uint8_t* l_valid, uint32_t* row0_out, uint8_t* valid_out, hipStream_t st) {
    const uint64_t n = 1ull << logn;
    // G'_j = −vinv/n · Σ_i ω^{-ij} H_i : the h query in the Lagrange basis of the domain, times −1/Z(g)
    DevBuf<uint32_t> g_row0(n * AFF1);
    DevBuf<uint8_t> g_valid(n);
    ec_inverse_dft(h_row0, h_valid, n_h, logn, Fr::one(), neg(mul(vanishing_inv, inv(fr_from_u64(n)))), g_row0.p, g_valid.p, st);
    // P_k = Σ_j C_jk · G'_j : the terms of C^T, summed per wire
    DevBuf<uint32_t> p_sums(M * ACC1);
    fill_zero(p_sums.p, p_sums.bytes(), st);
    const uint64_t nnz = t.view.nnz;
    if (nnz && num_constraints) {
        DevCsr ct;
        ct.upload(t.view, M, num_constraints, st, false);
        std::vector<uint32_t> keys_h(nnz);
        for (uint64_t k = 0; k < M; ++k)
            for (uint64_t e = t.ptr[k]; e < t.ptr[k + 1]; ++e) keys_h[e] = (uint32_t)k;
        DevBuf<uint32_t> keys(nnz), pts(nnz * ACC1);
        CG_HIP(hipMemcpyAsync(keys.p, keys_h.data(), nnz * 4, hipMemcpyHostToDevice, st));
        k_ec_terms<<<ceil_div(nnz, 256), 256, 0, st>>>(ct.col.p, ct.coef_idx.p, ct.dict.p, nnz, g_row0.p, g_valid.p, pts.p);
        CG_KERNEL_CHECK();
        sum_xyzz_by_key(keys.p, pts.p, nnz, p_sums.p, st);
        CG_HIP(hipStreamSynchronize(st));    // keys_h, ct and the device temporaries above outlive the kernels
    }
    k_ec_fold_l<<<ceil_div(M, 256), 256, 0, st>>>(p_sums.p, l_row0, l_valid, num_inputs, M, row0_out, valid_out);
    CG_KERNEL_CHECK();
    CG_HIP(hipStreamSynchronize(st));
}

}  // namespace cg
