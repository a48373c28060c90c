// Radix-2 NTT over BN254 Fr and the R1CS->QAP witness map on the GPU (internal C++ interface).
// Restates ark-poly's Radix2EvaluationDomain as used by
// forks/groth16/src/r1cs_to_qap.rs:150-213 (ω = 5^((r-1)/2^k), coset offset g = 5).
#pragma once
#include "common.hpp"

namespace cg {

struct NttDomain {
    int logn = 0;
    uint64_t n = 0;
    DevBuf<Fr> tw_fwd;     // ω^e,  e < n/2          (Montgomery)
    DevBuf<Fr> tw_inv;     // ω^-e, e < n/2
    DevBuf<Fr> coset_br;   // g^{rev(p)} / n  at position p  (coset pre-scale for bit-reversed coeffs)
    DevBuf<Fr> icoset_br;  // g^{-rev(p)} / n at position p
    Fr vanishing_inv;      // (g^n - 1)^-1, Montgomery (r1cs_to_qap.rs:201-204)
    void build(int logn, bool with_coset, hipStream_t st);
};

// host-side Fr helpers (Montgomery form)
Fr fr_from_u64(uint64_t v);
Fr fr_pow_u64(const Fr& a, uint64_t e);
Fr fr_root_of_unity(int logn);  // primitive 2^logn-th root, = 5^((r-1)/2^28) ^ 2^(28-logn)

// natural order in -> bit-reversed order out (decimation in frequency).  `premul` (optional) is
// multiplied into element p as it is first loaded.
void ntt_dif(Fr* data, const NttDomain& d, bool inverse, const Fr* premul, hipStream_t st);
// bit-reversed order in -> natural order out (decimation in time)
void ntt_dit(Fr* data, const NttDomain& d, bool inverse, const Fr* premul, hipStream_t st);

// out[rev(p)] = in[p] * scale[p] (scale optional), optionally leaving Montgomery form
void ntt_unbitrev_scale(const Fr* in, Fr* out, const Fr* scale, int logn, bool to_canonical, hipStream_t st);

// elementwise helpers (n elements)
void fr_to_mont(const Fr* in, Fr* out, uint64_t n, hipStream_t st);      // canonical -> Montgomery
void fr_from_mont(const Fr* in, Fr* out, uint64_t n, hipStream_t st);    // Montgomery -> canonical
void fr_mul_vec(Fr* a, const Fr* b, uint64_t n, hipStream_t st);          // a[i] *= b[i]
void fr_pow_table(Fr* out, const Fr& base, const Fr& scale, uint64_t n, bool bitrev, int logn, hipStream_t st);
// ab[i] = (a[i]*b[i] - c[i]) * vinv     (r1cs_to_qap.rs:187,205-208)
void qap_pointwise(const Fr* a, const Fr* b, const Fr* c, Fr* out, const Fr& vinv, uint64_t n, hipStream_t st);

// CSR sparse matrix (device) with a coefficient dictionary: coef_idx 0 is the literal one
// (the reference's `coeff.is_one()` shortcut, r1cs_to_qap.rs:31-35).
struct DevCsr {
    uint64_t rows = 0, nnz = 0;
    DevBuf<uint32_t> row_ptr;   // rows + 1  (nnz < 2^32 enforced at load)
    DevBuf<uint32_t> col;
    DevBuf<uint32_t> coef_idx;
    DevBuf<Fr> dict;            // Montgomery
    void upload(const cg_csr& m, uint64_t rows, uint64_t num_variables);
};
// out[i] = <M_i, w> for i < rows (w Montgomery), rows..n_out zero-filled except the caller's patch
void spmv(const DevCsr& m, const Fr* w, Fr* out, hipStream_t st);

}  // namespace cg
