// Domain constants of ark-poly's Radix2EvaluationDomain as used by forks/groth16/src/r1cs_to_qap.rs:150-213
// (ω = 5^((r-1)/2^k), coset offset g = 5), CSR matrices on the device, and saturated-limb Fr helpers
// (internal C++ interface).  The transforms run on 29-bit limbs (wmap29.hpp).
#pragma once
#include "common.hpp"
#include "csr_host.hpp"

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

// out[p] = scale * base^(bitrev ? rev(p) : p), p < n (Montgomery form)
void fr_pow_table(Fr* out, const Fr& base, const Fr& scale, uint64_t n, bool bitrev, int logn, hipStream_t st);

// (the sliced layout and its constants are described in csr_host.hpp, which prepares it on the host)
struct SellLevel {
    uint32_t n_pieces = 0, n_partials = 0;   // partials written to the scratch vector by this level
    DevBuf<uint32_t> slice_ptr;  // n_slices + 1 slot offsets (multiples of 64)
    DevBuf<uint32_t> col;        // level 0: wire index; level k > 0: partial index in the previous level's scratch
    DevBuf<uint32_t> cidx;       // dictionary index (0 = the literal one), SELL_PAD for padding
    DevBuf<uint32_t> dst;        // per piece: row | SELL_FINAL, or its slot in this level's scratch vector
};

// CSR sparse matrix (device) with a coefficient dictionary: coef_idx 0 is the literal one
// (the reference's `coeff.is_one()` shortcut, r1cs_to_qap.rs:31-35).
struct DevCsr {
    uint64_t rows = 0, nnz = 0;
    DevBuf<uint32_t> row_ptr;   // rows + 1  (nnz < 2^32 enforced at load)
    DevBuf<uint32_t> long_rows; // rows with more than 4096 terms (saturated spmv: one workgroup each)
    uint64_t n_long_rows = 0;
    DevBuf<uint32_t> col;
    DevBuf<uint32_t> coef_idx;
    DevBuf<Fr> dict;            // Montgomery
    SellLevel sell[8];          // the sliced layout, level by level (rows of up to 8^8 terms)
    int n_sell = 0;
    uint32_t sell_scratch = 0;  // largest partial count of any level
    // sliced = false: skip the sliced layout (the generator's transposed matrices only run the saturated product)
    // st: the loader's stream (the host arrays are copied on it and waited for)
    void upload(const cg_csr& m, uint64_t rows, uint64_t num_variables, hipStream_t st, bool sliced = true);
};
// host-side transpose of a CSR view (rows x cols): CSR of the transpose, terms of one column kept in row order
struct HostCsc {
    std::vector<uint64_t> ptr;
    std::vector<uint32_t> row;
    std::vector<uint8_t> coeff;
    cg_csr view;
};
void csr_transpose(const cg_csr& m, uint64_t rows, uint64_t cols, HostCsc& out);

// out[i] = <M_i, w> for i < rows (w Montgomery), rows..n_out zero-filled except the caller's patch
void spmv(const DevCsr& m, const Fr* w, Fr* out, hipStream_t st);

}  // namespace cg
