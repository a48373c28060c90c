// Saturated-limb (8 x 32-bit, R = 2^256) Fr plumbing around the transforms: the domain constants
// (twiddles, coset powers, vanishing-polynomial inverse) of ark-poly's radix-2 domain as driven by
// forks/groth16/src/r1cs_to_qap.rs:150-213, the CSR upload with its coefficient dictionary, and the
// saturated sparse product used by cg_setup.  The transforms themselves run on 29-bit limbs: csrc/wmap29.hip.

#include "ntt.hpp"

#include <algorithm>

namespace cg {

// ---------------------------------------------------------------------------------------------
// host-side Fr helpers
// ---------------------------------------------------------------------------------------------
Fr fr_from_u64(uint64_t v) {
    Fr a = Fr::zero();
    a.l[0] = (uint32_t)v;
    a.l[1] = (uint32_t)(v >> 32);
    return to_mont(a);
}
Fr fr_pow_u64(const Fr& a, uint64_t e) {
    Fr r = Fr::one();
    for (int b = 63; b >= 0; --b) {
        r = sqr(r);
        if ((e >> b) & 1ull) r = mul(r, a);
    }
    return r;
}
Fr fr_root_of_unity(int logn) {
    // 5^((r-1)/2^28); the exponent (r-1) >> 28 is derived from FrP::N to avoid a transcription error
    uint32_t e[8];
    uint32_t nm1[8];
    for (int i = 0; i < 8; ++i) nm1[i] = FrP::N[i];
    nm1[0] -= 1u;  // r is odd, no borrow
    for (int i = 0; i < 8; ++i) {
        uint32_t lo = nm1[i] >> 28;
        uint32_t hi = (i + 1 < 8) ? (nm1[i + 1] << 4) : 0u;
        e[i] = lo | hi;
    }
    Fr g = fr_from_u64(5);
    Fr w = pow_limbs(g, e);  // primitive 2^28-th root
    for (int i = 28; i > logn; --i) w = sqr(w);
    return w;
}

// ---------------------------------------------------------------------------------------------
// elementwise kernels
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bitrev32(uint32_t x, int bits) { return __brev(x) >> (32 - bits); }

// out[p] = scale * base^(bitrev ? rev(p) : p)
__global__ void __launch_bounds__(256) k_pow_table(Fr* out, Fr base, Fr scale, uint64_t n, int bitrev, int logn) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t e = bitrev ? bitrev32((uint32_t)p, logn) : (uint32_t)p;
    Fr r = scale;
    Fr b = base;
    while (e) {
        if (e & 1u) r = mul(r, b);
        b = sqr(b);
        e >>= 1;
    }
    out[p] = r;
}
static inline dim3 grid_for(uint64_t n) { return dim3(ceil_div(n, 256)); }

__global__ void __launch_bounds__(256) k_fill_zero(uint4* __restrict__ p, uint64_t n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) p[i] = z;
}
void fill_zero(void* dst, size_t bytes, hipStream_t st) {
    if (!bytes) return;
    if (bytes % 16) { CG_HIP(hipMemsetAsync(dst, 0, bytes, st)); return; }
    const uint64_t n16 = bytes / 16;
    uint32_t blocks = ceil_div(n16, 256 * 8);
    if (blocks > 256 * 16) blocks = 256 * 16;
    k_fill_zero<<<blocks ? blocks : 1, 256, 0, st>>>(reinterpret_cast<uint4*>(dst), n16);
    CG_KERNEL_CHECK();
}

void fr_pow_table(Fr* out, const Fr& base, const Fr& scale, uint64_t n, bool bitrev, int logn, hipStream_t st) {
    if (!n) return;
    k_pow_table<<<grid_for(n), 256, 0, st>>>(out, base, scale, n, bitrev ? 1 : 0, logn);
    CG_KERNEL_CHECK();
}
// ---------------------------------------------------------------------------------------------
// domain tables
// ---------------------------------------------------------------------------------------------
void NttDomain::build(int logn_, bool with_coset, hipStream_t st) {
    if (logn_ > 28) throw HipError(CG_ERR_POLY_DEGREE_TOO_LARGE, "domain larger than 2^28 (Fr two-adicity)");
    logn = logn_;
    n = 1ull << logn;
    Fr w = fr_root_of_unity(logn);
    Fr wi = inv(w);
    uint64_t half = n > 1 ? n / 2 : 1;
    tw_fwd.alloc(half);
    tw_inv.alloc(half);
    fr_pow_table(tw_fwd.p, w, Fr::one(), half, false, logn, st);
    fr_pow_table(tw_inv.p, wi, Fr::one(), half, false, logn, st);
    Fr g = fr_from_u64(5);  // F::GENERATOR of ark-bn254 Fr (coset offset, r1cs_to_qap.rs:182)
    Fr ninv = inv(fr_from_u64(n));
    if (with_coset) {
        coset_br.alloc(n);
        icoset_br.alloc(n);
        fr_pow_table(coset_br.p, g, ninv, n, true, logn, st);
        fr_pow_table(icoset_br.p, inv(g), ninv, n, true, logn, st);
    }
    // (g^n - 1)^-1
    Fr gn = fr_pow_u64(g, n);
    vanishing_inv = inv(sub(gn, Fr::one()));
}

// ---------------------------------------------------------------------------------------------
// sparse matrix-vector product (evaluate_constraint, r1cs_to_qap.rs:16-45)
// ---------------------------------------------------------------------------------------------
void csr_transpose(const cg_csr& m, uint64_t rows, uint64_t cols, HostCsc& out) {
    out.ptr.assign(cols + 1, 0);
    for (uint64_t t = 0; t < m.nnz; ++t) {
        if (m.col[t] >= cols) throw HipError(CG_ERR_INVALID_ARGUMENT, "column index out of range");
        out.ptr[m.col[t] + 1]++;
    }
    for (uint64_t j = 0; j < cols; ++j) out.ptr[j + 1] += out.ptr[j];
    out.row.resize(m.nnz ? m.nnz : 1);
    out.coeff.resize((m.nnz ? m.nnz : 1) * 32);
    std::vector<uint64_t> cur(out.ptr.begin(), out.ptr.end() - 1);
    for (uint64_t i = 0; i < rows; ++i)
        for (uint64_t t = m.row_ptr[i]; t < m.row_ptr[i + 1]; ++t) {
            uint64_t pos = cur[m.col[t]]++;
            out.row[pos] = (uint32_t)i;
            memcpy(&out.coeff[pos * 32], m.coeff + 32 * t, 32);
        }
    out.view.row_ptr = out.ptr.data();
    out.view.col = out.row.data();
    out.view.coeff = out.coeff.data();
    out.view.nnz = m.nnz;
}


void DevCsr::upload(const cg_csr& m, uint64_t rows_, uint64_t num_variables, hipStream_t st, bool sliced) {
    // host side (csr_host.hpp, all host threads): validation, coefficient dictionary, sliced layout
    static const bool plain = CG_TUNE_ENV("SELL_PLAIN") != nullptr;       // A/B aid (tuning builds): round 2's layout (terms as given, pieces by length)
    HostCsr h;
    csr_prepare_host(m, rows_, num_variables, sliced, !plain, h);
    rows = rows_;
    nnz = m.nnz;
    row_ptr.alloc(rows + 1);
    n_long_rows = h.long_rows.size();
    long_rows.alloc(h.long_rows.size() ? h.long_rows.size() : 1);
    col.alloc(nnz ? nnz : 1);
    coef_idx.alloc(nnz ? nnz : 1);
    dict.alloc(h.dict_mont.size());
    // the copies go out back to back and are waited for once (the host arrays live until then)
    auto h2d = [&](void* dst, const void* src, size_t bytes) { if (bytes) CG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st)); };
    h2d(long_rows.p, h.long_rows.data(), h.long_rows.size() * 4);
    h2d(row_ptr.p, h.rp.data(), (rows + 1) * 4);
    h2d(col.p, m.col, nnz * 4);
    h2d(coef_idx.p, h.idx.p, nnz * 4);
    h2d(dict.p, h.dict_mont.data(), h.dict_mont.size() * sizeof(Fr));
    n_sell = (int)h.levels.size();
    sell_scratch = h.sell_scratch;
    for (int k = 0; k < n_sell; ++k) {
        const HostSellLevel& H = h.levels[k];
        SellLevel& L = sell[k];
        L.n_pieces = H.n_pieces;
        L.n_partials = H.n_partials;
        L.slice_ptr.alloc(H.slice_ptr.size()); L.col.alloc(H.n_slots); L.cidx.alloc(H.n_slots); L.dst.alloc(H.dst.size());
        h2d(L.slice_ptr.p, H.slice_ptr.data(), H.slice_ptr.size() * 4);
        h2d(L.col.p, H.col.p, H.n_slots * 4);
        h2d(L.cidx.p, H.cidx.p, H.n_slots * 4);
        h2d(L.dst.p, H.dst.p, H.dst.size() * 4);
    }
    CG_HIP(hipStreamSynchronize(st));
}

static constexpr uint32_t SPMV_LONG_ROW = 4096;

__global__ void __launch_bounds__(256) k_spmv(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                              const uint32_t* __restrict__ cidx, const Fr* __restrict__ dict,
                                              const Fr* __restrict__ w, Fr* __restrict__ out, uint64_t rows) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    uint32_t b = row_ptr[i], e = row_ptr[i + 1];
    if (e - b > SPMV_LONG_ROW) return;            // k_spmv_long's
    Fr acc = Fr::zero();
    for (uint32_t t = b; t < e; ++t) {
        Fr v = w[col[t]];
        uint32_t ci = cidx[t];
        if (ci != 0) v = mul(v, dict[ci]);
        acc = add(acc, v);
    }
    out[i] = acc;
}
// one workgroup per long row (the transposed matrices of cg_setup have a few: the constant-one wire sits in a large share
// of all constraints), strided partial sums and an LDS tree
__global__ void __launch_bounds__(256) k_spmv_long(const uint32_t* __restrict__ long_rows, const uint32_t* __restrict__ row_ptr,
                                                   const uint32_t* __restrict__ col, const uint32_t* __restrict__ cidx,
                                                   const Fr* __restrict__ dict, const Fr* __restrict__ w, Fr* __restrict__ out) {
    __shared__ Fr part[256];
    const uint32_t i = long_rows[blockIdx.x];
    const uint32_t b = row_ptr[i], e = row_ptr[i + 1];
    Fr acc = Fr::zero();
    for (uint32_t t = b + threadIdx.x; t < e; t += 256) {
        Fr v = w[col[t]];
        uint32_t ci = cidx[t];
        if (ci != 0) v = mul(v, dict[ci]);
        acc = add(acc, v);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) part[threadIdx.x] = add(part[threadIdx.x], part[threadIdx.x + s2]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = part[0];
}

void spmv(const DevCsr& m, const Fr* w, Fr* out, hipStream_t st) {
    if (!m.rows) return;
    k_spmv<<<grid_for(m.rows), 256, 0, st>>>(m.row_ptr.p, m.col.p, m.coef_idx.p, m.dict.p, w, out, m.rows);
    CG_KERNEL_CHECK();
    if (m.n_long_rows) {
        k_spmv_long<<<(uint32_t)m.n_long_rows, 256, 0, st>>>(m.long_rows.p, m.row_ptr.p, m.col.p, m.coef_idx.p, m.dict.p, w, out);
        CG_KERNEL_CHECK();
    }
}

}  // namespace cg
