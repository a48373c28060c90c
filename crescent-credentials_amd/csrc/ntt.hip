// Saturated-limb (8 x 32-bit, R = 2^256) Fr plumbing around the transforms: the domain constants
// (twiddles, coset powers, vanishing-polynomial inverse) of ark-poly's radix-2 domain as driven by
// forks/groth16/src/r1cs_to_qap.rs:150-213, the CSR upload with its coefficient dictionary, and the
// saturated sparse product used by cg_setup.  The transforms themselves run on 29-bit limbs: csrc/wmap29.hip.
#include <unordered_map>

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
    rows = rows_;
    nnz = m.nnz;
    if (nnz >= (1ull << 32)) throw HipError(CG_ERR_INVALID_ARGUMENT, "matrix with >= 2^32 non-zeros");
    if (m.row_ptr[0] != 0 || m.row_ptr[rows] != nnz) throw HipError(CG_ERR_INVALID_ARGUMENT, "row_ptr does not span [0, nnz]");
    std::vector<uint32_t> rp(rows + 1);
    for (uint64_t i = 0; i <= rows; ++i) {
        if (i && m.row_ptr[i] < m.row_ptr[i - 1]) throw HipError(CG_ERR_INVALID_ARGUMENT, "row_ptr not monotone");
        rp[i] = (uint32_t)m.row_ptr[i];
    }
    // coefficient dictionary (circom matrices repeat a handful of constants millions of times)
    struct Key { uint64_t w[4]; bool operator==(const Key& o) const { return !memcmp(w, o.w, 32); } };
    struct KeyHash { size_t operator()(const Key& k) const { return (size_t)(k.w[0] * 0x9e3779b97f4a7c15ull ^ k.w[1] ^ (k.w[2] << 1) ^ (k.w[3] << 7)); } };
    std::unordered_map<Key, uint32_t, KeyHash> map;
    std::vector<Fr> dict_h;
    std::vector<uint32_t> idx(nnz);
    {
        Key one{};
        one.w[0] = 1;
        map.emplace(one, 0u);
        Fr o = Fr::zero(); o.l[0] = 1;
        dict_h.push_back(o);
    }
    for (uint64_t t = 0; t < nnz; ++t) {
        if (m.col[t] >= num_variables) throw HipError(CG_ERR_INVALID_ARGUMENT, "column index out of range");
        Key k;
        memcpy(k.w, m.coeff + 32 * t, 32);
        auto it = map.find(k);
        if (it == map.end()) {
            Fr c = fp_from_bytes<Fr>(m.coeff + 32 * t);
            if (!fp_is_canonical(c)) throw HipError(CG_ERR_INVALID_ARGUMENT, "non-canonical matrix coefficient");
            it = map.emplace(k, (uint32_t)dict_h.size()).first;
            dict_h.push_back(c);
        }
        idx[t] = it->second;
    }
    for (auto& c : dict_h) c = to_mont(c);
    row_ptr.alloc(rows + 1);
    {
        std::vector<uint32_t> lr;
        for (uint64_t i = 0; i < rows; ++i) if (rp[i + 1] - rp[i] > 4096u) lr.push_back((uint32_t)i);
        n_long_rows = lr.size();
        long_rows.alloc(lr.size() ? lr.size() : 1);
        if (!lr.empty()) h2d_sync(long_rows.p, lr.data(), lr.size() * 4, st);
    }
    col.alloc(nnz ? nnz : 1);
    coef_idx.alloc(nnz ? nnz : 1);
    dict.alloc(dict_h.size());
    h2d_sync(row_ptr.p, rp.data(), (rows + 1) * 4, st);
    if (nnz) {
        h2d_sync(col.p, m.col, nnz * 4, st);
        h2d_sync(coef_idx.p, idx.data(), nnz * 4, st);
    }
    h2d_sync(dict.p, dict_h.data(), dict_h.size() * sizeof(Fr), st);
    if (sliced) build_sell(rp, m.col, idx, st);
}

// the sliced layout of ntt.hpp's SellLevel, built on the host once per matrix
void DevCsr::build_sell(const std::vector<uint32_t>& rp, const uint32_t* col_h, const std::vector<uint32_t>& idx, hipStream_t st) {
    struct Item { uint32_t row, first, len; };              // a row of the current level: `len` terms from `first`
    std::vector<uint32_t> cur_col(col_h, col_h + nnz), cur_idx(idx);
    std::vector<Item> items;
    items.reserve(rows);
    for (uint64_t i = 0; i < rows; ++i)
        if (rp[i + 1] > rp[i]) items.push_back({(uint32_t)i, rp[i], rp[i + 1] - rp[i]});
    // Inside a row the terms with a coefficient other than one come first (a sum does not care), so a piece is "k
    // products, then plain additions", and the pieces are sorted by k before they are sliced: the 64 lanes of a slice
    // then agree on which of their steps multiply.  Circom rows are mostly unit coefficients with the powers of two
    // concentrated in the adder rows (gate mix: 82 % of A's rows carry no other coefficient at all), and a wave pays the
    // 207-instruction product at every step at which ANY of its lanes needs it.
    static const bool plain = CG_TUNE_ENV("SELL_PLAIN") != nullptr;       // A/B aid (tuning builds): round 2's layout (terms as given, pieces by length)
    for (const Item& it : items) {
        if (plain) break;
        uint32_t w = it.first;
        for (uint32_t t = it.first; t < it.first + it.len; ++t)
            if (cur_idx[t] != 0) {
                std::swap(cur_idx[t], cur_idx[w]);
                std::swap(cur_col[t], cur_col[w]);
                ++w;
            }
    }
    n_sell = 0;
    sell_scratch = 0;
    while (!items.empty()) {
        if (n_sell >= 8) throw HipError(CG_ERR_INVALID_ARGUMENT, "matrix row too long for the sliced layout");
        SellLevel& L = sell[n_sell++];
        struct Piece { uint32_t first, len, dst; };
        std::vector<Piece> pieces;
        std::vector<Item> next_items;
        std::vector<uint32_t> next_col;
        uint32_t partials = 0;
        for (const Item& it : items) {
            const uint32_t np = (it.len + SELL_PIECE - 1) / SELL_PIECE;
            if (np == 1) {
                pieces.push_back({it.first, it.len, it.row | SELL_FINAL});
                continue;
            }
            next_items.push_back({it.row, (uint32_t)next_col.size(), np});
            for (uint32_t k = 0; k < np; ++k) {
                const uint32_t b = it.first + k * SELL_PIECE;
                const uint32_t l = it.len - k * SELL_PIECE < SELL_PIECE ? it.len - k * SELL_PIECE : SELL_PIECE;
                pieces.push_back({b, l, partials});
                next_col.push_back(partials++);
            }
        }
        // most products first, then longest first: a slice holds pieces of (nearly) one shape
        auto products_of = [&](const Piece& p) {
            uint32_t k = 0;
            if (plain) return k;
            for (uint32_t t = 0; t < p.len; ++t) k += cur_idx[p.first + t] != 0;
            return k;
        };
        constexpr uint32_t NK = (SELL_PIECE + 1) * (SELL_PIECE + 1);
        auto key_of = [&](const Piece& p) { return (SELL_PIECE - products_of(p)) * (SELL_PIECE + 1) + (SELL_PIECE - p.len); };
        std::vector<uint32_t> start(NK + 1, 0);
        for (const Piece& p : pieces) start[key_of(p) + 1]++;
        for (uint32_t k = 0; k < NK; ++k) start[k + 1] += start[k];
        std::vector<Piece> sorted(pieces.size());
        for (const Piece& p : pieces) sorted[start[key_of(p)]++] = p;
        const uint32_t np = (uint32_t)sorted.size(), ns = (np + 63) / 64;
        std::vector<uint32_t> sp(ns + 1, 0), dst(np);
        for (uint32_t s = 0; s < ns; ++s) {
            uint32_t longest = 0;
            for (uint32_t p = s * 64; p < np && p < (s + 1) * 64; ++p) longest = std::max(longest, sorted[p].len);
            sp[s + 1] = sp[s] + 64 * longest;
        }
        std::vector<uint32_t> lc(sp[ns] ? sp[ns] : 1, 0), li(sp[ns] ? sp[ns] : 1, SELL_PAD);
        for (uint32_t p = 0; p < np; ++p) {
            const Piece& pc = sorted[p];
            dst[p] = pc.dst;
            const uint32_t base = sp[p / 64] + (p & 63);
            for (uint32_t t = 0; t < pc.len; ++t) {
                lc[base + t * 64] = cur_col[pc.first + t];
                li[base + t * 64] = cur_idx[pc.first + t];
            }
        }
        L.n_pieces = np;
        L.n_partials = partials;
        if (partials > sell_scratch) sell_scratch = partials;
        L.slice_ptr.alloc(ns + 1); L.col.alloc(lc.size()); L.cidx.alloc(li.size()); L.dst.alloc(np ? np : 1);
        h2d_sync(L.slice_ptr.p, sp.data(), (ns + 1) * 4, st);
        h2d_sync(L.col.p, lc.data(), lc.size() * 4, st);
        h2d_sync(L.cidx.p, li.data(), li.size() * 4, st);
        if (np) h2d_sync(L.dst.p, dst.data(), np * 4, st);
        // the next level sums the partials: unit coefficients over this level's scratch vector
        items.swap(next_items);
        cur_col.swap(next_col);
        cur_idx.assign(cur_col.size(), 0u);
    }
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
