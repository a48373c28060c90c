// Radix-2 NTT over BN254 Fr + R1CS->QAP witness-map kernels for gfx950.
//
// Restates (does not translate) ark-poly's radix-2 domain as driven by
// forks/groth16/src/r1cs_to_qap.rs:150-213.  A transform of size n = 2^logn runs as a few LDS
// passes: each workgroup stages a 1024-element tile (32 KiB) in LDS and performs up to ten
// butterfly stages on it before the tile goes back to HBM, so a 2^21 transform moves the array
// through HBM three times instead of twenty-one.  Strided passes gather 16 adjacent columns per
// group so every global access is a 512-byte contiguous run.
#include <unordered_map>

#include "ntt.hpp"

namespace cg {

static constexpr int TS_MAX = 10;        // log2 of the LDS tile
static constexpr int STRIDED_MAX_S = 6;  // stages per strided pass (16 columns of 32 B)

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

__global__ void __launch_bounds__(256) k_to_mont(const Fr* in, Fr* out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = to_mont(in[i]);
}
__global__ void __launch_bounds__(256) k_from_mont(const Fr* in, Fr* out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = from_mont(in[i]);
}
__global__ void __launch_bounds__(256) k_mul_vec(Fr* a, const Fr* b, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = mul(a[i], b[i]);
}
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
__global__ void __launch_bounds__(256) k_pointwise(const Fr* a, const Fr* b, const Fr* c, Fr* out, Fr vinv, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mul(sub(mul(a[i], b[i]), c[i]), vinv);
}
__global__ void __launch_bounds__(256) k_unbitrev_scale(const Fr* in, Fr* out, const Fr* scale, int logn, int to_canonical) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (1ull << logn)) return;
    Fr x = in[p];
    if (scale) x = mul(x, scale[p]);
    if (to_canonical) x = from_mont(x);
    out[bitrev32((uint32_t)p, logn)] = x;
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

void fr_to_mont(const Fr* in, Fr* out, uint64_t n, hipStream_t st) {
    if (!n) return;
    k_to_mont<<<grid_for(n), 256, 0, st>>>(in, out, n);
    CG_KERNEL_CHECK();
}
void fr_from_mont(const Fr* in, Fr* out, uint64_t n, hipStream_t st) {
    if (!n) return;
    k_from_mont<<<grid_for(n), 256, 0, st>>>(in, out, n);
    CG_KERNEL_CHECK();
}
void fr_mul_vec(Fr* a, const Fr* b, uint64_t n, hipStream_t st) {
    if (!n) return;
    k_mul_vec<<<grid_for(n), 256, 0, st>>>(a, b, n);
    CG_KERNEL_CHECK();
}
void fr_pow_table(Fr* out, const Fr& base, const Fr& scale, uint64_t n, bool bitrev, int logn, hipStream_t st) {
    if (!n) return;
    k_pow_table<<<grid_for(n), 256, 0, st>>>(out, base, scale, n, bitrev ? 1 : 0, logn);
    CG_KERNEL_CHECK();
}
void qap_pointwise(const Fr* a, const Fr* b, const Fr* c, Fr* out, const Fr& vinv, uint64_t n, hipStream_t st) {
    k_pointwise<<<grid_for(n), 256, 0, st>>>(a, b, c, out, vinv, n);
    CG_KERNEL_CHECK();
}
void ntt_unbitrev_scale(const Fr* in, Fr* out, const Fr* scale, int logn, bool to_canonical, hipStream_t st) {
    k_unbitrev_scale<<<grid_for(1ull << logn), 256, 0, st>>>(in, out, scale, logn, to_canonical ? 1 : 0);
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
// LDS pass kernel
//   local element e (ts bits) = [extra | g (S bits) | col (cbits)],  ts = extra_bits + S + cbits
//   global index  = col | (Tlo << cbits) | (g << gbit_lo) | (Thi << (gbit_lo + S))
//   with T = tile * 2^extra_bits + extra,  Tlo = low (gbit_lo - cbits) bits of T, Thi the rest.
// ---------------------------------------------------------------------------------------------
struct PassParams {
    int logn, ts, S, cbits, gbit_lo;
    int q0;  // first stage index of the pass (DIF: stage q pairs bit logn-1-q; DIT: stage q pairs bit q)
};

__device__ __forceinline__ uint32_t local_to_global(uint32_t e, uint32_t tile, const PassParams& pp) {
    const int extra_bits = pp.ts - pp.S - pp.cbits;
    uint32_t col = e & ((1u << pp.cbits) - 1u);
    uint32_t g = (e >> pp.cbits) & ((1u << pp.S) - 1u);
    uint32_t extra = e >> (pp.cbits + pp.S);
    uint32_t T = (tile << extra_bits) | extra;
    const int lo_bits = pp.gbit_lo - pp.cbits;
    uint32_t Tlo = T & ((1u << lo_bits) - 1u);
    uint32_t Thi = T >> lo_bits;
    return col | (Tlo << pp.cbits) | (g << pp.gbit_lo) | (Thi << (pp.gbit_lo + pp.S));
}

struct alignas(16) Half { uint32_t w[4]; };

__device__ __forceinline__ Fr lds_load(const Half* lo, const Half* hi, uint32_t e) {
    Fr r;
    Half a = lo[e], b = hi[e];
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.l[i] = a.w[i]; r.l[4 + i] = b.w[i]; }
    return r;
}
__device__ __forceinline__ void lds_store(Half* lo, Half* hi, uint32_t e, const Fr& x) {
    Half a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a.w[i] = x.l[i]; b.w[i] = x.l[4 + i]; }
    lo[e] = a;
    hi[e] = b;
}

template <bool DIT>
__global__ void __launch_bounds__(256) k_ntt_pass(Fr* __restrict__ data, const Fr* __restrict__ tw,
                                                  const Fr* __restrict__ premul, PassParams pp) {
    __shared__ Half s_lo[1 << TS_MAX];
    __shared__ Half s_hi[1 << TS_MAX];
    const uint32_t tile = blockIdx.x;
    const uint32_t tsize = 1u << pp.ts;
    // load (coalesced: consecutive e -> consecutive col -> consecutive addresses within a run)
    for (uint32_t e = threadIdx.x; e < tsize; e += 256) {
        uint32_t gi = local_to_global(e, tile, pp);
        Fr x = data[gi];
        if (premul) x = mul(x, premul[gi]);
        lds_store(s_lo, s_hi, e, x);
    }
    __syncthreads();
    const uint32_t nbf = tsize >> 1;
    for (int j = 0; j < pp.S; ++j) {
        // DIF walks the pass's bits from the top, DIT from the bottom
        const int q = pp.q0 + j;
        const int gb = DIT ? q : (pp.logn - 1 - q);       // global bit paired by this stage
        const int lb = pp.cbits + (gb - pp.gbit_lo);        // its position inside the tile
        const uint32_t lmask = (1u << lb) - 1u;
        const int tw_shift = DIT ? (pp.logn - 1 - q) : q;
        for (uint32_t b = threadIdx.x; b < nbf; b += 256) {
            uint32_t e0 = ((b & ~lmask) << 1) | (b & lmask);
            uint32_t e1 = e0 | (1u << lb);
            uint32_t gi = local_to_global(e0, tile, pp);
            uint32_t k = gi & ((1u << gb) - 1u);
            Fr u = lds_load(s_lo, s_hi, e0);
            Fr v = lds_load(s_lo, s_hi, e1);
            Fr w = tw[(uint64_t)k << tw_shift];
            if (DIT) {
                v = mul(v, w);
                lds_store(s_lo, s_hi, e0, add(u, v));
                lds_store(s_lo, s_hi, e1, sub(u, v));
            } else {
                lds_store(s_lo, s_hi, e0, add(u, v));
                lds_store(s_lo, s_hi, e1, mul(sub(u, v), w));
            }
        }
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < tsize; e += 256) {
        uint32_t gi = local_to_global(e, tile, pp);
        data[gi] = lds_load(s_lo, s_hi, e);
    }
}

// Pass plan: DIF = strided passes over the high bits, then one contiguous pass over the low
// min(ts, logn) bits; DIT is the mirror image.
static void run_passes(Fr* data, const NttDomain& d, bool dit, bool inverse, const Fr* premul, hipStream_t st) {
    const int logn = d.logn;
    if (logn == 0) {
        if (premul) fr_mul_vec(data, premul, 1, st);
        return;
    }
    const Fr* tw = inverse ? d.tw_inv.p : d.tw_fwd.p;
    const int ts = logn < TS_MAX ? logn : TS_MAX;
    const int s_cont = ts;            // stages in the contiguous pass
    const int rest = logn - s_cont;   // stages left for strided passes
    const int npass = rest ? (rest + STRIDED_MAX_S - 1) / STRIDED_MAX_S : 0;
    std::vector<PassParams> plan;
    // contiguous pass parameters (pairs global bits [0, s_cont))
    PassParams cont{logn, ts, s_cont, 0, 0, dit ? 0 : logn - s_cont};
    // strided passes cover global bits [s_cont, logn), split evenly
    std::vector<PassParams> strided;
    {
        int done = 0;
        for (int p = 0; p < npass; ++p) {
            int S = (rest - done + (npass - p) - 1) / (npass - p);
            PassParams pp;
            pp.logn = logn; pp.ts = ts; pp.S = S; pp.cbits = ts - S;
            pp.gbit_lo = s_cont + done;             // lowest global bit of this group (DIT order)
            pp.q0 = dit ? pp.gbit_lo : (logn - (pp.gbit_lo + S));
            strided.push_back(pp);
            done += S;
        }
    }
    if (dit) {
        plan.push_back(cont);
        for (auto& p : strided) plan.push_back(p);
    } else {
        for (int i = (int)strided.size() - 1; i >= 0; --i) plan.push_back(strided[i]);
        plan.push_back(cont);
    }
    const uint32_t tiles = (uint32_t)(d.n >> ts);
    for (size_t i = 0; i < plan.size(); ++i) {
        const Fr* pm = (i == 0) ? premul : nullptr;
        if (dit) k_ntt_pass<true><<<tiles, 256, 0, st>>>(data, tw, pm, plan[i]);
        else k_ntt_pass<false><<<tiles, 256, 0, st>>>(data, tw, pm, plan[i]);
        CG_KERNEL_CHECK();
    }
}

void ntt_dif(Fr* data, const NttDomain& d, bool inverse, const Fr* premul, hipStream_t st) {
    run_passes(data, d, false, inverse, premul, st);
}
void ntt_dit(Fr* data, const NttDomain& d, bool inverse, const Fr* premul, hipStream_t st) {
    run_passes(data, d, true, inverse, premul, st);
}

// ---------------------------------------------------------------------------------------------
// sparse matrix-vector product (evaluate_constraint, r1cs_to_qap.rs:16-45)
// ---------------------------------------------------------------------------------------------
void DevCsr::upload(const cg_csr& m, uint64_t rows_, uint64_t num_variables) {
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
    col.alloc(nnz ? nnz : 1);
    coef_idx.alloc(nnz ? nnz : 1);
    dict.alloc(dict_h.size());
    CG_HIP(hipMemcpy(row_ptr.p, rp.data(), (rows + 1) * 4, hipMemcpyHostToDevice));
    if (nnz) {
        CG_HIP(hipMemcpy(col.p, m.col, nnz * 4, hipMemcpyHostToDevice));
        CG_HIP(hipMemcpy(coef_idx.p, idx.data(), nnz * 4, hipMemcpyHostToDevice));
    }
    CG_HIP(hipMemcpy(dict.p, dict_h.data(), dict_h.size() * sizeof(Fr), hipMemcpyHostToDevice));
}

__global__ void __launch_bounds__(256) k_spmv(const uint32_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                              const uint32_t* __restrict__ cidx, const Fr* __restrict__ dict,
                                              const Fr* __restrict__ w, Fr* __restrict__ out, uint64_t rows) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    uint32_t b = row_ptr[i], e = row_ptr[i + 1];
    Fr acc = Fr::zero();
    for (uint32_t t = b; t < e; ++t) {
        Fr v = w[col[t]];
        uint32_t ci = cidx[t];
        if (ci != 0) v = mul(v, dict[ci]);
        acc = add(acc, v);
    }
    out[i] = acc;
}

void spmv(const DevCsr& m, const Fr* w, Fr* out, hipStream_t st) {
    if (!m.rows) return;
    k_spmv<<<grid_for(m.rows), 256, 0, st>>>(m.row_ptr.p, m.col.p, m.coef_idx.p, m.dict.p, w, out, m.rows);
    CG_KERNEL_CHECK();
}

}  // namespace cg
