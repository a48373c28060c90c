// R1CS -> QAP witness map on 29-bit-limb arithmetic (see wmap29.hpp).
//
// Restates LibsnarkReduction::witness_map_from_matrices (forks/groth16/src/r1cs_to_qap.rs:150-213):
//   a, b, c = A·w, B·w, C·w (rows m..m+l of a hold w[0..l))            :164-177,191-196
//   ifft; coset fft (offset g = 5)                                       :179-185,198-199
//   ab = (a∘b - c) / Z(g)                                                :187,201-208
//   coset ifft -> h                                                      :210
// as seven decimation-in-time transforms of three LDS passes each, with the bit reversals, the coset
// scalings, the 1/n factors, the pointwise step and the exit from Montgomery form all fused into the
// loads and stores of those passes.
#include "wmap29.hpp"

namespace cg {

// log2 of the LDS tile.  10: 1024 elements x 9 limbs = 36 KiB + padding per block of 256 threads, four blocks per CU, a
// 2^21 transform in THREE passes (10 + 6 + 5 stages).  11: 2048 elements = 73 KiB per block of 512 threads, two blocks per
// CU - the same four waves per SIMD - and a 2^21 transform in TWO passes (11 + 10 stages): a third of the pass ends (unpack,
// reduce, pack) and one global round trip fewer, at the price of 64-byte instead of 512-byte runs in the strided pass.
// Measured (profiles/r03_h_ntt_tiles.txt): stand-alone the two are equal at 2^21 (0.301 against 0.304 ms - the passes are
// bound by instruction issue, not by their ends) and the big tile is ~5 % ahead at 2^20 and 2^24, behind at 2^22 (no pass
// saved) and below 2^20 (too few tiles to fill the chip); in the proof pipeline it is worth +0.8 %.  So the big tile is
// taken where it saves a pass on a transform of at least 2^20 elements.
static constexpr int TS29_SMALL = 10, TS29_BIG = 11;
// most stages a strided pass takes (its tile holds 2^(tile - S) columns of 2^S rows)
static int strided_max_stages(int tile_log) { return tile_log == TS29_BIG ? 10 : 6; }
static int ntt_passes(int logn, int tile_log) {
    if (logn <= tile_log) return 1;
    const int smax = strided_max_stages(tile_log);
    return 1 + (logn - tile_log + smax - 1) / smax;
}
static int ntt_tile_log(int logn) {
    static const int forced = [] {
        const char* e = CG_TUNE_ENV("NTT_TILE");          // A/B aid (tuning builds)
        return (e && (atoi(e) == TS29_SMALL || atoi(e) == TS29_BIG)) ? atoi(e) : 0;
    }();
    if (forced) return (forced == TS29_BIG && logn > TS29_SMALL) ? TS29_BIG : TS29_SMALL;
    return (logn >= 20 && ntt_passes(logn, TS29_BIG) < ntt_passes(logn, TS29_SMALL)) ? TS29_BIG : TS29_SMALL;
}

// ---- packed (8 x u32) global accesses ---------------------------------------------------------------------
__device__ __forceinline__ Fr29 load_packed29(const uint32_t* __restrict__ base, uint64_t idx) {
    const uint4* p = reinterpret_cast<const uint4*>(base + idx * 8);
    uint4 a = p[0], b = p[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return unpack29<Fr29P>(w);
}
__device__ __forceinline__ void store_packed29(uint32_t* __restrict__ base, uint64_t idx, const Fr29& canonical_value) {
    uint32_t w[8];
    pack29(canonical_value, w);
    uint4* p = reinterpret_cast<uint4*>(base + idx * 8);
    p[0] = make_uint4(w[0], w[1], w[2], w[3]);
    p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ uint32_t brev(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// ---- table conversion ----------------------------------------------------------------------------------------
// Montgomery(2^256) -> packed R' form, or (plain = 1) -> packed plain integer
__global__ void __launch_bounds__(256) k_to_packed29(const Fr* __restrict__ in, uint32_t* __restrict__ out, uint64_t n, int plain,
                                                     int unbitrev_logn) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = in[i];
    uint64_t dst = unbitrev_logn ? brev((uint32_t)i, unbitrev_logn) : i;
    if (plain) {
        Fr c = from_mont(x);
        uint4* p = reinterpret_cast<uint4*>(out + dst * 8);
        p[0] = make_uint4(c.l[0], c.l[1], c.l[2], c.l[3]);
        p[1] = make_uint4(c.l[4], c.l[5], c.l[6], c.l[7]);
    } else {
        store_packed29(out, dst, from_mont256<Fr29P>(x));
    }
}

// twiddles are kept unpacked: nine 29-bit limbs in a 48-byte record (three 16-byte loads, no unpacking per butterfly)
__global__ void __launch_bounds__(256) k_to_limbs12(const Fr* __restrict__ in, uint32_t* __restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr29 x = from_mont256<Fr29P>(in[i]);
    uint4* p = reinterpret_cast<uint4*>(out + i * 12);
    p[0] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
    p[1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    p[2] = make_uint4(x.l[8], 0u, 0u, 0u);
}
__device__ __forceinline__ Fr29 load_tw(const uint32_t* __restrict__ tw, uint64_t idx) {
    const uint4* p = reinterpret_cast<const uint4*>(tw + idx * 12);
    uint4 a = p[0], b = p[1], c = p[2];
    Fr29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = c.x;
    return r;
}

// v (normalised, value < 2^261) -> r ≡ v (mod N), normalised, r < 3N < 2^256: a one-word quotient estimate instead of
// a full Montgomery product; used where a value only has to fit the packed 32-byte form again.
__device__ __forceinline__ Fr29 weak_reduce(const Fr29& v) {
    // q = floor(l[8] * MU / 2^48) with MU = floor(2^280 / N) and l[8] = floor(v / 2^232): never above v / N,
    // short of it by less than 2
    constexpr uint32_t MU = 88753990u;   // floor(2^280 / r), r = BN254 scalar modulus
    const uint32_t q = (uint32_t)(((uint64_t)v.l[8] * MU) >> 48);
    const int32_t nq = -(int32_t)q;
    Fr29 r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        c += (int64_t)v.l[i];
        c += (int64_t)nq * (int64_t)(int32_t)Fr29P::N[i];
        if (i < 8) { r.l[i] = (uint32_t)c & M29; c >>= 29; }
        else r.l[i] = (uint32_t)c;
    }
    return r;
}

void Wm29Domain::build(const NttDomain& d, hipStream_t st) {
    logn = d.logn;
    n = d.n;
    const uint64_t half = n > 1 ? n / 2 : 1;
    tw_fwd.alloc(half * 12);
    tw_inv.alloc(half * 12);
    coset.alloc(n * 8);
    icoset.alloc(n * 8);
    k_to_limbs12<<<ceil_div(half, 256), 256, 0, st>>>(d.tw_fwd.p, tw_fwd.p, half);
    k_to_limbs12<<<ceil_div(half, 256), 256, 0, st>>>(d.tw_inv.p, tw_inv.p, half);
    if (d.coset_br.n) {   // a domain built without coset tables serves plain transforms only (Wm29Strided::sub)
        // NttDomain keeps the coset tables at bit-reversed positions; here they are indexed naturally
        k_to_packed29<<<ceil_div(n, 256), 256, 0, st>>>(d.coset_br.p, coset.p, n, 0, logn);
        k_to_packed29<<<ceil_div(n, 256), 256, 0, st>>>(d.icoset_br.p, icoset.p, n, 1, logn);
    }
    CG_KERNEL_CHECK();
    Fr29 v = from_mont256<Fr29P>(d.vanishing_inv);   // host arithmetic
    pack29(v, vinv);
    Fr vp = from_mont(d.vanishing_inv);
    memcpy(vinv_plain, vp.l, 32);
}

void Wm29Strided::build(const NttDomain& big, int logs_, int rank, hipStream_t st) {
    logs = logs_;
    d = big.n >> logs;
    NttDomain small;
    small.build(big.logn - logs, false, st);
    sub.build(small, st);
    // s^e / n, s = g·ω^rank (ω the root of the size-n domain)
    const Fr s = mul(fr_from_u64(5), fr_pow_u64(fr_root_of_unity(big.logn), (uint64_t)rank));
    DevBuf<Fr> t(big.n);
    fr_pow_table(t.p, s, inv(fr_from_u64(big.n)), big.n, false, big.logn, st);
    fold.alloc(big.n * 8);
    k_to_packed29<<<ceil_div(big.n, 256), 256, 0, st>>>(t.p, fold.p, big.n, 0, 0);
    CG_KERNEL_CHECK();
    CG_HIP(hipStreamSynchronize(st));   // `small` and `t` are released on return
}

// out[rev_d(i)] = Σ_t a[i + t·d]·T[i + t·d], i < d, t < 2^logs (Wm29Strided): a the unscaled output of the inverse
// transform (packed, < 2^256), T canonical.  A product is below 1.04 N; limbs are renormalised every fourth term.
__global__ void __launch_bounds__(256) k_fold29(const uint32_t* __restrict__ a, const uint32_t* __restrict__ T, uint32_t* __restrict__ out,
                                                uint32_t d, int logd, uint32_t terms) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    Fr29 acc = Fr29::zero();
    for (uint32_t t = 0; t < terms; ++t) {
        const uint64_t e = (uint64_t)i + (uint64_t)t * d;
        acc = add(acc, mul(load_packed29(a, e), load_packed29(T, e)));
        if ((t & 3u) == 3u) acc = normalize(acc);
    }
    store_packed29(out, brev(i, logd), weak_reduce(normalize(acc)));
}

void Csr29::build(const DevCsr& m, hipStream_t st) {
    dict.alloc(m.dict.n * 8);
    k_to_packed29<<<ceil_div(m.dict.n, 256), 256, 0, st>>>(m.dict.p, dict.p, m.dict.n, 0, 0);
    CG_KERNEL_CHECK();
}

__device__ __forceinline__ bool fr_lt_modulus(const Fr& x) {
    bool lt = false, decided = false;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        if (!decided && x.l[k] != FrP::N[k]) { lt = x.l[k] < FrP::N[k]; decided = true; }
    }
    return lt;
}

// ---- witness -> R' form; sparse products --------------------------------------------------------------------
// Also places the instance wires behind A's rows (a[m + i] = w_i, r1cs_to_qap.rs:173-177) at the bit-reversed index the
// first transform wants: `va` must have been zeroed before this kernel and is written below row m + l by nothing else.
__global__ void __launch_bounds__(256) k_w_to29(const Fr* __restrict__ w, uint32_t* __restrict__ out, uint64_t n,
                                                uint32_t* __restrict__ bad_input, uint32_t* __restrict__ va, uint64_t m, uint64_t l, int logn) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = w[i];
    // the reference's scalars are field elements by type; across a C ABI they are bytes, so check (x < r)
    if (!fr_lt_modulus(x)) *bad_input = 1u;
    // x·R' as the product leaves it: normalised and below 2N (x < 2^256, R'^2 mod N < N), which is all the packed form
    // and the sparse product's lazy sums ask for; the canonical representative would cost a second product
    const Fr29 v = mul(unpack29<Fr29P>(x.l), Fr29::from_limbs(Fr29P::R2));
    store_packed29(out, i, v);
    if (i < l) store_packed29(va, brev((uint32_t)(m + i), logn), v);
}


// out[rev(i)] = <M_i, w>, i < rows  (evaluate_constraint, r1cs_to_qap.rs:16-45); the output vector is the
// bit-reversed input the first transform wants.
// One level of the sliced sparse product (ntt.hpp SellLevel): lane = piece, at most SELL_PIECE terms, the slice's
// index arrays read 64 lanes wide.  src: the witness (level 0) or the previous level's partial sums, packed R' form.
__global__ void __launch_bounds__(256) k_sell29(const uint32_t* __restrict__ slice_ptr, const uint32_t* __restrict__ col,
                                                const uint32_t* __restrict__ cidx, const uint32_t* __restrict__ dst,
                                                const uint32_t* __restrict__ dict, const uint32_t* __restrict__ src,
                                                uint32_t* __restrict__ out, uint32_t* __restrict__ scratch, uint32_t n_pieces, int logn) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pieces) return;
    const uint32_t s = p >> 6, lane = p & 63u;
    const uint32_t base = slice_ptr[s], len = (slice_ptr[s + 1] - base) >> 6;
    Fr29 acc = Fr29::zero();
    uint32_t cnt = 0;
#if !defined(CG_SELL_BATCH)
#define CG_SELL_BATCH 0
#endif
#if CG_SELL_BATCH == 0
    for (uint32_t t = 0; t < len; ++t) {
        const uint32_t ci = cidx[base + t * 64 + lane];
        if (ci == SELL_PAD) continue;
        Fr29 v = load_packed29(src, col[base + t * 64 + lane]);
        if (ci != 0) v = mul(v, load_packed29(dict, ci));      // index 0 is the literal one (is_one() shortcut :31-35)
        acc = add(acc, v);
        if ((++cnt & 3u) == 0) acc = normalize(acc);            // limbs stay below 5·2^29
    }
#else
    // (experiment, -DCG_SELL_BATCH=2 / 4; measured in round 5 and NOT the default)  A piece has at most SELL_PIECE = 8 terms,
    // and a term is two DEPENDENT reads (its column index, then the gathered vector element): walked term by term a lane makes
    // up to sixteen memory round trips one after the other.  Here all index pairs of the piece are requested first, then the
    // gathers CG_SELL_BATCH at a time: 376 -> 320 us per proof stand-alone (45 -> 120 VGPRs), and no difference in the
    // pipeline - 201.5 / 202.0 against 202.1 proofs/s over four alternating rounds (profiles/
    // r05_m_twiddle_ahead_and_sparse_batches.txt): the kernel's waits are filled by the other proofs' kernels.
    constexpr uint32_t BS = CG_SELL_BATCH;
    uint32_t ci[SELL_PIECE], cl[SELL_PIECE];
#pragma unroll
    for (uint32_t t = 0; t < SELL_PIECE; ++t) {
        const bool in = t < len;                                // uniform over the slice's 64 lanes
        ci[t] = in ? cidx[base + t * 64 + lane] : SELL_PAD;
        cl[t] = in ? col[base + t * 64 + lane] : 0u;
    }
#pragma unroll
    for (uint32_t h = 0; h < SELL_PIECE; h += BS) {
        if (h >= len) break;
        uint4 rv[BS][2], rd[BS][2];
#pragma unroll
        for (uint32_t u = 0; u < BS; ++u) {
            const uint32_t t = h + u;
            if (ci[t] == SELL_PAD) continue;
            const uint4* pv = reinterpret_cast<const uint4*>(src + (uint64_t)cl[t] * 8);
            rv[u][0] = pv[0]; rv[u][1] = pv[1];
            if (ci[t] != 0) {
                const uint4* pd = reinterpret_cast<const uint4*>(dict + (uint64_t)ci[t] * 8);
                rd[u][0] = pd[0]; rd[u][1] = pd[1];
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < BS; ++u) {
            const uint32_t t = h + u;
            if (ci[t] == SELL_PAD) continue;
            const uint32_t wv[8] = {rv[u][0].x, rv[u][0].y, rv[u][0].z, rv[u][0].w, rv[u][1].x, rv[u][1].y, rv[u][1].z, rv[u][1].w};
            Fr29 v = unpack29<Fr29P>(wv);
            if (ci[t] != 0) {                                   // index 0 is the literal one (is_one() shortcut :31-35)
                const uint32_t wd[8] = {rd[u][0].x, rd[u][0].y, rd[u][0].z, rd[u][0].w, rd[u][1].x, rd[u][1].y, rd[u][1].z, rd[u][1].w};
                v = mul(v, unpack29<Fr29P>(wd));
            }
            acc = add(acc, v);
            if ((++cnt & 3u) == 0) acc = normalize(acc);        // limbs stay below 5·2^29
        }
    }
#endif
    const uint32_t d = dst[p];
    const Fr29 r = weak_reduce(normalize(acc));     // below 3N: fits the packed form; nothing downstream needs the canonical value
    if (d & SELL_FINAL) store_packed29(out, brev(d & ~SELL_FINAL, logn), r);
    else store_packed29(scratch, d, r);
}

// ---- LDS pass ---------------------------------------------------------------------------------------------------
struct Pass29 {
    int logn, ts, S, cbits, gbit_lo, q0;
};
__device__ __forceinline__ uint32_t l2g29(uint32_t e, uint32_t tile, const Pass29& pp) {
    const int extra_bits = pp.ts - pp.S - pp.cbits;
    uint32_t colv = e & ((1u << pp.cbits) - 1u);
    uint32_t g = (e >> pp.cbits) & ((1u << pp.S) - 1u);
    uint32_t extra = e >> (pp.cbits + pp.S);
    uint32_t T = (tile << extra_bits) | extra;
    const int lo_bits = pp.gbit_lo - pp.cbits;
    uint32_t Tlo = T & ((1u << lo_bits) - 1u);
    uint32_t Thi = T >> lo_bits;
    return colv | (Tlo << pp.cbits) | (g << pp.gbit_lo) | (Thi << (pp.gbit_lo + pp.S));
}
// element e of the tile sits at word e*9 + e/32.  ds_read2_b32 / ds_write_b32 are serviced a 32-lane half at a time over
// 32 banks ((address / 4) mod 32): the odd stride spreads 32 consecutive elements over all of them, and the extra word per
// 32 elements separates the four 32-element runs a half-wave touches when its radix-4 groups are 4 or 16 elements apart.
// Rounds 1-5 padded a word per SIXTEEN elements - right for a 64-bank picture of the LDS and wrong for this one: lanes l + 9 and
// 16 + l of every half then met on one bank in EVERY phase, and SQ_LDS_BANK_CONFLICT read half of SQ_LDS_IDX_ACTIVE on all
// four passes (profiles/r05_am_lds_bank_conflicts.md).  Modelled over the stage pairs of both pass shapes (tools/lds_model.py):
// 1.0 extra cycle per access before, 0.29-0.33 now (two-way conflicts left at the pairs 2-4 / 1-3 bits up).
__device__ __forceinline__ uint32_t lds_off(uint32_t e) { return e * 9u + (e >> 5); }
__device__ __forceinline__ Fr29 lds_get(const uint32_t* s, uint32_t e) {
    Fr29 r;
    const uint32_t o = lds_off(e);
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = s[o + i];
    return r;
}
__device__ __forceinline__ void lds_put(uint32_t* s, uint32_t e, const Fr29& x) {
    const uint32_t o = lds_off(e);
#pragma unroll
    for (int i = 0; i < 9; ++i) s[o + i] = x.l[i];
}

struct Packed8 { uint32_t w[8]; };

// DIT butterfly on lazy values: t = v·w;  (u, v) <- (u + t, u - t + 3N).  Limbs are NOT renormalised here.
__device__ __forceinline__ void bfly(Fr29& u, Fr29& v, const Fr29& w) {
    Fr29 t = mul(v, w);
    v = sub<3, 1>(u, t);
    u = add(u, t);
}

// LOAD: 0 = one input vector; 1 = (a∘b - c)·vinv from three vectors (r1cs_to_qap.rs:187,201-208)
// STORE: 0 = value reduced just enough to pack; 1 = multiplied by scale[natural index] and made canonical
//        (coset factor, or the exit from Montgomery form); 2 = multiplied by the constant passed in `vinv_p`
//        and made canonical (unit-level transforms: plain 1 or plain 1/n; never combined with LOAD = 1);
//        3 = the transformed vector is b on the coset and `in_b` holds vinv·a there as PLAIN integers (natural order):
//        stores the canonical products vinv·a_i·b_i — the coset values of the quotient's a∘b part (r1cs_to_qap.rs:187,
//        201-208; c's part is folded into the l query at load, msm.hpp)
// Stages are taken two at a time as radix-4 groups held in registers (three twiddle loads and ONE carry
// propagation per element for two stages, half the barriers); an odd last stage runs radix-2.
// Lazy-value bounds: a pass starts below 6N (packed inputs are < 2^256 = 5.3N), every stage adds at most 3N — the
// product-free first group of a transform ends below 22.6N instead — so after ten stages values stay under 47N
// (representable: 2^261 = 169N) and limbs, renormalised every second stage, under 2^32.
template <int LOAD, int STORE, int TSL>
__global__ void __launch_bounds__(256 << (TSL - 10)) k_ntt29_pass(const uint32_t* __restrict__ in_a, const uint32_t* __restrict__ in_b,
                                                    const uint32_t* __restrict__ in_c, uint32_t* __restrict__ out,
                                                    const uint32_t* __restrict__ tw, const uint32_t* __restrict__ scale,
                                                    Packed8 vinv_p, Pass29 pp, int store_bitrev) {
    __shared__ uint32_t sm[(1 << TSL) * 9 + (1 << TSL) / 16];
    constexpr uint32_t NT = 256u << (TSL - 10);
    const uint32_t tile = blockIdx.x;
    const uint32_t tsize = 1u << pp.ts;
    for (uint32_t e = threadIdx.x; e < tsize; e += NT) {
        uint32_t gi = l2g29(e, tile, pp);
        Fr29 x = load_packed29(in_a, gi);
        if (LOAD == 1) {
            Fr29 bb = load_packed29(in_b, gi), cc = load_packed29(in_c, gi);
            Fr29 vinv = unpack29<Fr29P>(vinv_p.w);
            x = mul(normalize(sub<7, 1>(mul(x, bb), cc)), vinv);     // inputs < 5.3N each
        }
        lds_put(sm, e, x);
    }
    // The FIRST twiddle of the next stage pair is requested ahead (round 5).  A thread takes exactly one radix-4 group per
    // pair (tsize / 4 <= NT), so its twiddles depend on nothing but indices: the one its first two butterflies need (w1; for
    // the product-free first pair, the only one there is) is requested as soon as the current pair's results are on their
    // way to LDS, BEFORE the barrier - its L2 round trip then runs under the barrier and the next pair's LDS reads instead
    // of after them (31 % of the passes' wave-cycles sat in s_waitcnt) - and the other two are requested at the top of
    // the pair and arrive under the first two butterflies.  Nine more registers live across the barrier; holding all three
    // ahead costs 129-152 VGPRs and a block per CU.
    auto group_of = [&](int jj, uint32_t& k, int& sh1, int& sh2) {
        const int q = pp.q0 + jj;
        const int lb = pp.cbits + (q - pp.gbit_lo);
        const uint32_t lmask = (1u << lb) - 1u;
        sh1 = pp.logn - 1 - q; sh2 = pp.logn - 2 - q;
        const uint32_t bb = threadIdx.x;
        const uint32_t e00 = ((bb & ~lmask) << 2) | (bb & lmask);
        k = l2g29(e00, tile, pp) & ((1u << q) - 1u);
    };
    auto first_twiddle_of = [&](int jj, Fr29& w) {
        if (jj + 1 >= pp.S || threadIdx.x >= (tsize >> 2)) return;
        uint32_t k; int sh1, sh2;
        group_of(jj, k, sh1, sh2);
        w = (pp.q0 + jj == 0) ? load_tw(tw, (uint64_t)1 << sh2) : load_tw(tw, (uint64_t)k << sh1);
    };
    Fr29 w_first;
    first_twiddle_of(0, w_first);
    __syncthreads();
    int j = 0;
    for (; j + 1 < pp.S; j += 2) {                      // radix-4: stages q and q+1
        const int q = pp.q0 + j;
        const int lb = pp.cbits + (q - pp.gbit_lo);
        const uint32_t lmask = (1u << lb) - 1u;
        const int sh2 = pp.logn - 2 - q;
        for (uint32_t b = threadIdx.x; b < (tsize >> 2); b += NT) {
            const uint32_t e00 = ((b & ~lmask) << 2) | (b & lmask);
            const uint32_t e01 = e00 | (1u << lb), e10 = e00 | (2u << lb), e11 = e00 | (3u << lb);
            const uint32_t gi = l2g29(e00, tile, pp);
            const uint32_t k = gi & ((1u << q) - 1u);
            Fr29 x0 = lds_get(sm, e00), x1 = lds_get(sm, e01), x2 = lds_get(sm, e10), x3 = lds_get(sm, e11);
            if (q == 0) {
                // the first two stages of a transform: three of the four twiddles are 1 (k = 0), the fourth is ω^(n/4).
                // Inputs are freshly unpacked (< 2^256 < 6N, normalised), so the three products are skipped and the lazy
                // bounds become: after stage 0 < 10.6N / 12.3N, after stage 1 < 22.6N (instead of 11.3N).
                Fr29 t = x1;
                x1 = sub<7, 1>(x0, t);
                x0 = add(x0, t);
                t = x3;
                x3 = sub<7, 1>(x2, t);
                x2 = add(x2, t);
                t = x2;
                x2 = sub<12, 2>(x0, t);
                x0 = add(x0, t);
                bfly(x1, x3, w_first);
            } else {
                const Fr29 w2a = load_tw(tw, (uint64_t)k << sh2);                   // arrive under the two w1 butterflies
                const Fr29 w2b = load_tw(tw, (uint64_t)(k + (1u << q)) << sh2);
                bfly(x0, x1, w_first);
                bfly(x2, x3, w_first);
                bfly(x0, x2, w2a);
                bfly(x1, x3, w2b);
            }
            lds_put(sm, e00, normalize(x0));
            lds_put(sm, e01, normalize(x1));
            lds_put(sm, e10, normalize(x2));
            lds_put(sm, e11, normalize(x3));
        }
        first_twiddle_of(j + 2, w_first);
        // (experiment, -DCG_NTT_WAVE_LOCAL; measured in round 5 and NOT the default)  Between two stage pairs whose radix-4 groups
        // both lie inside the 256 consecutive elements a WAVE owns (64 lanes x one group of four; a group of the pair at bit lb
        // spans 4 << lb elements, so this holds while the NEXT pair's span is at most 256), the elements a wave reads next are
        // exactly the ones it has just written: no other wave's data is involved, the wave's own LDS operations complete in
        // order, and the block-wide barrier - three of the six of a 2048-element first pass - can be a wavefront-scope fence.
        // Parity-green, the first pass 443 -> 436 us per proof stand-alone, and -0.6 % in the pipeline in four of four
        // alternating pairs (196.8 -> 195.6, profiles/r05_o_ntt_wave_local_barriers.txt): waves that drift apart inside a
        // block make its remaining barriers longer.
#if defined(CG_NTT_WAVE_LOCAL)
        const int lb_next = lb + 2;
        const bool wave_local = (j + 3 < pp.S) && ((4u << lb_next) <= 256u) && (tsize >> 2) >= 64u;
        if (wave_local) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        else __syncthreads();
#else
        __syncthreads();
#endif
    }
    if (j < pp.S) {                                     // odd stage count: one radix-2 stage
        const int q = pp.q0 + j;
        const int lb = pp.cbits + (q - pp.gbit_lo);
        const uint32_t lmask = (1u << lb) - 1u;
        const int tw_shift = pp.logn - 1 - q;
        for (uint32_t b = threadIdx.x; b < (tsize >> 1); b += NT) {
            uint32_t e0 = ((b & ~lmask) << 1) | (b & lmask);
            uint32_t e1 = e0 | (1u << lb);
            uint32_t gi = l2g29(e0, tile, pp);
            uint32_t k = gi & ((1u << q) - 1u);
            Fr29 u = lds_get(sm, e0), v = lds_get(sm, e1);
            bfly(u, v, load_tw(tw, (uint64_t)k << tw_shift));
            lds_put(sm, e0, normalize(u));
            lds_put(sm, e1, normalize(v));
        }
        __syncthreads();
    }
    const uint32_t smask = (1u << pp.S) - 1u;
    // (Round 5 also tried requesting all of a thread's four elements - and the last pass's per-element factors - before the
    // first is used, with the loops unrolled over a constant trip count: +3 % instructions from the predicated unrolled
    // bodies, every pass 4-7 % slower stand-alone, -0.7 % in the pipeline: profiles/r05_l_ntt_batched_loads.txt.  The two
    // co-resident blocks of a CU already cover each other's load and store phases.)
    for (uint32_t f = threadIdx.x; f < tsize; f += NT) {
        uint32_t e = f;
        if (store_bitrev) {   // walk the tile so that consecutive lanes hit consecutive bit-reversed destinations
            uint32_t gprime = f & smask, colp = f >> pp.S;
            e = (brev(gprime, pp.S) << pp.cbits) | colp;
        }
        uint32_t gi = l2g29(e, tile, pp);
        Fr29 x = lds_get(sm, e);
        Fr29 y;
        if (STORE == 1) y = cond_sub_n(mul(x, load_packed29(scale, gi)));
        else if (STORE == 2) y = cond_sub_n(mul(x, unpack29<Fr29P>(vinv_p.w)));   // one constant for every element
        else if (STORE == 3) y = cond_sub_n(mul(load_packed29(in_b, gi), x));     // x = b on the coset, in_b = vinv·a (plain)
        else y = weak_reduce(x);
        store_packed29(out, store_bitrev ? brev(gi, pp.logn) : gi, y);
    }
}

template <int LOAD, int STORE>
static void launch_pass(uint32_t tiles, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* out, const uint32_t* tw,
                        const uint32_t* scale, const Packed8& vinv, const Pass29& pp, int store_bitrev, int tile_log, hipStream_t st) {
    if (tile_log == TS29_BIG) k_ntt29_pass<LOAD, STORE, TS29_BIG><<<tiles, 512, 0, st>>>(a, b, c, out, tw, scale, vinv, pp, store_bitrev);
    else k_ntt29_pass<LOAD, STORE, TS29_SMALL><<<tiles, 256, 0, st>>>(a, b, c, out, tw, scale, vinv, pp, store_bitrev);
    CG_KERNEL_CHECK();
}

// One DIT transform.  The first pass reads `in_a` (or the three pointwise operands) and writes `work`; middle
// passes run in place on `work`; the last pass writes `dst`.  A bit-reversing store permutes across tiles, so
// it must not be in place: callers give dst != work (and != in_a for a single-pass transform) in that case.
static void dit29(const Wm29Domain& d, const uint32_t* tw, const uint32_t* in_a, const uint32_t* in_b, const uint32_t* in_c,
                  bool pointwise, uint32_t* work, uint32_t* dst, const uint32_t* scale, bool store_bitrev, hipStream_t st,
                  const uint32_t* const_scale = nullptr, const uint32_t* quot_a = nullptr) {
    const int logn = d.logn;
    Packed8 vinv;
    memcpy(vinv.w, const_scale ? const_scale : d.vinv, 32);
    std::vector<Pass29> plan;
    const int tile_log = ntt_tile_log(logn);
    const int ts = logn < tile_log ? logn : tile_log;
    plan.push_back(Pass29{logn, ts, ts, 0, 0, 0});
    const int rest = logn - ts;
    const int smax = strided_max_stages(tile_log);
    const int npass = rest ? (rest + smax - 1) / smax : 0;
    int done = 0;
    for (int p = 0; p < npass; ++p) {
        int S = (rest - done + (npass - p) - 1) / (npass - p);
        plan.push_back(Pass29{logn, ts, S, ts - S, ts + done, ts + done});
        done += S;
    }
    const uint32_t tiles = (uint32_t)(d.n >> ts);
    for (size_t i = 0; i < plan.size(); ++i) {
        const bool first = (i == 0), last = (i + 1 == plan.size());
        const uint32_t* a = first ? in_a : work;
        uint32_t* o = last ? dst : work;
        const bool pw = first && pointwise;
        const bool sc = last && scale != nullptr;
        const int sb = last && store_bitrev ? 1 : 0;
        if (last && quot_a) launch_pass<0, 3>(tiles, a, quot_a, nullptr, o, tw, nullptr, vinv, plan[i], sb, tile_log, st);
        else if (last && const_scale) launch_pass<0, 2>(tiles, a, nullptr, nullptr, o, tw, nullptr, vinv, plan[i], sb, tile_log, st);
        else if (pw && sc) launch_pass<1, 1>(tiles, a, in_b, in_c, o, tw, scale, vinv, plan[i], sb, tile_log, st);
        else if (pw) launch_pass<1, 0>(tiles, a, in_b, in_c, o, tw, scale, vinv, plan[i], sb, tile_log, st);
        else if (sc) launch_pass<0, 1>(tiles, a, nullptr, nullptr, o, tw, scale, vinv, plan[i], sb, tile_log, st);
        else launch_pass<0, 0>(tiles, a, nullptr, nullptr, o, tw, scale, vinv, plan[i], sb, tile_log, st);
    }
}

void wm29_run(const Wm29Domain& dom, const DevCsr& A, const DevCsr& B, const DevCsr& C, const Csr29& dA, const Csr29& dB,
              const Csr29& dC, Wm29Buffers& buf, const Fr* w_canon, uint64_t M, uint64_t m, uint64_t l, Fr* h_out,
              hipStream_t st, bool coset_values, const Wm29Strided* strided, int half) {
    const uint64_t D = dom.n;
    const int logn = dom.logn;
    // the flag is HOST memory the kernel writes only when it meets a non-canonical element: no memset, no copy back
    buf.h_bad_input.p[0] = 0;
    uint32_t* v[3] = {buf.va.p, buf.vb.p, buf.vc.p};
    fill_zero(v[0], D * 32, st);
    k_w_to29<<<ceil_div(M, 256), 256, 0, st>>>(w_canon, buf.w29.p, M, buf.h_bad_input.dev(), buf.va.p, m, l, logn);
    CG_KERNEL_CHECK();
    const DevCsr* mats[3] = {&A, &B, &C};
    const Csr29* dicts[3] = {&dA, &dB, &dC};
    const int nvec = coset_values ? 2 : 3;       // c's share of the quotient lives in the folded l query
    for (int k = 0; k < nvec; ++k) {
        if (half && k != half - 1) continue;       // one side of the quotient only (wmap29.hpp): the other matrix is somebody else's
        if (k) fill_zero(v[k], D * 32, st);
        const uint32_t* src = buf.w29.p;
        for (int lv = 0; lv < mats[k]->n_sell; ++lv) {
            const SellLevel& L = mats[k]->sell[lv];
            uint32_t* scratch = (lv & 1) ? buf.sp_b.p : buf.sp_a.p;
            if (L.n_partials > buf.sp_cap) throw HipError(CG_ERR_INVALID_ARGUMENT, "sparse-product scratch smaller than the matrix needs");
            if (L.n_pieces) {
                k_sell29<<<ceil_div(L.n_pieces, 256), 256, 0, st>>>(L.slice_ptr.p, L.col.p, L.cidx.p, L.dst.p, dicts[k]->dict.p, src, v[k],
                                                                    scratch, L.n_pieces, logn);
                CG_KERNEL_CHECK();
            }
            src = scratch;
        }
    }
    if (coset_values && strided) {
        // a shard's own coset points only (Wm29Strided): two transforms of size D, two of size d
        const Wm29Strided& S = *strided;
        const uint32_t d = (uint32_t)S.d;
        const int logd = S.sub.logn;
        const uint32_t terms = 1u << S.logs;
        dit29(dom, dom.tw_inv.p, buf.va.p, nullptr, nullptr, false, buf.va.p, buf.vt.p, nullptr, false, st);          // n·a_e, natural order
        k_fold29<<<ceil_div(d, 256), 256, 0, st>>>(buf.vt.p, S.fold.p, buf.va.p, d, logd, terms);
        CG_KERNEL_CHECK();
        dit29(S.sub, S.sub.tw_fwd.p, buf.va.p, nullptr, nullptr, false, buf.va.p, buf.vc.p, nullptr, false, st, dom.vinv_plain);   // vinv·a, plain
        dit29(dom, dom.tw_inv.p, buf.vb.p, nullptr, nullptr, false, buf.vb.p, buf.vt.p, nullptr, false, st);
        k_fold29<<<ceil_div(d, 256), 256, 0, st>>>(buf.vt.p, S.fold.p, buf.vb.p, d, logd, terms);
        CG_KERNEL_CHECK();
        dit29(S.sub, S.sub.tw_fwd.p, buf.vb.p, nullptr, nullptr, false, buf.vb.p, reinterpret_cast<uint32_t*>(h_out), nullptr, false, st, nullptr,
              buf.vc.p);
        return;
    }
    if (coset_values && half) {
        // ONE side of the quotient's a∘b part on the coset, as plain canonical integers in natural order: vinv·a(gω^j) (half 1)
        // or b(gω^j) (half 2) - two transforms and one sparse product; q_j is their product mod r (fr_mul_plain29)
        static const uint32_t one_plain[8] = {1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        uint32_t* x = half == 1 ? buf.va.p : buf.vb.p;
        dit29(dom, dom.tw_inv.p, x, nullptr, nullptr, false, x, buf.vt.p, dom.coset.p, true, st);
        dit29(dom, dom.tw_fwd.p, buf.vt.p, nullptr, nullptr, false, buf.vt.p, reinterpret_cast<uint32_t*>(h_out), nullptr, false, st,
              half == 1 ? dom.vinv_plain : one_plain);
        return;
    }
    if (coset_values) {
        // four transforms: q_j = vinv·a(gω^j)·b(gω^j), the scalars of the h MSM over the transformed h query
        dit29(dom, dom.tw_inv.p, buf.va.p, nullptr, nullptr, false, buf.va.p, buf.vt.p, dom.coset.p, true, st);
        dit29(dom, dom.tw_fwd.p, buf.vt.p, nullptr, nullptr, false, buf.vt.p, buf.va.p, nullptr, false, st, dom.vinv_plain);   // vinv·a, plain
        dit29(dom, dom.tw_inv.p, buf.vb.p, nullptr, nullptr, false, buf.vb.p, buf.vt.p, dom.coset.p, true, st);
        dit29(dom, dom.tw_fwd.p, buf.vt.p, nullptr, nullptr, false, buf.vt.p, reinterpret_cast<uint32_t*>(h_out), nullptr, false, st, nullptr,
              buf.va.p);
        return;
    }
    for (int k = 0; k < 3; ++k) {
        // ifft (bit-reversed in), then x g^i / n, stored bit-reversed for the next transform      :179-185,198-199
        dit29(dom, dom.tw_inv.p, v[k], nullptr, nullptr, false, v[k], buf.vt.p, dom.coset.p, true, st);
        // fft on the coset; stored bit-reversed so the pointwise load below feeds the last transform directly
        dit29(dom, dom.tw_fwd.p, buf.vt.p, nullptr, nullptr, false, buf.vt.p, v[k], nullptr, true, st);
    }
    // (a∘b - c)/Z(g) on load; coset ifft; x g^-i / n and out of Montgomery form on store           :187,201-210
    dit29(dom, dom.tw_inv.p, buf.va.p, buf.vb.p, buf.vc.p, true, buf.va.p, reinterpret_cast<uint32_t*>(h_out), dom.icoset.p, false, st);
}

// out[i] = a[i]·b[i] mod r for plain canonical operands (two halves of the quotient's a∘b part -> the h MSM's scalars); a
// non-canonical operand sets *bad_input
__global__ void __launch_bounds__(256) k_mul_plain29(const Fr* __restrict__ a, const Fr* __restrict__ b, uint32_t* __restrict__ out, uint64_t n,
                                                     uint32_t* __restrict__ bad_input) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr x = a[i], y = b[i];
    if (!fr_lt_modulus(x) || !fr_lt_modulus(y)) *bad_input = 1u;
    const Fr29 v = mul(unpack29<Fr29P>(x.l), Fr29::from_limbs(Fr29P::R2));      // x·R', below 2N
    store_packed29(out, i, cond_sub_n(mul(v, unpack29<Fr29P>(y.l))));            // x·y, canonical
}
void fr_mul_plain29(const Fr* a, const Fr* b, Fr* out, uint64_t n, uint32_t* bad_input_dev, hipStream_t st) {
    if (!n) return;
    k_mul_plain29<<<ceil_div(n, 256), 256, 0, st>>>(a, b, reinterpret_cast<uint32_t*>(out), n, bad_input_dev);
    CG_KERNEL_CHECK();
}

// ---- unit-level transform (cg_ntt_*): canonical natural-order data in place -----------------------------------------
// in[i] canonical -> R' form (times g^i for the forward coset transform), stored at the bit-reversed index
__global__ void __launch_bounds__(256) k_unit_in29(const Fr* __restrict__ in, uint32_t* __restrict__ out, uint64_t n, int logn,
                                                   const uint32_t* __restrict__ premul, uint32_t* __restrict__ bad_input) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = in[i];
    if (!fr_lt_modulus(x)) *bad_input = 1u;
    Fr29 v = from_canonical_bytes<Fr29P>(x);
    if (premul) v = mul(v, load_packed29(premul, i));
    store_packed29(out, brev((uint32_t)i, logn), v);
}

void Ntt29Unit::build(int logn_, hipStream_t st) {
    logn = logn_;
    n = 1ull << logn;
    NttDomain d;
    d.build(logn, true, st);
    dom.build(d, st);
    // g^i in R' form at natural index i (pre-scale of the forward coset transform, r1cs_to_qap.rs:182-185)
    DevBuf<Fr> g(n);
    fr_pow_table(g.p, fr_from_u64(5), Fr::one(), n, false, logn, st);
    gpow.alloc(n * 8);
    k_to_packed29<<<ceil_div(n, 256), 256, 0, st>>>(g.p, gpow.p, n, 0, 0);
    CG_KERNEL_CHECK();
    work.alloc(n * 8);
    h_bad_input.alloc(1);
    // plain 1 and plain 1/n as packed constants
    memset(one_plain, 0, 32);
    one_plain[0] = 1;
    Fr ninv = from_mont(inv(fr_from_u64(n)));
    memcpy(ninv_plain, ninv.l, 32);
    CG_HIP(hipStreamSynchronize(st));   // d and g are released on return
}

bool Ntt29Unit::run(Fr* data_dev, bool inverse, bool coset, hipStream_t st) {
    h_bad_input.p[0] = 0;
    k_unit_in29<<<ceil_div(n, 256), 256, 0, st>>>(data_dev, work.p, n, logn, (!inverse && coset) ? gpow.p : nullptr, h_bad_input.dev());
    CG_KERNEL_CHECK();
    uint32_t* out = reinterpret_cast<uint32_t*>(data_dev);
    if (!inverse) dit29(dom, dom.tw_fwd.p, work.p, nullptr, nullptr, false, work.p, out, nullptr, false, st, one_plain);
    else if (!coset) dit29(dom, dom.tw_inv.p, work.p, nullptr, nullptr, false, work.p, out, nullptr, false, st, ninv_plain);
    else dit29(dom, dom.tw_inv.p, work.p, nullptr, nullptr, false, work.p, out, dom.icoset.p, false, st);
    CG_HIP(hipStreamSynchronize(st));
    return h_bad_input.p[0] == 0;
}

}  // namespace cg
