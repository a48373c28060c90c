// Pippenger MSM over BN254 G1/G2 for gfx950.
//
// Computes Σ s_i·P_i, the value of ark-ec's `VariableBaseMSM::msm_bigint` at the call sites
// forks/groth16/src/prover.rs:66,74,266.  The sum is a unique group element, so the window size,
// the signed-digit recoding and the order of additions are free design choices; only the final
// affine point is observable (prover.rs:131-135).
//
// Pipeline (all on one stream, no step waits for the host):
//   group    : every canonical scalar -> W signed c-bit digits; zero digits and identity bases are dropped;
//              survivors become one 64-bit entry each (bucket << 32 | table index | sign) and are PLACED
//              next to the other entries of their bucket by a two-level counting partition with the digit
//              extraction fused into both of its passes (k_part_*): no sort, no compaction scan.
//   accumulate: the grouped list is cut into equal segments, one per lane, so every lane of every
//              wave performs the same number of mixed additions regardless of how skewed the
//              buckets are (circom witnesses are dominated by 0/1 wires).  A lane flushes runs that
//              lie wholly inside its segment straight to the bucket array and hands the first/last
//              run up as a piece; pieces are combined 64 segments per wave (k_combine_wave).
//   reduce   : Σ (b+1)·S_b through row and column sums of the bucket matrix and per-bit sums of those,
//              folded on the host.
// All curve arithmetic here runs on the lazy 29-bit-limb representation (field29.hpp / curve29.hpp).
#include "msm.hpp"
#include "g2pair.hpp"

#include <chrono>
#include <cstring>
#include <type_traits>

namespace cg {

static constexpr int SCALAR_BITS = 255;  // c*W must reach bit 254 plus the recoding carry

// ---------------------------------------------------------------------------------------------
// window choice
// ---------------------------------------------------------------------------------------------
int msm_default_window(uint64_t n, bool precomputed) {
    double best = 1e300;
    int best_c = 4;
    for (int c = 4; c <= 22; ++c) {
        int W = (SCALAR_BITS + c - 1) / c;
        if ((double)W * (double)n >= 2147483648.0) continue;
        double buckets = (double)(1u << (c - 1)) * (precomputed ? 1 : W);
        if (buckets > (double)(1u << 24)) continue;
        // mixed add ~ 12 field products, bucket-reduction add ~ 16, two per bucket
        double cost = (double)n * W * 12.0 + buckets * 2.0 * 16.0;
        if (cost < best) { best = cost; best_c = c; }
    }
    return best_c;
}

int msm_best_window(uint64_t n_bases, double nz_small, double nz_full) {
    double best = 1e300;
    int best_c = 4;
    for (int c = 6; c <= 22; ++c) {
        int W = (SCALAR_BITS + c - 1) / c;
        if ((double)W * (double)n_bases >= 2147483648.0) continue;
        double entries = nz_small + nz_full * W;
        double cost = entries * 12.0 + (double)(1u << (c - 1)) * 2.0 * 16.0;
        if (cost < best) { best = cost; best_c = c; }
    }
    return best_c;
}

// ---------------------------------------------------------------------------------------------
// base import / window tables
// ---------------------------------------------------------------------------------------------
template <class F> struct Coords;
template <> struct Coords<Fq> { static constexpr int N = 2; };
template <> struct Coords<Fq2> { static constexpr int N = 4; };

// raw[i] holds 2 (G1) or 4 (G2) 32-byte field elements; identity = all zero
template <class F>
__global__ void __launch_bounds__(256) k_import(const Fq* __restrict__ raw, Affine<F>* __restrict__ out, uint64_t n, int to_mont_form) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int NC = Coords<F>::N;
    Fq c[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        c[k] = raw[i * NC + k];
        if (to_mont_form) c[k] = to_mont(c[k]);
    }
    Affine<F> p;
    if constexpr (NC == 2) { p.x = c[0]; p.y = c[1]; }
    else { p.x = {c[0], c[1]}; p.y = {c[2], c[3]}; }
    out[i] = p;
}

template <class F>
void import_bases(const uint8_t* host_bytes, uint32_t coord_form, uint64_t n, Affine<F>* out_dev, hipStream_t st) {
    if (!n) return;
    constexpr int NC = Coords<F>::N;
    DevBuf<Fq> raw(n * NC);
    CG_HIP(hipMemcpyAsync(raw.p, host_bytes, n * NC * 32, hipMemcpyHostToDevice, st));
    k_import<F><<<ceil_div(n, 256), 256, 0, st>>>(raw.p, out_dev, n, coord_form == CG_FORM_CANONICAL ? 1 : 0);
    CG_KERNEL_CHECK();
    CG_HIP(hipStreamSynchronize(st));  // raw is freed on return
}
template void import_bases<Fq>(const uint8_t*, uint32_t, uint64_t, Affine<Fq>*, hipStream_t);
template void import_bases<Fq2>(const uint8_t*, uint32_t, uint64_t, Affine<Fq2>*, hipStream_t);

// row 0: Montgomery(2^256) points -> packed table points (x·2^261 mod q)
template <class F>
__global__ void __launch_bounds__(256) k_table_first(const Affine<F>* __restrict__ bases, uint32_t* __restrict__ table,
                                                     uint8_t* __restrict__ valid, uint64_t n) {
    typedef typename To29<F>::type F29T;
    constexpr int AFF = Words29<F29T>::AFF;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = bases[i];
    uint32_t w[AFF];
    const bool inf = p.is_inf();
    if (inf) {
#pragma unroll
        for (int k = 0; k < AFF; ++k) w[k] = 0;
    } else {
        pack_table_point(p, w);
    }
    uint4* dst = reinterpret_cast<uint4*>(table + i * AFF);
#pragma unroll
    for (int k = 0; k < AFF / 4; ++k) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    valid[i] = inf ? 0 : 1;
}

// row j = 2^c * row (j-1)
template <class F29T>
__global__ void __launch_bounds__(256) k_table_next(uint32_t* __restrict__ table, const uint8_t* __restrict__ valid,
                                                    uint64_t n, int j, int c) {
    constexpr int AFF = Words29<F29T>::AFF;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* dstw = table + ((uint64_t)j * n + i) * AFF;
    uint32_t w[AFF];
    if (!valid[i]) {
#pragma unroll
        for (int k = 0; k < AFF; ++k) w[k] = 0;
    } else {
        Affine29<F29T> p = load_table_point<F29T>(table + (uint64_t)(j - 1) * n * AFF, (uint32_t)i, false);
        XYZZ29<F29T> a = dbl_affine29(p);
        for (int k = 1; k < c; ++k) a = dbl29(a);       // a point of odd prime order never doubles to the identity
        store_table_point_from_xyzz(a, w);
    }
    uint4* dst = reinterpret_cast<uint4*>(dstw);
#pragma unroll
    for (int k = 0; k < AFF / 4; ++k) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

template <class F>
void MsmBases<F>::alloc_rows(uint64_t n_, int c_, bool precompute) {
    n = n_;
    c = c_;
    W = (SCALAR_BITS + c - 1) / c;
    precomputed = precompute;
    if ((uint64_t)W * n >= (1ull << 31)) throw HipError(CG_ERR_INVALID_ARGUMENT, "MSM too large for 31-bit table indices");
    const uint64_t rows = precompute ? (uint64_t)W : 1;
    table.alloc(n ? rows * n * AFF : 4);
    valid.alloc(n ? n : 1);
}
template <class F>
void MsmBases<F>::expand_rows(hipStream_t st) {
    if (!precomputed || !n) return;
    for (int j = 1; j < W; ++j) {
        k_table_next<F29T><<<ceil_div(n, 256), 256, 0, st>>>(table.p, valid.p, n, j, c);
        CG_KERNEL_CHECK();
    }
}

template <class F>
void MsmBases<F>::build(const Affine<F>* bases_dev, uint64_t n_, int c_, bool precompute, hipStream_t st) {
    alloc_rows(n_, c_, precompute);
    if (!n) return;
    k_table_first<F><<<ceil_div(n, 256), 256, 0, st>>>(bases_dev, table.p, valid.p, n);
    CG_KERNEL_CHECK();
    expand_rows(st);
}

template <class F>
void MsmBases<F>::build_from_row0(const uint32_t* row0_dev, const uint8_t* valid_dev, uint64_t n_, int c_, hipStream_t st) {
    alloc_rows(n_, c_, true);
    if (!n) return;
    CG_HIP(hipMemcpyAsync(table.p, row0_dev, n * AFF * 4, hipMemcpyDeviceToDevice, st));
    CG_HIP(hipMemcpyAsync(valid.p, valid_dev, n, hipMemcpyDeviceToDevice, st));
    expand_rows(st);
}

template <class F>
int MsmBases<F>::rebuild(int c_new, hipStream_t st) {
    if (!precomputed || c_new == c || !n) return 0;
    const int W_new = (SCALAR_BITS + c_new - 1) / c_new;
    if ((uint64_t)W_new * n >= (1ull << 31)) return 0;
    {   // the old and the new table coexist until the swap: never re-tune into an out-of-memory failure
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return -1;
        const uint64_t need = (uint64_t)W_new * n * AFF * 4;
        if (need + (2ull << 30) > free_b) return -1;
    }
    DevBuf<uint32_t> t2((uint64_t)W_new * n * AFF);
    CG_HIP(hipMemcpyAsync(t2.p, table.p, n * AFF * 4, hipMemcpyDeviceToDevice, st));   // row 0 = the bases themselves
    for (int j = 1; j < W_new; ++j) {
        k_table_next<F29T><<<ceil_div(n, 256), 256, 0, st>>>(t2.p, valid.p, n, j, c_new);
        CG_KERNEL_CHECK();
    }
    CG_HIP(hipStreamSynchronize(st));
    table = std::move(t2);
    c = c_new;
    W = W_new;
    return 1;
}

// ---------------------------------------------------------------------------------------------
// signed-digit extraction
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t limb_at(const uint32_t s[8], int idx) {
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r = (idx == i) ? s[i] : r;
    return r;
}
// Calls f(j, digit) for every window with a non-zero digit candidate; digit in [-2^(c-1), 2^(c-1)], c <= 22.
// The scalar's bits stream through a 64-bit buffer that is refilled one limb at a time (a dynamic limb index costs
// eight selects, so it is paid per limb, not per window), and the walk ends as soon as nothing is left above the
// current window — after one step for the 0/1 wires that make up most of a circom witness.
template <class Fn>
__device__ __forceinline__ void for_each_digit(const uint32_t s[8], int c, int W, Fn f) {
    int hl = -1;                                   // highest non-zero limb
#pragma unroll
    for (int i = 0; i < 8; ++i) hl = s[i] ? i : hl;
    if (hl < 0) return;
    uint64_t buf = (uint64_t)s[0] | ((uint64_t)s[1] << 32);
    int avail = 64, next = 2;
    uint32_t carry = 0;
    const uint32_t mask = (1u << c) - 1u, half = 1u << (c - 1);
    for (int j = 0; j < W; ++j) {
        const uint32_t raw = ((uint32_t)buf & mask) + carry;
        buf >>= c;
        avail -= c;
        if (avail <= 32 && next < 8) {
            buf |= (uint64_t)limb_at(s, next) << avail;
            avail += 32;
            ++next;
        }
        int32_t d;
        if (raw > half) { d = (int32_t)raw - (int32_t)(1u << c); carry = 1; }
        else { d = (int32_t)raw; carry = 0; }
        f(j, d);
        if (buf == 0 && carry == 0 && next > hl) break;
    }
}

// The same walk with the window size known at compile time: every digit's limb index and shift are constants, so a digit
// costs one funnel shift, the mask, the carry and the sign - no bit buffer, no refills, no dynamic limb selection (the
// generic walk spends ~40 instructions per digit, most of them on those).  The walk still ends where the scalar does.
template <int I, int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// (The walk is unrolled by template recursion, so every limb index below is a literal.)
template <int C, class Fn>
__device__ __forceinline__ void for_each_digit_c(const uint32_t (&s)[8], Fn f) {
    constexpr int W = (SCALAR_BITS + C - 1) / C;
    constexpr uint32_t mask = (1u << C) - 1u, half = 1u << (C - 1);
    int hl = -1;
    uint32_t top = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hl = s[i] ? i : hl;
        top = s[i] ? s[i] : top;
    }
    if (hl < 0) return;
    const int topbit = hl * 32 + 31 - __clz((int)top);         // index of the scalar's highest set bit
    uint32_t carry = 0;
    bool live = true;
    static_for<0, W>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int bit = C * j, lo = bit >> 5, sh = bit & 31;
        if (!live) return;
        if (bit > topbit && carry == 0) { live = false; return; }      // nothing left above this window
        // the window's bits lie in limb lo and, when it straddles, limb lo + 1: one funnel shift (v_alignbit_b32) of the two
        // limbs.  Rounds 2-4 wrote it as `(s[lo] | (uint64_t)s[lo + 1] << 32) >> sh`: the optimiser merges the two adjacent
        // 4-byte reads into one 8-byte read at offset 4·lo of the scalar, an access the pass that moves small arrays into
        // registers does not split, so the four scalars of a thread stayed in SCRATCH in every level-1 kernel but the
        // C = 16 one: 144 bytes per thread, 230 MB written and read back per proof
        // (profiles/r04_n_hbm_write_per_proof_gates.md; `.private_segment_fixed_size` is 0 now, tests/test_abi.py)
        uint32_t w = 0;
        if constexpr (lo < 8) {
            if constexpr (sh == 0) w = s[lo];
            else if constexpr (lo + 1 < 8) w = __builtin_amdgcn_alignbit(s[lo + 1], s[lo], sh);
            else w = s[lo] >> sh;
        }
        const uint32_t raw = (w & mask) + carry;
        int32_t d;
        if (raw > half) { d = (int32_t)raw - (int32_t)(1u << C); carry = 1; }
        else { d = (int32_t)raw; carry = 0; }
        f(j, d);
    });
}

// ---------------------------------------------------------------------------------------------
// grouping the digit entries by bucket: a two-level counting partition, no sort
//
// The accumulation only needs the entries of a bucket to be contiguous; their order inside the bucket and the order of
// the buckets do not matter (the sum is a group element).  So the entries are never sorted: they are PLACED.
//   level 1  k_part_count   every block walks its tile of scalars, extracts the signed digits and counts them by the
//                           high bits of the bucket key in LDS; the block's counts go to its row of `blk_hist` and,
//                           with one atomic per non-empty bin, to the global histogram
//            k_part_plan    one block: entry total, segment length and count for the accumulation, bin starts
//                           (exclusive scan), and the chunk table of level 2
//            k_part_place   the same walk again; a block reserves its range of every bin with one global atomic and
//                           hands out slots inside it with LDS atomics: digit extraction is fused into the placement,
//                           so the entry list is written once and never read back by a "sort"
//   level 2  k_part_count2 / k_part_place2   the same on the low key bits inside every level-1 bin, chunk by chunk
//                           (a chunk is at most PART_CHUNK entries of one bin, so a dominant bucket - the 0/1 wires
//                           of a circom witness - is spread over many blocks)
// Key spaces of at most PART_MAX_BITS bits take level 1 only.  Everything the later kernels need to know about the
// entry count lives in `plan` on the device: no kernel launch waits for the host.
// ---------------------------------------------------------------------------------------------
static constexpr int PART_MAX_BITS = 12;          // bins of one level: at most 4096 LDS counters
static constexpr uint32_t PART_TILE = 4096;       // scalars per block, level 1 (few, large blocks: one global atomic per bin per block)
static constexpr uint32_t PART_THREADS = 1024;    // threads of a level-1 block
static constexpr uint32_t PART_CHUNK = 8192;      // entries per block, level 2
static constexpr uint32_t PART_PAD = 16;          // level-1 global counters sit 64 B apart: atomics of different bins do not share a line
enum { PLAN_N = 0, PLAN_L = 1, PLAN_T = 2, PLAN_NONZERO = 3, PLAN_CHUNKS = 4, PLAN_DONE = 5, PLAN_WORDS = 8 };   // PLAN_DONE: counting blocks that have finished

struct PartShape {
    uint32_t n;            // scalars
    int c, W;              // window bits, windows
    int precomputed;       // 1: key = |d| - 1, value = j * row_stride + i;  0: key = j * nb + |d| - 1, value = i
    uint32_t row_stride;
    int bits1, bits2;      // key bits taken by level 1 (high) and level 2 (low; 0 = single level)
    int staged;            // host side only: level-1 placement through the LDS staging area (latency contexts)
};

// A scalar's eight words as eight VALUES.  (`Fr sc = ok ? scalars[i] : Fr::zero()` is a 32-byte copy from one of two
// addresses; the optimiser keeps such an object in memory, and every level-1 kernel instantiated for a window size carried
// 144 bytes of scratch per thread for its four scalars - 230 MB written and read back per proof, profiles/
// r04_n_hbm_write_per_proof_gates.md.  Two 16-byte loads into named words stay in registers.)
__device__ __forceinline__ Fr load_scalar_or_zero(const Fr* __restrict__ scalars, uint32_t i, bool ok) {
    uint4 a = make_uint4(0u, 0u, 0u, 0u), b = a;
    if (ok) {
        const uint4* p = reinterpret_cast<const uint4*>(scalars + i);
        a = p[0];
        b = p[1];
    }
    Fr r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}

// C = the window size when the kernel was instantiated for it, 0 = taken from the shape at run time
template <int C, class Fn>
__device__ __forceinline__ void for_each_entry(const PartShape& sh, const Fr& s, uint32_t i, Fn f) {
    const uint32_t nb = 1u << (sh.c - 1);
    auto emit = [&](int j, int32_t d) {
        if (d == 0) return;
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        const uint32_t sign = d < 0 ? 0x80000000u : 0u;
        if (sh.precomputed) f(mag - 1u, ((uint32_t)j * sh.row_stride + i) | sign);
        else f((uint32_t)j * nb + (mag - 1u), i | sign);
    };
    if constexpr (C > 0) for_each_digit_c<C>(s.l, emit);
    else for_each_digit(s.l, sh.c, sh.W, emit);
}

// what the plan needs besides the histogram (k_part_count's last block writes the plan: part_plan_block)
struct PlanArgs {
    uint32_t* start1;          // B1 + 1: first slot of every level-1 bin
    uint32_t* chunk0;          // B1 + 1: first level-2 chunk of every bin
    uint32_t target_threads, min_L;
    int two_level;
};
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t* sc, uint32_t n);
__device__ __forceinline__ void part_plan_block(int bits1, const PlanArgs& pa, const uint32_t* hist1, uint32_t* plan, uint32_t* lds);

template <int C>
__global__ void __launch_bounds__(PART_THREADS) k_part_count(PartShape sh, const Fr* __restrict__ scalars, const uint8_t* __restrict__ valid,
                                                    uint32_t* __restrict__ blk_hist, uint32_t* hist1, uint32_t* plan, PlanArgs pa) {
    extern __shared__ uint32_t lds[];
    const uint32_t B1 = 1u << sh.bits1;
    for (uint32_t k = threadIdx.x; k < B1; k += blockDim.x) lds[k] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * PART_TILE;
    uint32_t nonzero = 0;
    constexpr int PER = PART_TILE / PART_THREADS;       // scalars per thread: all loaded before any is walked
    Fr sc[PER];
    bool ok[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const uint32_t i = base + u * PART_THREADS + threadIdx.x;
        ok[u] = i < sh.n && valid[i] != 0;
        sc[u] = load_scalar_or_zero(scalars, i, ok[u]);
    }
    static_for<0, PER>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        if (!ok[u]) return;
        bool any = false;
        for_each_entry<C>(sh, sc[u], base + u * PART_THREADS + threadIdx.x, [&](uint32_t key, uint32_t) {
            atomicAdd(&lds[key >> sh.bits2], 1u);
            any = true;
        });
        nonzero += any;
    });
    // one atomic per block: same-address atomics serialise at ~10 ns each
    __shared__ uint32_t nz_block;
    if (threadIdx.x == 0) nz_block = 0;
    for (int off = 32; off > 0; off >>= 1) nonzero += __shfl_down((int)nonzero, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && nonzero) atomicAdd(&nz_block, nonzero);
    __syncthreads();
    if (threadIdx.x == 0 && nz_block) atomicAdd(&plan[PLAN_NONZERO], nz_block);
    uint32_t* row = blk_hist + (size_t)blockIdx.x * B1;
    for (uint32_t k = threadIdx.x; k < B1; k += blockDim.x) {
        const uint32_t v = lds[k];
        row[k] = v;
        if (v) atomicAdd(&hist1[(size_t)k * PART_PAD], v);
    }
    // The block that finishes LAST turns the histogram into the plan (entry total, segment geometry, bin starts, chunk
    // table) - what a one-block launch of its own did before.  Everything the blocks hand to it goes through agent-scope
    // ATOMICS (the histogram adds, the PLAN_DONE count, the last block's loads), which are performed at the coherence
    // point of the eight XCDs' L2s: each wave waits for its own adds to be ACKNOWLEDGED - an explicit `s_waitcnt vmcnt(0)`:
    // the no-return global atomics are vector-memory operations, and a workgroup-scope release alone only drains
    // lgkmcnt in non-tgsplit mode (round 4 shipped without it: the ISA went from the atomic loop straight into s_barrier,
    // so the last block could in principle read a short histogram; tests/test_abi.py now checks the disassembly) - the
    // block meets at the barrier, one thread counts the block in.  No __threadfence(): at agent scope that is an L2
    // write-back + invalidate, and with it in every wave of this kernel the pipeline lost 17 % (profiles/r04_f_fold_ab.txt);
    // the s_waitcnt writes nothing back.
    if (!pa.start1) return;            // tuning builds, CG_PLAN_LAUNCH=1 (A/B aid): the plan is made by a launch of its own
    __shared__ uint32_t is_last;
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0) (gfx9 encoding: expcnt and lgkmcnt left at their maxima)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0)
        is_last = __hip_atomic_fetch_add(&plan[PLAN_DONE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
    __syncthreads();
    if (!is_last) return;
    part_plan_block(sh.bits1, pa, hist1, plan, lds);
}

// exclusive scan of `vals` (n <= 4096, in LDS `sc` of n words) by one block; total returned to every thread
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t* sc, uint32_t n) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t grand;
    const uint32_t per = (n + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi; ++k) sum += sc[k];
    uint32_t incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
        if ((threadIdx.x & 63) >= (uint32_t)off) incl += v;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < (blockDim.x + 63) / 64; ++w) { uint32_t t = wave_tot[w]; wave_tot[w] = run; run += t; }
        grand = run;
    }
    __syncthreads();
    uint32_t run = wave_tot[threadIdx.x >> 6] + incl - sum;
    for (uint32_t k = lo; k < hi; ++k) { uint32_t t = sc[k]; sc[k] = run; run += t; }
    __syncthreads();
    return grand;
}

// start1[b] = first slot of level-1 bin b (B1 + 1 values); chunk0[b] = first level-2 chunk of bin b (B1 + 1 values).
// Run by ONE block (the last block of k_part_count); lds: B1 words.
__device__ __forceinline__ void part_plan_block(int bits1, const PlanArgs& pa, const uint32_t* hist1, uint32_t* plan, uint32_t* lds) {
    const uint32_t B1 = 1u << bits1;
    auto hist = [&](uint32_t k) { return __hip_atomic_load(&hist1[(size_t)k * PART_PAD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < B1; k += blockDim.x) lds[k] = hist(k);
    __syncthreads();
    const uint32_t N = block_exclusive_scan(lds, B1);
    for (uint32_t k = threadIdx.x; k < B1; k += blockDim.x) pa.start1[k] = lds[k];
    if (threadIdx.x == 0) {
        pa.start1[B1] = N;
        uint32_t L = (uint32_t)(((uint64_t)N + pa.target_threads - 1) / pa.target_threads);
        if (L < pa.min_L) L = pa.min_L;
        plan[PLAN_N] = N;
        plan[PLAN_L] = L;
        plan[PLAN_T] = N ? (N + L - 1) / L : 0;
    }
    if (!pa.two_level) return;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < B1; k += blockDim.x) lds[k] = (hist(k) + PART_CHUNK - 1) / PART_CHUNK;
    __syncthreads();
    const uint32_t chunks = block_exclusive_scan(lds, B1);
    for (uint32_t k = threadIdx.x; k < B1; k += blockDim.x) pa.chunk0[k] = lds[k];
    if (threadIdx.x == 0) { pa.chunk0[B1] = chunks; plan[PLAN_CHUNKS] = chunks; }
}

__global__ void __launch_bounds__(1024) k_part_plan(int bits1, PlanArgs pa, const uint32_t* hist1, uint32_t* plan) {
    extern __shared__ uint32_t lds[];
    part_plan_block(bits1, pa, hist1, plan, lds);
}

// a plan taken over with another engine's entries: the segment geometry is re-cut for THIS engine's accumulation kernel
// (the entry list itself does not care where it is cut)
__global__ void k_replan(uint32_t* __restrict__ plan, const uint32_t* __restrict__ src_plan, uint32_t target_segments, uint32_t min_L) {
    for (int k = 0; k < PLAN_WORDS; ++k) plan[k] = src_plan[k];      // entry count and statistics travel with the entries
    const uint32_t N = plan[PLAN_N];
    uint32_t L = (uint32_t)(((uint64_t)N + target_segments - 1) / target_segments);
    if (L < min_L) L = min_L;
    plan[PLAN_L] = L;
    plan[PLAN_T] = N ? (N + L - 1) / L : 0;
}

template <int C>
__global__ void __launch_bounds__(PART_THREADS) k_part_place(PartShape sh, const Fr* __restrict__ scalars, const uint8_t* __restrict__ valid,
                                                    const uint32_t* __restrict__ blk_hist, const uint32_t* __restrict__ start1,
                                                    uint32_t* __restrict__ cur1, uint64_t* __restrict__ out) {
    extern __shared__ uint32_t lds[];                 // [B1] next slot of the bin for this block
    const uint32_t B1 = 1u << sh.bits1;
    const uint32_t* row = blk_hist + (size_t)blockIdx.x * B1;
    for (uint32_t k = threadIdx.x; k < B1; k += blockDim.x) {
        const uint32_t cnt = row[k];
        lds[k] = cnt ? start1[k] + atomicAdd(&cur1[(size_t)k * PART_PAD], cnt) : 0u;
    }
    const uint32_t base = blockIdx.x * PART_TILE;
    constexpr int PER = PART_TILE / PART_THREADS;
    Fr sc[PER];
    bool ok[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const uint32_t i = base + u * PART_THREADS + threadIdx.x;
        ok[u] = i < sh.n && valid[i] != 0;
        sc[u] = load_scalar_or_zero(scalars, i, ok[u]);
    }
    __syncthreads();
    static_for<0, PER>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        if (!ok[u]) return;
        for_each_entry<C>(sh, sc[u], base + u * PART_THREADS + threadIdx.x, [&](uint32_t key, uint32_t val) {
            const uint32_t pos = atomicAdd(&lds[key >> sh.bits2], 1u);
            out[pos] = ((uint64_t)key << 32) | val;
        });
    });
}

// Level-1 placement staged through LDS, for the wide windows (C >= 18: at most 15 digits per scalar).  k_part_place hands every
// entry to its bin with one 8-byte global store per lane - 64 stores to 64 different bins per wave instruction, 27 M
// separate write transactions for the h MSM - which makes it the slowest kernel of the grouping (0.28 ms stand-alone
// against 0.05 ms for the counting pass that does the same walk).  Here a block takes its tile in sub-tiles of 1024
// scalars: the entries of a sub-tile stay in registers while they are counted by bin in LDS, are then placed bin by bin
// into an LDS staging area, and leave it in RUNS: consecutive lanes write consecutive slots of the same bin (13 entries
// = 104 contiguous bytes per bin and sub-tile at c = 20).  The block's range of every bin is reserved once, as before.
static constexpr uint32_t STAGE_SUB = 1024;        // scalars per sub-tile = threads per block
static constexpr int STAGE_MIN_C = 18;
template <int C>
__global__ void __launch_bounds__(PART_THREADS) k_part_place_staged(PartShape sh, const Fr* __restrict__ scalars, const uint8_t* __restrict__ valid,
                                                           const uint32_t* __restrict__ blk_hist, const uint32_t* __restrict__ start1,
                                                           uint32_t* __restrict__ cur1, uint64_t* __restrict__ out) {
    constexpr int W = (SCALAR_BITS + C - 1) / C;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t B1 = 1u << sh.bits1;
    uint32_t* gbase = lds;                 // [B1] next global slot of the bin for this block
    uint32_t* cnt = lds + B1;              // [B1] entries of the sub-tile per bin, then the fill cursor
    uint32_t* off = lds + 2 * B1;          // [B1] first staging slot of the bin
    uint64_t* stage = reinterpret_cast<uint64_t*>(lds + 3 * B1 + (B1 & 1u));      // [STAGE_SUB * W]
    const uint32_t* row = blk_hist + (size_t)blockIdx.x * B1;
    for (uint32_t k = threadIdx.x; k < B1; k += blockDim.x) {
        const uint32_t c = row[k];
        gbase[k] = c ? start1[k] + atomicAdd(&cur1[(size_t)k * PART_PAD], c) : 0u;
    }
    const uint32_t base = blockIdx.x * PART_TILE;
    for (uint32_t sub = 0; sub < PART_TILE / STAGE_SUB; ++sub) {
        const uint32_t i = base + sub * STAGE_SUB + threadIdx.x;
        for (uint32_t k = threadIdx.x; k < B1; k += blockDim.x) cnt[k] = 0;
        __syncthreads();
        uint64_t e[W];                     // entry of window j, or all ones: indexed by the unrolled walk's constant j
#pragma unroll
        for (int u = 0; u < W; ++u) e[u] = ~0ull;
        if (i < sh.n && valid[i] != 0) {
            const Fr sc = scalars[i];
            for_each_digit_c<C>(sc.l, [&](int j, int32_t d) {
                if (d == 0) return;
                const uint32_t key = (uint32_t)(d < 0 ? -d : d) - 1u;
                const uint32_t val = ((uint32_t)j * sh.row_stride + i) | (d < 0 ? 0x80000000u : 0u);
                e[j] = ((uint64_t)key << 32) | val;
                atomicAdd(&cnt[key >> sh.bits2], 1u);
            });
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < B1; k += blockDim.x) off[k] = cnt[k];
        __syncthreads();
        const uint32_t total = block_exclusive_scan(off, B1);
        for (uint32_t k = threadIdx.x; k < B1; k += blockDim.x) cnt[k] = off[k];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < W; ++u)
            if (e[u] != ~0ull) stage[atomicAdd(&cnt[(uint32_t)(e[u] >> 32) >> sh.bits2], 1u)] = e[u];
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < total; k += blockDim.x) {
            const uint64_t v = stage[k];
            const uint32_t b = (uint32_t)(v >> 32) >> sh.bits2;
            out[gbase[b] + (k - off[b])] = v;
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < B1; k += blockDim.x) gbase[k] += cnt[k] - off[k];
        // the next sub-tile's zeroing of cnt is behind the barrier at the top of the loop; gbase is only read after the
        // barriers that follow
        __syncthreads();
    }
}

// the two level-1 kernels, instantiated for the window sizes the prover meets (size-based defaults and re-tuned windows of
// 10 .. 22 bits); any other window takes the run-time walk
template <int C>
static void launch_part_level1_c(bool count, const PartShape& sh, uint32_t tiles, size_t lds, hipStream_t st, const Fr* scalars,
                                 const uint8_t* valid, uint32_t* blk_hist, uint32_t* hist1, uint32_t* plan, const uint32_t* start1,
                                 uint32_t* cur1, uint64_t* out, const PlanArgs& pa) {
    if (count) { k_part_count<C><<<tiles, PART_THREADS, lds, st>>>(sh, scalars, valid, blk_hist, hist1, plan, pa); return; }
    if constexpr (C >= STAGE_MIN_C) {
        // latency contexts only: the staging costs 15 M wave-instructions per proof more than it saves in stores (23.9 M against
        // 9.1 M for the h MSM), which is 0.1 ms off a lone proof and 0.7 % ON a proof in the pipeline.  (Tuning builds:
        // CG_PLACE_STAGED=1 / 0 forces either.)
        static const char* force = CG_TUNE_ENV("PLACE_STAGED");
        const bool staged = (force && (force[0] == '0' || force[0] == '1')) ? force[0] == '1' : sh.staged != 0;
        if (staged && sh.precomputed) {
            constexpr int W = (SCALAR_BITS + C - 1) / C;
            const uint32_t B1 = 1u << sh.bits1;
            const size_t bytes = (size_t)(3 * B1 + (B1 & 1u)) * 4 + (size_t)STAGE_SUB * W * 8;
            // (dynamic LDS above 64 KB needs no opt-in on this platform: k_part_place2 has always taken 80 KB)
            k_part_place_staged<C><<<tiles, PART_THREADS, bytes, st>>>(sh, scalars, valid, blk_hist, start1, cur1, out);
            return;
        }
    }
    k_part_place<C><<<tiles, PART_THREADS, lds, st>>>(sh, scalars, valid, blk_hist, start1, cur1, out);
}
static void launch_part_level1(bool count, const PartShape& sh, uint32_t tiles, size_t lds, hipStream_t st, const Fr* scalars,
                               const uint8_t* valid, uint32_t* blk_hist, uint32_t* hist1, uint32_t* plan, const uint32_t* start1,
                               uint32_t* cur1, uint64_t* out, const PlanArgs& pa = PlanArgs{}) {
#define CG_PART_CASE(C) case C: launch_part_level1_c<C>(count, sh, tiles, lds, st, scalars, valid, blk_hist, hist1, plan, start1, cur1, out, pa); break;
    switch (sh.c) {
        CG_PART_CASE(10) CG_PART_CASE(11) CG_PART_CASE(12) CG_PART_CASE(13) CG_PART_CASE(14) CG_PART_CASE(15) CG_PART_CASE(16)
        CG_PART_CASE(17) CG_PART_CASE(18) CG_PART_CASE(19) CG_PART_CASE(20) CG_PART_CASE(21) CG_PART_CASE(22)
        default: launch_part_level1_c<0>(count, sh, tiles, lds, st, scalars, valid, blk_hist, hist1, plan, start1, cur1, out, pa);
    }
#undef CG_PART_CASE
}

// the level-1 bin and the entry range of level-2 chunk `c`
__device__ __forceinline__ bool chunk_range(uint32_t c, int bits1, const uint32_t* __restrict__ chunk0, const uint32_t* __restrict__ start1,
                                            uint32_t& bin, uint32_t& beg, uint32_t& end) {
    const uint32_t B1 = 1u << bits1;
    if (c >= chunk0[B1]) return false;
    uint32_t lo = 0, hi = B1;                         // last bin with chunk0[bin] <= c
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (chunk0[mid] <= c) lo = mid; else hi = mid;
    }
    bin = lo;
    beg = start1[bin] + (c - chunk0[bin]) * PART_CHUNK;
    end = beg + PART_CHUNK < start1[bin + 1] ? beg + PART_CHUNK : start1[bin + 1];
    return true;
}

__global__ void __launch_bounds__(256) k_part_count2(int bits1, int bits2, const uint64_t* __restrict__ in,
                                                     const uint32_t* __restrict__ chunk0, const uint32_t* __restrict__ start1,
                                                     uint32_t* __restrict__ hist2) {
    extern __shared__ uint32_t lds[];
    uint32_t bin, beg, end;
    if (!chunk_range(blockIdx.x, bits1, chunk0, start1, bin, beg, end)) return;
    const uint32_t B2 = 1u << bits2;
    for (uint32_t k = threadIdx.x; k < B2; k += blockDim.x) lds[k] = 0;
    __syncthreads();
    for (uint32_t k0 = beg + threadIdx.x; k0 < end; k0 += 4 * blockDim.x) {      // four loads in flight per lane
        uint64_t e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t k = k0 + u * blockDim.x; e[u] = k < end ? in[k] : 0ull; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u * blockDim.x < end) atomicAdd(&lds[(uint32_t)(e[u] >> 32) & (B2 - 1u)], 1u);
    }
    __syncthreads();
    uint32_t* g = hist2 + ((size_t)bin << bits2);
    for (uint32_t k = threadIdx.x; k < B2; k += blockDim.x) {
        const uint32_t v = lds[k];
        if (v) atomicAdd(&g[k], v);
    }
}

// A chunk is staged in LDS grouped by fine bin and written out run by run: the lanes of a wave store consecutive
// slots of the same bin, instead of 64 slots of 64 bins.
__global__ void __launch_bounds__(256) k_part_place2(int bits1, int bits2, const uint64_t* __restrict__ in,
                                                     const uint32_t* __restrict__ chunk0, const uint32_t* __restrict__ start1,
                                                     const uint32_t* __restrict__ hist2, uint32_t* __restrict__ cur2,
                                                     uint64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t bin, beg, end;
    if (!chunk_range(blockIdx.x, bits1, chunk0, start1, bin, beg, end)) return;
    const uint32_t B2 = 1u << bits2;
    uint32_t* cnt = lds;                 // [B2] entries of this chunk per fine bin, then the fill cursor
    uint32_t* off = lds + B2;            // [B2] first staging slot of the bin
    uint32_t* st = lds + 2 * B2;         // [B2] the bin's count over the whole level-1 bin, then its start inside it
    uint32_t* gb = lds + 3 * B2;         // [B2] global slot of the first entry this chunk contributes to the bin
    uint64_t* stage = reinterpret_cast<uint64_t*>(lds + 4 * B2);      // [PART_CHUNK]
    for (uint32_t k = threadIdx.x; k < B2; k += blockDim.x) { cnt[k] = 0; st[k] = hist2[((size_t)bin << bits2) + k]; }
    __syncthreads();
    constexpr int PER = PART_CHUNK / 256;
    uint64_t e[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) { const uint32_t k = beg + u * 256 + threadIdx.x; e[u] = k < end ? in[k] : ~0ull; }
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (e[u] != ~0ull) atomicAdd(&cnt[(uint32_t)(e[u] >> 32) & (B2 - 1u)], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < B2; k += blockDim.x) off[k] = cnt[k];
    (void)block_exclusive_scan(st, B2);               // starts of the fine bins inside this level-1 bin
    (void)block_exclusive_scan(off, B2);              // starts of the fine bins inside the staging area
    const uint32_t bin_start = start1[bin];
    uint32_t* g = cur2 + ((size_t)bin << bits2);
    for (uint32_t k = threadIdx.x; k < B2; k += blockDim.x) {
        const uint32_t c = cnt[k];
        gb[k] = c ? bin_start + st[k] + atomicAdd(&g[k], c) : 0u;
        cnt[k] = off[k];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (e[u] != ~0ull) stage[atomicAdd(&cnt[(uint32_t)(e[u] >> 32) & (B2 - 1u)], 1u)] = e[u];
    __syncthreads();
    const uint32_t total = end - beg;
    for (uint32_t k = threadIdx.x; k < total; k += blockDim.x) {
        const uint64_t v = stage[k];
        const uint32_t f = (uint32_t)(v >> 32) & (B2 - 1u);
        out[gb[f] + (k - off[f])] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// bucket accumulation over equal segments of the sorted entry list
// ---------------------------------------------------------------------------------------------
template <class F29T>
__device__ __forceinline__ void flush_run(uint32_t key, const XYZZ29<F29T>& acc, bool inf, bool first, bool final_level,
                                          uint32_t t, uint32_t* __restrict__ bucket_sums, uint32_t* __restrict__ part_keys,
                                          uint32_t* __restrict__ part_pts) {
    constexpr int ACC = Words29<F29T>::ACC;
    if (first && !final_level) {
        part_keys[2 * t] = key;
        store_acc(part_pts + (size_t)(2 * t) * ACC, acc, inf);
    } else {
        store_acc(bucket_sums + (size_t)key * ACC, acc, inf);
    }
}

#if defined(CG_ACCUM_WAVES)      // A/B aid: force the occupancy target of the G1 accumulation (default: what 116 VGPRs give, 4)
#define CG_ACCUM_ATTR __attribute__((amdgpu_waves_per_eu(CG_ACCUM_WAVES, CG_ACCUM_WAVES)))
#else
#define CG_ACCUM_ATTR
#endif
template <class F29T>
__global__ void __launch_bounds__(256) CG_ACCUM_ATTR k_accum_affine(const uint64_t* __restrict__ entries, const uint32_t* __restrict__ plan,
                                                      const uint32_t* __restrict__ table,
                                                      uint32_t* __restrict__ bucket_sums, uint32_t* __restrict__ part_keys,
                                                      uint32_t* __restrict__ part_pts) {
    constexpr int ACC = Words29<F29T>::ACC;
    const uint32_t N = plan[PLAN_N], L = plan[PLAN_L], T = plan[PLAN_T];   // written by k_part_plan
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const bool final_level = (T == 1);
    uint32_t beg = t * L;
    uint32_t end = beg + L < N ? beg + L : N;
    XYZZ29<F29T> acc;
    bool inf = true;
    // the next entry is fetched an iteration ahead: the table gather, whose address it holds, can go out as soon as the
    // iteration starts instead of after a first memory round trip (CG_NO_ENTRY_PREFETCH: A/B aid)
#if defined(CG_NO_ENTRY_PREFETCH)
    uint32_t cur = (uint32_t)(entries[beg] >> 32);
#else
    uint64_t next_ent = entries[beg];
    uint32_t cur = (uint32_t)(next_ent >> 32);
#endif
    bool first = true;
    for (uint32_t k = beg; k < end; ++k) {
#if defined(CG_NO_ENTRY_PREFETCH)
        const uint64_t ent = entries[k];
#else
        const uint64_t ent = next_ent;
        if (k + 1 < end) next_ent = entries[k + 1];
#endif
        const uint32_t key = (uint32_t)(ent >> 32), v = (uint32_t)ent;
        if (key != cur) {
            flush_run(cur, acc, inf, first, final_level, t, bucket_sums, part_keys, part_pts);
            first = false;
            inf = true;
            cur = key;
        }
        Affine29<F29T> p = load_table_point_lazy_y(table, v & 0x7fffffffu, (v >> 31) != 0);
        madd29(acc, inf, p);
    }
    if (final_level) {
        store_acc(bucket_sums + (size_t)cur * ACC, acc, inf);
    } else if (first) {  // the whole segment is one run
        part_keys[2 * t] = cur;
        store_acc(part_pts + (size_t)(2 * t) * ACC, acc, inf);
        part_keys[2 * t + 1] = cur;
        store_acc(part_pts + (size_t)(2 * t + 1) * ACC, acc, true);
    } else {
        part_keys[2 * t + 1] = cur;
        store_acc(part_pts + (size_t)(2 * t + 1) * ACC, acc, inf);
    }
}

// ---- G1 on SIGNED limbs (round 5): the kernel the prove path runs ---------------------------------------------------------
// The same walk with the lane's running accumulator in the signed form of curve29.hpp (G1AccS / madd29s): the differences
// of the mixed addition are fused into the products that precede them, the sign of a digit enters as a multiplier instead
// of a negated y, and the sign of Y flips instead of being subtracted - ~140 of the ~2180 instructions of an addition go.
// A flushed run leaves as a SIGNED record (curve29.hpp store_acc_signed: the accumulator as it is, marked); load_acc brings
// it to the stored (unsigned) invariant in the kernels that read it - a flush runs for a lane or two of a wave in 71 % of the
// loop's iterations, so every instruction taken out of it is taken out of the loop.
template <int BLOCK = 256>
__global__ void __launch_bounds__(BLOCK) CG_ACCUM_ATTR k_accum_affine_g1s(const uint64_t* __restrict__ entries, const uint32_t* __restrict__ plan,
                                                          const uint32_t* __restrict__ table,
                                                          uint32_t* __restrict__ bucket_sums, uint32_t* __restrict__ part_keys,
                                                          uint32_t* __restrict__ part_pts) {
    typedef Fq29 F29T;
    constexpr int ACC = Words29<F29T>::ACC;
    const uint32_t N = plan[PLAN_N], L = plan[PLAN_L], T = plan[PLAN_T];
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    // -1 and -2 as values the optimiser cannot see through: "column -= limb" stays one v_mad_i64_i32 (limb, -1, column)
    int32_t neg1 = -1, neg2 = -2;
    asm volatile("" : "+s"(neg1), "+s"(neg2));
    const bool final_level = (T == 1);
    uint32_t beg = t * L;
    uint32_t end = beg + L < N ? beg + L : N;
    G1AccS acc;
    bool inf = true;
    uint64_t next_ent = entries[beg];                  // fetched an iteration ahead (see k_accum_affine)
    uint32_t cur = (uint32_t)(next_ent >> 32);
    bool first = true;
    for (uint32_t k = beg; k < end; ++k) {
        const uint64_t ent = next_ent;
        if (k + 1 < end) next_ent = entries[k + 1];
        const uint32_t key = (uint32_t)(ent >> 32), v = (uint32_t)ent;
        if (key != cur) {
            // the record leaves in the signed form (curve29.hpp store_acc_signed); its readers convert it
            if (first && !final_level) {
                part_keys[2 * t] = cur;
                store_acc_signed(part_pts + (size_t)(2 * t) * ACC, acc, inf);
            } else {
                store_acc_signed(bucket_sums + (size_t)cur * ACC, acc, inf);
            }
            first = false;
            inf = true;
            cur = key;
        }
        const Affine29<F29T> p = load_table_point_plain(table, v & 0x7fffffffu);
        madd29s(acc, inf, p, (int32_t)v >> 31 | 1, neg1, neg2);       // sigma = -1 for a negative digit, +1 otherwise
    }
    if (final_level) {
        store_acc_signed(bucket_sums + (size_t)cur * ACC, acc, inf);
    } else if (first) {  // the whole segment is one run
        part_keys[2 * t] = cur;
        store_acc_signed(part_pts + (size_t)(2 * t) * ACC, acc, inf);
        part_keys[2 * t + 1] = cur;
        store_acc_signed(part_pts + (size_t)(2 * t + 1) * ACC, acc, true);
    } else {
        part_keys[2 * t + 1] = cur;
        store_acc_signed(part_pts + (size_t)(2 * t + 1) * ACC, acc, inf);
    }
}

// ---- the same over Fq2 with the accumulator in LDS -------------------------------------------------------------------
// An Fq2 XYZZ accumulator is 72 VGPRs; held in registers next to the point and the temporaries of a mixed addition
// it pushes the kernel past 400 VGPRs (one wave per SIMD, spills).  Here the running accumulator lives in LDS
// (word-major, [word][lane]: conflict-free) and each coordinate is read where the formula needs it and written back
// when it is final, so the kernel fits two waves per SIMD.
__device__ __forceinline__ Fq2_29 lacc_ld(const uint32_t* sl, int f) {
    Fq2_29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.c0.l[i] = sl[(f * 18 + i) * 256];
        r.c1.l[i] = sl[(f * 18 + 9 + i) * 256];
    }
    return r;
}
__device__ __forceinline__ void lacc_st(uint32_t* sl, int f, const Fq2_29& v) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        sl[(f * 18 + i) * 256] = v.c0.l[i];
        sl[(f * 18 + 9 + i) * 256] = v.c1.l[i];
    }
}
// coordinate `which` (0 = x, 1 = y) of table point idx; y negated when `negate`
__device__ __forceinline__ Fq2_29 load_table_coord2(const uint32_t* __restrict__ table, uint32_t idx, int which, bool negate) {
    const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)idx * 32 + which * 16);
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint4 v = p[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    Fq2_29 c;
    load_coord(c, w);
    if (which == 1) {
        Fq2_29 nc = normalize(sub<2, 1>(Fq2_29::zero(), c));
        if (negate) c = nc;
    }
    return c;
}
// acc (in LDS) += table point   (madd-2008-s in the statement order of curve29.hpp's madd29)
__device__ __forceinline__ void madd29_lds(uint32_t* sl, bool& inf, const uint32_t* __restrict__ table, uint32_t idx, bool negate) {
    typedef Fq2_29 F;
    const F px = load_table_coord2(table, idx, 0, false);
    if (inf) {
        lacc_st(sl, 0, px);
        lacc_st(sl, 1, load_table_coord2(table, idx, 1, negate));
        lacc_st(sl, 2, F::one());
        lacc_st(sl, 3, F::one());
        inf = false;
        return;
    }
    const F zz = lacc_ld(sl, 2), x1 = lacc_ld(sl, 0);
    F P = normalize(sub<KX, 1>(mul(zz, px), x1));
    F PP = sqr_loose(P);
    F ZZ3 = mul(zz, PP);
    if (maybe_zero_mod(ZZ3) && is_zero_mod(ZZ3)) {   // same x: doubling or cancellation (rare)
        const F py = load_table_coord2(table, idx, 1, negate);
        F R0 = normalize(sub<KY, 1>(mul(lacc_ld(sl, 3), py), lacc_ld(sl, 1)));
        if (is_zero_mod(canonical(R0))) {
            XYZZ29<F> d2 = dbl_affine29(Affine29<F>{px, py});
            lacc_st(sl, 0, d2.x); lacc_st(sl, 1, d2.y); lacc_st(sl, 2, d2.zz); lacc_st(sl, 3, d2.zzz);
        } else {
            inf = true;
        }
        return;
    }
    lacc_st(sl, 2, ZZ3);
    F Q = mul(x1, PP);
    F PPP = mul(P, PP);
    const F py = load_table_coord2(table, idx, 1, negate);
    const F zzz = lacc_ld(sl, 3), y1 = lacc_ld(sl, 1);
    F R = normalize(sub<KY, 1>(mul(zzz, py), y1));
    lacc_st(sl, 3, mul(zzz, PPP));
    F X3 = normalize(sub<K2, 2>(sub<K1, 1>(sqr(R), PPP), dbl(Q)));
    lacc_st(sl, 0, X3);
    F d = normalize(sub<KX, 1>(Q, X3));
    lacc_st(sl, 1, mul_sub(d, R, y1, PPP));
}
__device__ __forceinline__ XYZZ29<Fq2_29> lacc_all(const uint32_t* sl) { return {lacc_ld(sl, 0), lacc_ld(sl, 1), lacc_ld(sl, 2), lacc_ld(sl, 3)}; }

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_accum_affine_g2(const uint64_t* __restrict__ entries, const uint32_t* __restrict__ plan, const uint32_t* __restrict__ table,
                  uint32_t* __restrict__ bucket_sums, uint32_t* __restrict__ part_keys, uint32_t* __restrict__ part_pts) {
    typedef Fq2_29 F29T;
    constexpr int ACC = Words29<F29T>::ACC;
    __shared__ uint32_t sm[ACC * 256];
    uint32_t* sl = sm + threadIdx.x;
    const uint32_t N = plan[PLAN_N], L = plan[PLAN_L], T = plan[PLAN_T];
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const bool final_level = (T == 1);
    uint32_t beg = t * L;
    uint32_t end = beg + L < N ? beg + L : N;
    bool inf = true;
    uint64_t next_ent = entries[beg];                  // fetched an iteration ahead, as in k_accum_affine
    uint32_t cur = (uint32_t)(next_ent >> 32);
    bool first = true;
    for (uint32_t k = beg; k < end; ++k) {
        const uint64_t ent = next_ent;
        if (k + 1 < end) next_ent = entries[k + 1];
        const uint32_t key = (uint32_t)(ent >> 32), v = (uint32_t)ent;
        if (key != cur) {
            flush_run(cur, lacc_all(sl), inf, first, final_level, t, bucket_sums, part_keys, part_pts);
            first = false;
            inf = true;
            cur = key;
        }
        madd29_lds(sl, inf, table, v & 0x7fffffffu, (v >> 31) != 0);
    }
    const XYZZ29<F29T> acc = lacc_all(sl);
    if (final_level) {
        store_acc(bucket_sums + (size_t)cur * ACC, acc, inf);
    } else if (first) {  // the whole segment is one run
        part_keys[2 * t] = cur;
        store_acc(part_pts + (size_t)(2 * t) * ACC, acc, inf);
        part_keys[2 * t + 1] = cur;
        store_acc(part_pts + (size_t)(2 * t + 1) * ACC, acc, true);
    } else {
        part_keys[2 * t + 1] = cur;
        store_acc(part_pts + (size_t)(2 * t + 1) * ACC, acc, inf);
    }
}

// ---- the same over Fq2 with every value split over a lane pair (g2pair.hpp): accumulator in registers, no LDS -----------
// Lanes 2p and 2p + 1 take segment p together; they share every branch (same entries, same keys), so the DPP exchanges
// inside pr_madd always find their partner active.
#if defined(CG_G2PAIR_WAVES)     // A/B aid: force the occupancy target
#define CG_G2PAIR_ATTR __attribute__((amdgpu_waves_per_eu(CG_G2PAIR_WAVES, CG_G2PAIR_WAVES)))
#else
#define CG_G2PAIR_ATTR
#endif
__global__ void __launch_bounds__(256) CG_G2PAIR_ATTR
k_accum_affine_g2_pair(const uint64_t* __restrict__ entries, const uint32_t* __restrict__ plan, const uint32_t* __restrict__ table,
                       uint32_t* __restrict__ bucket_sums, uint32_t* __restrict__ part_keys, uint32_t* __restrict__ part_pts) {
    constexpr int ACC = Words29<Fq2_29>::ACC;
    const uint32_t N = plan[PLAN_N], L = plan[PLAN_L], T = plan[PLAN_T];
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = gt >> 1, half = gt & 1u, odd = 0u - half;
    if (t >= T) return;
    const bool final_level = (T == 1);
    const uint32_t beg = t * L;
    const uint32_t end = beg + L < N ? beg + L : N;
    PairAcc acc;
    bool inf = true;
    uint64_t next_ent = entries[beg];                  // fetched an iteration ahead, as in k_accum_affine
    uint32_t cur = (uint32_t)(next_ent >> 32);
    bool first = true;
    for (uint32_t k = beg; k < end; ++k) {
        const uint64_t ent = next_ent;
        if (k + 1 < end) next_ent = entries[k + 1];
        const uint32_t key = (uint32_t)(ent >> 32), v = (uint32_t)ent;
        if (key != cur) {
            if (first && !final_level) {
                part_keys[2 * t] = cur;
                pr_store_acc(part_pts + (size_t)(2 * t) * ACC, acc, inf, half);
            } else {
                pr_store_acc(bucket_sums + (size_t)cur * ACC, acc, inf, half);
            }
            first = false;
            inf = true;
            cur = key;
        }
        pr_madd(acc, inf, table, v & 0x7fffffffu, half, (v >> 31) != 0, odd);
    }
    if (final_level) {
        pr_store_acc(bucket_sums + (size_t)cur * ACC, acc, inf, half);
    } else if (first) {  // the whole segment is one run
        part_keys[2 * t] = cur;
        pr_store_acc(part_pts + (size_t)(2 * t) * ACC, acc, inf, half);
        part_keys[2 * t + 1] = cur;
        pr_store_acc(part_pts + (size_t)(2 * t + 1) * ACC, acc, true, half);
    } else {
        part_keys[2 * t + 1] = cur;
        pr_store_acc(part_pts + (size_t)(2 * t + 1) * ACC, acc, inf, half);
    }
}

// Which G2 accumulation kernel an engine runs.  The lane-pair kernel (g2pair.hpp) is the faster one when it has the GPU to
// itself - 3.15 against 3.35 ms on a full 2^20 G2 MSM, 0.55 against 0.66 ms on the 1.8 M entries of a circom-like
// witness, i.e. 80 % against 68 % of its instruction floor at full occupancy - but it executes ~10 % more instructions
// per addition (the lane exchanges and operand selects), and with a dozen proofs in flight every issue slot the one-lane
// kernel leaves idle is taken by another proof's kernel: all-uniform witnesses prove at 77.2 proofs/s with the pair kernel
// against 79.3 with the one-lane kernel, circom-like ones the same either way (profiles/r03_c_g2_lane_pair.txt).  So:
// pair kernel for latency contexts (a lone proof, a shard), one-lane kernel for throughput contexts.  (Tuning builds:
// CG_G2_PAIR=1 / 0 forces either.)
static bool g2_pair_kernel(bool latency_mode) {
    static const char* e = CG_TUNE_ENV("G2_PAIR");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    return latency_mode;
}
}  // namespace cg
// the batch-affine pair rounds (an experiment that measured 16-21 % slower, profiles/r03_s_batch_affine.txt) are only in
// builds made with -DCG_WITH_BATCH_AFFINE (tools/ab_build.sh); the shipped library does not carry them
#ifdef CG_WITH_BATCH_AFFINE
#include "batchaff.hpp"
#endif
namespace cg {
// threads per workgroup of the wave-per-unit kernels around the accumulation (k_combine_wave, k_bucket_chunks,
// k_bucket_chunk_sums: no barrier, no LDS - any multiple of 64 is legal).  FOUR waves, not one (round 5): a 256-thread
// workgroup lands one wave on each SIMD of a CU, single-wave workgroups are placed unevenly and their long dependent chains
// then sit on a quarter of the SIMDs - +1.8 % on the rate in three of three alternating rounds
// (profiles/r05_ac_tail_workgroups.txt; the same effect the other way round: the accumulation in 64-thread workgroups,
// -3.5 %).  (Tuning builds: CG_TAIL_BLOCK / CG_TAIL_BLOCK_G2 = 64 | 128 | 256.)
template <class F29T>
static uint32_t tail_block() {
    static const uint32_t v = [] {
        const char* e = Words29<F29T>::NF == 2 ? CG_TUNE_ENV("TAIL_BLOCK_G2") : CG_TUNE_ENV("TAIL_BLOCK");
        const int x = e ? atoi(e) : 0;
        return (uint32_t)(x == 64 || x == 128 || x == 256 ? x : 256);
    }();
    return v;
}
// T_max: the largest segment count the plan can hold for this engine (lanes beyond the plan's T return at once)
template <class F29T>
static void launch_accum_affine(const uint64_t* entries, const uint32_t* plan, uint32_t T_max, const uint32_t* table, uint32_t* bucket_sums,
                                uint32_t* part_keys, uint32_t* part_pts, bool latency_mode, hipStream_t st) {
    if constexpr (Words29<F29T>::NF == 2) {
        if (!g2_pair_kernel(latency_mode)) k_accum_affine_g2<<<ceil_div(T_max, 256), 256, 0, st>>>(entries, plan, table, bucket_sums, part_keys, part_pts);
        else k_accum_affine_g2_pair<<<ceil_div(2ull * T_max, 256), 256, 0, st>>>(entries, plan, table, bucket_sums, part_keys, part_pts);
    } else {
        // (tuning builds: CG_ACCUM_UNSIGNED=1 runs round 4's kernel on unsigned limbs - the A/B reference)
        static const bool unsigned_ref = CG_TUNE_ENV("ACCUM_UNSIGNED") != nullptr && CG_TUNE_ENV("ACCUM_UNSIGNED")[0] == '1';
        if (unsigned_ref) k_accum_affine<F29T><<<ceil_div(T_max, 256), 256, 0, st>>>(entries, plan, table, bucket_sums, part_keys, part_pts);
        else {
#ifdef CG_TUNING
            // CG_ACCUM_BLOCK=64 / 128 / 512 / 1024: the kernel has no barrier, so any workgroup size is legal - smaller ones measure
            // -3.5 % in the pipeline, 512 -0.6 %, 1024 -2 % (profiles/r05_ab_accum_workgroup_size.txt: a 256-thread workgroup puts one
            // wave on each SIMD of a CU)
            static const uint32_t blk = [] { const char* e = CG_TUNE_ENV("ACCUM_BLOCK"); const int v = e ? atoi(e) : 0; return (uint32_t)(v == 64 || v == 128 || v == 512 || v == 1024 ? v : 0); }();
            if (blk == 512) k_accum_affine_g1s<512><<<ceil_div(T_max, 512u), 512, 0, st>>>(entries, plan, table, bucket_sums, part_keys, part_pts);
            else if (blk == 1024) k_accum_affine_g1s<1024><<<ceil_div(T_max, 1024u), 1024, 0, st>>>(entries, plan, table, bucket_sums, part_keys, part_pts);
            else if (blk) k_accum_affine_g1s<256><<<ceil_div(T_max, blk), blk, 0, st>>>(entries, plan, table, bucket_sums, part_keys, part_pts);
            else k_accum_affine_g1s<256><<<ceil_div(T_max, 256), 256, 0, st>>>(entries, plan, table, bucket_sums, part_keys, part_pts);
#else
            k_accum_affine_g1s<256><<<ceil_div(T_max, 256), 256, 0, st>>>(entries, plan, table, bucket_sums, part_keys, part_pts);
#endif
        }
    }
}

template <class F29T>
__global__ void __launch_bounds__(256) k_accum_xyzz(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ pts,
                                                    uint32_t N, uint32_t L, uint32_t T, uint32_t* __restrict__ bucket_sums,
                                                    uint32_t* __restrict__ part_keys, uint32_t* __restrict__ part_pts) {
    constexpr int ACC = Words29<F29T>::ACC;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const bool final_level = (T == 1);
    uint32_t beg = t * L;
    uint32_t end = beg + L < N ? beg + L : N;
    XYZZ29<F29T> acc;
    bool inf = true;
    uint32_t cur = keys[beg];
    bool first = true;
    for (uint32_t k = beg; k < end; ++k) {
        uint32_t key = keys[k];
        if (key != cur) {
            flush_run(cur, acc, inf, first, final_level, t, bucket_sums, part_keys, part_pts);
            first = false;
            inf = true;
            cur = key;
        }
        XYZZ29<F29T> q;
        bool qinf = load_acc(pts + (size_t)k * ACC, q);
        add29(acc, inf, q, qinf);
    }
    if (final_level) {
        store_acc(bucket_sums + (size_t)cur * ACC, acc, inf);
    } else if (first) {
        part_keys[2 * t] = cur;
        store_acc(part_pts + (size_t)(2 * t) * ACC, acc, inf);
        part_keys[2 * t + 1] = cur;
        store_acc(part_pts + (size_t)(2 * t + 1) * ACC, acc, true);
    } else {
        part_keys[2 * t + 1] = cur;
        store_acc(part_pts + (size_t)(2 * t + 1) * ACC, acc, inf);
    }
}

// ---------------------------------------------------------------------------------------------
// combining the per-segment partials: one wave per 64 consecutive segments
//
// Every segment of the level below leaves two pieces: F (its first run) and L (its last run); a segment that is a
// single run leaves F = its total and L = empty under the same key.  Laid side by side, the pieces of 64 segments are
// a key-sorted list of 128, and what is wanted is its sum by key: runs that lie inside the wave go to the bucket
// array, the run touching the wave's left edge and the one touching its right edge go up as this wave's own (F, L).
// With uniform keys a run spans two segments, so the whole step is ONE addition per lane (L of the lane to the left
// + own F); where a key covers whole segments (the 0/1 wires of a circom witness all share a bucket) the open run is
// carried across the lanes by a segmented scan, taken only by waves that contain such a segment.  A list of T
// segments is finished in ceil(log64 T) launches of one addition's depth each, where the segment-per-lane scheme
// (k_accum_xyzz, still used at load time) needed log4 T launches of eight.
// ---------------------------------------------------------------------------------------------
static constexpr uint32_t NO_KEY = 0xfffffffeu;   // key of the padding lanes of the last wave; never emitted

template <class F29T>
__device__ __forceinline__ XYZZ29<F29T> shfl_up_acc(const XYZZ29<F29T>& a, unsigned d) {
    constexpr int ACC = Words29<F29T>::ACC;
    uint32_t w[ACC];
    store_limbs(a.x, w);
    store_limbs(a.y, w + ACC / 4);
    store_limbs(a.zz, w + ACC / 2);
    store_limbs(a.zzz, w + 3 * ACC / 4);
#pragma unroll
    for (int i = 0; i < ACC; ++i) w[i] = (uint32_t)__shfl_up((int)w[i], d, 64);
    XYZZ29<F29T> r;
    load_limbs(r.x, w);
    load_limbs(r.y, w + ACC / 4);
    load_limbs(r.zz, w + ACC / 2);
    load_limbs(r.zzz, w + 3 * ACC / 4);
    return r;
}

// One launch per level; level k reads the pieces of level k - 1 and writes its own region of keys_b / pts_b (level k at
// record offset 2·(W_0 + .. + W_{k-1}), W_k = waves of level k).  (Round 4 tried all levels in ONE launch - a wave publishes
// its pair, takes a ticket of its 64-wave group, and the group's last wave goes on as the next level's: correct, and 23 %
// slower in the pipeline together with the same pattern in k_part_count, because every __threadfence() at agent scope is an
// L2 write-back + invalidate on a chip whose eight XCDs have an L2 each (profiles/r04_f_fold_ab.txt).  Cross-workgroup
// hand-offs inside a kernel are not free on MI355X; a kernel boundary does the same flush once.)
template <class F29T>
__global__ void __launch_bounds__(256) k_combine_wave(const uint32_t* __restrict__ keys_a, const uint32_t* __restrict__ pts_a,
                                                       const uint32_t* __restrict__ plan, uint32_t* bucket_sums,
                                                       uint32_t* keys_b, uint32_t* pts_b, int level) {
    constexpr int ACC = Words29<F29T>::ACC;
    uint32_t T = plan[PLAN_T];
    if (T <= 1) return;                            // nothing to combine: the accumulation wrote the buckets itself
    // a wave per 64 segments; a workgroup is 1 or 4 such waves (no barrier, no LDS: tail_block())
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t* in_keys = keys_a;
    const uint32_t* in_pts = pts_a;
    uint32_t out_rec = 0;                          // first record of this level's output region
    for (int k = 0; k < level; ++k) {
        if (T <= 64) return;
        const uint32_t W = (T + 63u) >> 6;
        in_keys = keys_b + out_rec;
        in_pts = pts_b + (size_t)out_rec * ACC;
        out_rec += 2u * W;
        T = W;
    }
    if (wv * 64u >= T) return;
    {
        const uint32_t t = wv * 64u + lane;
        const bool final_level = T <= 64;
        uint32_t* out_keys = keys_b + out_rec;
        uint32_t* out_pts = pts_b + (size_t)out_rec * ACC;
        uint32_t kF = NO_KEY, kL = NO_KEY;
        XYZZ29<F29T> A, B;
        bool Ainf = true, Binf = true;
        if (t < T) {
            kF = in_keys[2 * t];
            kL = in_keys[2 * t + 1];
            Ainf = load_acc(in_pts + (size_t)(2 * t) * ACC, A);
            Binf = load_acc(in_pts + (size_t)(2 * t + 1) * ACC, B);
        }
        const bool single = kF == kL;                 // one run covers the segment: F carries it, L is empty
        // C = the run still open at the right end of this segment
        XYZZ29<F29T> Cv = single ? A : B;
        bool Cinf = single ? Ainf : Binf;
        if (single && !Binf) add29(Cv, Cinf, B, Binf);
        const uint32_t kPrev = (uint32_t)__shfl_up((int)kL, 1, 64);
        const bool joinL = lane > 0 && kPrev == kF;   // the run open at the end of the lane to the left continues here
        bool f = single && joinL;                     // ... and runs on through this whole segment
        if (__ballot(f)) {
            for (unsigned d = 1; d < 64; d <<= 1) {   // segmented inclusive scan of C; f true at step d implies lane >= d
                XYZZ29<F29T> Cu = shfl_up_acc(Cv, d);
                const bool Cuinf = __shfl_up((int)Cinf, d, 64) != 0;
                const bool fu = __shfl_up((int)f, d, 64) != 0;
                if (f) add29(Cv, Cinf, Cu, Cuinf);
                f = f && fu;
            }
        }
        XYZZ29<F29T> Pv = shfl_up_acc(Cv, 1);         // the open run of the lane to the left, after the scan
        const bool Pinf = __shfl_up((int)Cinf, 1, 64) != 0;
        // lanes 0..j all single and chained <=> the run open at the end of lane j started at or before the wave's left edge
        const unsigned long long chain = __ballot(single && (lane == 0 || joinL));
        auto open_left = [&](unsigned j) { return (~chain & (j >= 63 ? ~0ull : ((2ull << j) - 1ull))) == 0ull; };
        auto emit = [&](uint32_t key, const XYZZ29<F29T>& v, bool vinf, bool touches_left) {
            if (key == NO_KEY) return;
            if (touches_left && !final_level) {
                out_keys[2 * wv] = key;
                store_acc(out_pts + (size_t)(2 * wv) * ACC, v, vinf);
            } else {
                store_acc(bucket_sums + (size_t)key * ACC, v, vinf);
            }
        };
        if (lane > 0 && !joinL) emit(kPrev, Pv, Pinf, open_left(lane - 1));      // the left neighbour's open run ends at the boundary
        if (!single) {                                                           // this segment's first run ends inside it
            XYZZ29<F29T> H = A;
            bool Hinf = Ainf, left = lane == 0;
            if (joinL) { add29(H, Hinf, Pv, Pinf); left = open_left(lane - 1); }
            emit(kF, H, Hinf, left);
        }
        if (lane == 63) {                                                        // the run open at the wave's right edge
            if (final_level) {
                emit(kL, Cv, Cinf, false);
            } else if (open_left(63)) {                                          // the whole wave is one run
                emit(kL, Cv, Cinf, true);
                out_keys[2 * wv + 1] = kL;
                store_acc(out_pts + (size_t)(2 * wv + 1) * ACC, Cv, true);
            } else {
                out_keys[2 * wv + 1] = kL;
                store_acc(out_pts + (size_t)(2 * wv + 1) * ACC, Cv, Cinf);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bucket reduction: per window, Σ_b (b+1)·S_b
//
// The buckets of a window are viewed as an R x C matrix (b = r·C + col, C = 2^cb):
//     Σ_b (b+1)·S_b = C · Σ_r r·Row_r  +  Σ_col (col+1)·Col_col,    Row_r = Σ_col S[r,col],  Col_col = Σ_r S[r,col]
// Row and column sums are plain sums - 2 additions per bucket like the textbook running sum, but fully
// parallel - taken by ONE launch (a block per row and a block per column, an LDS tree inside each).  The two
// weighted sums Σ_r r·Row_r and Σ_col (col+1)·Col_col are taken bit by bit: Σ_i w_i·P_i = Σ_k 2^k·(Σ_{i: bit k of w_i} P_i),
// one block per bit in a second launch, and the log-many partial sums go to the host, which folds them with one
// doubling and one addition each while it normalises the result anyway.  Depth: two tree sums; the double-and-add
// per lane and the ~10 fan-in steps per axis of the earlier scheme are gone.
// ---------------------------------------------------------------------------------------------
// Σ over the block of each thread's (acc, inf); the result lands in thread 0's acc / inf.  sm: blockDim.x * ACC words.
template <class F29T>
__device__ __forceinline__ void block_tree_sum(XYZZ29<F29T>& acc, bool& inf, uint32_t* sm) {
    constexpr int ACC = Words29<F29T>::ACC;
    store_acc(sm + (size_t)threadIdx.x * ACC, acc, inf);
    __syncthreads();
    for (uint32_t s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            XYZZ29<F29T> b;
            bool bi = load_acc(sm + (size_t)(threadIdx.x + s) * ACC, b);
            add29(acc, inf, b, bi);
            store_acc(sm + (size_t)threadIdx.x * ACC, acc, inf);
        }
        __syncthreads();
    }
}

// blocks [0, R): Row_r = Σ_col S[r·C + col];  blocks [R, R + C): Col_col = Σ_r S[r·C + col];  blockIdx.y = window
template <class F29T>
__global__ void __launch_bounds__(256) k_bucket_rows_cols(const uint32_t* __restrict__ buckets, uint32_t R, uint32_t C,
                                                          uint32_t* __restrict__ rows, uint32_t* __restrict__ cols) {
    constexpr int ACC = Words29<F29T>::ACC;
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    const uint32_t w = blockIdx.y, b = blockIdx.x;
    const bool is_row = b < R;
    const uint32_t count = is_row ? C : R, base = is_row ? b * C : b - R, stride = is_row ? 1u : C;
    const uint32_t* src = buckets + (size_t)w * R * C * ACC;
    XYZZ29<F29T> acc;
    bool inf = true;
    for (uint32_t k = threadIdx.x; k < count; k += blockDim.x) {
        XYZZ29<F29T> q;
        bool qinf = load_acc(src + ((size_t)base + (size_t)k * stride) * ACC, q);
        add29(acc, inf, q, qinf);
    }
    block_tree_sum(acc, inf, sm);
    if (threadIdx.x == 0) {
        uint32_t* dst = is_row ? rows + ((size_t)w * R + b) * ACC : cols + ((size_t)w * C + (b - R)) * ACC;
        store_acc(dst, acc, inf);
    }
}

// The same sums with no tree at all, for contexts that prove several proofs at a time (instruction count matters, depth
// does not): a LANE adds up a chunk of RED_CHUNK consecutive buckets of one row, or of one column, serially, and a
// second launch adds up the chunks of every row and column the same way.  2·nb + (nb / RED_CHUNK)·2 additions, every
// lane of every wave busy, where the block-per-row scheme spends a third to a half of its additions in half-empty
// tree levels (1.6 - 2.1 wave-additions per 64 buckets against 1.03 here).  Column lanes of a wave read adjacent
// buckets; row lanes read RED_CHUNK buckets apart (the kernel is bound by the 3000-instruction additions, not by HBM).
static constexpr uint32_t RED_CHUNK = 32;
// One WAVE per RED_CHUNK x RED_CHUNK tile of the bucket matrix: lanes 0..31 add up the tile's rows (-> rowp[r][tile column]),
// lanes 32..63 its columns (-> colp[tile row][col]).  Both halves have read the whole tile when the loop ends, and nobody
// else reads it, so the wave then ZEROES the tile (zero_after): the bucket array is left empty for the next MSM and the
// fill launch that opened every accumulation is gone.
template <class F29T>
__global__ void __launch_bounds__(256) k_bucket_chunks(uint32_t* __restrict__ buckets, uint32_t R, uint32_t C,
                                                      uint32_t* __restrict__ rowp, uint32_t* __restrict__ colp, int zero_after) {
    constexpr int ACC = Words29<F29T>::ACC;
    const uint32_t w = blockIdx.y;
    const uint32_t KC = (C + RED_CHUNK - 1) / RED_CHUNK, KR = (R + RED_CHUNK - 1) / RED_CHUNK;
    const uint32_t tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // a wave per tile, 1 or 4 waves per workgroup
    if (tile >= KR * KC) return;
    const uint32_t tr = tile / KC, tc = tile - tr * KC;
    const uint32_t lane = threadIdx.x & 63u, i = lane & 31u;
    const bool row_lane = lane < 32u;
    uint32_t* src = buckets + (size_t)w * R * C * ACC;
    const uint32_t r0 = tr * RED_CHUNK, c0 = tc * RED_CHUNK;
    const uint32_t nr = R - r0 < RED_CHUNK ? R - r0 : RED_CHUNK, nc = C - c0 < RED_CHUNK ? C - c0 : RED_CHUNK;
    XYZZ29<F29T> acc;
    bool inf = true;
    const bool live = row_lane ? i < nr : i < nc;
    const uint32_t steps = row_lane ? nc : nr;
    for (uint32_t s = 0; s < RED_CHUNK; ++s) {
        if (live && s < steps) {
            const uint32_t r = row_lane ? r0 + i : r0 + s, col = row_lane ? c0 + s : c0 + i;
            XYZZ29<F29T> q;
            bool qinf = load_acc(src + ((size_t)r * C + col) * ACC, q);
            add29(acc, inf, q, qinf);
        }
    }
    if (live) {
        if (row_lane) store_acc(rowp + ((size_t)w * R * KC + (size_t)(r0 + i) * KC + tc) * ACC, acc, inf);
        else store_acc(colp + ((size_t)w * KR * C + (size_t)tr * C + (c0 + i)) * ACC, acc, inf);
    }
    if (zero_after) {
        // every load of the tile above has returned (its value fed an addition); rows of the tile are nc * ACC contiguous words
        static_assert(ACC % 4 == 0, "accumulator records are whole uint4s");
        const uint32_t row_vecs = nc * (ACC / 4);
        for (uint32_t r = 0; r < nr; ++r) {
            uint4* dst = reinterpret_cast<uint4*>(src + ((size_t)(r0 + r) * C + c0) * ACC);
            for (uint32_t k = lane; k < row_vecs; k += 64u) dst[k] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}
// Row_r = Σ_k rowp[r][k] (KC chunks), Col_col = Σ_k colp[k][col] (KR chunks): one lane each
template <class F29T>
__global__ void __launch_bounds__(256) k_bucket_chunk_sums(const uint32_t* __restrict__ rowp, const uint32_t* __restrict__ colp, uint32_t R,
                                                          uint32_t C, uint32_t* __restrict__ rows, uint32_t* __restrict__ cols) {
    constexpr int ACC = Words29<F29T>::ACC;
    const uint32_t w = blockIdx.y;
    const uint32_t KC = (C + RED_CHUNK - 1) / RED_CHUNK, KR = (R + RED_CHUNK - 1) / RED_CHUNK;
    const uint32_t n_row = (R + 63u) & ~63u;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ29<F29T> acc;
    bool inf = true;
    if (t < n_row) {
        if (t >= R) return;
        const uint32_t* src = rowp + ((size_t)w * R + t) * KC * ACC;
        for (uint32_t k = 0; k < KC; ++k) {
            XYZZ29<F29T> q;
            bool qinf = load_acc(src + (size_t)k * ACC, q);
            add29(acc, inf, q, qinf);
        }
        store_acc(rows + ((size_t)w * R + t) * ACC, acc, inf);
    } else {
        const uint32_t col = t - n_row;
        if (col >= C) return;
        const uint32_t* src = colp + (size_t)w * KR * C * ACC;
        for (uint32_t k = 0; k < KR; ++k) {
            XYZZ29<F29T> q;
            bool qinf = load_acc(src + ((size_t)k * C + col) * ACC, q);
            add29(acc, inf, q, qinf);
        }
        store_acc(cols + ((size_t)w * C + col) * ACC, acc, inf);
    }
}

// block b < rbits: Σ_{r: bit b of r} Row_r;  block rbits + k: Σ_{col: bit k of (col + 1)} Col_col;  blockIdx.y = window
// `out` and `plan_out` are HOST memory (PinnedBuf::dev): the MSM's last kernel hands its few KB of per-bit sums and the
// plan's statistics straight to the host.
// zero_words != 0 (engines whose MSMs follow each other on one stream): this being the MSM's last kernel, it also leaves the
// partition counters (`plan`, zero_words words: plan | histograms | cursors) zeroed for the next MSM - the fill launch that
// opened every MSM before.
template <class F29T>
__global__ void __launch_bounds__(256) k_bit_sums(const uint32_t* __restrict__ rows, uint32_t R, uint32_t rbits,
                                                  const uint32_t* __restrict__ cols, uint32_t C, uint32_t* __restrict__ out,
                                                  uint32_t* plan, uint32_t* __restrict__ plan_out, uint32_t zero_words) {
    constexpr int ACC = Words29<F29T>::ACC;
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    const uint32_t w = blockIdx.y, b = blockIdx.x;
    if (w == 0 && b == 0 && threadIdx.x < PLAN_WORDS) {
        plan_out[threadIdx.x] = plan[threadIdx.x];
        if (zero_words) plan[threadIdx.x] = 0;
    }
    if (zero_words) {
        const uint32_t nthr = gridDim.x * gridDim.y * blockDim.x, me = (w * gridDim.x + b) * blockDim.x + threadIdx.x;
        for (uint32_t k = PLAN_WORDS + me; k < zero_words; k += nthr) plan[k] = 0;
    }
    const bool is_row = b < rbits;
    const uint32_t bit = is_row ? b : b - rbits, n = is_row ? R : C, offset = is_row ? 0u : 1u;
    const uint32_t* src = is_row ? rows + (size_t)w * R * ACC : cols + (size_t)w * C * ACC;
    XYZZ29<F29T> acc;
    bool inf = true;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        if (((i + offset) >> bit) & 1u) {
            XYZZ29<F29T> q;
            bool qinf = load_acc(src + (size_t)i * ACC, q);
            add29(acc, inf, q, qinf);
        }
    }
    block_tree_sum(acc, inf, sm);
    if (threadIdx.x == 0) store_acc(out + ((size_t)w * gridDim.x + b) * ACC, acc, inf);
}

// ---------------------------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------------------------
static constexpr uint32_t ACC_TARGET_THREADS = 256u * 4u * 4u * 64u;  // CUs x SIMDs x waves x lanes: one fully resident round
// the G2 accumulation runs a lane PAIR per segment at CG_G2PAIR_WAVES waves per SIMD (g2pair.hpp): one resident round
#if !defined(CG_G2PAIR_WAVES)
#define CG_G2PAIR_RESIDENT_WAVES 2u
#else
#define CG_G2PAIR_RESIDENT_WAVES ((uint32_t)(CG_G2PAIR_WAVES))
#endif
static constexpr uint32_t ACC_TARGET_PAIRS = 256u * 4u * CG_G2PAIR_RESIDENT_WAVES * 64u / 2u;
// segments of one fully resident round of the engine's accumulation kernel
template <class F29T> static uint32_t acc_target_segments(bool latency_mode) {
    // CG_ACC_QUARTERS (tuning aid, throughput contexts): lanes of the largest accumulation launch in quarters of a resident
    // round - 3 leaves one wave slot per SIMD to whatever else is in flight, 8 is two rounds
    static const uint32_t quarters = [] { const char* e = CG_TUNE_ENV("ACC_QUARTERS"); const int v = e ? atoi(e) : 4; return (uint32_t)(v >= 1 && v <= 16 ? v : 4); }();
    const uint32_t full = latency_mode ? ACC_TARGET_THREADS : ACC_TARGET_THREADS / 4u * quarters;
    if (Words29<F29T>::NF != 2) return full;
    return g2_pair_kernel(latency_mode) ? ACC_TARGET_PAIRS : full;
}
// shortest segment.  Throughput: 64 - fewer, longer lanes for the small MSMs, whose pieces cost a wave-wide addition
// each to combine (they run beside the h MSM, which fills the chip).  Latency (a shard of one proof): 16.
static constexpr uint32_t ACC_MIN_L = 64, ACC_MIN_L_LATENCY = 16;
static constexpr uint32_t ACC_LEVEL_L = 8;   // segment length of the partial-combining levels

// shape of the bucket matrix of a window: nb = R·C buckets, C = 2^cbits columns (at most 1024)
static int red_cbits(int c) { return (c - 1 + 1) / 2 > 10 ? 10 : (c - 1 + 1) / 2; }

// key bits of an engine's entries and their split over the two partition levels
static void part_bits(int c, int W, bool precomputed, int& bits1, int& bits2) {
    int key_bits = c - 1;
    if (!precomputed) key_bits += ilog2_ceil((uint64_t)W);      // key = window * nb + |d| - 1, with nb a power of two
    if (key_bits < 1) key_bits = 1;
    if (key_bits <= PART_MAX_BITS) { bits1 = key_bits; bits2 = 0; return; }
    bits1 = (key_bits + 1) / 2;
    // tuning builds, CG_PART_BITS1_DELTA=-2..2 (experiment): fewer level-1 bins write longer runs per bin and sub-tile, more of them
    // leave shorter runs at level 2
    if (const char* e = CG_TUNE_ENV("PART_BITS1_DELTA")) {
        const int b = bits1 + atoi(e);
        if (b >= 1 && b <= PART_MAX_BITS && key_bits - b >= 1 && key_bits - b <= PART_MAX_BITS) bits1 = b;
    }
    bits2 = key_bits - bits1;
}

template <class F>
void MsmEngine<F>::init(const MsmBases<F>* b, hipStream_t zero_stream) {
    bases = b;
    const uint64_t n = b->n;
    const int W = b->W;
    cap_entries = n * (uint64_t)W;
    if (cap_entries == 0) cap_entries = 1;
    const uint32_t nb = 1u << (b->c - 1);
    nbuckets_total = b->precomputed ? nb : nb * (uint32_t)W;
    part_bits(b->c, W, b->precomputed, bits1, bits2);
    if (bits1 > PART_MAX_BITS || bits2 > PART_MAX_BITS) throw HipError(CG_ERR_INVALID_ARGUMENT, "bucket key space too large");
    const uint32_t B1 = 1u << bits1;
    const uint64_t tiles = n ? ceil_div(n, PART_TILE) : 1;
    blk_hist.alloc(tiles * B1);
    // zeroed per MSM in one fill: plan | level-1 histogram | level-1 cursors | level-2 histogram | level-2 cursors
    const size_t keyspace = bits2 ? ((size_t)1 << (bits1 + bits2)) : 0;
    counters.alloc(PLAN_WORDS + 2 * (size_t)B1 * PART_PAD + 2 * keyspace);
    starts.alloc(2 * ((size_t)B1 + 1) + 2);                    // start1[B1 + 1] | chunk0[B1 + 1]
    max_chunks = bits2 ? (uint32_t)(cap_entries / PART_CHUNK) + B1 + 1 : 0;
    bucket_sums.alloc(((size_t)nbuckets_total + 1) * ACC);
    min_L = latency_mode ? ACC_MIN_L_LATENCY : ACC_MIN_L;
    if (const char* e = CG_TUNE_ENV("MIN_SEGMENT")) {        // tuning builds only
        const int v = atoi(e);
        if (v >= 1 && v <= 4096) min_L = (uint32_t)v;
    }
    uint64_t t1 = (cap_entries + min_L - 1) / min_L;
    if (t1 > acc_target_segments<F29T>(latency_mode)) t1 = acc_target_segments<F29T>(latency_mode);
    max_segments = (uint32_t)t1;
    // two pieces per segment; then two per wave of every combine level, the levels' regions one after another
    uint64_t pa = 2 * t1, pb = 0;
    for (uint64_t w = ceil_div(t1, 64); ; w = ceil_div(w, 64)) { pb += 2 * w; if (w <= 1) break; }
    mem().reserve(cap_entries, bits2 ? cap_entries : 1, pa, pa * ACC, pb, pb * ACC);
    const uint32_t wins = b->precomputed ? 1u : (uint32_t)W;
    {
        const int cbits = red_cbits(b->c);
        const uint32_t C = 1u << cbits, R = nb / C;
        red_rbits = ilog2_ceil((uint64_t)R);      // weights r < R
        red_cbits1 = cbits + 1;                   // weights col + 1 <= C
        rows_buf.alloc((size_t)R * wins * ACC);
        cols_buf.alloc((size_t)C * wins * ACC);
        if (!latency_mode) {     // chunk sums of the tree-free reduction
            rowp_buf.alloc((size_t)R * ceil_div(C, RED_CHUNK) * wins * ACC);
            colp_buf.alloc((size_t)ceil_div(R, RED_CHUNK) * C * wins * ACC);
        }
    }
#ifdef CG_WITH_BATCH_AFFINE
    ba_rounds = 0;
    if constexpr (Words29<F29T>::NF == 1) {
        if (const char* e = CG_TUNE_ENV("BA_ROUNDS"); e && ba_allowed) ba_rounds = atoi(e) < 0 ? 0 : (atoi(e) > 6 ? 6 : atoi(e));
        if (const char* e = CG_TUNE_ENV("BA_SLOTS")) { const int v = atoi(e); if (v >= 1 && v <= 1024) ba_B = (uint32_t)v; }
    }
    if (ba_rounds) {
        uint64_t ncap = cap_entries, out1 = 0, out2 = 0;
        for (int r = 0; r < ba_rounds; ++r) {
            const uint64_t scap = (ncap + 1) / 2;
            ba_tcap[r] = (uint32_t)((ceil_div(scap, (uint64_t)ba_B) + 63) & ~(uint64_t)63);
            uint64_t ocap = scap + nbuckets_total;
            if (ocap > ncap) ocap = ncap;
            if (r == 0) out1 = ocap;
            if (r == 1) out2 = ocap;
            ncap = ocap;
        }
        const size_t lanes = ba_tcap[0], slots = (size_t)lanes * ba_B;
        ba_prefix.alloc(slots * 9);
        ba_split.alloc(slots / 64 + 1);
        ba_exc.alloc(slots / 64 + 1);
        ba_wpre.alloc(slots / 64 + 1);
        ba_totals.alloc(lanes * 9);
        ba_inv.alloc(lanes * 9);
        ba_chain.alloc((lanes + BA_GROUP) * 9);
        ba_rec_a.alloc((out1 + 1) * BA_REC);
        ba_rec_b.alloc((out2 + 1) * BA_REC);
        ba_plan.alloc((size_t)(ba_rounds + 1) * BAP_WORDS + PLAN_WORDS);
    }
#endif
    // both start out zeroed (load time: a synchronous memset); per MSM they are either re-zeroed by the MSM's own last
    // kernels (zero_at_end) or filled at its start
    // (On a stream of its own and WAITED for: hipMemset on device memory returns before the fill has run, and the legacy
    // default stream it runs on does not order itself against this library's non-blocking streams - the first MSM of a
    // freshly loaded context could start before the fill and have its counters zeroed under it.  Seen as wrong proofs and
    // memory faults in test_contexts_come_and_go_while_others_prove, 10 runs in 12.)
    if (zero_stream) {        // the caller waits for it (once, for all the engines it makes)
        CG_HIP(hipMemsetAsync(counters.p, 0, counters.bytes(), zero_stream));
        CG_HIP(hipMemsetAsync(bucket_sums.p, 0, bucket_sums.bytes(), zero_stream));
    } else {
        hipStream_t zs = nullptr;
        CG_HIP(hipStreamCreateWithFlags(&zs, hipStreamNonBlocking));
        hipError_t e1 = hipMemsetAsync(counters.p, 0, counters.bytes(), zs);
        hipError_t e2 = hipMemsetAsync(bucket_sums.p, 0, bucket_sums.bytes(), zs);
        hipError_t e3 = hipStreamSynchronize(zs);
        (void)hipStreamDestroy(zs);
        CG_HIP(e1); CG_HIP(e2); CG_HIP(e3);
    }
    counters_clean = buckets_clean = true;
    h_plan.alloc(PLAN_WORDS);
    for (int k = 0; k < PLAN_WORDS; ++k) h_plan.p[k] = 0;
    h_result.alloc((size_t)wins * (red_rbits + red_cbits1) * ACC);   // per window: the per-bit sums of rows, then of columns
    if (!ev_t[0])
        for (auto& e : ev_t) CG_HIP(hipEventCreate(&e));
}

template <class F>
void MsmEngine<F>::device_bytes(uint64_t& entries, uint64_t& pieces, uint64_t& other) const {
    entries += own_mem.entry_bytes();
    pieces += own_mem.piece_bytes();
    other += blk_hist.bytes() + counters.bytes() + starts.bytes() + bucket_sums.bytes() + rows_buf.bytes() + cols_buf.bytes() +
             rowp_buf.bytes() + colp_buf.bytes();
#ifdef CG_WITH_BATCH_AFFINE
    other += ba_prefix.bytes() + ba_totals.bytes() + ba_inv.bytes() + ba_chain.bytes() + ba_wpre.bytes() + ba_rec_a.bytes() +
             ba_rec_b.bytes() + ba_plan.bytes() + ba_split.bytes() + ba_exc.bytes();
#endif
}

static float elapsed_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.f;
    return ms;
}
template <class F> float MsmEngine<F>::ms_total() const { return elapsed_ms(ev_t[0], ev_t[5]); }
template <class F> float MsmEngine<F>::ms_sort() const { return n_scalars ? elapsed_ms(ev_t[1], ev_t[2]) : 0.f; }
template <class F> float MsmEngine<F>::ms_accum() const { return n_scalars ? elapsed_ms(ev_t[3], ev_t[4]) : 0.f; }

template <class F>
MsmEngine<F>::~MsmEngine() {
    for (auto& e : ev_t) if (e) (void)hipEventDestroy(e);
}

// phase 1: the entries of `n` canonical scalars, grouped by bucket (see "grouping the digit entries by bucket")
template <class F>
void MsmEngine<F>::digits(const Fr* scalars_dev, uint64_t n, hipStream_t st) {
    if (n > bases->n) n = bases->n;  // msm_bigint zips and truncates to the shorter operand
    n_scalars = n;
    adopted = nullptr;
    for (int k = 0; k < PLAN_WORDS; ++k) h_plan.p[k] = 0;
    CG_HIP(hipEventRecord(ev_t[0], st));
    if (!n) {   // the reduction's last kernel copies the device plan to the host: leave it a zeroed one
        if (!counters_clean) fill_zero(counters.p, (counters.bytes() + 15) & ~(size_t)15, st);
        counters_clean = zero_at_end;     // what accumulate()'s last kernel leaves behind
        return;
    }
    const uint32_t B1 = 1u << bits1;
    uint32_t* plan = counters.p;
    uint32_t* hist1 = plan + PLAN_WORDS;
    uint32_t* cur1 = hist1 + (size_t)B1 * PART_PAD;
    uint32_t* hist2 = cur1 + (size_t)B1 * PART_PAD;
    uint32_t* cur2 = hist2 + (bits2 ? ((size_t)1 << (bits1 + bits2)) : 0);
    uint32_t* start1 = starts.p;
    uint32_t* chunk0 = starts.p + B1 + 1;
    PartShape sh;
    sh.n = (uint32_t)n; sh.c = bases->c; sh.W = bases->W; sh.precomputed = bases->precomputed ? 1 : 0;
    sh.row_stride = (uint32_t)bases->n;          // table row j starts at j * bases->n
    sh.bits1 = bits1; sh.bits2 = bits2;
    sh.staged = latency_mode ? 1 : 0;
    const uint32_t tiles = ceil_div(n, PART_TILE);
    CG_HIP(hipEventRecord(ev_t[1], st));
    // the partition counters are zero when the MSM starts: left so by the last kernel of this engine's previous MSM
    // (one-stream engines, zero_at_end) or filled here
    if (!counters_clean) fill_zero(counters.p, (counters.bytes() + 15) & ~(size_t)15, st);
    counters_clean = false;
    PlanArgs pa;
    pa.start1 = start1; pa.chunk0 = chunk0; pa.target_threads = acc_target_segments<F29T>(latency_mode); pa.min_L = min_L;
    pa.two_level = bits2 ? 1 : 0;
    static const bool plan_launch = CG_TUNE_ENV("PLAN_LAUNCH") != nullptr && CG_TUNE_ENV("PLAN_LAUNCH")[0] == '1';   // tuning builds: the reference path of the A/B
    if (plan_launch) {
        PlanArgs none = pa;
        none.start1 = nullptr;
        launch_part_level1(true, sh, tiles, (size_t)B1 * 4, st, scalars_dev, bases->valid.p, blk_hist.p, hist1, plan, nullptr, nullptr, nullptr, none);
        CG_KERNEL_CHECK();
        k_part_plan<<<1, 1024, (size_t)B1 * 4, st>>>(bits1, pa, hist1, plan);
    } else {
        launch_part_level1(true, sh, tiles, (size_t)B1 * 4, st, scalars_dev, bases->valid.p, blk_hist.p, hist1, plan, nullptr, nullptr, nullptr, pa);
    }
    CG_KERNEL_CHECK();
    launch_part_level1(false, sh, tiles, (size_t)B1 * 4, st, scalars_dev, bases->valid.p, blk_hist.p, nullptr, nullptr, start1, cur1, mem().ent_a.p);
    CG_KERNEL_CHECK();
    if (bits2) {
        const uint32_t B2 = 1u << bits2;
        k_part_count2<<<max_chunks, 256, (size_t)B2 * 4, st>>>(bits1, bits2, mem().ent_a.p, chunk0, start1, hist2);
        CG_KERNEL_CHECK();
        k_part_place2<<<max_chunks, 256, (size_t)B2 * 16 + (size_t)PART_CHUNK * 8, st>>>(bits1, bits2, mem().ent_a.p, chunk0, start1, hist2, cur2, mem().ent_b.p);
        CG_KERNEL_CHECK();
    }
    CG_HIP(hipEventRecord(ev_t[2], st));
}

template <class F>
void MsmEngine<F>::adopt(const uint64_t* grouped_entries, const uint32_t* plan_dev, uint64_t n, hipStream_t st) {
    if (n > bases->n) n = bases->n;
    n_scalars = n;
    for (int k = 0; k < PLAN_WORDS; ++k) h_plan.p[k] = 0;
    CG_HIP(hipEventRecord(ev_t[0], st));
    CG_HIP(hipEventRecord(ev_t[1], st));
    // the plan (entry count, segment geometry, statistics) travels with the entries; everything downstream reads it from
    // this engine's own counters as usual
    k_replan<<<1, 1, 0, st>>>(counters.p, plan_dev, acc_target_segments<F29T>(latency_mode), min_L);
    counters_clean = false;
    CG_KERNEL_CHECK();
    CG_HIP(hipEventRecord(ev_t[2], st));
    adopted = n ? grouped_entries : nullptr;
}

// phase 2: accumulate the grouped entries into the buckets, combine the segments, reduce the buckets; the per-bit sums
// and the plan (entry count, statistics) are copied to pinned memory.  Nothing here waits for the host.
template <class F>
void MsmEngine<F>::accumulate(hipStream_t st) {
    // the bucket array is empty when the accumulation starts: left so by the reduction of this engine's previous MSM
    // (k_bucket_chunks, zero_at_end) or filled here
    if (!buckets_clean) fill_zero(bucket_sums.p, bucket_sums.bytes(), st);
    buckets_clean = false;
    if (n_scalars) {
        const uint32_t* plan = counters.p;
        const uint64_t* grouped = adopted ? adopted : this->grouped();
        CG_HIP(hipEventRecord(ev_t[3], st));
#ifdef CG_WITH_BATCH_AFFINE
        bool ba_done = false;
        if constexpr (Words29<F29T>::NF == 1) {
            if (ba_rounds) {
                // R pair rounds (batchaff.hpp), then the XYZZ accumulation over what is left under its own plan
                uint32_t* plan2 = ba_plan.p + (size_t)(ba_rounds + 1) * BAP_WORDS;
                k_ba_begin<<<1, 1, 0, st>>>(plan, ba_plan.p, ba_B);
                CG_KERNEL_CHECK();
                const uint32_t* recs_in = nullptr;
                for (int r = 0; r < ba_rounds; ++r) {
                    const uint32_t* bp = ba_plan.p + (size_t)r * BAP_WORDS;
                    const bool last = r + 1 == ba_rounds;
                    uint32_t* out = (r & 1) ? ba_rec_b.p : ba_rec_a.p;
                    const uint32_t grid = ceil_div(ba_tcap[r], 256u);
                    if (r == 0) k_ba_forward<0><<<grid, 256, 0, st>>>(grouped, bases->table.p, nullptr, bp, ba_B, ba_prefix.p, ba_totals.p, ba_split.p, ba_exc.p);
                    else k_ba_forward<1><<<grid, 256, 0, st>>>(nullptr, nullptr, recs_in, bp, ba_B, ba_prefix.p, ba_totals.p, ba_split.p, ba_exc.p);
                    CG_KERNEL_CHECK();
                    k_ba_scan<<<1, 1024, 0, st>>>(ba_split.p, ba_wpre.p, bp, last ? nullptr : ba_plan.p + (size_t)(r + 1) * BAP_WORDS, ba_B,
                                                  last ? plan2 : nullptr, acc_target_segments<F29T>(latency_mode), min_L);
                    CG_KERNEL_CHECK();
                    k_ba_invert<<<ceil_div(ceil_div(ba_tcap[r], BA_GROUP), 64u), 64, 0, st>>>(ba_totals.p, bp, ba_chain.p, ba_inv.p);
                    CG_KERNEL_CHECK();
                    if (r == 0) k_ba_backward<0><<<grid, 256, 0, st>>>(grouped, bases->table.p, nullptr, bp, ba_B, ba_prefix.p, ba_inv.p, ba_split.p, ba_exc.p, ba_wpre.p, out);
                    else k_ba_backward<1><<<grid, 256, 0, st>>>(nullptr, nullptr, recs_in, bp, ba_B, ba_prefix.p, ba_inv.p, ba_split.p, ba_exc.p, ba_wpre.p, out);
                    CG_KERNEL_CHECK();
                    recs_in = out;
                }
                k_accum_records<<<ceil_div(max_segments, 256u), 256, 0, st>>>(recs_in, plan2, bucket_sums.p, mem().part_keys_a.p, mem().part_pts_a.p);
                CG_KERNEL_CHECK();
                plan = plan2;
                ba_done = true;
            }
        }
        if (!ba_done)
#endif
        // tuning builds: KNOCK_ACCUM leaves the accumulation AND the combine levels out (the pieces would be stale),
        // KNOCK_COMBINE the combine levels, KNOCK_TAIL the bucket reduction - wrong sums, the time of the rest
        static const bool knock_accum = CG_TUNE_ENV("KNOCK_ACCUM") != nullptr, knock_combine = CG_TUNE_ENV("KNOCK_COMBINE") != nullptr;
        if (!knock_accum)
        launch_accum_affine<F29T>(grouped, plan, max_segments, bases->table.p, bucket_sums.p, mem().part_keys_a.p, mem().part_pts_a.p, latency_mode, st);
        CG_KERNEL_CHECK();
        CG_HIP(hipEventRecord(ev_t[4], st));
        // combine the segments' pieces wave by wave until one wave covers them all (k_combine_wave); the grids cover the
        // largest plan this engine can see, waves beyond the actual one return at once
        if (max_segments > 1 && !knock_accum && !knock_combine) {
            MsmScratch& S = mem();
            uint32_t segs = max_segments;
            for (int level = 0; segs > 1; ++level) {
                const uint32_t waves = ceil_div(segs, 64u);
                k_combine_wave<F29T><<<ceil_div(waves, tail_block<F29T>() / 64u), tail_block<F29T>(), 0, st>>>(S.part_keys_a.p, S.part_pts_a.p, plan, bucket_sums.p, S.part_keys_b.p, S.part_pts_b.p, level);
                CG_KERNEL_CHECK();
                segs = waves;
            }
        }
    }
    // bucket reduction (see the comment above block_tree_sum): two launches, the second writing the per-bit sums to host memory, of a fixed
    // shape for a given window size - captured once into a HIP graph and replayed as one submission.
    // The reduction's two or three launches go out directly.  Rounds 1-2 replayed them as a captured HIP graph, which is
    // worth ~0.03 ms on a lone proof and nothing on the rate (and costs the host twice the CPU: 1.4 against 0.7 CPUs busy
    // with sixteen proofs in flight) - and a capture is invalidated by anything that synchronises the device from ANOTHER
    // thread while it is open (a context being freed or loaded next to a context's first proof: "operation failed due
    // to a previous error during capture", met by tests/test_gpu_e2e_files.py under concurrent callers).  No capture, no
    // such window (profiles/r03_j_reduction_graph.txt).
    enqueue_reduction(st);
    CG_HIP(hipEventRecord(ev_t[5], st));
}

template <class F>
void MsmEngine<F>::enqueue_reduction(hipStream_t st) {
    const uint32_t wins = bases->precomputed ? 1u : (uint32_t)bases->W;
    const uint32_t nb = 1u << (bases->c - 1);
    const uint32_t C = 1u << red_cbits(bases->c), R = nb / C;
    const size_t lds = (size_t)256 * ACC * 4;
    static const bool force_tree = CG_TUNE_ENV("RED_TREE") != nullptr;      // A/B aid (tuning builds)
    static const bool knock_tail = CG_TUNE_ENV("KNOCK_TAIL") != nullptr;
    if (knock_tail) return;             // buckets_clean / counters_clean stay false: the next MSM fills them
    if (latency_mode || force_tree) {   // a block per row / column with an LDS tree: depth log, more additions
        k_bucket_rows_cols<F29T><<<dim3(R + C, wins), 256, lds, st>>>(bucket_sums.p, R, C, rows_buf.p, cols_buf.p);
        CG_KERNEL_CHECK();
    } else {                     // a lane per chunk of 32 buckets, then a lane per row / column: fewest additions
        const uint32_t KC = ceil_div(C, RED_CHUNK), KR = ceil_div(R, RED_CHUNK);
        const uint32_t lanes_b = ((R + 63u) & ~63u) + C;
        const uint32_t tb = tail_block<F29T>();
        k_bucket_chunks<F29T><<<dim3(ceil_div(KR * KC, tb / 64u), wins), tb, 0, st>>>(bucket_sums.p, R, C, rowp_buf.p, colp_buf.p, zero_at_end ? 1 : 0);
        CG_KERNEL_CHECK();
        buckets_clean = zero_at_end;
        k_bucket_chunk_sums<F29T><<<dim3(ceil_div(lanes_b, tb), wins), tb, 0, st>>>(rowp_buf.p, colp_buf.p, R, C, rows_buf.p, cols_buf.p);
        CG_KERNEL_CHECK();
    }
    const uint32_t nbits = (uint32_t)(red_rbits + red_cbits1);
    k_bit_sums<F29T><<<dim3(nbits, wins), 256, lds, st>>>(rows_buf.p, R, (uint32_t)red_rbits, cols_buf.p, C, h_result.dev(), counters.p,
                                                          h_plan.dev(), zero_at_end ? (uint32_t)counters.n : 0u);
    CG_KERNEL_CHECK();
    counters_clean = zero_at_end;
}

// ---- host: lazy 29-bit accumulator -> saturated Montgomery(2^256) XYZZ ---------------------------------
// value = Σ l_i 2^(29 i) < 2^261 is x·2^261 mod q up to a multiple of q; reduce, then x·2^256 = value / 32.
static Fq fq_from_limbs29(const uint32_t l[9]) {
    typedef unsigned __int128 u128;
    uint64_t v[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 9; ++i) {
        int bit = 29 * i, w = bit >> 6, sh = bit & 63;
        u128 x = (u128)l[i] << sh;
        u128 s = (u128)v[w] + (uint64_t)x;
        v[w] = (uint64_t)s;
        u128 carry = (s >> 64) + (x >> 64);
        for (int k = w + 1; k < 5 && carry; ++k) {
            u128 t = (u128)v[k] + (uint64_t)carry;
            v[k] = (uint64_t)t;
            carry = t >> 64;
        }
    }
    uint64_t n[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) n[i] = (uint64_t)FqP::N[2 * i] | ((uint64_t)FqP::N[2 * i + 1] << 32);
    auto ge = [&]() {
        for (int i = 4; i >= 0; --i) {
            if (v[i] > n[i]) return true;
            if (v[i] < n[i]) return false;
        }
        return true;
    };
    while (ge()) {
        u128 br = 0;
        for (int i = 0; i < 5; ++i) {
            u128 d = (u128)v[i] - n[i] - (uint64_t)br;
            v[i] = (uint64_t)d;
            br = (d >> 64) & 1;
        }
    }
    Fq a;
    for (int i = 0; i < 4; ++i) { a.l[2 * i] = (uint32_t)v[i]; a.l[2 * i + 1] = (uint32_t)(v[i] >> 32); }
    Fq c2 = Fq::zero();
    c2.l[7] = 0x08000000u;   // 2^251 = (2^256)^2 / 2^261
    return mul(a, c2);       // a · 2^251 / 2^256 = a / 32
}
static void coord_from_limbs29(Fq& out, const uint32_t* w) { out = fq_from_limbs29(w); }
static void coord_from_limbs29(Fq2& out, const uint32_t* w) { out.c0 = fq_from_limbs29(w); out.c1 = fq_from_limbs29(w + 9); }

template <class F>
static XYZZ<F> xyzz_from_words(const uint32_t* w, int acc_words) {
    const int q = acc_words / 4;
    bool inf = true;
    for (int i = 0; i < q; ++i) if (w[2 * q + i]) inf = false;
    if (inf) return XYZZ<F>::inf();
    XYZZ<F> r;
    coord_from_limbs29(r.x, w);
    coord_from_limbs29(r.y, w + q);
    coord_from_limbs29(r.zz, w + 2 * q);
    coord_from_limbs29(r.zzz, w + 3 * q);
    return r;
}

// Σ_k 2^k·B_k over the per-bit sums B_0 .. B_{n-1} (Horner from the top bit)
template <class F>
static XYZZ<F> fold_bits(const uint32_t* bit_sums, int n, int acc_words) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int k = n - 1; k >= 0; --k) {
        acc = dbl(acc);
        add(acc, xyzz_from_words<F>(bit_sums + (size_t)k * acc_words, acc_words));
    }
    return acc;
}
// one window's sum: 2^cbits · Σ_r r·Row_r + Σ_col (col+1)·Col_col from the per-bit sums of rows and columns
template <class F>
static XYZZ<F> window_value(const uint32_t* bit_sums, int acc_words, int rbits, int cbits1) {
    XYZZ<F> rows = fold_bits<F>(bit_sums, rbits, acc_words);
    XYZZ<F> cols = fold_bits<F>(bit_sums + (size_t)rbits * acc_words, cbits1, acc_words);
    for (int k = 0; k < cbits1 - 1; ++k) rows = dbl(rows);
    add(rows, cols);
    return rows;
}

template <class F>
XYZZ<F> MsmEngine<F>::value() const {
    const int nbits = red_rbits + red_cbits1;
    if (bases->precomputed) return window_value<F>(h_result.p, ACC, red_rbits, red_cbits1);
    // Horner over the windows: Σ_j 2^(c j) S_j
    const int W = bases->W, c = bases->c;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int j = W - 1; j >= 0; --j) {
        for (int k = 0; k < c; ++k) acc = dbl(acc);
        add(acc, window_value<F>(h_result.p + (size_t)j * nbits * ACC, ACC, red_rbits, red_cbits1));
    }
    return acc;
}

// Σ of XYZZ accumulators per key (keys sorted ascending), added into sums[key]: the partial-combining levels of the
// bucket accumulation (k_accum_xyzz) run on their own
void sum_xyzz_by_key(const uint32_t* keys, const uint32_t* pts, uint64_t count, uint32_t* sums, hipStream_t st) {
    typedef Fq29 F29T;
    constexpr int ACC = Words29<F29T>::ACC;
    if (!count) return;
    const uint64_t pa = 2 * ceil_div(count, ACC_LEVEL_L), pb = 2 * ceil_div(pa, ACC_LEVEL_L);
    DevBuf<uint32_t> ka(pa), kb(pb), qa(pa * ACC), qb(pb * ACC);
    const uint32_t* ik = keys;
    const uint32_t* ip = pts;
    bool to_a = true;
    uint32_t cnt = (uint32_t)count;
    while (cnt) {
        const uint32_t Tk = ceil_div(cnt, ACC_LEVEL_L);
        uint32_t* ok = to_a ? ka.p : kb.p;
        uint32_t* op = to_a ? qa.p : qb.p;
        k_accum_xyzz<F29T><<<ceil_div(Tk, 256), 256, 0, st>>>(ik, ip, cnt, ACC_LEVEL_L, Tk, sums, ok, op);
        CG_KERNEL_CHECK();
        cnt = (Tk == 1) ? 0 : 2 * Tk;
        ik = ok;
        ip = op;
        to_a = !to_a;
    }
    CG_HIP(hipStreamSynchronize(st));
}

// packed table points and validity flags first, first + stride, first + 2·stride, ... -> contiguous
__global__ void __launch_bounds__(256) k_gather_points(const uint32_t* __restrict__ row0, const uint8_t* __restrict__ valid, uint64_t first,
                                                       uint64_t stride, uint64_t count, uint32_t* __restrict__ out, uint8_t* __restrict__ out_valid) {
    constexpr int AFF = MsmBases<Fq>::AFF;
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const uint64_t j = first + k * stride;
    const uint4* src = reinterpret_cast<const uint4*>(row0 + j * AFF);
    uint4* dst = reinterpret_cast<uint4*>(out + k * AFF);
#pragma unroll
    for (int i = 0; i < AFF / 4; ++i) dst[i] = src[i];
    out_valid[k] = valid[j];
}

static float ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// h query -> coset evaluation basis (ecntt.hip), then the usual window rows over the slice h_first + k·h_stride, k < h_count
void build_h_bases_folded(MsmBases<Fq>& out_h, const uint32_t* h_row0, const uint8_t* h_valid, uint64_t n_h, int logn, uint64_t h_first,
                          uint64_t h_stride, uint64_t h_count, int c_h, hipStream_t st, float* ms_fold, float* ms_tables) {
    constexpr int AFF = MsmBases<Fq>::AFF;
    const uint64_t n = 1ull << logn;
    if (!h_stride || (h_count && h_first + (h_count - 1) * h_stride >= n)) throw HipError(CG_ERR_INVALID_ARGUMENT, "range outside the query");
    auto t0 = std::chrono::steady_clock::now();
    DevBuf<uint32_t> row0t(n * AFF);
    DevBuf<uint8_t> validt(n);
    ec_transform_h_bases(h_row0, h_valid, n_h, logn, row0t.p, validt.p, st);       // synchronises st
    if (ms_fold) *ms_fold += ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    if (h_stride == 1) {
        out_h.build_from_row0(row0t.p + h_first * AFF, validt.p + h_first, h_count, c_h, st);
    } else {
        DevBuf<uint32_t> row0s(h_count ? h_count * AFF : 4);
        DevBuf<uint8_t> valids(h_count ? h_count : 1);
        if (h_count) k_gather_points<<<ceil_div(h_count, 256), 256, 0, st>>>(row0t.p, validt.p, h_first, h_stride, h_count, row0s.p, valids.p);
        CG_KERNEL_CHECK();
        out_h.build_from_row0(row0s.p, valids.p, h_count, c_h, st);
        CG_HIP(hipStreamSynchronize(st));   // the gathered copies are released at the end of this scope
    }
    CG_HIP(hipStreamSynchronize(st));
    if (ms_tables) *ms_tables += ms_since(t0);
}

// C matrix -> l query (ecntt.hip), then the window rows over [l_first, l_first + l_count) of the M folded bases
void build_l_bases_folded(MsmBases<Fq>& out_l, const uint32_t* h_row0, const uint8_t* h_valid, uint64_t n_h, int logn,
                          const uint32_t* l_row0, const uint8_t* l_valid, uint64_t num_inputs, uint64_t M, const HostCsc& c_transposed,
                          uint64_t num_constraints, const Fr& vanishing_inv, uint64_t l_first, uint64_t l_count,
                          const std::function<int()>& pick_c_l, hipStream_t st, float* ms_fold, float* ms_tables) {
    constexpr int AFF = MsmBases<Fq>::AFF;
    if (l_first + l_count > M) throw HipError(CG_ERR_INVALID_ARGUMENT, "range outside the query");
    auto t0 = std::chrono::steady_clock::now();
    DevBuf<uint32_t> row0f(M * AFF);
    DevBuf<uint8_t> validf(M);
    ec_fold_ct_into_l(h_row0, h_valid, n_h, logn, vanishing_inv, c_transposed, num_constraints, num_inputs, M, l_row0, l_valid,
                      row0f.p, validf.p, st);                                                // synchronises st
    if (ms_fold) *ms_fold += ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    out_l.build_from_row0(row0f.p + l_first * AFF, validf.p + l_first, l_count, pick_c_l(), st);
    CG_HIP(hipStreamSynchronize(st));
    if (ms_tables) *ms_tables += ms_since(t0);
}

// both, from the key's affine points (the synchronous load)
void build_hl_bases_folded(MsmBases<Fq>& out_h, MsmBases<Fq>& out_l, const Affine<Fq>* h_bases_dev, uint64_t n_h, int logn,
                           const Affine<Fq>* l_bases_dev, uint64_t num_inputs, uint64_t M, const cg_csr& c_matrix,
                           uint64_t num_constraints, const Fr& vanishing_inv, uint64_t h_first, uint64_t h_stride, uint64_t h_count, int c_h,
                           uint64_t l_first, uint64_t l_count, int c_l, hipStream_t st, float* ms_fold, float* ms_tables) {
    constexpr int AFF = MsmBases<Fq>::AFF;
    const uint64_t n_l = M - num_inputs;
    DevBuf<uint32_t> h_row0(n_h ? n_h * AFF : 4), l_row0(n_l ? n_l * AFF : 4);
    DevBuf<uint8_t> h_valid(n_h ? n_h : 1), l_valid(n_l ? n_l : 1);
    if (n_h) k_table_first<Fq><<<ceil_div(n_h, 256), 256, 0, st>>>(h_bases_dev, h_row0.p, h_valid.p, n_h);
    if (n_l) k_table_first<Fq><<<ceil_div(n_l, 256), 256, 0, st>>>(l_bases_dev, l_row0.p, l_valid.p, n_l);
    CG_KERNEL_CHECK();
    HostCsc t;
    if (c_matrix.nnz && num_constraints) csr_transpose(c_matrix, num_constraints, M, t);
    else { t.ptr.assign(M + 1, 0); t.view = cg_csr{t.ptr.data(), nullptr, nullptr, 0}; }
    build_h_bases_folded(out_h, h_row0.p, h_valid.p, n_h, logn, h_first, h_stride, h_count, c_h, st, ms_fold, ms_tables);
    build_l_bases_folded(out_l, h_row0.p, h_valid.p, n_h, logn, l_row0.p, l_valid.p, num_inputs, M, t, num_constraints, vanishing_inv,
                         l_first, l_count, [c_l] { return c_l; }, st, ms_fold, ms_tables);
}

template struct MsmBases<Fq>;
template struct MsmBases<Fq2>;
template struct MsmEngine<Fq>;
template struct MsmEngine<Fq2>;

}  // namespace cg
