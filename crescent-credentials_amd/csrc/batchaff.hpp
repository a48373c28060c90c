// Batch-affine pair rounds in front of the G1 bucket accumulation (included by msm.hip).
//
// The grouped entry list (sorted by bucket) is halved R times before the XYZZ accumulation sees it: in every round the
// elements at positions 2g and 2g + 1 are added when they lie in the same bucket, as AFFINE points,
//     λ = (y2 − y1)/(x2 − x1),  x3 = λ² − x1 − x2,  y3 = λ·(x1 − x3) − y1,
// with the inversions shared by Montgomery's trick over ALL the additions of the round (a lane chains `B` slots, the
// lanes' totals are chained again, `G` to a lane, and only those few products are inverted).  An addition then costs
// 5 products + 1 square against the 9 + 2 of the XYZZ mixed addition; what it costs instead is memory traffic - the
// operands are read twice (forward pass: x only; backward pass: everything), the prefix products are written and read
// back, the sums are written - all of it in lane order (slot g of a round belongs to lane g mod T), so apart from the
// first round's table gathers every access is coalesced.
//
// Lists.  Round 0 reads the engine's 64-bit entries and the window table; every round writes RECORDS of 20 words
// (x as nine 29-bit limbs, the bucket key, y as nine limbs, one pad word: 80 B, 16-byte aligned), position = order, so
// buckets stay contiguous runs.  A slot whose two elements lie in different buckets ("split") passes both on; the
// output position of slot g is g + (number of split slots before g), from a popcount scan of the wave ballots.
// Values: record coordinates are normalised and below 3N (weak_reduce); the identity (P + (−P)) is a record with
// x.l[8] = BA_IDENT.  Slots that need anything but the generic addition - an identity operand, the odd element at the
// end of the list, equal x (doubling or cancellation) - are flagged by the forward pass in a second ballot mask, use
// the denominator 1 (2y for a doubling), and are finished on a slow path by the backward pass.
#pragma once
#include "curve29.hpp"

namespace cg {

static constexpr int BA_REC = 20;                   // words per record
static constexpr uint32_t BA_IDENT = 0xffffffffu;   // x.l[8] of the identity record
static constexpr uint32_t BA_GROUP = 32;            // lane totals per lane of the second-level chain
enum { BAP_N = 0, BAP_S = 1, BAP_T = 2, BAP_WORDS = 4 };   // per round: elements, slots, lanes (a multiple of 64)

// v (normalised, value < 2^261) -> the same residue, normalised, below 3N (cf. weak_reduce in wmap29.hip)
__device__ __forceinline__ Fq29 ba_weak_reduce(const Fq29& v) {
    constexpr uint32_t MU = 88753990u;   // floor(2^280 / q)
    const uint32_t qh = (uint32_t)(((uint64_t)v.l[8] * MU) >> 48);
    const int32_t nq = -(int32_t)qh;
    Fq29 r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        c += (int64_t)v.l[i];
        c += (int64_t)nq * (int64_t)(int32_t)Fq29P::N[i];
        if (i < 8) { r.l[i] = (uint32_t)c & M29; c >>= 29; }
        else r.l[i] = (uint32_t)c;
    }
    return r;
}

struct BaElem {          // one list element as the passes see it
    Fq29 x, y;
    uint32_t key;
    bool ident;
};

// SRC 0: element i = entry i of the grouped list (bucket key | table index, sign); SRC 1: record i
template <int SRC, bool WITH_Y>
__device__ __forceinline__ BaElem ba_load(const uint64_t* __restrict__ entries, const uint32_t* __restrict__ table,
                                          const uint32_t* __restrict__ recs, uint32_t i) {
    BaElem e;
    e.ident = false;
    if (SRC == 0) {
        const uint64_t ent = entries[i];
        e.key = (uint32_t)(ent >> 32);
        const uint32_t v = (uint32_t)ent;
        const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)(v & 0x7fffffffu) * 16);
        uint32_t w[16];
        const uint4 a = p[0], b = p[1];
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        e.x = unpack29<Fq29P>(w);
        if (WITH_Y) {
            const uint4 c = p[2], d = p[3];
            w[8] = c.x; w[9] = c.y; w[10] = c.z; w[11] = c.w; w[12] = d.x; w[13] = d.y; w[14] = d.z; w[15] = d.w;
            e.y = unpack29<Fq29P>(w + 8);
            const Fq29 ny = normalize(sub<2, 1>(Fq29::zero(), e.y));
            if (v >> 31) e.y = ny;
        }
    } else {
        const uint4* p = reinterpret_cast<const uint4*>(recs + (size_t)i * BA_REC);
        const uint4 a = p[0], b = p[1], c = p[2];
        e.x.l[0] = a.x; e.x.l[1] = a.y; e.x.l[2] = a.z; e.x.l[3] = a.w;
        e.x.l[4] = b.x; e.x.l[5] = b.y; e.x.l[6] = b.z; e.x.l[7] = b.w;
        e.x.l[8] = c.x;
        e.key = c.y;
        e.ident = (c.x == BA_IDENT);
        if (WITH_Y) {
            const uint4 d = p[3], f = p[4];
            e.y.l[0] = c.z; e.y.l[1] = c.w;
            e.y.l[2] = d.x; e.y.l[3] = d.y; e.y.l[4] = d.z; e.y.l[5] = d.w;
            e.y.l[6] = f.x; e.y.l[7] = f.y; e.y.l[8] = f.z;
        }
    }
    return e;
}
__device__ __forceinline__ void ba_store(uint32_t* __restrict__ recs, uint32_t pos, const Fq29& x, const Fq29& y, uint32_t key, bool ident) {
    uint4* p = reinterpret_cast<uint4*>(recs + (size_t)pos * BA_REC);
    p[0] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
    p[1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    p[2] = make_uint4(ident ? BA_IDENT : x.l[8], key, y.l[0], y.l[1]);
    p[3] = make_uint4(y.l[2], y.l[3], y.l[4], y.l[5]);
    p[4] = make_uint4(y.l[6], y.l[7], y.l[8], 0u);
}
__device__ __forceinline__ Fq29 ba_load9(const uint32_t* __restrict__ p) {
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = p[i];
    return r;
}
__device__ __forceinline__ void ba_store9(uint32_t* __restrict__ p, const Fq29& v) {
#pragma unroll
    for (int i = 0; i < 9; ++i) p[i] = v.l[i];
}
// y1 ≡ y2 (mod N)?  both normalised, below 3N
__device__ __forceinline__ bool ba_same_y(const Fq29& y1, const Fq29& y2) {
    return is_zero_mod(canonical(normalize(sub<3, 1>(y2, y1))));
}

// round 0's geometry from the engine's plan
__global__ void k_ba_begin(const uint32_t* __restrict__ plan, uint32_t* __restrict__ bp, uint32_t B) {
    const uint32_t N = plan[0];     // PLAN_N
    const uint32_t S = (N + 1) / 2;
    bp[BAP_N] = N;
    bp[BAP_S] = S;
    bp[BAP_T] = ((S + B - 1) / B + 63u) & ~63u;
}

// forward pass: the running product of the slots' denominators down every lane, every prefix kept
template <int SRC>
__global__ void __launch_bounds__(256) k_ba_forward(const uint64_t* __restrict__ entries, const uint32_t* __restrict__ table,
                                                    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ bp, uint32_t B,
                                                    uint32_t* __restrict__ prefix, uint32_t* __restrict__ totals,
                                                    uint64_t* __restrict__ split_mask, uint64_t* __restrict__ exc_mask) {
    const uint32_t N = bp[BAP_N], S = bp[BAP_S], T = bp[BAP_T];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;                         // T is a multiple of 64: whole waves leave
    Fq29 run = Fq29::one();
    for (uint32_t j = 0; j < B; ++j) {
        const uint32_t g = j * T + t;
        bool split = false, exc = false;
        Fq29 dx = Fq29::one();
        BaElem e0, e1;
        const bool active = g < S;
        if (active) {
            if (2 * g + 1 >= N) {
                exc = true;                     // the odd element at the end of the list: passed on
            } else {
                e0 = ba_load<SRC, false>(entries, table, recs, 2 * g);
                e1 = ba_load<SRC, false>(entries, table, recs, 2 * g + 1);
                if (e0.key != e1.key) split = true;
                else if (e0.ident || e1.ident) exc = true;
                else dx = sub<3, 1>(e1.x, e0.x);
            }
        }
        Fq29 p = mul(run, dx);
        if (active && !split && !exc && maybe_zero_mod(p) && is_zero_mod(p)) {   // same x: doubling or cancellation (rare)
            exc = true;
            const BaElem f0 = ba_load<SRC, true>(entries, table, recs, 2 * g), f1 = ba_load<SRC, true>(entries, table, recs, 2 * g + 1);
            p = ba_same_y(f0.y, f1.y) ? mul(run, normalize(dbl(f0.y))) : run;
        }
        run = p;
        ba_store9(prefix + (size_t)g * 9, run);
        const uint64_t sm = __ballot(split), em = __ballot(exc);
        if ((threadIdx.x & 63) == 0) {
            split_mask[g >> 6] = sm;
            exc_mask[g >> 6] = em;
        }
    }
    ba_store9(totals + (size_t)t * 9, run);
}

// exclusive popcount prefix over the split masks of the round's slots; the next round's geometry (and, after the last
// round, the XYZZ accumulation's plan)
__global__ void __launch_bounds__(1024) k_ba_scan(const uint64_t* __restrict__ split_mask, uint32_t* __restrict__ wpre,
                                                  const uint32_t* __restrict__ bp, uint32_t* __restrict__ bp_next, uint32_t B_next,
                                                  uint32_t* __restrict__ plan_out, uint32_t target_segments, uint32_t min_L) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t S = bp[BAP_S];
    const uint32_t words = (S + 63) / 64;
    const uint32_t per = (words + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < words ? lo + per : words;
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi; ++k) sum += (uint32_t)__popcll(split_mask[k]);
    uint32_t incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
        if ((threadIdx.x & 63) >= (uint32_t)off) incl += v;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, grand = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; ++w) {
        if (w < (threadIdx.x >> 6)) base += wave_tot[w];
        grand += wave_tot[w];
    }
    uint32_t run = base + incl - sum;
    for (uint32_t k = lo; k < hi; ++k) {
        wpre[k] = run;
        run += (uint32_t)__popcll(split_mask[k]);
    }
    if (threadIdx.x == 0) {
        const uint32_t Nn = S + grand;
        if (bp_next) {
            const uint32_t Sn = (Nn + 1) / 2;
            bp_next[BAP_N] = Nn;
            bp_next[BAP_S] = Sn;
            bp_next[BAP_T] = ((Sn + B_next - 1) / B_next + 63u) & ~63u;
        }
        if (plan_out) {
            uint32_t L = (uint32_t)(((uint64_t)Nn + target_segments - 1) / target_segments);
            if (L < min_L) L = min_L;
            plan_out[0] = Nn;                      // PLAN_N, PLAN_L, PLAN_T
            plan_out[1] = L;
            plan_out[2] = Nn ? (Nn + L - 1) / L : 0;
        }
    }
}

// the inverses of the lanes' totals: a second chain of BA_GROUP totals per lane, one Fermat inversion per lane
__global__ void __launch_bounds__(64) k_ba_invert(const uint32_t* __restrict__ totals, const uint32_t* __restrict__ bp,
                                                  uint32_t* __restrict__ chain, uint32_t* __restrict__ inv_totals) {
    const uint32_t T = bp[BAP_T];
    const uint32_t U = (T + BA_GROUP - 1) / BA_GROUP;
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const uint32_t first = u * BA_GROUP;
    const uint32_t cnt = first + BA_GROUP <= T ? BA_GROUP : T - first;
    Fq29 run = Fq29::one();
    for (uint32_t i = 0; i < cnt; ++i) {
        run = mul(run, ba_load9(totals + (size_t)(first + i) * 9));
        ba_store9(chain + ((size_t)i * U + u) * 9, run);
    }
    Fq29 inv = inv29(run);
    for (uint32_t i = cnt; i-- > 0;) {
        const Fq29 pm = i ? ba_load9(chain + ((size_t)(i - 1) * U + u) * 9) : Fq29::one();
        ba_store9(inv_totals + (size_t)(first + i) * 9, mul(inv, pm));
        inv = mul(inv, ba_load9(totals + (size_t)(first + i) * 9));
    }
}

// backward pass: every slot's inverse from the chain, the affine sum, the output record(s)
template <int SRC>
__global__ void __launch_bounds__(256) k_ba_backward(const uint64_t* __restrict__ entries, const uint32_t* __restrict__ table,
                                                     const uint32_t* __restrict__ recs, const uint32_t* __restrict__ bp, uint32_t B,
                                                     const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ inv_totals,
                                                     const uint64_t* __restrict__ split_mask, const uint64_t* __restrict__ exc_mask,
                                                     const uint32_t* __restrict__ wpre, uint32_t* __restrict__ out) {
    const uint32_t N = bp[BAP_N], S = bp[BAP_S], T = bp[BAP_T];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const uint32_t lane = threadIdx.x & 63;
    Fq29 run = ba_load9(inv_totals + (size_t)t * 9);
    for (uint32_t j = B; j-- > 0;) {
        const uint32_t g = j * T + t;
        if (g >= S) continue;
        const uint64_t sm = split_mask[g >> 6], em = exc_mask[g >> 6];
        const uint32_t pos = g + wpre[g >> 6] + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
        const bool split = (sm >> lane) & 1ull, exc = (em >> lane) & 1ull;
        if (2 * g + 1 >= N) {               // the odd element at the end
            const BaElem a = ba_load<SRC, true>(entries, table, recs, 2 * g);
            ba_store(out, pos, a.x, a.y, a.key, a.ident);
            continue;
        }
        const BaElem a = ba_load<SRC, true>(entries, table, recs, 2 * g), b = ba_load<SRC, true>(entries, table, recs, 2 * g + 1);
        if (split) {
            ba_store(out, pos, a.x, a.y, a.key, a.ident);
            ba_store(out, pos + 1, b.x, b.y, b.key, b.ident);
            continue;
        }
        Fq29 num, dx, x2 = b.x;
        if (exc) {
            if (a.ident || b.ident) {
                const BaElem& s = a.ident ? b : a;
                ba_store(out, pos, s.x, s.y, a.key, s.ident);
                continue;
            }
            if (!ba_same_y(a.y, b.y)) {     // P + (−P)
                ba_store(out, pos, a.x, a.y, a.key, true);
                continue;
            }
            const Fq29 xx = sqr(a.x);       // doubling: λ = 3x² / 2y
            num = normalize(add(dbl(xx), xx));
            dx = normalize(dbl(a.y));
            x2 = a.x;
        } else {
            num = sub<3, 1>(b.y, a.y);
            dx = sub<3, 1>(b.x, a.x);
        }
        const Fq29 pm = j ? ba_load9(prefix + (size_t)(g - T) * 9) : Fq29::one();
        const Fq29 inv = mul(run, pm);
        run = mul(run, dx);
        const Fq29 lam = mul(inv, num);
        const Fq29 x3 = ba_weak_reduce(normalize(sub<6, 2>(sqr(lam), add(a.x, x2))));
        const Fq29 y3 = ba_weak_reduce(normalize(sub<3, 1>(mul(lam, sub<3, 1>(a.x, x3)), a.y)));
        ba_store(out, pos, x3, y3, a.key, false);
    }
}

// the XYZZ accumulation over the records of the last round: k_accum_affine with the list elements carrying their own
// points.  Identity records are skipped.
__global__ void __launch_bounds__(256) k_accum_records(const uint32_t* __restrict__ recs, const uint32_t* __restrict__ plan,
                                                       uint32_t* __restrict__ bucket_sums, uint32_t* __restrict__ part_keys,
                                                       uint32_t* __restrict__ part_pts) {
    constexpr int ACC = Words29<Fq29>::ACC;
    const uint32_t N = plan[0], L = plan[1], T = plan[2];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const bool final_level = (T == 1);
    const uint32_t beg = t * L;
    const uint32_t end = beg + L < N ? beg + L : N;
    XYZZ29<Fq29> acc;
    bool inf = true, first = true;
    BaElem nxt = ba_load<1, true>(nullptr, nullptr, recs, beg);
    uint32_t cur = nxt.key;
    for (uint32_t k = beg; k < end; ++k) {
        const BaElem e = nxt;
        if (k + 1 < end) nxt = ba_load<1, true>(nullptr, nullptr, recs, k + 1);
        if (e.key != cur) {
            flush_run(cur, acc, inf, first, final_level, t, bucket_sums, part_keys, part_pts);
            first = false;
            inf = true;
            cur = e.key;
        }
        if (e.ident) continue;
        Affine29<Fq29> p{e.x, e.y};
        madd29(acc, inf, p);
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

}  // namespace cg
