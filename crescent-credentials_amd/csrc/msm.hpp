// Pippenger multi-scalar multiplication over BN254 G1/G2 on the GPU (internal C++ interface).
// Computes the same group element as ark-ec's `VariableBaseMSM::msm_bigint`
// (call sites forks/groth16/src/prover.rs:66,74,266).
#pragma once
#include "common.hpp"
#include "curve29.hpp"
#include "ntt.hpp"

#include <functional>

namespace cg {

// Fixed bases of one query, expanded for every window: row j holds 2^(c*j) * P_i.  With all windows
// pre-shifted, every signed digit of every scalar lands in ONE shared set of 2^(c-1) buckets, so a whole
// MSM is a single bucket accumulation plus a single bucket reduction — the layout 288 GB of HBM makes
// affordable (13x the key size at c = 20).  A table point is its two (G1) or four (G2) coordinates as
// canonical x·2^261 mod q, eight u32 each: 64 B / 128 B, exactly the size of the key's own points.
template <class F>
struct MsmBases {
    typedef typename To29<F>::type F29T;
    static constexpr int AFF = Words29<F29T>::AFF;
    uint64_t n = 0;
    int c = 0;          // window bits
    int W = 0;          // number of windows = ceil(255 / c)
    bool precomputed = true;   // false: table holds only window 0 and keys carry the window index
    DevBuf<uint32_t> table;    // rows * n * AFF words, window-major
    DevBuf<uint8_t> valid;     // 1 = base is not the identity
    // bases_dev: n affine points in Montgomery(2^256) form on the device (identity = all zero)
    void build(const Affine<F>* bases_dev, uint64_t n, int c, bool precompute, hipStream_t st);
    // the same from packed table points (row 0) and their validity flags already on the device
    void build_from_row0(const uint32_t* row0_dev, const uint8_t* valid_dev, uint64_t n, int c, hipStream_t st);
    // re-expand the same bases (row 0 of the table) for another window size.  Returns 1 when the table was rebuilt, 0 when
    // there was nothing to do, -1 when the new table would not fit next to the old one in free device memory (the old
    // window stays in force: never re-tune into an out-of-memory failure)
    int rebuild(int c_new, hipStream_t st);
    void alloc_rows(uint64_t n, int c, bool precompute);
    void expand_rows(hipStream_t st);
};

int msm_default_window(uint64_t n, bool precomputed);
// window minimising (mixed additions + bucket-reduction additions) for a scalar population observed on a proof:
// `nz_small` scalars that contribute a single non-zero digit (0/1-like wires) and `nz_full` full-width scalars.
// g2 = the additions are over Fq2 (the ratio of the two addition kinds is the same, so only the bucket term's weight differs).
int msm_best_window(uint64_t n_bases, double nz_small, double nz_full);

// The memory an engine needs only from its digits() to the end of its accumulate(): the two entry lists and the pieces of
// the segments.  It is most of a proof slot (the entry lists are 8 B x bases x windows, twice), and in a throughput context
// the MSMs of a proof run one after another on one stream, so the five engines of a slot SHARE one MsmScratch sized for
// the largest of them (prover.hip); an engine that was given none owns one.
struct MsmScratch {
    DevBuf<uint64_t> ent_a, ent_b;            // entries grouped by the high key bits (level 1) / by the whole key (level 2)
    DevBuf<uint32_t> part_keys_a, part_keys_b;
    DevBuf<uint32_t> part_pts_a, part_pts_b;  // u32 words: engines over Fq and Fq2 store 36- and 72-word accumulators in it
    // grows what is too small (contents are not kept), never shrinks
    uint64_t entry_bytes() const { return ent_a.bytes() + ent_b.bytes(); }
    uint64_t piece_bytes() const { return part_keys_a.bytes() + part_keys_b.bytes() + part_pts_a.bytes() + part_pts_b.bytes(); }
    void reserve(size_t n_ent_a, size_t n_ent_b, size_t keys_a, size_t pts_a, size_t keys_b, size_t pts_b) {
        if (ent_a.n < n_ent_a) ent_a.alloc(n_ent_a);
        if (ent_b.n < n_ent_b) ent_b.alloc(n_ent_b);
        if (part_keys_a.n < keys_a) part_keys_a.alloc(keys_a);
        if (part_pts_a.n < pts_a) part_pts_a.alloc(pts_a);
        if (part_keys_b.n < keys_b) part_keys_b.alloc(keys_b);
        if (part_pts_b.n < pts_b) part_pts_b.alloc(pts_b);
    }
};

// Per-MSM working set; reusable across proofs.
template <class F>
struct MsmEngine {
    typedef typename To29<F>::type F29T;
    static constexpr int ACC = Words29<F29T>::ACC;   // u32 words of a stored XYZZ accumulator (144 B / 288 B)
    const MsmBases<F>* bases = nullptr;
    uint64_t cap_entries = 0;
    uint32_t nbuckets_total = 0;  // buckets per window * windows-in-key-space
    // digit entries: bucket key in the high word, table index | sign<<31 in the low word; mem().ent_a grouped by the high
    // key bits (level 1), mem().ent_b by the whole key (level 2; unused when one level covers the key)
    MsmScratch own_mem;
    MsmScratch* shared_mem = nullptr; // set before init(): scratch shared with engines whose MSMs never overlap this one's
    MsmScratch& mem() { return shared_mem ? *shared_mem : own_mem; }
    const MsmScratch& mem() const { return shared_mem ? *shared_mem : own_mem; }
    int bits1 = 0, bits2 = 0;         // key bits taken by the two partition levels
    DevBuf<uint32_t> blk_hist;        // per level-1 block: its counts per bin
    DevBuf<uint32_t> counters;        // plan | level-1 histogram, cursors | level-2 histogram, cursors (zeroed per MSM)
    DevBuf<uint32_t> starts;          // level-1 bin starts | first level-2 chunk of every bin
    uint32_t max_chunks = 0;          // launch bound of the level-2 kernels
    uint32_t min_L = 16;              // shortest segment (entries per lane) the plan may choose
    uint32_t max_segments = 0;        // launch bound of the accumulation (the plan's segment count is at most this)
    DevBuf<uint32_t> bucket_sums;     // nbuckets_total * ACC
    DevBuf<uint32_t> rows_buf, cols_buf;   // row / column sums of the bucket matrix (bucket reduction)
    DevBuf<uint32_t> rowp_buf, colp_buf;   // sums of 32-bucket chunks of the rows / columns (throughput contexts)
    int red_rbits = 0, red_cbits1 = 0;     // bits of the row weights r < R and of the column weights col + 1 <= C
    // host memory the reduction's last kernel writes directly:
    PinnedBuf<uint32_t> h_plan;       // the device plan of the last MSM: [0] entries, [3] scalars with a non-zero digit
    PinnedBuf<uint32_t> h_result;     // per window: red_rbits + red_cbits1 per-bit sums of the rows / the columns
    uint64_t n_scalars = 0;
    // set before init() for engines whose MSMs follow each other on ONE stream: the MSM's last kernels leave the partition
    // counters and the bucket array zeroed for the next MSM, and the two fill launches that opened an MSM are skipped.
    // (With an MSM spread over several streams another engine may still be reading this one's plan - MsmEngine::adopt - when
    // its last kernel runs, so those engines fill at the start.)  The *_clean flags say whether the invariant holds: an MSM
    // that was abandoned half-way (a failed call) leaves them false and the next MSM fills.
    bool zero_at_end = false;
    bool counters_clean = false, buckets_clean = false;
    bool latency_mode = false;        // set before init(): short segments (one proof at a time matters more than proofs per second)
#ifdef CG_WITH_BATCH_AFFINE
    // batch-affine pair rounds in front of the accumulation (batchaff.hpp; G1 only; 0 = off)
    bool ba_allowed = true;           // set before init()
    int ba_rounds = 0;
    uint32_t ba_B = 32;               // slots chained per lane
    uint32_t ba_tcap[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // launch bound (lanes) of every round
    DevBuf<uint32_t> ba_prefix, ba_totals, ba_inv, ba_chain, ba_wpre, ba_rec_a, ba_rec_b, ba_plan;
    DevBuf<uint64_t> ba_split, ba_exc;
#endif
    // valid once the stream has been synchronised
    uint32_t n_entries() const { return h_plan.p ? h_plan.p[0] : 0; }
    uint32_t n_nonzero() const { return h_plan.p ? h_plan.p[3] : 0; }
    void enqueue_reduction(hipStream_t st);
    // timing events (recorded on the MSM's own stream): digits start, sort begin/end, level-1
    // accumulation kernel begin/end, result ready
    hipEvent_t ev_t[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float ms_total() const;   // digits start -> result ready (valid after the stream is synchronised)
    float ms_sort() const;
    float ms_accum() const;

    // zero_stream: the stream the initial zero-fill of the counters and buckets is enqueued on; the CALLER waits for it before the
    // engine's first MSM (a loader that makes many engines waits once).  Null: a stream of the engine's own, waited for here.
    void init(const MsmBases<F>* b, hipStream_t zero_stream = nullptr);
    ~MsmEngine();
    // device memory this engine holds, by kind (cg_ctx_get_info): entry lists and segment pieces it OWNS (nothing when it
    // works in a shared scratch), and the rest (bucket array, reduction buffers, partition counters)
    void device_bytes(uint64_t& entries, uint64_t& pieces, uint64_t& other) const;
    // phase 1: signed-digit extraction of `n` canonical scalars -> (bucket, index) entries grouped by bucket
    void digits(const Fr* scalars_dev, uint64_t n, hipStream_t st);
    // phase 1 taken over from another engine that grouped the SAME scalars against bases with the same identity pattern,
    // window and count (the B query in G1 and in G2: b_i(τ)·G1 and b_i(τ)·G2 vanish together, generator.rs:162,168): its
    // entry list IS this engine's.  `src`'s digits() must be enqueued on a stream `st` already waits for.
    template <class G>
    bool can_adopt(const MsmEngine<G>& src) const {
        return bases->precomputed && src.bases->precomputed && bases->n == src.bases->n && bases->c == src.bases->c &&
               bits1 == src.bits1 && bits2 == src.bits2;
    }
    void adopt(const uint64_t* grouped_entries, const uint32_t* plan_dev, uint64_t n, hipStream_t st);
    const uint64_t* grouped() const { return bits2 ? mem().ent_b.p : mem().ent_a.p; }
    const uint64_t* adopted = nullptr;    // non-null: the grouped entries of the engine adopted for the current MSM
    // phase 2: accumulate, combine, reduce; per-bit sums and the plan copied to pinned memory.  Neither phase waits
    // for the host.
    void accumulate(hipStream_t st);
    // after the stream has been synchronised: the MSM value (host arithmetic, Montgomery 2^256 form)
    XYZZ<F> value() const;
};

// Two changes of basis made ONCE per key so that a proof needs four transforms instead of seven (G1 only).
//
// prover.rs:63-66 computes h_acc = Σ_i h_i·H_i with h = coset_ifft(((a∘b − c)/Z) on the coset) (r1cs_to_qap.rs:187-210).
// Z is the constant 1/vinv on the coset and coset_ifft is linear, so h = coset_ifft(vinv·a∘b) − vinv·coeffs(c), and
//   (1) Σ_i coset_ifft(q)_i·H_i = Σ_j q_j·H'_j,  H'_j = Σ_i ω^{-ij}·(g^-i / n)·H_i   (H_i = identity for i >= n_h: the key
//       holds n − 1 points) — the h query in the evaluation basis of the coset (ec_transform_h_bases): the coset values
//       q_j = vinv·a_j·b_j are the MSM's scalars and the seventh transform disappears;
//   (2) −vinv·Σ_i coeffs(c)_i·H_i = Σ_j c_j·G'_j with c_j = <C_j, w> and G'_j = −(vinv / n)·Σ_i ω^{-ij}·H_i, which is
//       Σ_k w_k·P_k,  P_k = Σ_j C_jk·G'_j: the C matrix folded into per-wire points that are simply added to the l query
//       (ec_fold_c_into_l) — the sparse product with C and both of c's transforms disappear.  The folded l query has one
//       base per wire (the ℓ instance wires included) and takes the full assignment as its scalars.
// Every step is an identity between group elements, so the proof bytes are unchanged for ANY assignment, satisfying
// or not.  Both are inverse DFTs over group elements (n/2·log n scalar multiplications by twiddles each, ≈1 s at 2^21).
void ec_inverse_dft(const uint32_t* row0_in, const uint8_t* valid_in, uint64_t n_in, int logn, const Fr& scale_base,
                    const Fr& scale_mult, uint32_t* row0_out, uint8_t* valid_out, hipStream_t st);
void ec_transform_h_bases(const uint32_t* row0_in, const uint8_t* valid_in, uint64_t n_in, int logn, uint32_t* row0_out,
                          uint8_t* valid_out, hipStream_t st);
// row0_out / valid_out: M packed table points: P_k for k < num_inputs, l_query[k − num_inputs] + P_k above
void ec_fold_c_into_l(const uint32_t* h_row0, const uint8_t* h_valid, uint64_t n_h, int logn, const Fr& vanishing_inv,
                      const cg_csr& c_matrix, uint64_t num_constraints, uint64_t num_inputs, uint64_t M,
                      const uint32_t* l_row0, const uint8_t* l_valid, uint32_t* row0_out, uint8_t* valid_out, hipStream_t st);
void ec_fold_ct_into_l(const uint32_t* h_row0, const uint8_t* h_valid, uint64_t n_h, int logn, const Fr& vanishing_inv,
                       const HostCsc& c_transposed, uint64_t num_constraints, uint64_t num_inputs, uint64_t M,
                       const uint32_t* l_row0, const uint8_t* l_valid, uint32_t* row0_out, uint8_t* valid_out, hipStream_t st);
void sum_xyzz_by_key(const uint32_t* keys, const uint32_t* pts, uint64_t count, uint32_t* sums, hipStream_t st);
// window tables of the transformed h query over the points h_first + k·h_stride, k < h_count, of 2^logn and of the folded
// l query over [l_first, l_first + l_count) of M
void build_hl_bases_folded(MsmBases<Fq>& out_h, MsmBases<Fq>& out_l, const Affine<Fq>* h_bases_dev, uint64_t n_h, int logn,
                           const Affine<Fq>* l_bases_dev, uint64_t num_inputs, uint64_t M, const cg_csr& c_matrix,
                           uint64_t num_constraints, const Fr& vanishing_inv, uint64_t h_first, uint64_t h_stride, uint64_t h_count,
                           int c_h, uint64_t l_first, uint64_t l_count, int c_l, hipStream_t st, float* ms_fold = nullptr,
                           float* ms_tables = nullptr);

// The two halves of the above from ROW-0 TABLE POINTS already on the device (a staged load keeps every query as its row 0
// and folds on a worker thread).  ms_fold / ms_tables (optional) accumulate the host-clock milliseconds spent in the
// change of basis and in the expansion of the window rows.  pick_c_l is asked for the l query's window AFTER the fold (the
// caller may have learnt a proof's digit statistics by then).
void build_h_bases_folded(MsmBases<Fq>& out_h, const uint32_t* h_row0, const uint8_t* h_valid, uint64_t n_h, int logn, uint64_t h_first,
                          uint64_t h_stride, uint64_t h_count, int c_h, hipStream_t st, float* ms_fold, float* ms_tables);
void build_l_bases_folded(MsmBases<Fq>& out_l, const uint32_t* h_row0, const uint8_t* h_valid, uint64_t n_h, int logn,
                          const uint32_t* l_row0, const uint8_t* l_valid, uint64_t num_inputs, uint64_t M, const HostCsc& c_transposed,
                          uint64_t num_constraints, const Fr& vanishing_inv, uint64_t l_first, uint64_t l_count,
                          const std::function<int()>& pick_c_l, hipStream_t st, float* ms_fold, float* ms_tables);

// import packed affine points (64 B / 128 B each, `coord_form`) into Montgomery Affine<F> on the device
template <class F>
void import_bases(const uint8_t* host_bytes, uint32_t coord_form, uint64_t n, Affine<F>* out_dev, hipStream_t st);

extern template struct MsmBases<Fq>;
extern template struct MsmBases<Fq2>;
extern template struct MsmEngine<Fq>;
extern template struct MsmEngine<Fq2>;

}  // namespace cg
