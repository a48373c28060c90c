// libcrescent_gpu: the C ABI of include/crescent_gpu.h.
//
// Host orchestration of the Groth16 prove path: the structure follows
// forks/groth16/src/prover.rs:26-136,256-274 and r1cs_to_qap.rs:150-213 (what is computed and in
// which algebraic order), re-laid-out for one MI355X: the proving key lives in HBM as per-window
// base tables, the five MSMs and the witness map run on separate HIP streams, and only the few
// scalar multiplications by r, s and the final additions run on the host.
#include <atomic>
#include <condition_variable>
#include <shared_mutex>
#include <memory>
#include <mutex>
#include <chrono>
#include <cstdlib>
#include <exception>
#include <functional>
#include <thread>

#include "msm.hpp"
#include "ntt.hpp"
#include "wmap29.hpp"

namespace cg {

const char* csr_view_problem(const cg_csr& m);   // unit.hip

std::string& last_error() {
    static thread_local std::string e;
    return e;
}
int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

// a > b as 256-bit integers (canonical limbs)
static bool limbs_gt(const uint32_t a[8], const uint32_t b[8]) {
    for (int i = 7; i >= 0; --i) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return false;
}
// ark-serialize SWFlags for the uncompressed form [ark-mem; SURVEY Appendix B]: bit 7 of the last byte
// is set when y is the larger of {y, -y}; bit 6 marks infinity (all-zero coordinates).  Isolated here.
static void g1_serialize_uncompressed(const G1Affine& p, uint8_t out[64]) {
    if (p.is_inf()) { memset(out, 0, 64); out[63] |= 0x40; return; }
    Fq y = from_mont(p.y), ny = from_mont(neg(p.y));
    g1_export_canonical(p, out);
    if (limbs_gt(y.l, ny.l)) out[63] |= 0x80;
}
static void g2_serialize_uncompressed(const G2Affine& p, uint8_t out[128]) {
    if (p.is_inf()) { memset(out, 0, 128); out[127] |= 0x40; return; }
    g2_export_canonical(p, out);
    Fq2 n = neg(p.y);
    Fq y1 = from_mont(p.y.c1), n1 = from_mont(n.c1), y0 = from_mont(p.y.c0), n0 = from_mont(n.c0);
    // QuadExtField ordering compares c1 first, then c0 [ark-mem]
    bool gt = limbs_gt(y1.l, n1.l) || (y1 == n1 && limbs_gt(y0.l, n0.l));
    if (gt) out[127] |= 0x80;
}


template <class F>
static XYZZ<F> scalar_mul_bytes(const Affine<F>& p, const uint8_t k[32]) {
    uint32_t e[8];
    memcpy(e, k, 32);
    return scalar_mul(XYZZ<F>::from_affine(p), e);
}
template <class F>
static XYZZ<F> scalar_mul_bytes(const XYZZ<F>& p, const uint8_t k[32]) {
    uint32_t e[8];
    memcpy(e, k, 32);
    return scalar_mul(p, e);
}

// Host-side fixed-base table of one point (delta_g1, delta_g2): 64 four-bit windows x 15 affine multiples, built once
// per circuit, so that r·delta, s·delta and (r·s)·delta cost 64 mixed additions instead of a 254-bit double-and-add.
template <class F>
struct FixedBase {
    std::vector<Affine<F>> t;   // t[j*15 + (d-1)] = d·16^j·P
    void build(const Affine<F>& p) {
        t.assign(64 * 15, Affine<F>::inf());
        if (p.is_inf()) return;
        XYZZ<F> base = XYZZ<F>::from_affine(p);
        for (int j = 0; j < 64; ++j) {
            const Affine<F> b = to_affine(base);
            XYZZ<F> acc = XYZZ<F>::from_affine(b);
            t[j * 15] = b;
            for (int d = 2; d <= 15; ++d) {
                madd(acc, b);
                t[j * 15 + d - 1] = to_affine(acc);
            }
            if (j + 1 < 64) { madd(acc, b); base = acc; }     // 16·16^j·P
        }
    }
    XYZZ<F> mul(const uint8_t k[32]) const {
        XYZZ<F> acc = XYZZ<F>::inf();
        if (t.empty()) return acc;
        for (int j = 0; j < 64; ++j) {
            const unsigned d = (k[j >> 1] >> ((j & 1) * 4)) & 15u;
            if (d) madd(acc, t[j * 15 + d - 1]);
        }
        return acc;
    }
};

struct Range { uint64_t lo, hi; };
static Range shard_range(uint64_t n, int rank, int count) {
    if (count <= 1) return {0, n};
    return {n * (uint64_t)rank / (uint64_t)count, n * (uint64_t)(rank + 1) / (uint64_t)count};
}

}  // namespace cg

using namespace cg;

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
// Per-proof working set: everything a proof in flight writes.  A context owns `n_slots` of them so that
// several proofs can overlap on one GPU (the latency-bound tails of one proof hide under the bulk kernels
// of another); the key tables, matrices and NTT tables are shared and read-only.
struct ProofSlot {
    // the slot's device and page-locked buffers are carved from ONE allocation each (common.hpp SlotArena) when the size is
    // known - every slot after a set's first; declared first: released after everything that points into it
    std::unique_ptr<SlotArena> arena;
    std::mutex busy;
    // one-stream slots: the five MSMs of a proof run one after another, so their entry lists and segment pieces live in
    // ONE scratch sized for the largest (declared before the engines that point into it: destroyed after them)
    MsmScratch scratch;
    MsmEngine<Fq> eh, el, ea, eb1;
    MsmEngine<Fq2> eb2;
    DevBuf<Fr> h_canon;
    DevBuf<Fr> q2;                     // cg_prove_partial_q_finish2: the second side's slice when it arrives in host memory (made on first use)
    const Fr* knock_h = nullptr;       // tuning builds (KNOCK & 16): the h scalars of this slot's first proof, reused
    Wm29Buffers wm;
    hipStream_t st[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // 0: witness map + h, 1: l, 2: a, 3: b1, 4: b2
    bool st_borrowed[5] = {false, false, false, false, false};          // a lone slot runs on streams of the one-stream slots
    hipEvent_t ev_w = nullptr;
    hipEvent_t ev_b1 = nullptr;        // b1's entries are grouped (the G2 MSM adopts them)
    hipEvent_t ev_done = nullptr;      // one-stream slots: recorded behind the proof's last kernel and polled (wait_sleeping)
    hipEvent_t ev_fin[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // five-stream slots: behind the proof's last kernel on each stream
    bool one_stream = false;
    // the extra slot of a throughput context, taken by a proof that arrives when no other is in flight (cg_ctx::acquire): five
    // streams and the latency arrangement of the engines, as a latency context's only slot has them
    bool lone = false;
    hipEvent_t ev_t[2] = {nullptr, nullptr};
    ~ProofSlot() {
        for (int i = 0; i < 5; ++i)
            if (st[i] && !st_borrowed[i] && (i == 0 || st[i] != st[0])) (void)hipStreamDestroy(st[i]);
        if (ev_w) (void)hipEventDestroy(ev_w);
        if (ev_b1) (void)hipEventDestroy(ev_b1);
        if (ev_done) (void)hipEventDestroy(ev_done);
        for (auto& e : ev_fin) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_t) if (e) (void)hipEventDestroy(e);
    }
};

// The device copy of one assignment on its way into a proof.  A context owns two more of these than proof slots and
// each has a copy-only stream, so a caller's upload overlaps the proofs in flight WITHOUT holding one of their working
// sets, and never puts a copy in front of another proof's kernels (see upload_assignment).
struct Upload {
    std::mutex busy;
    DevBuf<Fr> w;
    hipStream_t st = nullptr;          // one of the context's few copy-only streams (cg_ctx::up_streams: owned there)
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipEvent_t ev_done = nullptr;      // recorded behind the copy and polled (wait_sleeping)
    ~Upload() {
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        if (ev_done) (void)hipEventDestroy(ev_done);
    }
};

// The context's reader / writer gate: proofs hold it shared from slot acquisition to their last stream synchronisation; a
// window re-tune, the swap that ends a staged load and cg_circuit_free hold it exclusively.  WRITER-PREFERRING, unlike
// std::shared_mutex on glibc: with a dozen callers proving back to back some proof always holds the gate shared, and a
// writer that only gets in when no reader is there would wait for as long as the host keeps the context busy.  A waiting
// writer stops new readers; the readers in flight (one proof each, a few tens of ms) drain; the writer runs.
// (No thread takes it twice: every entry point takes it once, and the re-tune check runs after its proof has let go.)
class TuneGate {
    std::mutex mu;
    std::condition_variable cv;
    int readers = 0, writers_waiting = 0;
    bool writer = false;
public:
    void lock_shared() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !writer && writers_waiting == 0; });
        ++readers;
    }
    // For a reader that may ALREADY hold the gate through another handle (cg_prove_partial_q_begin: a host opens proof k + 1
    // before it finishes proof k): it passes a WAITING writer - queueing behind it would wait for the writer, which waits for
    // this caller's first handle - and waits only for an ACTIVE one.
    void lock_shared_passing_waiting_writers() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !writer; });
        ++readers;
    }
    void unlock_shared() {
        std::lock_guard<std::mutex> lk(mu);
        if (--readers == 0) cv.notify_all();
    }
    void lock() {
        std::unique_lock<std::mutex> lk(mu);
        ++writers_waiting;
        cv.wait(lk, [&] { return !writer && readers == 0; });
        --writers_waiting;
        writer = true;
    }
    void unlock() {
        std::lock_guard<std::mutex> lk(mu);
        writer = false;
        cv.notify_all();
    }
};

// statistics of one finished proof's assignment-driven MSMs (window re-tune; the window choice of a staged load)
struct TuneStats {
    bool valid = false;
    struct Q { uint64_t n_scalars = 0; double nonzero = 0, entries = 0; int W0 = 0; } l, a, b1, b2;
};

struct cg_ctx {
    ~cg_ctx() {
        uploads.clear();
        for (hipStream_t us : up_streams) if (us) (void)hipStreamDestroy(us);
    }
    int device = 0;
    uint64_t l = 0, m = 0, M = 0, D = 0;
    int logD = 0;
    int shard_rank = 0, shard_count = 1;
    int span_lo = 0, span_hi = 0;       // cg_options.shard_span (1/10000 of every query); 0, 0 = equal parts
    // h query held in the coset evaluation basis and the C matrix folded into the l query (msm.hpp): four transforms
    // per proof instead of seven, no sparse product with C; false = the reference's arrangement
    bool folded = true;
    // host copies of the single points the finishing step needs (Montgomery)
    G1Affine alpha_g1, beta_g1, delta_g1, a0, b1_0;
    G2Affine beta_g2, delta_g2, b2_0;
    FixedBase<Fq> fb_delta_g1;
    FixedBase<Fq2> fb_delta_g2;
    // scalar ranges (into the MSM operand numbering) owned by this shard
    Range rh, rl, ra;   // h: [0, D-1), l: [0, M-l), a/b: [0, M-1)
    MsmBases<Fq> bh, bl, ba, bb1;
    MsmBases<Fq2> bb2;
    DevCsr A, B, C;
    Csr29 dA, dB, dC;
    NttDomain dom;
    Wm29Domain wdom;
    // a shard of a folded key over a power-of-two shard count owns the coset points j ≡ shard_rank (mod shard_count) of
    // the h MSM instead of a contiguous range (wmap29.hpp Wm29Strided): half of its transforms shrink by the shard count
    bool h_strided = false;
    Wm29Strided wstr;
    std::vector<std::unique_ptr<ProofSlot>> slots;
    // The upload buffers share a FEW copy-only streams (round 5).  With a stream per buffer a context held proof_slots + 2 of
    // them next to its proof streams - 34 streams on 16 hardware queues - so an upload's barrier packet (the copy engine's
    // completion, ~1 ms) sat in a hardware queue in front of some other proof's kernels: the 2 % between the host-memory and
    // the device-resident rate.  Four streams carry 200 uploads of 1 ms a second with room to spare, and with the proof
    // streams they fit the hardware queues one each (cg_init asks for 20).
    std::vector<hipStream_t> up_streams;
    std::vector<std::unique_ptr<Upload>> uploads;
    std::mutex up_mu;
    std::condition_variable up_cv;
    Upload* acquire_upload() {
        std::unique_lock<std::mutex> lk(up_mu);
        for (;;) {
            for (auto& u : uploads)
                if (u->busy.try_lock()) return u.get();
            up_cv.wait(lk);
        }
    }
    void release_upload(Upload* u) {
        u->busy.unlock();
        std::lock_guard<std::mutex> lk(up_mu);
        up_cv.notify_one();
    }
    // window tuning: the window of each assignment-driven query is re-chosen once from the digit statistics of
    // the first proof (circom witnesses are mostly 0/1 wires, for which the size-based default is far too wide)
    bool fixed_window = false;
    bool b_same_identities = false;   // b_g1_query and b_g2_query are the identity at the same indices (true for a generated key)
    std::atomic<bool> tuned{false};   // read outside tune_mu by every finished proof
    std::atomic<int> retune_attempts{0};        // finished proofs whose statistics were looked at for the re-tune
    std::atomic<int> retune_skipped_memory{0};  // queries whose re-tuned table did not fit next to the old one
    // a re-tune that failed AFTER a table was rebuilt, while the slots' engines were being re-sized for it: some engines
    // are cut for the old window, the table is the new one.  Every later proof is refused (reload the circuit)
    std::atomic<bool> broken{false};
    TuneGate tune_mu;            // proofs hold it shared; a retune, the staged load's swap and cg_circuit_free hold it exclusively
    // ---- staged load (CG_FLAG_STAGED_LOAD): the context proves in the warm-up arrangement (coefficient basis, row-0
    // tables, one bucket set per window) while `worker` builds the final one and swaps it in (staged_worker) ----
    std::atomic<bool> warmup{false};          // the warm-up arrangement is in force
    std::thread worker;
    std::atomic<bool> cancel{false};          // cg_circuit_free: the worker stops at its next step
    int n_slots_final = 1;
    int window_opt = 0;                       // cg_options.window_bits
    HostCsc c_transposed;                     // C^T for the fold, made while the caller's arrays were still there
    std::mutex ready_mu;
    std::condition_variable ready_cv;
    bool ready = true;                        // the final arrangement is in force (guarded by ready_mu)
    int bg_status = 0;
    std::string bg_error;
    std::mutex warm_mu;                       // warm_stats
    TuneStats warm_stats;                     // the first representative warm-up proof's digit statistics
    std::atomic<int> warmup_proofs{0};
    cg_load_timings lt{};
    std::chrono::steady_clock::time_point t_load0;
    // entry points that are still somewhere inside this context (CallGuard): a call lets go of tune_mu before its re-tune
    // check and its host finish, both of which read the context, so cg_circuit_free waits for this count as well
    std::atomic<int> calls_inside{0};
    bool latency = false;
    bool spin_wait = false;      // CG_FLAG_SPIN_WAIT
    bool external_q = false;     // CG_FLAG_H_SCALARS_EXTERNAL: no witness-map resources; the h scalars arrive with every proof
    // device bytes that stay resident (cg_ctx_get_info): window tables + validity flags | matrices and domain tables | one slot
    // (slot_bytes is the sum of slot_part: entry lists, segment pieces, bucket arrays and reduction buffers, the witness
    // map's vectors + the h MSM's scalars, one upload buffer - account_slot)
    int64_t table_bytes = 0, matrix_bytes = 0, slot_bytes = 0;
    uint64_t slot_part[5] = {0, 0, 0, 0, 0};
    std::mutex pick_mu;
    std::condition_variable pick_cv;
    int in_flight = 0;           // slots taken (pick_mu)
    // Blocks until a slot is free; returns it locked.  A throughput context keeps its proofs on one stream each because the
    // overlap comes from the OTHER proofs in flight; a proof that finds none (a server between requests: one task per credential,
    // sample/client_helper/src/main.rs:177-216) would run its seventy launches back to back on an otherwise empty chip - 10.9 ms
    // where a latency context takes 6.  Such a proof (`may_run_alone`, and fewer than n_lone proofs in flight) gets one of the
    // context's LONE slots.
    ProofSlot* acquire(bool may_run_alone = false) {
        std::unique_lock<std::mutex> lk(pick_mu);
        for (;;) {
            if (may_run_alone && in_flight < n_lone)
                for (auto& sp : slots)
                    if (sp->lone && sp->busy.try_lock()) { ++in_flight; return sp.get(); }
            for (auto& sp : slots)
                if (!sp->lone && sp->busy.try_lock()) { ++in_flight; return sp.get(); }
            pick_cv.wait(lk);
        }
    }
    void release(ProofSlot* sl) {
        sl->busy.unlock();
        std::lock_guard<std::mutex> lk(pick_mu);
        --in_flight;
        pick_cv.notify_one();
    }
    int64_t lone_slot_bytes = 0;
    int n_lone = 0;              // lone slots wanted: a throughput context with more than one slot, unless CG_FLAG_NO_LONE_SLOT
};
// what ONE proof slot holds on the device, by kind, read off the first slot's buffers (the slots are identical); called
// when the slots are made and after a re-tune has re-sized them
static uint64_t slot_device_bytes(const ProofSlot& S, uint64_t part[4]) {
    uint64_t ent = S.scratch.entry_bytes(), pcs = S.scratch.piece_bytes(), oth = 0;
    S.eh.device_bytes(ent, pcs, oth); S.el.device_bytes(ent, pcs, oth); S.ea.device_bytes(ent, pcs, oth);
    S.eb1.device_bytes(ent, pcs, oth); S.eb2.device_bytes(ent, pcs, oth);
    part[0] = ent; part[1] = pcs; part[2] = oth;
    part[3] = S.wm.device_bytes() + S.h_canon.bytes();
    return ent + pcs + oth + part[3];
}
static void account_slot(cg_ctx* c) {
    uint64_t part[4] = {0, 0, 0, 0};
    c->lone_slot_bytes = 0;
    for (auto& sp : c->slots)
        if (sp->lone) { uint64_t lp[4]; c->lone_slot_bytes += (int64_t)slot_device_bytes(*sp, lp); }
    if (!c->slots.empty() && !c->slots[0]->lone) (void)slot_device_bytes(*c->slots[0], part);
    for (int k = 0; k < 4; ++k) c->slot_part[k] = part[k];
    c->slot_part[4] = c->uploads.empty() ? 0 : c->uploads[0]->w.bytes();      // about one upload buffer per proof in flight
    c->slot_bytes = 0;
    for (uint64_t v : c->slot_part) c->slot_bytes += (int64_t)v;
}

struct CallGuard {
    cg_ctx* c;
    explicit CallGuard(cg_ctx* ctx) : c(ctx) { if (c) c->calls_inside.fetch_add(1, std::memory_order_acq_rel); }
    CallGuard(const CallGuard&) = delete;
    CallGuard& operator=(const CallGuard&) = delete;
    ~CallGuard() { if (c) c->calls_inside.fetch_sub(1, std::memory_order_acq_rel); }
};
struct UploadGuard {
    cg_ctx* c = nullptr;
    Upload* u = nullptr;
    UploadGuard() = default;
    UploadGuard(const UploadGuard&) = delete;
    UploadGuard& operator=(const UploadGuard&) = delete;
    void take(cg_ctx* ctx) { c = ctx; u = ctx->acquire_upload(); }
    ~UploadGuard() {
        if (!u) return;
        (void)hipStreamSynchronize(u->st);      // a failed call may leave its copy in flight
        c->release_upload(u);
    }
};
struct SlotGuard {
    cg_ctx* c;
    ProofSlot* s;
    int exceptions;
    explicit SlotGuard(cg_ctx* ctx, bool may_run_alone = false) : c(ctx), s(ctx->acquire(may_run_alone)), exceptions(std::uncaught_exceptions()) {}
    ~SlotGuard() {
        // a failure part-way through a proof may leave kernels queued on the slot's streams: drain them before the
        // working set is handed to the next proof
        if (std::uncaught_exceptions() > exceptions)
            for (auto st : s->st) if (st) (void)hipStreamSynchronize(st);
        c->release(s);
    }
};

namespace cg {
int translate_current_exception() {
    try {
        throw;
    } catch (const HipError& e) {
        last_error() = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        last_error() = "host allocation failed";
        return CG_ERR_OUT_OF_MEMORY;
    } catch (const std::exception& e) {
        last_error() = e.what();
        return CG_ERR_HIP;
    }
}
}  // namespace cg
static int translate_exception() { return cg::translate_current_exception(); }

extern "C" const char* cg_last_error(void) { return last_error().c_str(); }

extern "C" const char* cg_version(void) {
#ifdef CG_WITH_BATCH_AFFINE
    return "crescent_gpu 0.1 (gfx950; BN254 Groth16 prove path: MSM G1/G2 + NTT + witness map) [experiment build: batch-affine]";
#else
    return "crescent_gpu 0.1 (gfx950; BN254 Groth16 prove path: MSM G1/G2 + NTT + witness map)";
#endif
}

extern "C" int cg_init(int n_devices, const int* device_ids) {
    // The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); kernels of streams that share a
    // queue cannot overlap, and a process with more than ~24 user queues is time-sliced by the hardware scheduler (a
    // lone proof then meets 15 ms stalls: profiles/r03_a_streams_and_queues.txt).  20 holds sixteen one-stream proofs in
    // flight and a context's four copy-only upload streams one queue each (round 5), without either effect.  Effective only if HIP has not initialised yet in
    // this process; a host that initialises HIP earlier should export the variable itself.
    (void)setenv("GPU_MAX_HW_QUEUES", "20", 0);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return fail(CG_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    for (int i = 0; i < n_devices; ++i)
        if (!device_ids || device_ids[i] < 0 || device_ids[i] >= count)
            return fail(CG_ERR_INVALID_ARGUMENT, "device id out of range (have %d devices)", count);
    return CG_OK;
}

extern "C" int cg_set_device(int32_t device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) return fail(CG_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(CG_ERR_INVALID_ARGUMENT, "device id out of range (have %d devices)", count);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(CG_ERR_HIP, "hipSetDevice(%d) failed: %s", (int)device, hipGetErrorString(e));
    return CG_OK;
}

extern "C" uint64_t cg_domain_size(const cg_ctx* ctx) { return ctx ? ctx->D : 0; }

static float ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// one query of the key -> its tables.  precompute = false keeps row 0 only (the warm-up arrangement of a staged load: keys
// carry the window, one bucket set per window).  ms_copy / ms_tables: host-clock milliseconds of the copy + import and of
// the table rows.
template <class F>
static void load_query(MsmBases<F>& bases, const uint8_t* bytes, uint32_t form, uint64_t first, uint64_t count, int window_bits,
                       bool precompute, hipStream_t st, float* ms_copy, float* ms_tables) {
    constexpr size_t PT = sizeof(Affine<F>);
    auto t0 = std::chrono::steady_clock::now();
    DevBuf<Affine<F>> tmp(count ? count : 1);
    import_bases<F>(bytes + first * PT, form, count, tmp.p, st);
    if (ms_copy) *ms_copy += ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    int c = window_bits > 0 ? window_bits : msm_default_window(count ? count : 1, precompute);
    bases.build(tmp.p, count, c, precompute, st);
    CG_HIP(hipStreamSynchronize(st));
    if (ms_tables) *ms_tables += ms_since(t0);
}

// One proof slot (working set + streams) over the given tables.  Everything else it is cut for - the domain, the matrices'
// scratch, the mode - is the context's.
// zs: the stream the engines' initial zero-fills go out on; the caller waits for it ONCE after its last slot (a wait per
// engine is a wait for a fill kernel to be scheduled, and behind a busy context that is tens of ms each: eighty of them made
// a sixteen-slot set take 2 s on a loaded GPU).
// rec: what a slot of this set asks for, measured on the set's first slot (built buffer by buffer) and used to size the ONE
// device and ONE page-locked allocation every later slot is carved from.
struct SlotRecipe { size_t dev_bytes = 0, host_bytes = 0; };
static std::unique_ptr<ProofSlot> make_slot(cg_ctx* c, const MsmBases<Fq>* bh, const MsmBases<Fq>* bl, const MsmBases<Fq>* ba,
                                            const MsmBases<Fq>* bb1, const MsmBases<Fq2>* bb2, hipStream_t zs, SlotRecipe& rec, bool lone = false,
                                            const std::vector<hipStream_t>& borrow = {}) {
    CG_HIP(hipSetDevice(c->device));
    std::unique_ptr<ProofSlot> sl(new ProofSlot());
    sl->lone = lone;
    AllocMeter meter;
    if (rec.dev_bytes) {
        sl->arena.reset(new SlotArena());
        DevBuf<uint8_t> whole(rec.dev_bytes);                 // (DevBuf for its out-of-memory report; the arena takes the memory over)
        sl->arena->dev = whole.p; sl->arena->dev_size = rec.dev_bytes;
        whole.p = nullptr; whole.n = 0;
        if (rec.host_bytes) {
            CG_HIP(hipHostMalloc((void**)&sl->arena->host, rec.host_bytes, hipHostMallocDefault));
            sl->arena->host_size = rec.host_bytes;
        }
    }
    ArenaScope carve(sl->arena.get(), rec.dev_bytes ? nullptr : &meter);
    // A throughput context runs every proof on ONE stream: with a dozen proofs in flight the overlap comes from the
    // other proofs, and twelve streams fit the hardware queues one each, where 60 share them (and anything above
    // ~24 user queues per process is time-sliced by the hardware scheduler in 15 ms quanta): 192 proofs/s on 16
    // queues against 188 with five streams per proof on 32 (profiles/r03_a_streams_and_queues.txt).  A latency
    // context (one proof at a time, or a shard of one) spreads its five MSMs and the witness map over five streams.
    // (CG_FLAG_THROUGHPUT_MODE with one slot is the profiling arrangement: a kernel trace of one proof at a time on one
    // stream shows stand-alone durations of the kernels the pipelined run launches.  Tuning builds: CG_SERIAL_STREAMS=1 / 0
    // decouples the stream count from the mode.)
    bool serial = !c->latency && !lone;
    if (const char* e = CG_TUNE_ENV("SERIAL_STREAMS")) serial = e[0] == '1' && !lone;
    // tuning builds, CG_CHAIN_PRIORITY=1 (experiment): the witness-map -> h-MSM chain, which sets a lone proof's latency,
    // on a high-priority stream
    const bool chain_prio = CG_TUNE_ENV("CHAIN_PRIORITY") && CG_TUNE_ENV("CHAIN_PRIORITY")[0] == '1';
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    for (int i = 0; i < 5; ++i) {
        if (serial && i) sl->st[i] = sl->st[0];
        else if ((size_t)i < borrow.size()) { sl->st[i] = borrow[i]; sl->st_borrowed[i] = true; }
        else if (chain_prio && i == 0) CG_HIP(hipStreamCreateWithPriority(&sl->st[i], hipStreamNonBlocking, prio_hi));
        else CG_HIP(hipStreamCreateWithFlags(&sl->st[i], hipStreamNonBlocking));
    }
    CG_HIP(hipEventCreateWithFlags(&sl->ev_w, hipEventDisableTiming));
    CG_HIP(hipEventCreateWithFlags(&sl->ev_b1, hipEventDisableTiming));
    CG_HIP(hipEventCreateWithFlags(&sl->ev_done, hipEventDisableTiming));
    for (auto& e : sl->ev_fin) CG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    sl->one_stream = serial;
    for (auto& e : sl->ev_t) CG_HIP(hipEventCreate(&e));
    const bool latency = c->latency || lone;
    sl->eh.latency_mode = sl->el.latency_mode = sl->ea.latency_mode = sl->eb1.latency_mode = sl->eb2.latency_mode = latency;
#ifdef CG_WITH_BATCH_AFFINE
    if (CG_TUNE_ENV("BA_H_ONLY")) sl->el.ba_allowed = sl->ea.ba_allowed = sl->eb1.ba_allowed = false;   // experiment switch
#endif
    if (serial) {
        sl->eh.shared_mem = sl->el.shared_mem = sl->ea.shared_mem = sl->eb1.shared_mem = sl->eb2.shared_mem = &sl->scratch;
        const bool zero_at_end = !(CG_TUNE_ENV("NO_ZERO_AT_END") && CG_TUNE_ENV("NO_ZERO_AT_END")[0] == '1');     // A/B aid (tuning builds)
        sl->eh.zero_at_end = sl->el.zero_at_end = sl->ea.zero_at_end = sl->eb1.zero_at_end = sl->eb2.zero_at_end = zero_at_end;
    }
    sl->eh.init(bh, zs); sl->el.init(bl, zs); sl->ea.init(ba, zs); sl->eb1.init(bb1, zs); sl->eb2.init(bb2, zs);
    if (c->external_q) {      // only the landing buffer of a slice that arrives in host memory, and the input flag
        sl->h_canon.alloc(c->rh.hi - c->rh.lo ? c->rh.hi - c->rh.lo : 1);
        sl->wm.h_bad_input.alloc(1);
    } else {
        sl->h_canon.alloc(c->D);
        sl->wm.alloc(c->M, c->D, std::max(c->A.sell_scratch, std::max(c->B.sell_scratch, c->C.sell_scratch)));
    }
    if (!rec.dev_bytes) { rec.dev_bytes = meter.dev_bytes + 4096; rec.host_bytes = meter.host_bytes + 4096; }
    return sl;
}

// The lone slots of a throughput context (cg_ctx::acquire).  A device with no room left for them does without: the context
// then proves every proof on a one-stream slot.
static void add_lone_slot(cg_ctx* c, std::vector<std::unique_ptr<ProofSlot>>& slots, const MsmBases<Fq>* bh, const MsmBases<Fq>* bl,
                          const MsmBases<Fq>* ba, const MsmBases<Fq>* bb1, const MsmBases<Fq2>* bb2, hipStream_t zs) {
    if (c->n_lone <= 0) return;
    try {
        SlotRecipe own;                       // its buffers are a latency slot's: measured, not the one-stream slots' recipe
        // A lone slot is in use when (next to) nothing else is: it runs on streams the one-stream slots own - the LAST slots',
        // which a proof takes only when all the others are busy - so the context has no more streams than it had, one per
        // hardware queue (cg_init).
        size_t next = slots.size();
        for (int k = 0; k < c->n_lone; ++k) {
            std::vector<hipStream_t> borrow;
            while (borrow.size() < 5 && next > 2 && !slots[next - 1]->lone) borrow.push_back(slots[--next]->st[0]);   // (never the first two slots')
            slots.push_back(make_slot(c, bh, bl, ba, bb1, bb2, zs, own, true, borrow));
        }
    } catch (const HipError& e) {
        if (e.code != CG_ERR_OUT_OF_MEMORY) throw;
        (void)hipGetLastError();
    }
}

// threads of a loader that are joined on every way out of its scope
struct JoinAll {
    std::vector<std::thread> th;
    ~JoinAll() { for (auto& t : th) if (t.joinable()) t.join(); }
};

// ---------------------------------------------------------------------------------------------
// staged load: the worker that builds the final arrangement behind the first proofs
// ---------------------------------------------------------------------------------------------
struct WorkerCancelled {};
static void staged_worker(cg_ctx* c) {
    const auto t0 = std::chrono::steady_clock::now();
    int status = 0;
    std::string err;
    bool done = false, from_proof = false;
    float ms_fold = 0.f, ms_tables = 0.f, ms_slots = 0.f, ms_wait = 0.f;
    try {
        CG_HIP(hipSetDevice(c->device));
        // The final arrangement under construction; after the swap these hold the WARM-UP tables and slots, which are
        // released when this scope ends - outside the gate.
        MsmBases<Fq> bh, bl, ba, bb1;
        MsmBases<Fq2> bb2;
        std::vector<std::unique_ptr<ProofSlot>> slots;
        int64_t table_bytes = 0;
        auto stop = [&] { if (c->cancel.load()) throw WorkerCancelled(); };
        const uint64_t D = c->D, M = c->M, l = c->l;
        const int wb = c->window_opt;
        TuneStats ts;
        bool all_from_proof = true;
        // the window of an assignment-driven query: from a finished warm-up proof's digit statistics when one is there
        // (msm_best_window, as the one-time re-tune of a synchronous load chooses it), else by size
        auto window_for = [&](uint64_t n, int which) {
            if (wb > 0) return wb;
            if (!ts.valid) {
                std::lock_guard<std::mutex> lk(c->warm_mu);
                ts = c->warm_stats;
            }
            // a proof with r = 0 skips b1 (prover.rs:102-112): same scalars and identity pattern as b2
            const TuneStats::Q& q = which == 0 ? ts.l : which == 1 ? ts.a : (which == 2 && ts.b1.n_scalars) ? ts.b1 : ts.b2;
            if (!ts.valid) all_from_proof = false;
            if (ts.valid && q.n_scalars && q.W0 > 0) {
                double nz = q.nonzero, nz_full = q.W0 > 1 ? (q.entries - nz) / (double)(q.W0 - 1) : 0.0;
                if (nz_full < 0) nz_full = 0;
                if (nz_full > nz) nz_full = nz;
                return msm_best_window(n ? n : 1, nz - nz_full, nz_full);
            }
            return msm_default_window(n ? n : 1, true);
        };
        // The FIRST proof goes first.  A host that loads and proves once (create_client_state) is waiting for exactly that
        // proof, and the change of basis below fills every wave slot of the chip for seconds: started at once it triples the
        // first proof's time (70-85 ms against ~25 alone at the rs256 size).  So the worker lets the first warm-up proof
        // finish - or a tenth of a second pass without one, for a host that loads now and proves later - and only then starts.
        for (int waited = 0; waited < 100 && c->warmup_proofs.load() == 0; ++waited) {
            stop();
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        {
            AllocScope booking(&table_bytes);
            ScopedStream st;
            // The h query first.  Its scalars are the quotient's values - uniform whatever the witness - so its window is the
            // size-based one; and its change of basis is the longest step: by the time it is done the first warm-up proof
            // has normally finished and the other four windows can be chosen from it.
            build_h_bases_folded(bh, c->bh.table.p, c->bh.valid.p, D - 1, c->logD, 0, 1, D, wb > 0 ? wb : msm_default_window(D, true), st,
                                 &ms_fold, &ms_tables);
            stop();
            // tuning builds, CG_FAULT_STAGED=1 (fault injection for tests/fault_retune_child.py): fail here as an allocation of the
            // final tables would.  The shipped library carries no such switch.
            if (const char* f = CG_TUNE_ENV("FAULT_STAGED")) if (f[0] == '1') throw HipError(CG_ERR_OUT_OF_MEMORY, "injected: out of device memory while building the final arrangement");
            build_l_bases_folded(bl, c->bh.table.p, c->bh.valid.p, D - 1, c->logD, c->bl.table.p, c->bl.valid.p, l, M, c->c_transposed, c->m,
                                 c->dom.vanishing_inv, 0, M,
                                 [&] { return window_for(M, 0); }, st, &ms_fold, &ms_tables);
            stop();
            const auto tt = std::chrono::steady_clock::now();
            ba.build_from_row0(c->ba.table.p, c->ba.valid.p, c->ba.n, window_for(c->ba.n, 1), st);
            CG_HIP(hipStreamSynchronize(st));
            stop();
            bb1.build_from_row0(c->bb1.table.p, c->bb1.valid.p, c->bb1.n, window_for(c->bb1.n, 2), st);
            CG_HIP(hipStreamSynchronize(st));
            stop();
            bb2.build_from_row0(c->bb2.table.p, c->bb2.valid.p, c->bb2.n, window_for(c->bb2.n, 3), st);
            CG_HIP(hipStreamSynchronize(st));
            ms_tables += ms_since(tt);
            from_proof = wb == 0 && ts.valid && all_from_proof;
        }
        stop();
        {
            const auto tt = std::chrono::steady_clock::now();
            ScopedStream zs;
            SlotRecipe rec;
            for (int k = 0; k < c->n_slots_final; ++k) {
                slots.push_back(make_slot(c, &bh, &bl, &ba, &bb1, &bb2, zs, rec));
                stop();
            }
            add_lone_slot(c, slots, &bh, &bl, &ba, &bb1, &bb2, zs);
            CG_HIP(hipStreamSynchronize(zs));
            ms_slots = ms_since(tt);
        }
        {
            const auto tw = std::chrono::steady_clock::now();
            std::unique_lock<TuneGate> lk(c->tune_mu);       // the proofs in flight drain; new ones wait for the swap
            ms_wait = ms_since(tw);
            std::swap(c->bh, bh); std::swap(c->bl, bl); std::swap(c->ba, ba); std::swap(c->bb1, bb1); std::swap(c->bb2, bb2);
            c->slots.swap(slots);
            for (auto& sl : c->slots) {       // the engines were cut for the tables where they were built
                sl->eh.bases = &c->bh; sl->el.bases = &c->bl; sl->ea.bases = &c->ba; sl->eb1.bases = &c->bb1; sl->eb2.bases = &c->bb2;
            }
            c->folded = true;
            c->rh = {0, D};
            c->rl = {0, M};
            c->table_bytes = table_bytes;
            c->tuned = from_proof;             // windows already chosen from a proof: no re-tune follows
            account_slot(c);
            c->warmup = false;
            {
                std::lock_guard<std::mutex> pl(c->pick_mu);
                c->pick_cv.notify_all();
            }
            // The warm-up slots and row-0 tables are released HERE, while the gate is held: hipFree waits for the whole
            // device, and with callers proving back to back each of the ~150 releases would wait for - and stall - the
            // pipeline in turn (9 s of them measured behind a sixteen-slot context under four callers).  With the gate held
            // nothing is in flight and they take microseconds each.
            slots.clear();
            bh = MsmBases<Fq>(); bl = MsmBases<Fq>(); ba = MsmBases<Fq>(); bb1 = MsmBases<Fq>(); bb2 = MsmBases<Fq2>();
            std::vector<uint64_t>().swap(c->c_transposed.ptr);
            std::vector<uint32_t>().swap(c->c_transposed.row);
            std::vector<uint8_t>().swap(c->c_transposed.coeff);
        }
        done = true;
    } catch (const WorkerCancelled&) {
        status = 0;
    } catch (...) {
        status = cg::translate_current_exception();
        err = last_error();
        (void)hipGetLastError();
    }
    {
        std::lock_guard<std::mutex> lk(c->ready_mu);
        c->lt.fold_ms = ms_fold; c->lt.window_tables_ms += ms_tables; c->lt.final_slots_ms = ms_slots; c->lt.swap_wait_ms = ms_wait;
        c->lt.background_ms = ms_since(t0);
        c->lt.ready_after_ms = ms_since(c->t_load0);
        c->lt.windows_from_proof = from_proof ? 1 : 0;
        c->ready = done;
        c->bg_status = status;
        c->bg_error = err;
    }
    c->ready_cv.notify_all();
}

extern "C" int cg_circuit_load(cg_ctx** out, const cg_proving_key* pk, const cg_csr abc[3], uint64_t num_inputs,
                               uint64_t num_constraints, uint64_t num_variables, const cg_options* opt) {
    if (!out || !pk || !abc) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (num_inputs == 0 || num_inputs > num_variables) return fail(CG_ERR_INVALID_ARGUMENT, "need 1 <= num_inputs <= num_variables");
    if (pk->coord_form != CG_FORM_CANONICAL && pk->coord_form != CG_FORM_MONTGOMERY) return fail(CG_ERR_INVALID_ARGUMENT, "bad coord_form");
    const uint64_t l = num_inputs, m = num_constraints, M = num_variables;
    const uint64_t dom_in = m + l;
    const int logD = ilog2_ceil(dom_in);
    if (logD > 28) return fail(CG_ERR_POLY_DEGREE_TOO_LARGE, "num_constraints + num_inputs = %llu exceeds 2^28", (unsigned long long)dom_in);
    const uint64_t D = 1ull << logD;
    // query lengths fixed by the generator (generator.rs:140,162,168,174-179,185)
    if (pk->a_len != M || pk->b_g1_len != M || pk->b_g2_len != M) return fail(CG_ERR_MALFORMED_KEY, "a/b query length must equal num_variables");
    if (pk->l_len != M - l) return fail(CG_ERR_MALFORMED_KEY, "l_query length must equal num_variables - num_inputs");
    if (pk->h_len != D - 1) return fail(CG_ERR_MALFORMED_KEY, "h_query length must equal domain_size - 1 = %llu", (unsigned long long)(D - 1));
    // options are checked before the GPU is touched
    const int shard_count = (opt && opt->shard_count > 1) ? opt->shard_count : 1;
    const int shard_rank = (opt && shard_count > 1) ? opt->shard_rank : 0;
    if (shard_rank < 0 || shard_rank >= shard_count) return fail(CG_ERR_INVALID_ARGUMENT, "shard_rank out of range");
    const int wb = opt ? opt->window_bits : 0;
    if (wb < 0 || wb == 1 || wb > 22) return fail(CG_ERR_INVALID_ARGUMENT, "window_bits must be 0 (automatic) or in [2, 22]");
    if (opt && opt->proof_slots < 0) return fail(CG_ERR_INVALID_ARGUMENT, "proof_slots must not be negative");
    if (opt && opt->hw_queues < 0) return fail(CG_ERR_INVALID_ARGUMENT, "hw_queues must not be negative");
    const int span_lo = opt ? (opt->shard_span & 0xffff) : 0, span_hi = opt ? ((opt->shard_span >> 16) & 0xffff) : 0;
    if (opt && opt->shard_span != 0 && (shard_count <= 1 || span_lo >= span_hi || span_hi > 10000))
        return fail(CG_ERR_INVALID_ARGUMENT, "shard_span needs a sharded context and 0 <= lo < hi <= 10000");
    constexpr int32_t KNOWN_FLAGS = CG_FLAG_H_COEFFICIENT_BASIS | CG_FLAG_LATENCY_MODE | CG_FLAG_THROUGHPUT_MODE | CG_FLAG_SPIN_WAIT |
                                    CG_FLAG_CONTIGUOUS_H_SHARDS | CG_FLAG_H_SCALARS_EXTERNAL | CG_FLAG_STAGED_LOAD | CG_FLAG_NO_LONE_SLOT;
    if (opt && (opt->flags & ~KNOWN_FLAGS)) return fail(CG_ERR_INVALID_ARGUMENT, "unknown bits in flags");
    if (opt && (opt->flags & CG_FLAG_H_SCALARS_EXTERNAL) && ((opt->flags & CG_FLAG_H_COEFFICIENT_BASIS) || shard_count <= 1))
        return fail(CG_ERR_INVALID_ARGUMENT, "flags: CG_FLAG_H_SCALARS_EXTERNAL needs a sharded context over the folded key");
    if (opt && (opt->flags & CG_FLAG_LATENCY_MODE) && (opt->flags & CG_FLAG_THROUGHPUT_MODE))
        return fail(CG_ERR_INVALID_ARGUMENT, "flags: CG_FLAG_LATENCY_MODE and CG_FLAG_THROUGHPUT_MODE are exclusive");
    // every pointer the structs carry is checked before anything is read through it (a half-filled struct from the
    // C or Rust side must come back as an error, not a fault)
    if (!pk->alpha_g1 || !pk->beta_g1 || !pk->delta_g1 || !pk->beta_g2 || !pk->delta_g2)
        return fail(CG_ERR_INVALID_ARGUMENT, "null key point (alpha_g1 / beta_g1 / delta_g1 / beta_g2 / delta_g2)");
    if ((pk->a_len && !pk->a_query) || (pk->b_g1_len && !pk->b_g1_query) || (pk->b_g2_len && !pk->b_g2_query) ||
        (pk->h_len && !pk->h_query) || (pk->l_len && !pk->l_query))
        return fail(CG_ERR_INVALID_ARGUMENT, "null query pointer with a non-zero length");
    for (int k = 0; k < 3; ++k)
        if (const char* why = csr_view_problem(abc[k])) return fail(CG_ERR_INVALID_ARGUMENT, "matrix %d: %s", k, why);
    try {
        const auto T0 = std::chrono::steady_clock::now();
        int dev = (opt && opt->device >= 0) ? opt->device : -1;
        if (dev < 0) CG_HIP(hipGetDevice(&dev));
        CG_HIP(hipSetDevice(dev));
        std::unique_ptr<cg_ctx> c(new cg_ctx());
        c->t_load0 = T0;
        c->device = dev;
        c->l = l; c->m = m; c->M = M; c->D = D; c->logD = logD;
        c->shard_count = shard_count;
        c->shard_rank = shard_rank;
        c->span_lo = span_lo; c->span_hi = span_hi;
        c->fixed_window = wb > 0;
        c->window_opt = wb;
        int n_slots = (opt && opt->proof_slots > 0) ? opt->proof_slots : 1;
        if (n_slots > 16) n_slots = 16;
        c->n_slots_final = n_slots;
        c->folded = !(opt && (opt->flags & CG_FLAG_H_COEFFICIENT_BASIS));
        c->external_q = opt && (opt->flags & CG_FLAG_H_SCALARS_EXTERNAL);
        // a staged load proves in the reference's arrangement first (see CG_FLAG_STAGED_LOAD); sharded contexts and contexts
        // that keep that arrangement for good load synchronously
        const bool staged = opt && (opt->flags & CG_FLAG_STAGED_LOAD) && c->folded && shard_count == 1;
        // the loader's stream: every load-time copy and kernel of THIS thread runs on it; destroyed on every way out of this
        // function - a load that runs out of device memory half-way gives back its stream with everything else (the
        // context's buffers are RAII members of `c`, the temporaries are scoped DevBufs)
        ScopedStream s0_guard;
        const hipStream_t s0 = s0_guard;
        const uint32_t form = pk->coord_form;
        // ---- the three matrices on a host thread of their own (validation, coefficient dictionary, sliced layout: host work
        // over 17 M terms at the rs256 size, every pass of it on all host threads - csr_host.hpp), next to the key on this
        // thread.  Every thread books what it leaves resident.
        struct MatJob { std::exception_ptr err; int64_t bytes = 0; float ms = 0.f; } mj[4];
        DevCsr* mats[3] = {&c->A, &c->B, &c->C};
        cg_ctx* cp = c.get();
        {
            JoinAll jobs;
            jobs.th.emplace_back([&] {
                try {
                    const auto t = std::chrono::steady_clock::now();
                    CG_HIP(hipSetDevice(dev));
                    AllocScope booking(&mj[0].bytes);
                    ScopedStream st;
                    // validates the CSR views (monotone row_ptr, column range, canonical coefficients); a context that
                    // never runs the witness map skips the sliced layout
                    for (int k = 0; k < 3; ++k) mats[k]->upload(abc[k], m, M, st, !cp->external_q);
                    mj[0].ms = ms_since(t);
                } catch (...) {
                    mj[0].err = std::current_exception();
                }
            });
            if (staged)      // C^T for the fold, made now: the worker must not read the caller's arrays after this call returns
                jobs.th.emplace_back([&] {
                    try {
                        if (abc[2].nnz && m) csr_transpose(abc[2], m, M, cp->c_transposed);
                        else { cp->c_transposed.ptr.assign(M + 1, 0); cp->c_transposed.view = cg_csr{cp->c_transposed.ptr.data(), nullptr, nullptr, 0}; }
                    } catch (...) {
                        mj[3].err = std::current_exception();
                    }
                });
            // ---- this thread: the single points, the domain, the five queries
            c->alpha_g1 = g1_import(pk->alpha_g1, form);
            c->beta_g1 = g1_import(pk->beta_g1, form);
            c->delta_g1 = g1_import(pk->delta_g1, form);
            c->beta_g2 = g2_import(pk->beta_g2, form);
            c->delta_g2 = g2_import(pk->delta_g2, form);
            c->a0 = g1_import(pk->a_query, form);            // query[0] of calculate_coeff (prover.rs:265)
            c->b1_0 = g1_import(pk->b_g1_query, form);
            c->b2_0 = g2_import(pk->b_g2_query, form);
            c->fb_delta_g1.build(c->delta_g1);
            c->fb_delta_g2.build(c->delta_g2);
            {
                const int logs = ilog2_ceil((uint64_t)c->shard_count);
                c->h_strided = c->folded && c->shard_count > 1 && (1 << logs) == c->shard_count && logD - logs >= 4 &&
                               !(opt && (opt->flags & CG_FLAG_CONTIGUOUS_H_SHARDS)) && span_hi == 0;     // a span is a contiguous range
            }
            const bool folded_now = c->folded && !staged;     // the arrangement this call leaves in force
            // this shard's part of a query of n entries: the rank-th of count equal parts, or the span it was given
            auto part_of = [&](uint64_t n) {
                return span_hi ? Range{n * (uint64_t)span_lo / 10000u, n * (uint64_t)span_hi / 10000u} : shard_range(n, c->shard_rank, c->shard_count);
            };
            c->rh = part_of(folded_now ? D : D - 1);
            if (c->h_strided) c->rh = {0, D / (uint64_t)c->shard_count};     // positions in the shard's own list of points
            c->rl = part_of(folded_now ? M : M - l);
            c->ra = part_of(M - 1);
            {
                const auto t = std::chrono::steady_clock::now();
                AllocScope booking(&c->matrix_bytes);
                c->dom.build(logD, true, s0);
                CG_HIP(hipStreamSynchronize(s0));
                c->lt.domain_ms = ms_since(t);
            }
            AllocScope booking(&c->table_bytes);
            if (staged) {
                // every query as its row-0 table: what the warm-up arrangement proves on and what the worker expands
                load_query<Fq>(c->bh, pk->h_query, form, 0, D - 1, 0, false, s0, &c->lt.key_copy_ms, &c->lt.key_copy_ms);
                load_query<Fq>(c->bl, pk->l_query, form, 0, M - l, 0, false, s0, &c->lt.key_copy_ms, &c->lt.key_copy_ms);
                load_query<Fq>(c->ba, pk->a_query, form, 1, M - 1, 0, false, s0, &c->lt.key_copy_ms, &c->lt.key_copy_ms);
                load_query<Fq>(c->bb1, pk->b_g1_query, form, 1, M - 1, 0, false, s0, &c->lt.key_copy_ms, &c->lt.key_copy_ms);
                load_query<Fq2>(c->bb2, pk->b_g2_query, form, 1, M - 1, 0, false, s0, &c->lt.key_copy_ms, &c->lt.key_copy_ms);
            } else {
                if (c->folded) {
                    // every shard transforms the whole queries (the DFT mixes all points) and keeps its own ranges of the results
                    const auto t = std::chrono::steady_clock::now();
                    DevBuf<G1Affine> th(D), tl(M - l ? M - l : 1);
                    import_bases<Fq>(pk->h_query, form, D - 1, th.p, s0);
                    import_bases<Fq>(pk->l_query, form, M - l, tl.p, s0);
                    c->lt.key_copy_ms += ms_since(t);
                    const uint64_t nh = c->rh.hi - c->rh.lo, nl = c->rl.hi - c->rl.lo;
                    build_hl_bases_folded(c->bh, c->bl, th.p, D - 1, logD, tl.p, l, M, abc[2], m, c->dom.vanishing_inv,
                                          c->h_strided ? (uint64_t)c->shard_rank : c->rh.lo, c->h_strided ? (uint64_t)c->shard_count : 1, nh,
                                          wb > 0 ? wb : msm_default_window(nh ? nh : 1, true), c->rl.lo, nl,
                                          wb > 0 ? wb : msm_default_window(nl ? nl : 1, true), s0, &c->lt.fold_ms, &c->lt.window_tables_ms);
                } else {
                    load_query<Fq>(c->bh, pk->h_query, form, c->rh.lo, c->rh.hi - c->rh.lo, wb, true, s0, &c->lt.key_copy_ms, &c->lt.window_tables_ms);
                    load_query<Fq>(c->bl, pk->l_query, form, c->rl.lo, c->rl.hi - c->rl.lo, wb, true, s0, &c->lt.key_copy_ms, &c->lt.window_tables_ms);
                }
                load_query<Fq>(c->ba, pk->a_query, form, 1 + c->ra.lo, c->ra.hi - c->ra.lo, wb, true, s0, &c->lt.key_copy_ms, &c->lt.window_tables_ms);    // query[1..] (prover.rs:266)
                load_query<Fq>(c->bb1, pk->b_g1_query, form, 1 + c->ra.lo, c->ra.hi - c->ra.lo, wb, true, s0, &c->lt.key_copy_ms, &c->lt.window_tables_ms);
                load_query<Fq2>(c->bb2, pk->b_g2_query, form, 1 + c->ra.lo, c->ra.hi - c->ra.lo, wb, true, s0, &c->lt.key_copy_ms, &c->lt.window_tables_ms);
            }
            {   // the G2 MSM may take over b1's grouped entries only if the two queries vanish together (generator.rs:162,168
                // makes them b_i(τ)·G1 and b_i(τ)·G2; a key from elsewhere is not trusted to)
                const uint64_t nb = c->bb1.n;
                c->b_same_identities = nb == c->bb2.n;
                if (c->b_same_identities && nb) {
                    std::vector<uint8_t> v1(nb), v2(nb);
                    CG_HIP(hipMemcpyAsync(v1.data(), c->bb1.valid.p, nb, hipMemcpyDeviceToHost, s0));
                    CG_HIP(hipMemcpyAsync(v2.data(), c->bb2.valid.p, nb, hipMemcpyDeviceToHost, s0));
                    CG_HIP(hipStreamSynchronize(s0));
                    c->b_same_identities = v1 == v2;
                }
            }
        }   // the matrix threads are joined here
        for (const MatJob& j : mj)
            if (j.err) std::rethrow_exception(j.err);
        for (int k = 0; k < 3; ++k) {
            c->matrix_bytes += mj[k].bytes;
            c->lt.matrices_ms = std::max(c->lt.matrices_ms, mj[k].ms);
        }
        {
            AllocScope booking(&c->matrix_bytes);
            if (!c->external_q) {
                c->wdom.build(c->dom, s0);
                if (c->h_strided) c->wstr.build(c->dom, ilog2_ceil((uint64_t)c->shard_count), c->shard_rank, s0);
                c->dA.build(c->A, s0); c->dB.build(c->B, s0); c->dC.build(c->C, s0);
            }
            CG_HIP(hipStreamSynchronize(s0));
            if (c->external_q) {      // the matrices were uploaded for their validation and for the load-time fold only
                c->A = DevCsr(); c->B = DevCsr(); c->C = DevCsr();
            }
            // the saturated-form tables were only the source of the packed ones
            c->dom.tw_fwd.release(); c->dom.tw_inv.release(); c->dom.coset_br.release(); c->dom.icoset_br.release();
            c->A.dict.release(); c->B.dict.release(); c->C.dict.release();
        }
        // a context that proves one proof at a time (whole, or its shard of one) is a latency job; several proofs in
        // flight - whole proofs, or this rank's shards of several proofs (distributed.ShardedProver, proofs_in_flight) -
        // are a throughput job
        c->latency = n_slots == 1;
        if (opt && (opt->flags & CG_FLAG_LATENCY_MODE)) c->latency = true;
        if (opt && (opt->flags & CG_FLAG_THROUGHPUT_MODE)) c->latency = false;
        c->spin_wait = opt && (opt->flags & CG_FLAG_SPIN_WAIT);
        if (const char* e = CG_TUNE_ENV("LATENCY_MODE")) c->latency = e[0] == '1';    // tuning builds: force either segment length
        const auto t_slots = std::chrono::steady_clock::now();
        // (slots are accounted by kind from their buffers: account_slot.)  A staged load starts with a few warm-up slots -
        // callers beyond them wait their turn, as with any context whose slots are all busy - and the worker makes the
        // final ones.
        const int n_now = staged ? std::min(n_slots, 4) : n_slots;
        if (staged) c->folded = false;                    // the arrangement in force until the swap
        SlotRecipe slot_recipe;
        for (int k = 0; k < n_now; ++k) c->slots.push_back(make_slot(c.get(), &c->bh, &c->bl, &c->ba, &c->bb1, &c->bb2, s0, slot_recipe));
        CG_HIP(hipStreamSynchronize(s0));
        // Four shared copy-only streams when the runtime's hardware queues hold them beside the proof streams one each
        // (GPU_MAX_HW_QUEUES is the HIP runtime's own variable; cg_init asks for 20); with fewer queues four shared streams
        // would only concentrate the blocking (measured -5 % on 16 queues), so every buffer keeps a stream of its own there.
        // (cg_options.hw_queues when the host states it; else the runtime's own variable - which cg_init set to 20 if the host
        // had not, effective only if HIP was not initialised before cg_init: a host that initialises HIP first exports the
        // variable itself or passes hw_queues)
        const char* hwq_env = getenv("GPU_MAX_HW_QUEUES");
        const int hwq = (opt && opt->hw_queues > 0) ? opt->hw_queues : (hwq_env ? atoi(hwq_env) : 4);
        const size_t n_copy_streams = hwq >= n_slots + 4 ? 4 : (size_t)n_slots + 2;
        for (int k = 0; k < n_slots + 2; ++k) {
            if (c->up_streams.size() < n_copy_streams) {
                hipStream_t us = nullptr;
                CG_HIP(hipStreamCreateWithFlags(&us, hipStreamNonBlocking));
                c->up_streams.push_back(us);
            }
            std::unique_ptr<Upload> u(new Upload());
            u->w.alloc(M);
            u->st = c->up_streams[(size_t)k % c->up_streams.size()];
            for (auto& e : u->ev) CG_HIP(hipEventCreate(&e));
            CG_HIP(hipEventCreateWithFlags(&u->ev_done, hipEventDisableTiming));
            c->uploads.push_back(std::move(u));
        }
        // two of them: with two proofs in flight both still gain (145 against 123-131 proofs/s at 2^21); a third buys nothing
        c->n_lone = (!c->latency && n_slots > 1 && !(opt && (opt->flags & CG_FLAG_NO_LONE_SLOT))) ? std::min(2, n_slots - 1) : 0;
        if (const char* e = CG_TUNE_ENV("LONE_SLOTS")) if (c->n_lone) c->n_lone = atoi(e);      // tuning builds (A/B aid)
        if (!staged) {                                    // (a staged load's warm-up slots are few and short-lived: the worker adds it)
            add_lone_slot(c.get(), c->slots, &c->bh, &c->bl, &c->ba, &c->bb1, &c->bb2, s0);
            CG_HIP(hipStreamSynchronize(s0));
        }
        c->lt.slots_ms = ms_since(t_slots);
        account_slot(c.get());
        c->lt.staged = staged ? 1 : 0;
        c->lt.total_ms = ms_since(T0);
        if (staged) {
            c->ready = false;
            c->warmup = true;
            c->worker = std::thread(staged_worker, c.get());
        }
        *out = c.release();
        return CG_OK;
    } catch (...) {
        return translate_exception();
    }
}

extern "C" int cg_ctx_get_load_timings(cg_ctx* ctx, cg_load_timings* out) {
    if (!ctx || !out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lk(ctx->ready_mu);
    *out = ctx->lt;
    out->ready = ctx->ready ? 1 : 0;
    out->warmup_proofs = ctx->warmup_proofs.load();
    out->background_status = ctx->bg_status;
    return CG_OK;
}

extern "C" int cg_ctx_wait_ready(cg_ctx* ctx, int32_t timeout_ms) {
    if (!ctx) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    std::unique_lock<std::mutex> lk(ctx->ready_mu);
    auto settled = [&] { return ctx->ready || ctx->bg_status != 0; };
    if (timeout_ms < 0) ctx->ready_cv.wait(lk, settled);
    else if (!ctx->ready_cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), settled)) return 1;
    if (ctx->bg_status != 0) return fail(ctx->bg_status, "the background part of the staged load failed (the context keeps proving in the warm-up arrangement): %s", ctx->bg_error.c_str());
    return CG_OK;
}

extern "C" void cg_circuit_free(cg_ctx* ctx) {
    if (!ctx) return;
    // a staged load's worker stops at its next step (between two table builds at the latest) and is waited for: it works on
    // the context's row-0 tables
    ctx->cancel = true;
    if (ctx->worker.joinable()) ctx->worker.join();
    {   // proofs hold tune_mu shared from slot acquisition to their last stream synchronisation: taking it exclusively
        // waits for the GPU part of every cg_prove* still inside this context ...
        std::unique_lock<TuneGate> drain(ctx->tune_mu);
    }
    // ... and the count of calls inside covers their tails, which run without the lock (the re-tune check takes it
    // exclusively itself; the host finish reads the context's fixed points).  A caller must not START a call after this
    // one, as with any handle.
    while (ctx->calls_inside.load(std::memory_order_acquire) > 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    delete ctx;
}

// ---------------------------------------------------------------------------------------------
// witness map on stream st: w_mont must be ready; result h (canonical, natural order) in c->h_canon
// (LibsnarkReduction::witness_map_from_matrices, r1cs_to_qap.rs:150-213)
// ---------------------------------------------------------------------------------------------
static void run_witness_map(cg_ctx* c, ProofSlot* S, const Fr* w_canon_dev, hipStream_t st, bool coset_values) {
    wm29_run(c->wdom, c->A, c->B, c->C, c->dA, c->dB, c->dC, S->wm, w_canon_dev, c->M, c->m, c->l, S->h_canon.p, st, coset_values,
             coset_values && c->h_strided ? &c->wstr : nullptr);
}

// x >= r for any of n scalars -> *bad = 1 (the witness map's own input check, for proofs that skip the witness map)
__global__ void __launch_bounds__(256) k_flag_non_canonical(const Fr* __restrict__ s, uint64_t n, uint32_t* __restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr x = s[i];
    bool lt = false, decided = false;
#pragma unroll
    for (int k = 7; k >= 0; --k)
        if (!decided && x.l[k] != FrP::N[k]) { lt = x.l[k] < FrP::N[k]; decided = true; }
    if (!lt) *bad = 1u;
}
// out[p·d + k] = in[p + k·count]: the coset values in shard-major order for strided shards (d = n / count)
__global__ void __launch_bounds__(256) k_shard_major(const Fr* __restrict__ in, Fr* __restrict__ out, uint64_t n, uint32_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t d = n / count, p = i / d, k = i - p * d;
    out[i] = in[p + k * (uint64_t)count];
}

// The scalars of this context's share of the h MSM, on stream st: out of the witness map (h_canon: coefficients of h, or the
// coset values vinv·a·b for a folded key), or - q_dev given - the caller's, with the input check the witness map would
// have made (canonical assignment) and the same check of the slice.
static const Fr* witness_map_or_check(cg_ctx* c, ProofSlot* S, const Fr* w_dev, const Fr* q_dev, hipStream_t st) {
    if (!q_dev) {
        run_witness_map(c, S, w_dev, st, c->folded);
        return S->h_canon.p + c->rh.lo;
    }
    S->wm.h_bad_input.p[0] = 0;
    const uint64_t nq = c->rh.hi - c->rh.lo;
    k_flag_non_canonical<<<ceil_div(c->M, 256), 256, 0, st>>>(w_dev, c->M, S->wm.h_bad_input.dev());
    if (nq) k_flag_non_canonical<<<ceil_div(nq, 256), 256, 0, st>>>(q_dev, nq, S->wm.h_bad_input.dev());
    CG_KERNEL_CHECK();
    return q_dev;
}

struct Partials {
    G1Affine h, l, a, b1;
    G2Affine b2;
};

static float ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

// How a calling thread waits for the GPU.  A throughput context has a dozen or more proofs in flight, one calling thread
// each, and every one of them waits ~80 ms for its proof: spinning in hipStreamSynchronize they would hold as many CPUs
// as there are proofs in flight (and a multi-rank host multiplies that by its ranks, against whatever CPU quota it has).
// Measured: 16.0 CPUs busy for 16 proofs in flight - the whole cgroup quota of the pool's hosts - and no different with
// hipEventBlockingSync events, which this runtime also waits for actively.  So the waiters POLL the event and sleep in
// between (hipEventQuery is a read of the completion signal): a 250 us nap costs a proof 0.3 % of its time in flight and
// nothing of the GPU's, which the other proofs keep busy (wait_sleeping: shorter naps early on, for small circuits).  A
// latency context (one proof at a time: the wait IS the latency) keeps the spinning synchronise.  CG_FLAG_SPIN_WAIT makes
// every caller of a context spin.
static bool spin_wait(const cg_ctx* c) { return c->spin_wait; }
// Polls `ev` until it is done: spinning for the first 200 us (a small circuit's proof is over by then), after that napping
// an eighth of the time already waited, at most `max_nap_us` - the overshoot stays below an eighth of the wait whatever
// the circuit's size, and a long wait costs next to no CPU.
static void wait_sleeping(hipEvent_t ev, unsigned max_nap_us) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) CG_HIP(e);
        const long long waited = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        if (waited < 200) { std::this_thread::yield(); continue; }
        long long nap = waited / 8;
        if (nap < 20) nap = 20;
        if (nap > (long long)max_nap_us) nap = max_nap_us;
        std::this_thread::sleep_for(std::chrono::microseconds(nap));
    }
}

// Waits for what THIS proof queued on its slot's streams - not for the streams themselves: a lone slot runs on streams that
// belong to one-stream slots, and a proof of theirs that queued behind this one (the context filled up meanwhile) would be
// waited for as well: +77 ms on the two proofs that open a burst, measured.  The calling thread spins in the runtime.
static void wait_for_slot(ProofSlot* S) {
    for (int i = 0; i < 5; ++i)
        if (i == 0 || S->st[i] != S->st[0]) CG_HIP(hipEventRecord(S->ev_fin[i], S->st[i]));
    for (int i = 0; i < 5; ++i)
        if (i == 0 || S->st[i] != S->st[0]) CG_HIP(hipEventSynchronize(S->ev_fin[i]));
}

// Host -> device copy of one assignment into `u`, on u's copy-only stream; THIS THREAD waits for it.  Every kernel of
// the proof needs the assignment, so the proof loses nothing, and the other proofs in flight (other threads) keep the GPU
// busy meanwhile.  Enqueued on the proof's own stream instead, the copy becomes a barrier packet in a hardware queue that
// several streams share and stalls kernels of OTHER proofs behind it: 171 proofs/s against 184-187 with the wait here
// (profiles/r03_a_host_witness.txt).  Page-locked source (cg_host_alloc / cg_host_register): one DMA at PCIe speed;
// pageable source: staged by the runtime through its own pinned buffers, inside the call.  Returns the copy's ms (timed).
static float upload_assignment(cg_ctx* c, Upload* u, const void* host_assignment, bool timed) {
    CG_HIP(hipSetDevice(c->device));
    if (timed) CG_HIP(hipEventRecord(u->ev[0], u->st));
    CG_HIP(hipMemcpyAsync(u->w.p, host_assignment, c->M * 32, hipMemcpyHostToDevice, u->st));
    if (timed) CG_HIP(hipEventRecord(u->ev[1], u->st));
    if (spin_wait(c) || c->latency) {
        CG_HIP(hipStreamSynchronize(u->st));
    } else {
        CG_HIP(hipEventRecord(u->ev_done, u->st));
        wait_sleeping(u->ev_done, 50);             // a 48 MB copy takes ~1 ms
    }
    return timed ? ev_ms(u->ev[0], u->ev[1]) : 0.f;
}

// w_dev: the assignment in this context's device memory (the caller's own buffer, or an Upload's)
// q_dev (optional): this shard's h scalars, supplied by the caller (cg_prove_partial_q) - the witness map is skipped
static int prove_partial_impl(cg_ctx* c, ProofSlot* S, const Fr* w_dev, bool skip_b1, Partials& P, cg_timings* tm,
                              const std::function<void()>* while_gpu_runs = nullptr, const Fr* q_dev = nullptr) {
    CG_HIP(hipSetDevice(c->device));
    auto t0 = std::chrono::steady_clock::now();
    const uint64_t M = c->M, l = c->l;
    (void)M;
    hipStream_t s0 = S->st[0];
    // assignment-driven MSMs: operands (prover.rs:70-74, 84-89, 265-266)
    //   l: l_query[i] x w[l + i];  a, b1, b2: query[1 + i] x w[1 + i]
    const Fr* w_l = w_dev + (c->folded ? 0 : l) + c->rl.lo;          // folded l query: one base per wire
    const uint64_t n_l = c->rl.hi - c->rl.lo, n_a = c->ra.hi - c->ra.lo;
    const Fr* w_a = w_dev + 1 + c->ra.lo;
    // b1 and b2 take the same scalars against bases that vanish together: with equal windows the grouped entry list of
    // one IS the other's, so the G2 MSM skips its own grouping (five launches, ~0.9 % of a proof's instructions)
    static const bool no_share = CG_TUNE_ENV("NO_SHARE_B") != nullptr;        // A/B aid (tuning builds)
    // tuning builds: KNOCK is a mask of parts of a proof to leave out (1 l, 2 a, 4 b1, 8 b2, 16 the witness map after a slot's
    // first proof, 32 h) - the proof is wrong, the time is what the rest costs in the pipeline (profiles/r05_w_knock_outs.md)
    static const int knock = [] { const char* e = CG_TUNE_ENV("KNOCK"); return e ? atoi(e) : 0; }();
    if (knock & 4) skip_b1 = true;
    if (knock && tm) { memset(tm, 0, sizeof(*tm)); tm = nullptr; }       // an engine left out has no events to read
    const bool b2_adopts = !skip_b1 && !no_share && c->b_same_identities && S->eb2.can_adopt(S->eb1) && n_a > 0 && !(knock & 8);
    if (knock && S->one_stream) {
        if (!(knock & 1)) { S->el.digits(w_l, n_l, s0); S->el.accumulate(s0); }
        if (!(knock & 2)) { S->ea.digits(w_a, n_a, s0); S->ea.accumulate(s0); }
        if (!skip_b1) S->eb1.digits(w_a, n_a, s0);
        if (b2_adopts) S->eb2.adopt(S->eb1.grouped(), S->eb1.counters.p, n_a, s0);
        if (!skip_b1) S->eb1.accumulate(s0);
        if (!(knock & 8)) { if (!b2_adopts) S->eb2.digits(w_a, n_a, s0); S->eb2.accumulate(s0); }
        if (tm) CG_HIP(hipEventRecord(S->ev_t[0], s0));
        const Fr* h_scalars = (knock & 16) && S->knock_h ? S->knock_h : witness_map_or_check(c, S, w_dev, q_dev, s0);
        S->knock_h = h_scalars;
        if (tm) CG_HIP(hipEventRecord(S->ev_t[1], s0));
        if (!(knock & 32)) { S->eh.digits(h_scalars, c->rh.hi - c->rh.lo, s0); S->eh.accumulate(s0); }
    } else if (S->one_stream) {
        // everything on one stream, every MSM grouped and accumulated before the next one starts: the engines share the
        // slot's scratch (entry lists, segment pieces), which is what a slot's memory mostly is
        S->el.digits(w_l, n_l, s0); S->el.accumulate(s0);
        S->ea.digits(w_a, n_a, s0); S->ea.accumulate(s0);
        if (!skip_b1) S->eb1.digits(w_a, n_a, s0);
        if (b2_adopts) S->eb2.adopt(S->eb1.grouped(), S->eb1.counters.p, n_a, s0);    // b1's list is still in the scratch ...
        if (!skip_b1) S->eb1.accumulate(s0);
        if (!b2_adopts) S->eb2.digits(w_a, n_a, s0);
        S->eb2.accumulate(s0);                                                         // ... until here
        if (tm) CG_HIP(hipEventRecord(S->ev_t[0], s0));
        const Fr* h_scalars = witness_map_or_check(c, S, w_dev, q_dev, s0);
        if (tm) CG_HIP(hipEventRecord(S->ev_t[1], s0));
        S->eh.digits(h_scalars, c->rh.hi - c->rh.lo, s0);
        S->eh.accumulate(s0);
    } else {
        CG_HIP(hipEventRecord(S->ev_w, s0));
        for (int i = 1; i < 5; ++i) CG_HIP(hipStreamWaitEvent(S->st[i], S->ev_w, 0));
        S->el.digits(w_l, n_l, S->st[1]);
        S->ea.digits(w_a, n_a, S->st[2]);
        if (!skip_b1) S->eb1.digits(w_a, n_a, S->st[3]);
        if (b2_adopts) {
            CG_HIP(hipEventRecord(S->ev_b1, S->st[3]));
            CG_HIP(hipStreamWaitEvent(S->st[4], S->ev_b1, 0));
            S->eb2.adopt(S->eb1.grouped(), S->eb1.counters.p, n_a, S->st[4]);
        } else {
            S->eb2.digits(w_a, n_a, S->st[4]);
        }
        // witness map, then h digits, on stream 0
        if (tm) CG_HIP(hipEventRecord(S->ev_t[0], s0));
        const Fr* h_scalars = witness_map_or_check(c, S, w_dev, q_dev, s0);
        if (tm) CG_HIP(hipEventRecord(S->ev_t[1], s0));
        S->eh.digits(h_scalars, c->rh.hi - c->rh.lo, s0);
        // second phase (each waits for its own entry count)
        S->el.accumulate(S->st[1]);
        S->ea.accumulate(S->st[2]);
        if (!skip_b1) S->eb1.accumulate(S->st[3]);
        S->eb2.accumulate(S->st[4]);
        S->eh.accumulate(s0);
    }
    if (while_gpu_runs) (*while_gpu_runs)();      // host work that needs no MSM value
    if (S->one_stream && !spin_wait(c)) {
        CG_HIP(hipEventRecord(S->ev_done, s0));
        wait_sleeping(S->ev_done, 250);            // a proof with fifteen others in flight takes ~80 ms
    } else {
        wait_for_slot(S);
    }
    if (S->wm.h_bad_input.p[0])
        return fail(CG_ERR_INVALID_ARGUMENT, q_dev ? "full_assignment or the h-scalar slice holds a value >= the scalar field modulus"
                                                   : "full_assignment holds a value >= the scalar field modulus");
    P.h = to_affine(S->eh.value());
    P.l = to_affine(S->el.value());
    P.a = to_affine(S->ea.value());
    P.b1 = skip_b1 ? G1Affine::inf() : to_affine(S->eb1.value());
    P.b2 = to_affine(S->eb2.value());
    if (tm) {
        memset(tm, 0, sizeof(*tm));
        tm->witness_map_ms = ev_ms(S->ev_t[0], S->ev_t[1]);
        tm->msm_h_ms = S->eh.ms_total();
        tm->msm_l_ms = S->el.ms_total();
        tm->msm_a_ms = S->ea.ms_total();
        tm->msm_b1_ms = skip_b1 ? 0.f : S->eb1.ms_total();
        tm->msm_b2_ms = S->eb2.ms_total();
        tm->accum_g1_ms = S->eh.ms_accum() + S->el.ms_accum() + S->ea.ms_accum() + (skip_b1 ? 0.f : S->eb1.ms_accum());
        tm->accum_g2_ms = S->eb2.ms_accum();
        tm->sort_ms = S->eh.ms_sort() + S->el.ms_sort() + S->ea.ms_sort() + (skip_b1 ? 0.f : S->eb1.ms_sort()) + S->eb2.ms_sort();
        tm->entries_g1 = (uint64_t)S->eh.n_entries() + S->el.n_entries() + S->ea.n_entries() + (skip_b1 ? 0 : S->eb1.n_entries());
        tm->entries_g2 = S->eb2.n_entries();
        tm->accum_g1_launches = (S->eh.n_entries() != 0) + (S->el.n_entries() != 0) + (S->ea.n_entries() != 0) + (!skip_b1 && S->eb1.n_entries() != 0);
        tm->accum_g2_launches = S->eb2.n_entries() != 0;
        tm->msm_g1_pairs = S->eh.n_scalars + S->el.n_scalars + S->ea.n_scalars + (skip_b1 ? 0 : S->eb1.n_scalars);
        tm->msm_g2_pairs = S->eb2.n_scalars;
        tm->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return CG_OK;
}

// The scalar multiples of delta that do not depend on the MSM values (prover.rs:76-80,94,104,116): fixed-base, and
// computed while the GPU is still working on the proof.
struct DeltaMultiples {
    G1XYZZ r_g1, rs_delta, s_g1;
    G2XYZZ s_g2;
};
static DeltaMultiples delta_multiples(const cg_ctx* c, const uint8_t r[32], const uint8_t s[32]) {
    DeltaMultiples d;
    d.r_g1 = c->fb_delta_g1.mul(r);                              // :94
    // r_s_delta_g1 = (delta_g1 * r) * s = (r·s mod n)·delta_g1   (:76-80)
    uint8_t rs[32];
    fp_to_bytes(from_mont(mul(to_mont(fp_from_bytes<Fr>(r)), to_mont(fp_from_bytes<Fr>(s)))), rs);
    d.rs_delta = c->fb_delta_g1.mul(rs);
    d.s_g1 = c->fb_delta_g1.mul(s);                              // :104
    d.s_g2 = c->fb_delta_g2.mul(s);                              // :116
    return d;
}

// prover.rs:76-135 with the MSM values given
static void assemble_impl(const cg_ctx* c, const Partials& S, const uint8_t r[32], const uint8_t s[32], uint8_t proof_out[256],
                          const DeltaMultiples* pre = nullptr) {
    const bool r_zero = scalar_is_zero(r);
    DeltaMultiples local;
    if (!pre) { local = delta_multiples(c, r, s); pre = &local; }
    const G1XYZZ& r_g1 = pre->r_g1;
    const G1XYZZ& rs_delta = pre->rs_delta;
    // A = r*delta + a_query[0] + msm_a + alpha   (:96, 256-274)
    G1XYZZ g_a = r_g1;
    madd(g_a, c->a0);
    madd(g_a, S.a);
    madd(g_a, c->alpha_g1);
    G1XYZZ s_g_a = scalar_mul_bytes(g_a, s);                   // :98
    // B in G1 (:102-112)
    G1XYZZ g1_b = G1XYZZ::inf();
    if (!r_zero) {
        g1_b = pre->s_g1;
        madd(g1_b, c->b1_0);
        madd(g1_b, S.b1);
        madd(g1_b, c->beta_g1);
    }
    // B in G2 (:116-117)
    G2XYZZ g2_b = pre->s_g2;
    madd(g2_b, c->b2_0);
    madd(g2_b, S.b2);
    madd(g2_b, c->beta_g2);
    G1XYZZ r_g1_b = scalar_mul_bytes(g1_b, r);                 // :118
    // C = s*A + r*B1 - r*s*delta + l_aux_acc + h_acc (:123-128)
    G1XYZZ g_c = s_g_a;
    add(g_c, r_g1_b);
    add(g_c, neg(rs_delta));
    madd(g_c, S.l);
    madd(g_c, S.h);
    g1_serialize_uncompressed(to_affine(g_a), proof_out);          // Proof { a, b, c } (:131-135; data_structures.rs:7-14)
    g2_serialize_uncompressed(to_affine(g2_b), proof_out + 64);
    g1_serialize_uncompressed(to_affine(g_c), proof_out + 192);
}

static void partials_to_bytes(const Partials& P, uint8_t out[384]) {
    g1_export_canonical(P.h, out);
    g1_export_canonical(P.l, out + 64);
    g1_export_canonical(P.a, out + 128);
    g1_export_canonical(P.b1, out + 192);
    g2_export_canonical(P.b2, out + 256);
}

// One-time window re-tuning from the digit statistics of a finished proof.  The statistics are copied out of the
// proof's slot while the proof still owns it; whether they are worth a re-tune is decided WITHOUT the exclusive lock, so a
// stream of unrepresentative assignments never drains the pipeline, and after RETUNE_MAX_ATTEMPTS looks the context keeps
// its size-based windows for good.
static constexpr int RETUNE_MAX_ATTEMPTS = 8;
template <class F>
static TuneStats::Q tune_stats_of(const MsmEngine<F>& e) {
    TuneStats::Q q;
    q.n_scalars = e.n_scalars; q.nonzero = e.n_nonzero(); q.entries = e.n_entries(); q.W0 = e.bases->W;
    return q;
}
static void snapshot_tune_stats(const cg_ctx* c, const ProofSlot* S, bool skip_b1, TuneStats& ts) {
    if (c->fixed_window || c->tuned || c->retune_attempts >= RETUNE_MAX_ATTEMPTS) return;
    ts.l = tune_stats_of(S->el); ts.a = tune_stats_of(S->ea); ts.b2 = tune_stats_of(S->eb2);
    if (!skip_b1) ts.b1 = tune_stats_of(S->eb1);
    ts.valid = true;
}
// a query's table changed size: every slot's engine for it is re-sized (the first slot's change is booked)
template <class F>
static void reinit_engines(cg_ctx* c, MsmEngine<F> ProofSlot::*eng, const MsmBases<F>& bases) {
    try {
        // tuning builds, CG_FAULT_RETUNE=1 (fault injection for tests/test_gpu_host_and_ranks.py): fail here as an allocation
        // would, table rebuilt and engines not yet re-sized.  The shipped library carries no such switch.
        if (const char* f = CG_TUNE_ENV("FAULT_RETUNE")) if (f[0] == '1') throw HipError(CG_ERR_OUT_OF_MEMORY, "injected: out of device memory while re-sizing the proof slots");
        for (size_t k = 0; k < c->slots.size(); ++k) ((*c->slots[k]).*eng).init(&bases);
    } catch (...) {
        c->broken = true;     // the table is already the new one: engines and table no longer agree
        throw;
    }
}
template <class F>
static int rebuild_booked(cg_ctx* c, MsmBases<F>& bases, int window, hipStream_t st) {
    AllocScope booking(&c->table_bytes);
    return bases.rebuild(window, st);
}
template <class F>
static void retune_query(cg_ctx* c, MsmBases<F>& bases, MsmEngine<F> ProofSlot::*eng, const TuneStats::Q& q, hipStream_t st) {
    if (!q.n_scalars || !bases.n) return;
    const int W0 = q.W0 > 0 ? q.W0 : bases.W;      // the window count of the tables the statistics were taken on
    double nz = q.nonzero, N = q.entries;
    double nz_full = W0 > 1 ? (N - nz) / (double)(W0 - 1) : 0.0;
    if (nz_full < 0) nz_full = 0;
    if (nz_full > nz) nz_full = nz;
    const int best = msm_best_window(bases.n, nz - nz_full, nz_full);
    if (best == bases.c) return;
    const int rc = rebuild_booked(c, bases, best, st);
    if (rc < 0) { c->retune_skipped_memory++; return; }
    if (rc > 0) reinit_engines(c, eng, bases);
}
static void maybe_retune(cg_ctx* c, const TuneStats& ts) {
    if (c->warmup.load()) {
        // a staged load's warm-up arrangement: nothing to re-tune (its tables are row 0 only).  The first representative
        // proof's statistics go to the worker, which chooses the FINAL windows from them before it expands those tables.
        c->warmup_proofs++;
        if (ts.valid && ts.a.nonzero * 64 >= (double)ts.a.n_scalars) {
            std::lock_guard<std::mutex> lk(c->warm_mu);
            if (!c->warm_stats.valid) c->warm_stats = ts;
        }
        return;
    }
    if (!ts.valid || c->fixed_window || c->tuned) return;
    if (!ts.l.n_scalars && !ts.a.n_scalars) return;
    // A degenerate assignment (all zero, or next to it) says nothing about the proofs to come: its statistics would pick
    // the narrowest window and the widest tables for good.  Keep the size-based windows and wait for a representative
    // proof - a bounded number of times.
    if (ts.a.nonzero * 64 < (double)ts.a.n_scalars) { c->retune_attempts++; return; }
    std::unique_lock<TuneGate> lk(c->tune_mu);     // waits for the proofs in flight to drain
    if (c->tuned) return;
    c->retune_attempts++;
    try {
        CG_HIP(hipSetDevice(c->device));
        hipStream_t st = c->slots[0]->st[0];
        retune_query<Fq>(c, c->bl, &ProofSlot::el, ts.l, st);
        retune_query<Fq>(c, c->ba, &ProofSlot::ea, ts.a, st);
        if (ts.b1.n_scalars) retune_query<Fq>(c, c->bb1, &ProofSlot::eb1, ts.b1, st);
        retune_query<Fq2>(c, c->bb2, &ProofSlot::eb2, ts.b2, st);
        if (!ts.b1.n_scalars && c->bb1.c != c->bb2.c) {         // b1 was skipped (r = 0): same scalars and identity pattern as b2
            const int rc = rebuild_booked(c, c->bb1, c->bb2.c, st);
            if (rc < 0) c->retune_skipped_memory++;
            if (rc > 0) reinit_engines(c, &ProofSlot::eb1, c->bb1);
        }
        c->tuned = true;
        account_slot(c);
    } catch (...) {
        account_slot(c);
        // the proof this call belongs to is already computed: a failed re-tune (out of memory beside another tenant of the
        // GPU, say) leaves the size-based windows in force and is retried by a later proof, never reported as that
        // proof's failure
        (void)hipGetLastError();
    }
}

static const char* const BROKEN_CONTEXT = "a window re-tune ran out of device memory while re-sizing the proof slots: free this context and load the circuit again";
static int check_rs(const uint8_t r[32], const uint8_t s[32]) {
    if (!r || !s) return fail(CG_ERR_INVALID_ARGUMENT, "null r/s");
    if (!scalar_is_canonical(r) || !scalar_is_canonical(s)) return fail(CG_ERR_INVALID_ARGUMENT, "r/s not canonical (>= field modulus)");
    return CG_OK;
}

static int prove_common(cg_ctx* ctx, const void* assignment, bool on_device, const uint8_t r[32], const uint8_t s[32],
                        uint8_t proof_out[256], cg_timings* tm) {
    if (!ctx || !assignment || !proof_out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    if (int e = check_rs(r, s)) return e;
    if (ctx->shard_count != 1) return fail(CG_ERR_INVALID_ARGUMENT, "context is a shard; use cg_prove_partial + cg_assemble");
    if (ctx->broken) return fail(CG_ERR_OUT_OF_MEMORY, "%s", BROKEN_CONTEXT);
    CallGuard inside(ctx);
    try {
        Partials P;
        DeltaMultiples pre;
        TuneStats ts;
        const std::function<void()> overlap = [&]() { pre = delta_multiples(ctx, r, s); };
        int e;
        {
            std::shared_lock<TuneGate> tl(ctx->tune_mu);
            // a caller that passed the check above can have waited here for another thread's re-tune, and that re-tune can
            // have failed half-way: the engines of the slots are then cut for the old window against the rebuilt table
            if (ctx->broken) return fail(CG_ERR_OUT_OF_MEMORY, "%s", BROKEN_CONTEXT);
            UploadGuard up;                       // declared before the slot: released after it
            float upload_ms = 0.f;
            const Fr* w_dev = (const Fr*)assignment;
            if (!on_device) {
                up.take(ctx);
                upload_ms = upload_assignment(ctx, up.u, assignment, tm != nullptr);
                w_dev = up.u->w.p;
            }
            // (a TIMED proof stays on a one-stream slot: its phases are then stand-alone durations that add up)
            SlotGuard g(ctx, tm == nullptr);
            e = prove_partial_impl(ctx, g.s, w_dev, scalar_is_zero(r), P, tm, &overlap);
            if (!e) snapshot_tune_stats(ctx, g.s, scalar_is_zero(r), ts);
            if (!e && tm) { tm->upload_ms = upload_ms; tm->total_ms += upload_ms; }
        }
        if (e) return e;
        maybe_retune(ctx, ts);
        auto t0 = std::chrono::steady_clock::now();
        assemble_impl(ctx, P, r, s, proof_out, &pre);
        if (tm) {
            tm->finish_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            tm->total_ms += tm->finish_ms;
        }
        return CG_OK;
    } catch (...) {
        return translate_exception();
    }
}

extern "C" int cg_prove(cg_ctx* ctx, const uint8_t* full_assignment, const uint8_t r[32], const uint8_t s[32],
                        uint8_t proof_out[256], cg_timings* timings) {
    return prove_common(ctx, full_assignment, false, r, s, proof_out, timings);
}
extern "C" int cg_prove_dev(cg_ctx* ctx, const void* d_full_assignment, const uint8_t r[32], const uint8_t s[32],
                            uint8_t proof_out[256], cg_timings* timings) {
    return prove_common(ctx, d_full_assignment, true, r, s, proof_out, timings);
}

static int prove_partial_common(cg_ctx* ctx, const void* full_assignment, int assignment_on_device, const void* q_slice, int q_on_device,
                                bool with_q, const uint8_t r[32], uint8_t out_partials[384], cg_timings* timings) {
    if (!ctx || !full_assignment || !out_partials || !r || (with_q && !q_slice)) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    if (!scalar_is_canonical(r)) return fail(CG_ERR_INVALID_ARGUMENT, "r not canonical");
    if (with_q && (!ctx->folded || ctx->shard_count <= 1))
        return fail(CG_ERR_INVALID_ARGUMENT, "cg_prove_partial_q needs a sharded context over the folded key");
    if (!with_q && ctx->external_q)
        return fail(CG_ERR_INVALID_ARGUMENT, "context loaded with CG_FLAG_H_SCALARS_EXTERNAL: its h scalars must be supplied (cg_prove_partial_q)");
    if (ctx->broken) return fail(CG_ERR_OUT_OF_MEMORY, "%s", BROKEN_CONTEXT);
    CallGuard inside(ctx);
    try {
        Partials P;
        TuneStats ts;
        int e;
        {
            std::shared_lock<TuneGate> tl(ctx->tune_mu);
            if (ctx->broken) return fail(CG_ERR_OUT_OF_MEMORY, "%s", BROKEN_CONTEXT);     // as in prove_common
            UploadGuard up;
            float upload_ms = 0.f;
            const Fr* w_dev = (const Fr*)full_assignment;
            if (!assignment_on_device) {
                up.take(ctx);
                upload_ms = upload_assignment(ctx, up.u, full_assignment, timings != nullptr);
                w_dev = up.u->w.p;
            }
            SlotGuard g(ctx, timings == nullptr);        // (as prove_common: a shard of a proof that arrives alone, untimed)
            const Fr* q_dev = nullptr;
            if (with_q) {
                q_dev = (const Fr*)q_slice;
                const uint64_t nq = ctx->rh.hi - ctx->rh.lo;
                if (!q_on_device && nq) {        // lands in the slot's h vector, in front of the proof's kernels on its stream
                    CG_HIP(hipSetDevice(ctx->device));
                    Fr* dst = ctx->external_q ? g.s->h_canon.p : g.s->h_canon.p + ctx->rh.lo;
                    CG_HIP(hipMemcpyAsync(dst, q_slice, nq * 32, hipMemcpyHostToDevice, g.s->st[0]));
                    q_dev = dst;
                }
            }
            e = prove_partial_impl(ctx, g.s, w_dev, scalar_is_zero(r), P, timings, nullptr, q_dev);
            if (!e) snapshot_tune_stats(ctx, g.s, scalar_is_zero(r), ts);
            if (!e && timings) { timings->upload_ms = upload_ms; timings->total_ms += upload_ms; }
        }
        if (e) return e;
        maybe_retune(ctx, ts);
        partials_to_bytes(P, out_partials);
        return CG_OK;
    } catch (...) {
        return translate_exception();
    }
}

extern "C" int cg_prove_partial(cg_ctx* ctx, const void* full_assignment, int assignment_on_device, const uint8_t r[32],
                                uint8_t out_partials[384], cg_timings* timings) {
    return prove_partial_common(ctx, full_assignment, assignment_on_device, nullptr, 0, false, r, out_partials, timings);
}

extern "C" int cg_prove_partial_q(cg_ctx* ctx, const void* full_assignment, int assignment_on_device, const void* q_slice, int q_on_device,
                                  const uint8_t r[32], uint8_t out_partials[384], cg_timings* timings) {
    return prove_partial_common(ctx, full_assignment, assignment_on_device, q_slice, q_on_device, true, r, out_partials, timings);
}

// ---------------------------------------------------------------------------------------------
// a sharded proof in two calls (cg_prove_partial_q_begin / _finish): the assignment-driven MSMs run while the h scalars are
// still on their way
// ---------------------------------------------------------------------------------------------
struct cg_partial {
    cg_ctx* c = nullptr;
    ProofSlot* S = nullptr;
    Upload* up = nullptr;
    const Fr* w_dev = nullptr;
    bool skip_b1 = false;
    bool gate_held = false;
    std::chrono::steady_clock::time_point t0;
    // everything the open proof holds is given back exactly once, whichever call ends it
    void close() {
        if (S) {
            for (auto st : S->st) if (st) (void)hipStreamSynchronize(st);
            c->release(S);
            S = nullptr;
        }
        if (up) {
            (void)hipStreamSynchronize(up->st);
            c->release_upload(up);
            up = nullptr;
        }
        if (gate_held) { c->tune_mu.unlock_shared(); gate_held = false; }
        if (c) { c->calls_inside.fetch_sub(1, std::memory_order_acq_rel); c = nullptr; }
    }
};

// all coset values (half = 0) or one side of them (1: vinv·a, 2: b) for `w_dev`, on the slot's stream 0, shard-major for strided
// shards, to host or device memory; waits for them.  -> CG_OK or CG_ERR_INVALID_ARGUMENT (a non-canonical assignment element)
static int coset_values_to(cg_ctx* ctx, ProofSlot* S, const Fr* w_dev, int half, void* q_out, int q_on_device) {
    hipStream_t s0 = S->st[0];
    wm29_run(ctx->wdom, ctx->A, ctx->B, ctx->C, ctx->dA, ctx->dB, ctx->dC, S->wm, w_dev, ctx->M, ctx->m, ctx->l, S->h_canon.p, s0, true, nullptr, half);
    const Fr* src = S->h_canon.p;
    if (ctx->h_strided) {        // shard-major: shard p's scalars q_{p + k·count} become the contiguous slice p
        Fr* tmp = reinterpret_cast<Fr*>(half == 1 ? S->wm.vb.p : S->wm.va.p);      // a vector the half just computed does not use
        if (!half) tmp = reinterpret_cast<Fr*>(S->wm.vt.p);
        k_shard_major<<<ceil_div(ctx->D, 256), 256, 0, s0>>>(S->h_canon.p, tmp, ctx->D, (uint32_t)ctx->shard_count);
        CG_KERNEL_CHECK();
        src = tmp;
    }
    CG_HIP(hipMemcpyAsync(q_out, src, ctx->D * 32, q_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s0));
    CG_HIP(hipStreamSynchronize(s0));
    if (S->wm.h_bad_input.p[0]) return fail(CG_ERR_INVALID_ARGUMENT, "full_assignment holds a value >= the scalar field modulus");
    return CG_OK;
}

extern "C" int cg_prove_partial_q_begin(cg_ctx* ctx, const void* full_assignment, int assignment_on_device, const uint8_t r[32], cg_partial** out) {
    if (!ctx || !full_assignment || !r || !out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (!scalar_is_canonical(r)) return fail(CG_ERR_INVALID_ARGUMENT, "r not canonical");
    if (!ctx->folded || ctx->shard_count <= 1) return fail(CG_ERR_INVALID_ARGUMENT, "cg_prove_partial_q_begin needs a sharded context over the folded key");
    if (ctx->broken) return fail(CG_ERR_OUT_OF_MEMORY, "%s", BROKEN_CONTEXT);
    std::unique_ptr<cg_partial> p(new cg_partial());
    p->c = ctx;
    ctx->calls_inside.fetch_add(1, std::memory_order_acq_rel);
    try {
        ctx->tune_mu.lock_shared_passing_waiting_writers();
        p->gate_held = true;
        if (ctx->broken) { p->close(); return fail(CG_ERR_OUT_OF_MEMORY, "%s", BROKEN_CONTEXT); }
        CG_HIP(hipSetDevice(ctx->device));
        p->t0 = std::chrono::steady_clock::now();
        p->w_dev = (const Fr*)full_assignment;
        if (!assignment_on_device) {
            p->up = ctx->acquire_upload();
            (void)upload_assignment(ctx, p->up, full_assignment, false);
            p->w_dev = p->up->w.p;
        }
        p->S = ctx->acquire(true);
        p->skip_b1 = scalar_is_zero(r);
        ProofSlot* S = p->S;
        cg_ctx* c = ctx;
        const Fr* w_l = p->w_dev + c->rl.lo;                      // folded l query: one base per wire
        const uint64_t n_l = c->rl.hi - c->rl.lo, n_a = c->ra.hi - c->ra.lo;
        const Fr* w_a = p->w_dev + 1 + c->ra.lo;
        const bool b2_adopts = !p->skip_b1 && c->b_same_identities && S->eb2.can_adopt(S->eb1) && n_a > 0;
        hipStream_t s0 = S->st[0];
        if (S->one_stream) {
            S->el.digits(w_l, n_l, s0); S->el.accumulate(s0);
            S->ea.digits(w_a, n_a, s0); S->ea.accumulate(s0);
            if (!p->skip_b1) S->eb1.digits(w_a, n_a, s0);
            if (b2_adopts) S->eb2.adopt(S->eb1.grouped(), S->eb1.counters.p, n_a, s0);
            if (!p->skip_b1) S->eb1.accumulate(s0);
            if (!b2_adopts) S->eb2.digits(w_a, n_a, s0);
            S->eb2.accumulate(s0);
        } else {
            CG_HIP(hipEventRecord(S->ev_w, s0));
            for (int i = 1; i < 5; ++i) CG_HIP(hipStreamWaitEvent(S->st[i], S->ev_w, 0));
            S->el.digits(w_l, n_l, S->st[1]);
            S->ea.digits(w_a, n_a, S->st[2]);
            if (!p->skip_b1) S->eb1.digits(w_a, n_a, S->st[3]);
            if (b2_adopts) {
                CG_HIP(hipEventRecord(S->ev_b1, S->st[3]));
                CG_HIP(hipStreamWaitEvent(S->st[4], S->ev_b1, 0));
                S->eb2.adopt(S->eb1.grouped(), S->eb1.counters.p, n_a, S->st[4]);
            } else {
                S->eb2.digits(w_a, n_a, S->st[4]);
            }
            S->el.accumulate(S->st[1]);
            S->ea.accumulate(S->st[2]);
            if (!p->skip_b1) S->eb1.accumulate(S->st[3]);
            S->eb2.accumulate(S->st[4]);
        }
        *out = p.release();
        return CG_OK;
    } catch (...) {
        const int e = translate_exception();
        p->close();
        return e;
    }
}

static int partial_coset_values(cg_partial* p, int half, void* q_out, int q_on_device) {
    if (!p || !p->c || !p->S || !q_out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument or a closed handle");
    cg_ctx* ctx = p->c;
    if (ctx->external_q) return fail(CG_ERR_INVALID_ARGUMENT, "context loaded with CG_FLAG_H_SCALARS_EXTERNAL holds no witness-map resources");
    try {
        CG_HIP(hipSetDevice(ctx->device));
        return coset_values_to(ctx, p->S, p->w_dev, half, q_out, q_on_device);
    } catch (...) {
        return translate_exception();
    }
}
extern "C" int cg_partial_witness_map_coset(cg_partial* p, void* q_out, int q_on_device) { return partial_coset_values(p, 0, q_out, q_on_device); }
extern "C" int cg_partial_witness_map_coset_half(cg_partial* p, int which, void* out, int out_on_device) {
    if (which != 0 && which != 1) return fail(CG_ERR_INVALID_ARGUMENT, "which must be 0 (the a side) or 1 (the b side)");
    return partial_coset_values(p, which + 1, out, out_on_device);
}

// q_slice: this shard's h scalars; or, with b_slice, the a side's slice, the h scalars being the products of the two
static int partial_finish(cg_partial* p, const void* q_slice, const void* b_slice, bool two_sides, int q_on_device, uint8_t out_partials[384],
                          cg_timings* timings) {
    if (!p) return fail(CG_ERR_INVALID_ARGUMENT, "null handle");
    std::unique_ptr<cg_partial> own(p);
    if (!p->c || !p->S) return fail(CG_ERR_INVALID_ARGUMENT, "closed handle");
    if (!q_slice || !out_partials || (two_sides && !b_slice)) { p->close(); return fail(CG_ERR_INVALID_ARGUMENT, "null argument"); }
    cg_ctx* c = p->c;
    ProofSlot* S = p->S;
    int e = CG_OK;
    Partials P;
    TuneStats ts;
    try {
        CG_HIP(hipSetDevice(c->device));
        hipStream_t s0 = S->st[0];
        const uint64_t nq = c->rh.hi - c->rh.lo;
        const Fr* q_dev = (const Fr*)q_slice;
        Fr* const own_q = c->external_q ? S->h_canon.p : S->h_canon.p + c->rh.lo;      // this shard's place in the slot's h vector
        if (!q_on_device && nq) {
            CG_HIP(hipMemcpyAsync(own_q, q_slice, nq * 32, hipMemcpyHostToDevice, s0));
            q_dev = own_q;
        }
        // the input checks the witness map would have made (canonical assignment), and of what arrived instead of it
        S->wm.h_bad_input.p[0] = 0;
        k_flag_non_canonical<<<ceil_div(c->M, 256), 256, 0, s0>>>(p->w_dev, c->M, S->wm.h_bad_input.dev());
        if (two_sides && nq) {
            const Fr* b_dev = (const Fr*)b_slice;
            if (!q_on_device) {
                if (S->q2.n < nq) S->q2.alloc(nq);
                CG_HIP(hipMemcpyAsync(S->q2.p, b_slice, nq * 32, hipMemcpyHostToDevice, s0));
                b_dev = S->q2.p;
            }
            fr_mul_plain29(q_dev, b_dev, own_q, nq, S->wm.h_bad_input.dev(), s0);    // q_j = (vinv·a_j)·b_j; checks both operands
            q_dev = own_q;
        } else if (nq) {
            k_flag_non_canonical<<<ceil_div(nq, 256), 256, 0, s0>>>(q_dev, nq, S->wm.h_bad_input.dev());
        }
        CG_KERNEL_CHECK();
        const Fr* h_scalars = q_dev;
        S->eh.digits(h_scalars, nq, s0);
        S->eh.accumulate(s0);
        if (S->one_stream && !spin_wait(c)) {
            CG_HIP(hipEventRecord(S->ev_done, s0));
            wait_sleeping(S->ev_done, 250);
        } else {
            wait_for_slot(S);
        }
        if (S->wm.h_bad_input.p[0]) {
            e = fail(CG_ERR_INVALID_ARGUMENT, "full_assignment or the h-scalar slice holds a value >= the scalar field modulus");
        } else {
            P.h = to_affine(S->eh.value());
            P.l = to_affine(S->el.value());
            P.a = to_affine(S->ea.value());
            P.b1 = p->skip_b1 ? G1Affine::inf() : to_affine(S->eb1.value());
            P.b2 = to_affine(S->eb2.value());
            if (timings) {
                memset(timings, 0, sizeof(*timings));
                timings->msm_h_ms = S->eh.ms_total(); timings->msm_l_ms = S->el.ms_total(); timings->msm_a_ms = S->ea.ms_total();
                timings->msm_b1_ms = p->skip_b1 ? 0.f : S->eb1.ms_total(); timings->msm_b2_ms = S->eb2.ms_total();
                timings->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - p->t0).count();
            }
            snapshot_tune_stats(c, S, p->skip_b1, ts);
        }
    } catch (...) {
        e = translate_exception();
    }
    p->c->calls_inside.fetch_add(1, std::memory_order_acq_rel);    // this call's own tail (the re-tune check) outlives the handle's hold
    p->close();
    if (!e) {
        maybe_retune(c, ts);
        partials_to_bytes(P, out_partials);
    }
    c->calls_inside.fetch_sub(1, std::memory_order_acq_rel);
    return e;
}

extern "C" int cg_prove_partial_q_finish(cg_partial* p, const void* q_slice, int q_on_device, uint8_t out_partials[384], cg_timings* timings) {
    return partial_finish(p, q_slice, nullptr, false, q_on_device, out_partials, timings);
}
extern "C" int cg_prove_partial_q_finish2(cg_partial* p, const void* a_slice, const void* b_slice, int slices_on_device, uint8_t out_partials[384],
                                          cg_timings* timings) {
    return partial_finish(p, a_slice, b_slice, true, slices_on_device, out_partials, timings);
}

extern "C" void cg_prove_partial_q_abort(cg_partial* p) {
    if (!p) return;
    p->close();
    delete p;
}

extern "C" int cg_h_scalars_slice(const cg_ctx* ctx, uint32_t shard, uint64_t* offset, uint64_t* count) {
    if (!ctx || !offset || !count) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->folded) return fail(CG_ERR_INVALID_ARGUMENT, "context keeps the h query in the coefficient basis");
    if (shard >= (uint32_t)ctx->shard_count) return fail(CG_ERR_INVALID_ARGUMENT, "shard out of range");
    if (ctx->span_hi) {          // a context that was given its span knows its own shard only
        if ((int)shard != ctx->shard_rank) return fail(CG_ERR_INVALID_ARGUMENT, "a context loaded with shard_span answers for its own shard only");
        *offset = ctx->D * (uint64_t)ctx->span_lo / 10000u;
        *count = ctx->D * (uint64_t)ctx->span_hi / 10000u - *offset;
    } else if (ctx->h_strided) {
        *count = ctx->D / (uint64_t)ctx->shard_count;
        *offset = *count * shard;
    } else {
        const Range rg = shard_range(ctx->D, (int)shard, ctx->shard_count);
        *offset = rg.lo;
        *count = rg.hi - rg.lo;
    }
    return CG_OK;
}

static int witness_map_coset_common(cg_ctx* ctx, const void* full_assignment, int assignment_on_device, int half, void* q_out, int q_on_device) {
    if (!ctx || !full_assignment || !q_out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    if (!ctx->folded) return fail(CG_ERR_INVALID_ARGUMENT, "context keeps the h query in the coefficient basis: its h scalars are cg_witness_map's");
    if (ctx->external_q) return fail(CG_ERR_INVALID_ARGUMENT, "context loaded with CG_FLAG_H_SCALARS_EXTERNAL holds no witness-map resources");
    CallGuard inside(ctx);
    try {
        CG_HIP(hipSetDevice(ctx->device));
        std::shared_lock<TuneGate> tl(ctx->tune_mu);
        if (ctx->broken) return fail(CG_ERR_OUT_OF_MEMORY, "%s", BROKEN_CONTEXT);
        UploadGuard up;
        const Fr* w_dev = (const Fr*)full_assignment;
        if (!assignment_on_device) {
            up.take(ctx);
            (void)upload_assignment(ctx, up.u, full_assignment, false);
            w_dev = up.u->w.p;
        }
        SlotGuard g(ctx);
        // ALL coset values (the whole-domain arrangement, whatever this context's own share is)
        return coset_values_to(ctx, g.s, w_dev, half, q_out, q_on_device);
    } catch (...) {
        return translate_exception();
    }
}
extern "C" int cg_witness_map_coset(cg_ctx* ctx, const void* full_assignment, int assignment_on_device, void* q_out, int q_on_device) {
    return witness_map_coset_common(ctx, full_assignment, assignment_on_device, 0, q_out, q_on_device);
}
extern "C" int cg_witness_map_coset_half(cg_ctx* ctx, const void* full_assignment, int assignment_on_device, int which, void* out, int out_on_device) {
    if (which != 0 && which != 1) return fail(CG_ERR_INVALID_ARGUMENT, "which must be 0 (the a side) or 1 (the b side)");
    return witness_map_coset_common(ctx, full_assignment, assignment_on_device, which + 1, out, out_on_device);
}

extern "C" int cg_assemble(cg_ctx* ctx, const uint8_t* partials, uint32_t n_shards, const uint8_t r[32], const uint8_t s[32],
                           uint8_t proof_out[256]) {
    if (!ctx || !partials || !proof_out || n_shards == 0) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    if (int e = check_rs(r, s)) return e;
    CallGuard inside(ctx);
    try {
        G1XYZZ h = G1XYZZ::inf(), l = G1XYZZ::inf(), a = G1XYZZ::inf(), b1 = G1XYZZ::inf();
        G2XYZZ b2 = G2XYZZ::inf();
        for (uint32_t k = 0; k < n_shards; ++k) {
            const uint8_t* p = partials + (size_t)k * 384;
            madd(h, g1_import(p, CG_FORM_CANONICAL));
            madd(l, g1_import(p + 64, CG_FORM_CANONICAL));
            madd(a, g1_import(p + 128, CG_FORM_CANONICAL));
            madd(b1, g1_import(p + 192, CG_FORM_CANONICAL));
            madd(b2, g2_import(p + 256, CG_FORM_CANONICAL));
        }
        Partials S{to_affine(h), to_affine(l), to_affine(a), to_affine(b1), to_affine(b2)};
        assemble_impl(ctx, S, r, s, proof_out);
        return CG_OK;
    } catch (...) {
        return translate_exception();
    }
}

extern "C" int cg_witness_map(cg_ctx* ctx, const uint8_t* full_assignment, uint8_t* h_out) {
    if (!ctx || !full_assignment || !h_out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    if (ctx->external_q) return fail(CG_ERR_INVALID_ARGUMENT, "context loaded with CG_FLAG_H_SCALARS_EXTERNAL holds no witness-map resources");
    CallGuard inside(ctx);
    try {
        CG_HIP(hipSetDevice(ctx->device));
        std::shared_lock<TuneGate> tl(ctx->tune_mu);
        if (ctx->broken) return fail(CG_ERR_OUT_OF_MEMORY, "%s", BROKEN_CONTEXT);
        UploadGuard up;
        up.take(ctx);
        (void)upload_assignment(ctx, up.u, full_assignment, false);
        SlotGuard g(ctx);
        ProofSlot* S = g.s;
        hipStream_t s0 = S->st[0];
        run_witness_map(ctx, S, up.u->w.p, s0, false);   // the reference's result: coefficients
        CG_HIP(hipMemcpyAsync(h_out, S->h_canon.p, ctx->D * 32, hipMemcpyDeviceToHost, s0));
        CG_HIP(hipStreamSynchronize(s0));
        if (S->wm.h_bad_input.p[0]) return fail(CG_ERR_INVALID_ARGUMENT, "full_assignment holds a value >= the scalar field modulus");
        return CG_OK;
    } catch (...) {
        return translate_exception();
    }
}

// ---------------------------------------------------------------------------------------------
// page-locked host memory for assignments; context description
// ---------------------------------------------------------------------------------------------
extern "C" void* cg_host_alloc(uint64_t bytes) {
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)fail(e == hipErrorOutOfMemory ? CG_ERR_OUT_OF_MEMORY : CG_ERR_HIP, "hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes,
                   hipGetErrorString(e));
        return nullptr;
    }
    return p;
}
extern "C" void cg_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}
extern "C" int cg_host_register(void* p, uint64_t bytes) {
    if (!p || !bytes) return fail(CG_ERR_INVALID_ARGUMENT, "null buffer");
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) return fail(CG_ERR_HIP, "hipHostRegister failed: %s", hipGetErrorString(e));
    return CG_OK;
}
extern "C" int cg_host_unregister(void* p) {
    if (!p) return fail(CG_ERR_INVALID_ARGUMENT, "null buffer");
    hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) return fail(CG_ERR_HIP, "hipHostUnregister failed: %s", hipGetErrorString(e));
    return CG_OK;
}

extern "C" int cg_ctx_get_info(cg_ctx* ctx, cg_ctx_info* out) {
    if (!ctx || !out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    try {
        memset(out, 0, sizeof(*out));
        std::shared_lock<TuneGate> tl(ctx->tune_mu);     // not in the middle of a re-tune
        out->table_bytes = (uint64_t)ctx->table_bytes;
        out->matrix_bytes = (uint64_t)ctx->matrix_bytes;
        out->slot_bytes = (uint64_t)ctx->slot_bytes;
        int n_regular = 0;
        for (auto& sp : ctx->slots) n_regular += sp->lone ? 0 : 1;
        out->proof_slots = n_regular;
        out->lone_slots = (int32_t)ctx->slots.size() - n_regular;
        out->lone_slot_bytes = (uint64_t)ctx->lone_slot_bytes;
        out->total_bytes = out->table_bytes + out->matrix_bytes + out->slot_bytes * (uint64_t)out->proof_slots + out->lone_slot_bytes;
        CG_HIP(hipSetDevice(ctx->device));
        size_t free_b = 0, total_b = 0;
        CG_HIP(hipMemGetInfo(&free_b, &total_b));
        out->device_free_bytes = free_b;
        out->device_total_bytes = total_b;
        out->window_bits[0] = ctx->bh.c; out->window_bits[1] = ctx->bl.c; out->window_bits[2] = ctx->ba.c;
        out->window_bits[3] = ctx->bb1.c; out->window_bits[4] = ctx->bb2.c;
        out->tuned = ctx->tuned ? 1 : 0;
        out->retune_skipped_for_memory = ctx->retune_skipped_memory;
        out->retune_attempts = ctx->retune_attempts;
        out->shard_rank = ctx->shard_rank;
        out->shard_count = ctx->shard_count;
        out->latency_mode = ctx->latency ? 1 : 0;
        out->warmup = ctx->warmup.load() ? 1 : 0;
        out->slot_entry_bytes = ctx->slot_part[0]; out->slot_piece_bytes = ctx->slot_part[1]; out->slot_bucket_bytes = ctx->slot_part[2];
        out->slot_transform_bytes = ctx->slot_part[3]; out->slot_upload_bytes = ctx->slot_part[4];
        return CG_OK;
    } catch (...) {
        return translate_exception();
    }
}
