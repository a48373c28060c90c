// libcrescent_gpu: unit-level MSM and NTT entry points of include/crescent_gpu.h.
//
// The one-shot forms mirror the reference's own call shapes (`msm_bigint(bases, scalars)`,
// `domain.fft_in_place(&mut v)`; call sites forks/groth16/src/prover.rs:66,74,266 and
// r1cs_to_qap.rs:179-185,198-199,210).  The handle forms keep what a caller reuses - the expanded base
// tables of a fixed set of bases, the twiddle tables of a domain - resident in HBM, and accept operands that
// already live on the device; they run the same kernels as cg_prove (csrc/msm.hip, csrc/wmap29.hip).
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>

#include "msm.hpp"
#include "wmap29.hpp"

using namespace cg;

// ---------------------------------------------------------------------------------------------
// MSM over a resident set of bases
// ---------------------------------------------------------------------------------------------
struct cg_msm_ctx {
    int device = 0;
    int group = 1;
    uint64_t n = 0;
    MsmBases<Fq> b1;
    MsmBases<Fq2> b2;
    MsmEngine<Fq> e1;
    MsmEngine<Fq2> e2;
    DevBuf<Fr> scalars;
    DevBuf<uint32_t> bad;
    PinnedBuf<uint32_t> h_bad;
    hipStream_t st = nullptr;
    std::mutex mu;
    ~cg_msm_ctx() { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } }   // a failed load may leave work queued
};

__global__ void __launch_bounds__(256) k_check_canonical(const Fr* __restrict__ s, uint64_t n, uint32_t* __restrict__ bad) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = s[i];
    bool lt = false, decided = false;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        if (!decided && x.l[k] != FrP::N[k]) { lt = x.l[k] < FrP::N[k]; decided = true; }
    }
    if (!lt) *bad = 1u;
}

static int msm_load(cg_msm_ctx** out, int group, const uint8_t* bases, uint32_t form, uint64_t n, const cg_options* opt) {
    if (!out) return fail(CG_ERR_INVALID_ARGUMENT, "null out");
    *out = nullptr;
    if (n && !bases) return fail(CG_ERR_INVALID_ARGUMENT, "null bases");
    if (form != CG_FORM_CANONICAL && form != CG_FORM_MONTGOMERY) return fail(CG_ERR_INVALID_ARGUMENT, "bad coord_form");
    if (n >= (1ull << 31)) return fail(CG_ERR_INVALID_ARGUMENT, "more than 2^31 - 1 bases");
    try {
        int dev = (opt && opt->device >= 0) ? opt->device : -1;
        if (dev < 0) CG_HIP(hipGetDevice(&dev));
        CG_HIP(hipSetDevice(dev));
        std::unique_ptr<cg_msm_ctx> c(new cg_msm_ctx());
        c->device = dev;
        c->group = group;
        c->n = n;
        CG_HIP(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
        int wb = opt ? opt->window_bits : 0;
        if (wb < 0 || wb == 1 || wb > 22) return fail(CG_ERR_INVALID_ARGUMENT, "window_bits must be 0 or in [2, 22]");
        if (n) {
            const int cbits = wb > 0 ? wb : msm_default_window(n, true);
            if (group == 1) {
                DevBuf<G1Affine> tmp(n);
                import_bases<Fq>(bases, form, n, tmp.p, c->st);
                c->b1.build(tmp.p, n, cbits, true, c->st);
                CG_HIP(hipStreamSynchronize(c->st));
                c->e1.latency_mode = true;      // one MSM at a time (calls on a handle serialise): short segments, tree reduction
                c->e1.init(&c->b1);
            } else {
                DevBuf<G2Affine> tmp(n);
                import_bases<Fq2>(bases, form, n, tmp.p, c->st);
                c->b2.build(tmp.p, n, cbits, true, c->st);
                CG_HIP(hipStreamSynchronize(c->st));
                c->e2.latency_mode = true;
                c->e2.init(&c->b2);
            }
            c->scalars.alloc(n);
            c->bad.alloc(1);
            c->h_bad.alloc(1);
        }
        *out = c.release();
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}

extern "C" int cg_msm_load_g1(cg_msm_ctx** out, const uint8_t* bases, uint32_t coord_form, uint64_t n_bases, const cg_options* opt) {
    return msm_load(out, 1, bases, coord_form, n_bases, opt);
}
extern "C" int cg_msm_load_g2(cg_msm_ctx** out, const uint8_t* bases, uint32_t coord_form, uint64_t n_bases, const cg_options* opt) {
    return msm_load(out, 2, bases, coord_form, n_bases, opt);
}

extern "C" void cg_msm_free(cg_msm_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    delete ctx;
}

template <class F>
static void fill_msm_timings(const MsmEngine<F>& e, bool g2, cg_timings* tm) {
    memset(tm, 0, sizeof(*tm));
    (g2 ? tm->msm_b2_ms : tm->msm_h_ms) = e.ms_total();
    (g2 ? tm->accum_g2_ms : tm->accum_g1_ms) = e.ms_accum();
    tm->sort_ms = e.ms_sort();
    (g2 ? tm->msm_g2_pairs : tm->msm_g1_pairs) = e.n_scalars;
    (g2 ? tm->entries_g2 : tm->entries_g1) = e.n_entries();
    (g2 ? tm->accum_g2_launches : tm->accum_g1_launches) = e.n_entries() != 0;
}

extern "C" int cg_msm_run(cg_msm_ctx* ctx, const void* scalars, int scalars_on_device, uint64_t n_scalars, uint8_t* out,
                          cg_timings* timings) {
    if (!ctx || !out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    const size_t out_bytes = ctx->group == 1 ? 64 : 128;
    const uint64_t n = n_scalars < ctx->n ? n_scalars : ctx->n;     // msm_bigint zips bases with scalars
    if (timings) memset(timings, 0, sizeof(*timings));
    if (n == 0) { memset(out, 0, out_bytes); return CG_OK; }
    if (!scalars) return fail(CG_ERR_INVALID_ARGUMENT, "null scalars");
    try {
        std::lock_guard<std::mutex> lk(ctx->mu);
        CG_HIP(hipSetDevice(ctx->device));
        auto t0 = std::chrono::steady_clock::now();
        hipStream_t st = ctx->st;
        const Fr* sc = (const Fr*)scalars;
        if (!scalars_on_device) {
            CG_HIP(hipMemcpyAsync(ctx->scalars.p, scalars, n * 32, hipMemcpyHostToDevice, st));
            sc = ctx->scalars.p;
        }
        CG_HIP(hipMemsetAsync(ctx->bad.p, 0, 4, st));
        k_check_canonical<<<ceil_div(n, 256), 256, 0, st>>>(sc, n, ctx->bad.p);
        CG_KERNEL_CHECK();
        CG_HIP(hipMemcpyAsync(ctx->h_bad.p, ctx->bad.p, 4, hipMemcpyDeviceToHost, st));
        if (ctx->group == 1) {
            ctx->e1.digits(sc, n, st);
            ctx->e1.accumulate(st);
            CG_HIP(hipStreamSynchronize(st));
            if (ctx->h_bad.p[0]) return fail(CG_ERR_INVALID_ARGUMENT, "a scalar is not canonical (>= the scalar field modulus)");
            g1_export_canonical(to_affine(ctx->e1.value()), out);
            if (timings) fill_msm_timings(ctx->e1, false, timings);
        } else {
            ctx->e2.digits(sc, n, st);
            ctx->e2.accumulate(st);
            CG_HIP(hipStreamSynchronize(st));
            if (ctx->h_bad.p[0]) return fail(CG_ERR_INVALID_ARGUMENT, "a scalar is not canonical (>= the scalar field modulus)");
            g2_export_canonical(to_affine(ctx->e2.value()), out);
            if (timings) fill_msm_timings(ctx->e2, true, timings);
        }
        if (timings) timings->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}

// one-shot: bases used once, so only window 0 is tabulated and the keys carry the window index
template <class F>
static int msm_unit(const uint8_t* bases, uint32_t form, uint64_t n_bases, const uint8_t* scalars, uint64_t n_scalars,
                    int window_bits, Affine<F>& out) {
    uint64_t n = n_bases < n_scalars ? n_bases : n_scalars;
    out = Affine<F>::inf();
    if (n == 0) return CG_OK;
    if (!bases || !scalars) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    if (form != CG_FORM_CANONICAL && form != CG_FORM_MONTGOMERY) return fail(CG_ERR_INVALID_ARGUMENT, "bad coord_form");
    for (uint64_t i = 0; i < n; ++i)
        if (!scalar_is_canonical(scalars + 32 * i)) return fail(CG_ERR_INVALID_ARGUMENT, "scalar %llu not canonical", (unsigned long long)i);
    {
        ScopedStream st;            // destroyed on every way out (the callers translate exceptions)
        DevBuf<Affine<F>> pts(n);
        import_bases<F>(bases, form, n, pts.p, st);
        DevBuf<Fr> sc(n);
        CG_HIP(hipMemcpyAsync(sc.p, scalars, n * 32, hipMemcpyHostToDevice, st));
        MsmBases<F> mb;
        int c = window_bits > 0 ? window_bits : msm_default_window(n, false);
        if (c < 2 || c > 22) throw HipError(CG_ERR_INVALID_ARGUMENT, "window_bits must be in [2, 22]");
        mb.build(pts.p, n, c, false, st);
        MsmEngine<F> eng;
        eng.latency_mode = true;
        eng.init(&mb);
        eng.digits(sc.p, n, st);
        eng.accumulate(st);
        CG_HIP(hipStreamSynchronize(st));
        out = to_affine(eng.value());
    }
    return CG_OK;
}

extern "C" int cg_msm_g1(const uint8_t* bases, uint32_t coord_form, uint64_t n_bases, const uint8_t* scalars,
                         uint64_t n_scalars, int32_t window_bits, uint8_t out[64]) {
    if (!out) return fail(CG_ERR_INVALID_ARGUMENT, "null out");
    try {
        G1Affine r;
        int e = msm_unit<Fq>(bases, coord_form, n_bases, scalars, n_scalars, window_bits, r);
        if (e) return e;
        g1_export_canonical(r, out);
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}
extern "C" int cg_msm_g2(const uint8_t* bases, uint32_t coord_form, uint64_t n_bases, const uint8_t* scalars,
                         uint64_t n_scalars, int32_t window_bits, uint8_t out[128]) {
    if (!out) return fail(CG_ERR_INVALID_ARGUMENT, "null out");
    try {
        G2Affine r;
        int e = msm_unit<Fq2>(bases, coord_form, n_bases, scalars, n_scalars, window_bits, r);
        if (e) return e;
        g2_export_canonical(r, out);
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}

// ---------------------------------------------------------------------------------------------
// R1CS -> QAP witness map over resident matrices (no proving key): the reference's own plug point,
// `R1CSToQAP::witness_map_from_matrices` (forks/groth16/src/r1cs_to_qap.rs:49-98,150-213)
// ---------------------------------------------------------------------------------------------
struct cg_qap_ctx {
    int device = 0;
    uint64_t l = 0, m = 0, M = 0, D = 0;
    int logD = 0;
    DevCsr A, B, C;
    Csr29 dA, dB, dC;
    NttDomain dom;
    Wm29Domain wdom;
    Wm29Buffers wm;
    DevBuf<Fr> w_canon, h_canon;
    hipStream_t st = nullptr;
    std::mutex mu;
    ~cg_qap_ctx() { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } }
};

namespace cg {
// shared with cg_circuit_load: every pointer a cg_csr view must carry for `rows` rows
const char* csr_view_problem(const cg_csr& m) {
    if (!m.row_ptr) return "null row_ptr";
    if (m.nnz && (!m.col || !m.coeff)) return "null col/coeff with nnz > 0";
    return nullptr;
}
}  // namespace cg

extern "C" int cg_qap_load(cg_qap_ctx** out, const cg_csr abc[3], uint64_t num_inputs, uint64_t num_constraints,
                           uint64_t num_variables, int32_t device) {
    if (!out || !abc) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (num_inputs == 0 || num_inputs > num_variables) return fail(CG_ERR_INVALID_ARGUMENT, "need 1 <= num_inputs <= num_variables");
    for (int k = 0; k < 3; ++k)
        if (const char* why = csr_view_problem(abc[k])) return fail(CG_ERR_INVALID_ARGUMENT, "matrix %d: %s", k, why);
    const int logD = ilog2_ceil(num_constraints + num_inputs);
    if (logD > 28)                                              // r1cs_to_qap.rs:156-157
        return fail(CG_ERR_POLY_DEGREE_TOO_LARGE, "num_constraints + num_inputs = %llu exceeds 2^28",
                    (unsigned long long)(num_constraints + num_inputs));
    try {
        int dev = device;
        if (dev < 0) CG_HIP(hipGetDevice(&dev));
        CG_HIP(hipSetDevice(dev));
        std::unique_ptr<cg_qap_ctx> c(new cg_qap_ctx());
        c->device = dev;
        c->l = num_inputs; c->m = num_constraints; c->M = num_variables;
        c->logD = logD; c->D = 1ull << logD;
        CG_HIP(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
        c->A.upload(abc[0], c->m, c->M, c->st);
        c->B.upload(abc[1], c->m, c->M, c->st);
        c->C.upload(abc[2], c->m, c->M, c->st);
        c->dom.build(logD, true, c->st);
        CG_HIP(hipStreamSynchronize(c->st));
        c->wdom.build(c->dom, c->st);
        c->dA.build(c->A, c->st); c->dB.build(c->B, c->st); c->dC.build(c->C, c->st);
        CG_HIP(hipStreamSynchronize(c->st));
        c->dom.tw_fwd.release(); c->dom.tw_inv.release(); c->dom.coset_br.release(); c->dom.icoset_br.release();
        c->A.dict.release(); c->B.dict.release(); c->C.dict.release();
        c->wm.alloc(c->M, c->D, std::max(c->A.sell_scratch, std::max(c->B.sell_scratch, c->C.sell_scratch)));
        c->w_canon.alloc(c->M);
        c->h_canon.alloc(c->D);
        *out = c.release();
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}

extern "C" uint64_t cg_qap_domain_size(const cg_qap_ctx* ctx) { return ctx ? ctx->D : 0; }

extern "C" int cg_qap_witness_map(cg_qap_ctx* ctx, const void* full_assignment, int assignment_on_device, void* h_out,
                                  int h_on_device) {
    if (!ctx || !full_assignment || !h_out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    try {
        std::lock_guard<std::mutex> lk(ctx->mu);
        CG_HIP(hipSetDevice(ctx->device));
        const Fr* w = (const Fr*)full_assignment;
        if (!assignment_on_device) {
            CG_HIP(hipMemcpyAsync(ctx->w_canon.p, full_assignment, ctx->M * 32, hipMemcpyHostToDevice, ctx->st));
            w = ctx->w_canon.p;
        }
        Fr* h = h_on_device ? (Fr*)h_out : ctx->h_canon.p;
        wm29_run(ctx->wdom, ctx->A, ctx->B, ctx->C, ctx->dA, ctx->dB, ctx->dC, ctx->wm, w, ctx->M, ctx->m, ctx->l, h, ctx->st,
                 false);
        if (!h_on_device) CG_HIP(hipMemcpyAsync(h_out, ctx->h_canon.p, ctx->D * 32, hipMemcpyDeviceToHost, ctx->st));
        CG_HIP(hipStreamSynchronize(ctx->st));
        if (ctx->wm.h_bad_input.p[0]) return fail(CG_ERR_INVALID_ARGUMENT, "full_assignment holds a value >= the scalar field modulus");
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}

extern "C" void cg_qap_free(cg_qap_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    delete ctx;
}

// ---------------------------------------------------------------------------------------------
// NTT over a resident domain
// ---------------------------------------------------------------------------------------------
struct cg_ntt_ctx {
    int device = 0;
    uint32_t log_n = 0;
    Ntt29Unit unit;           // unused for log_n = 0
    DevBuf<Fr> staging;       // for host-resident data
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    std::mutex mu;
    ~cg_ntt_ctx() {
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        if (st) (void)hipStreamDestroy(st);
    }
};

extern "C" int cg_ntt_load(cg_ntt_ctx** out, uint32_t log_n, int32_t device) {
    if (!out) return fail(CG_ERR_INVALID_ARGUMENT, "null out");
    *out = nullptr;
    if (log_n > 28) return fail(CG_ERR_POLY_DEGREE_TOO_LARGE, "log_n > 28 (the two-adicity of the BN254 scalar field)");
    try {
        int dev = device;
        if (dev < 0) CG_HIP(hipGetDevice(&dev));
        CG_HIP(hipSetDevice(dev));
        std::unique_ptr<cg_ntt_ctx> c(new cg_ntt_ctx());
        c->device = dev;
        c->log_n = log_n;
        CG_HIP(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
        for (auto& e : c->ev) CG_HIP(hipEventCreate(&e));
        if (log_n > 0) c->unit.build((int)log_n, c->st);
        *out = c.release();
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}

extern "C" void cg_ntt_free(cg_ntt_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    delete ctx;
}

extern "C" int cg_ntt_run(cg_ntt_ctx* ctx, void* data, int data_on_device, int inverse, int coset, float* kernel_ms) {
    if (!ctx || !data) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    if (kernel_ms) *kernel_ms = 0.f;
    const uint64_t n = 1ull << ctx->log_n;
    if (ctx->log_n == 0) {     // the size-1 transform is the identity (g^0 = 1, 1/n = 1); only the operand check remains
        uint8_t x[32];
        if (data_on_device) {
            try {       // on the handle's own stream (no legacy-default-stream copy next to other contexts' non-blocking streams)
                std::lock_guard<std::mutex> lk(ctx->mu);
                CG_HIP(hipSetDevice(ctx->device));
                CG_HIP(hipMemcpyAsync(x, data, 32, hipMemcpyDeviceToHost, ctx->st));
                CG_HIP(hipStreamSynchronize(ctx->st));
            } catch (...) { return translate_current_exception(); }
        } else {
            memcpy(x, data, 32);
        }
        if (!scalar_is_canonical(x)) return fail(CG_ERR_INVALID_ARGUMENT, "an element is not canonical (>= the scalar field modulus)");
        return CG_OK;
    }
    try {
        std::lock_guard<std::mutex> lk(ctx->mu);
        CG_HIP(hipSetDevice(ctx->device));
        hipStream_t st = ctx->st;
        Fr* d = (Fr*)data;
        if (!data_on_device) {
            if (!ctx->staging.p) ctx->staging.alloc(n);
            CG_HIP(hipMemcpyAsync(ctx->staging.p, data, n * 32, hipMemcpyHostToDevice, st));
            d = ctx->staging.p;
        }
        CG_HIP(hipEventRecord(ctx->ev[0], st));
        const bool ok = ctx->unit.run(d, inverse != 0, coset != 0, st);
        CG_HIP(hipEventRecord(ctx->ev[1], st));
        if (!ok) {
            CG_HIP(hipStreamSynchronize(st));
            return fail(CG_ERR_INVALID_ARGUMENT, "an element is not canonical (>= the scalar field modulus)");
        }
        if (!data_on_device) CG_HIP(hipMemcpyAsync(data, d, n * 32, hipMemcpyDeviceToHost, st));
        CG_HIP(hipStreamSynchronize(st));
        if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, ctx->ev[0], ctx->ev[1]);
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}

extern "C" int cg_ntt(uint8_t* data, uint32_t log_n, int inverse, int coset) {
    cg_ntt_ctx* c = nullptr;
    int e = cg_ntt_load(&c, log_n, -1);
    if (e) return e;
    e = cg_ntt_run(c, data, 0, inverse, coset, nullptr);
    cg_ntt_free(c);
    return e;
}

// ---------------------------------------------------------------------------------------------
// shader-clock probe (diagnostic): the clock the chip actually holds while it is doing something else
// ---------------------------------------------------------------------------------------------
// One wave reads the shader-cycle counter (s_memtime, clock64) and the constant-rate 100 MHz counter (s_memrealtime,
// wall_clock64) `window_ticks` of the latter apart, sleeping in between: cycles / ticks x 100 MHz is the clock the
// shader engines ran at over that window, whatever else was running.  It occupies one wave slot and next to no issue
// slots.
__global__ void __launch_bounds__(64) k_clock_probe(uint64_t window_ticks, uint64_t* __restrict__ out) {
    if (threadIdx.x != 0) return;
    const uint64_t w0 = wall_clock64();
    const uint64_t c0 = clock64();
    uint64_t w1 = w0;
    for (int guard = 0; guard < (1 << 24) && w1 - w0 < window_ticks; ++guard) {   // bounded whatever the counters do
        __builtin_amdgcn_s_sleep(127);
        w1 = wall_clock64();
    }
    const uint64_t c1 = clock64();
    w1 = wall_clock64();
    out[0] = c1 - c0;
    out[1] = w1 - w0;
}

namespace {
// one stream and one result buffer per device for the life of the process: a probe must not allocate or free anything
// while proofs run (freeing host or device memory synchronises the device)
struct ClockProbe {
    std::mutex mu;
    hipStream_t st = nullptr;
    cg::PinnedBuf<uint64_t> res;
    uint64_t* res_dev = nullptr;
    int wall_khz = 0;
};
ClockProbe* const g_probe = new ClockProbe[64];   // never destroyed: the HIP runtime may be gone by static destruction time
}  // namespace

extern "C" int cg_probe_shader_clock(int32_t device, uint32_t window_us, double* ghz_out) {
    if (!ghz_out || window_us == 0 || window_us > 1000000u) return fail(CG_ERR_INVALID_ARGUMENT, "window_us must be in [1, 1000000]");
    try {
        if (device < 0) CG_HIP(hipGetDevice(&device));
        if (device >= 64) return fail(CG_ERR_INVALID_ARGUMENT, "device ordinal out of range");
        CG_HIP(hipSetDevice(device));
        ClockProbe& P = g_probe[device];
        std::lock_guard<std::mutex> lk(P.mu);
        if (!P.st) {
            CG_HIP(hipDeviceGetAttribute(&P.wall_khz, hipDeviceAttributeWallClockRate, device));
            if (P.wall_khz <= 0) P.wall_khz = 100000;
            P.res.alloc(2);
            P.res_dev = P.res.dev();
            CG_HIP(hipStreamCreateWithFlags(&P.st, hipStreamNonBlocking));
        }
        P.res.p[0] = P.res.p[1] = 0;
        k_clock_probe<<<1, 64, 0, P.st>>>((uint64_t)window_us * (uint64_t)P.wall_khz / 1000u, P.res_dev);
        CG_KERNEL_CHECK();
        for (;;) {                                  // the probe sleeps on the device: so does its caller (no spinning synchronise)
            const hipError_t e = hipStreamQuery(P.st);
            if (e == hipSuccess) break;
            if (e != hipErrorNotReady) CG_HIP(e);
            std::this_thread::sleep_for(std::chrono::microseconds(window_us < 2000 ? 50 : 500));
        }
        if (!P.res.p[1]) return fail(CG_ERR_HIP, "the constant-rate counter did not advance");
        *ghz_out = (double)P.res.p[0] / (double)P.res.p[1] * (double)P.wall_khz * 1e-6;
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}
