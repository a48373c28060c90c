// Shared host-side plumbing for libcrescent_gpu: error reporting, HIP checks, device buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/crescent_gpu.h"
#include "curve.hpp"
#include "errors.hpp"

namespace cg {

// thread-local last error (cg_last_error)
std::string& last_error();
int fail(int code, const char* fmt, ...);


#define CG_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            char _b[512];                                                                         \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                     __FILE__, __LINE__);                                                         \
            throw ::cg::HipError(_e == hipErrorOutOfMemory ? CG_ERR_OUT_OF_MEMORY : CG_ERR_HIP, _b); \
        }                                                                                         \
    } while (0)

#define CG_KERNEL_CHECK() CG_HIP(hipGetLastError())

// Tuning / A-B / fault-injection switches are read from the environment ONLY in builds made with -DCG_TUNING
// (`python crescent-credentials_amd/build.py --tuning` -> libcrescent_gpu_tuning.so, tools/ab_*.sh); the shipped library
// never looks at the host's environment: what a host may legitimately choose is a cg_options field or flag
// (include/crescent_gpu.h).  CG_TUNE_ENV("X") is getenv("CG_X") there and a null pointer here (no string in the binary;
// tests/test_abi.py checks `strings`).
#ifdef CG_TUNING
#define CG_TUNE_ENV(name) getenv("CG_" name)
#else
#define CG_TUNE_ENV(name) ((const char*)nullptr)
#endif

// Device-memory accounting (cg_ctx_get_info): while an AllocScope is alive on a thread, every DevBuf that thread
// allocates adds its bytes to the scope's counter and every DevBuf it releases subtracts them, so temporaries made and
// freed inside a scope cancel out and what is left is what stays resident.
struct AllocScope {
    static int64_t*& current() { static thread_local int64_t* cur = nullptr; return cur; }
    int64_t* prev;
    explicit AllocScope(int64_t* counter) : prev(current()) { current() = counter; }
    ~AllocScope() { current() = prev; }
    AllocScope(const AllocScope&) = delete;
    AllocScope& operator=(const AllocScope&) = delete;
    static void note(int64_t delta) { if (int64_t* c = current()) *c += delta; }
};

// One allocation that many buffers are carved from (a proof slot's ~56 device buffers and ~11 page-locked ones).  hipMalloc,
// hipHostMalloc and hipFree each wait for the device; next to a context with sixteen proofs in flight that is ~20 ms apiece, and
// building sixteen slots buffer by buffer took 19 s there (1.2 s per slot; 0.02 s on an idle GPU: tools/probe_load_under_load.py).
// While a SlotArena is installed on a thread (ArenaScope), DevBuf::alloc / PinnedBuf::alloc on that thread take their memory
// from it - 256-byte aligned, not owned: released with the arena - and fall back to an allocation of their own when it is
// full.  An AllocMeter installed instead adds up what a build asks for, so the next identical build can size its arena.
struct SlotArena {
    uint8_t* dev = nullptr;
    uint8_t* host = nullptr;
    size_t dev_size = 0, host_size = 0, dev_off = 0, host_off = 0;
    SlotArena() = default;
    SlotArena(const SlotArena&) = delete;
    SlotArena& operator=(const SlotArena&) = delete;
    ~SlotArena() {
        if (dev) (void)hipFree(dev);
        if (host) (void)hipHostFree(host);
    }
    static SlotArena*& current() { static thread_local SlotArena* a = nullptr; return a; }
    void* take_dev(size_t bytes) {
        const size_t at = (dev_off + 255) & ~(size_t)255;
        if (!dev || at + bytes > dev_size) return nullptr;
        dev_off = at + bytes;
        return dev + at;
    }
    void* take_host(size_t bytes) {
        const size_t at = (host_off + 255) & ~(size_t)255;
        if (!host || at + bytes > host_size) return nullptr;
        host_off = at + bytes;
        return host + at;
    }
};
struct AllocMeter {
    size_t dev_bytes = 0, host_bytes = 0;          // with the 256-byte alignment an arena would add
    static AllocMeter*& current() { static thread_local AllocMeter* m = nullptr; return m; }
    static void note_dev(size_t bytes) { if (AllocMeter* m = current()) m->dev_bytes = ((m->dev_bytes + 255) & ~(size_t)255) + bytes; }
    static void note_host(size_t bytes) { if (AllocMeter* m = current()) m->host_bytes = ((m->host_bytes + 255) & ~(size_t)255) + bytes; }
};
struct ArenaScope {      // installs an arena (or a meter) on this thread for one scope
    SlotArena* prev_a;
    AllocMeter* prev_m;
    ArenaScope(SlotArena* a, AllocMeter* m) : prev_a(SlotArena::current()), prev_m(AllocMeter::current()) {
        SlotArena::current() = a;
        AllocMeter::current() = m;
    }
    ~ArenaScope() { SlotArena::current() = prev_a; AllocMeter::current() = prev_m; }
    ArenaScope(const ArenaScope&) = delete;
    ArenaScope& operator=(const ArenaScope&) = delete;
};

// RAII device buffer
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool owned = true;          // false: carved from a SlotArena, which releases it
    size_t cap_bytes = 0;       // of an arena piece: what was carved (a later, smaller alloc() reuses the piece in place)
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), owned(o.owned), cap_bytes(o.cap_bytes) { o.p = nullptr; o.n = 0; o.owned = true; o.cap_bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            release();
            p = o.p; n = o.n; owned = o.owned; cap_bytes = o.cap_bytes;
            o.p = nullptr; o.n = 0; o.owned = true; o.cap_bytes = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        if (p && !owned && count && count * sizeof(T) <= cap_bytes) {      // an arena piece re-cut (a re-tune shrinks bucket arrays)
            AllocScope::note((int64_t)(count * sizeof(T)) - (int64_t)(n * sizeof(T)));
            n = count;
            return;
        }
        release();
        if (count) {
            AllocMeter::note_dev(count * sizeof(T));
            void* q = SlotArena::current() ? SlotArena::current()->take_dev(count * sizeof(T)) : nullptr;
            if (q) {
                p = (T*)q;
                owned = false;
                cap_bytes = count * sizeof(T);
            } else {
                const hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
                if (e != hipSuccess) {
                    p = nullptr;
                    (void)hipGetLastError();        // the failure is reported here, not left behind as the thread's sticky error
                    size_t free_b = 0, total_b = 0;
                    (void)hipMemGetInfo(&free_b, &total_b);
                    char b[256];
                    snprintf(b, sizeof(b), "device allocation of %zu bytes failed: %s (%zu of %zu bytes free on the device)",
                             count * sizeof(T), hipGetErrorString(e), free_b, total_b);
                    throw ::cg::HipError(e == hipErrorOutOfMemory ? CG_ERR_OUT_OF_MEMORY : CG_ERR_HIP, b);
                }
                owned = true;
            }
        }
        n = count;
        AllocScope::note((int64_t)(count * sizeof(T)));
    }
    void release() {
        if (p) {
            if (owned) (void)hipFree(p);
            AllocScope::note(-(int64_t)(n * sizeof(T)));
        }
        p = nullptr; n = 0; owned = true; cap_bytes = 0;
    }
    size_t bytes() const { return n * sizeof(T); }
};

template <class T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0;
    bool owned = true;
    PinnedBuf() = default;
    explicit PinnedBuf(size_t count) { alloc(count); }
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { free_now(); }
    void free_now() {
        if (p && owned) (void)hipHostFree(p);
        p = nullptr;
        owned = true;
    }
    void alloc(size_t count) {
        free_now();
        if (count) {
            AllocMeter::note_host(count * sizeof(T));
            void* q = SlotArena::current() ? SlotArena::current()->take_host(count * sizeof(T)) : nullptr;
            if (q) {
                p = (T*)q;
                owned = false;
            } else {
                CG_HIP(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault));
                owned = true;
            }
        }
        n = count;
    }
    // the address kernels use to write this host memory directly (results of a few KB: no copy kernel, no extra
    // submission on the stream; the stream synchronisation that precedes every host read makes the writes visible)
    T* dev() const {
        void* d = nullptr;
        if (p) CG_HIP(hipHostGetDevicePointer(&d, (void*)p, 0));
        return (T*)d;
    }
};

// A stream that lives for one scope (loaders): destroyed on every way out, an exception included.
struct ScopedStream {
    hipStream_t st = nullptr;
    ScopedStream() { CG_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); }
    ~ScopedStream() { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } }
    ScopedStream(const ScopedStream&) = delete;
    ScopedStream& operator=(const ScopedStream&) = delete;
    operator hipStream_t() const { return st; }
};

// Host -> device copy of a loader's temporary (a std::vector about to go out of scope): on the LOADER'S stream, and waited
// for.  No synchronous hipMemcpy / hipMemset in this library: they run on the legacy default stream, which does not order
// itself against the non-blocking streams every context works on ("stream-order audit").
inline void h2d_sync(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (!bytes) return;
    CG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    CG_HIP(hipStreamSynchronize(st));
}

// zero `bytes` (a multiple of 16) of device memory with full-width stores on the whole chip
// (hipMemsetAsync's fill kernel reaches ~0.2 TB/s on these sizes)
void fill_zero(void* dst, size_t bytes, hipStream_t st);

inline uint32_t ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
inline int ilog2_ceil(uint64_t n) { int l = 0; while ((1ull << l) < n) ++l; return l; }

// ---- host-side byte <-> field conversions --------------------------------------------------------
template <class F>
inline F fp_from_bytes(const uint8_t* b) {  // raw 32 LE bytes -> limbs (no form change)
    F r;
    memcpy(r.l, b, 32);
    return r;
}
template <class F>
inline void fp_to_bytes(const F& a, uint8_t* b) { memcpy(b, a.l, 32); }

template <class P>
inline bool fp_is_canonical(const Fp<P>& a) {  // a < N
    for (int i = 7; i >= 0; --i) {
        if (a.l[i] < P::N[i]) return true;
        if (a.l[i] > P::N[i]) return false;
    }
    return false;
}

// ---- canonical byte <-> Montgomery point conversions (host) ----------------------------------------
inline Fq fq_import(const uint8_t* b, uint32_t form) {
    Fq a = fp_from_bytes<Fq>(b);
    return form == CG_FORM_CANONICAL ? to_mont(a) : a;
}
inline G1Affine g1_import(const uint8_t* b, uint32_t form) { return {fq_import(b, form), fq_import(b + 32, form)}; }
inline G2Affine g2_import(const uint8_t* b, uint32_t form) {
    return {{fq_import(b, form), fq_import(b + 32, form)}, {fq_import(b + 64, form), fq_import(b + 96, form)}};
}
inline void g1_export_canonical(const G1Affine& p, uint8_t* out) {  // identity -> zeros
    fp_to_bytes(from_mont(p.x), out);
    fp_to_bytes(from_mont(p.y), out + 32);
}
inline void g2_export_canonical(const G2Affine& p, uint8_t* out) {
    fp_to_bytes(from_mont(p.x.c0), out);
    fp_to_bytes(from_mont(p.x.c1), out + 32);
    fp_to_bytes(from_mont(p.y.c0), out + 64);
    fp_to_bytes(from_mont(p.y.c1), out + 96);
}

inline bool scalar_is_zero(const uint8_t s[32]) {
    for (int i = 0; i < 32; ++i) if (s[i]) return false;
    return true;
}
inline bool scalar_is_canonical(const uint8_t s[32]) { return fp_is_canonical(fp_from_bytes<Fr>(s)); }

int translate_current_exception();   // HipError / bad_alloc / std::exception in flight -> cg_status + last_error()

}  // namespace cg
