// Host-side worker threads for the loaders (file parsers, matrix layouts): plain std::thread, joined before return.
//
// The files either side of the prove step are 0.6 GB each at the rs256 size (creds/test-vectors/README.md:5-10) and the
// reference reads them on every `create_client_state` (creds/src/lib.rs:257-268); a single thread walks them at a few
// hundred MB/s, which is most of a cold start once the GPU side of the load takes half a second.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <sys/mman.h>

namespace cg {

// threads a loader may start: the hardware threads visible, capped by the cgroup's CPU quota when there is one (a container
// that sees 256 threads and may use 16 is throttled, not sped up, by more runnable threads than its quota) and by 16
inline unsigned host_threads() {
    static const unsigned n = [] {
        unsigned hw = std::thread::hardware_concurrency();
        if (hw == 0) hw = 4;
        double quota = 0.0;
        {
            std::ifstream f("/sys/fs/cgroup/cpu.max");        // cgroup v2: "<quota|max> <period>"
            std::string q;
            double p = 0.0;
            if (f >> q >> p && q != "max" && p > 0.0) quota = atof(q.c_str()) / p;
        }
        if (quota <= 0.0) {
            std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");   // v1
            double q = 0.0, p = 0.0;
            if (fq >> q && fp >> p && q > 0.0 && p > 0.0) quota = q / p;
        }
        if (quota >= 1.0 && quota < (double)hw) hw = (unsigned)(quota + 0.5);
        return std::max(1u, std::min(hw, 16u));
    }();
    return n;
}

// fn(lo, hi) over [0, n) cut into one contiguous range per thread (ranges of at least `min_chunk` items: small inputs run
// on the calling thread).  An exception in any range is rethrown here - the one from the LOWEST range, so that what a
// parser reports does not depend on thread timing.
template <class Fn>
inline void parallel_ranges(uint64_t n, uint64_t min_chunk, Fn fn) {
    if (!n) return;
    uint64_t parts = std::min<uint64_t>(host_threads(), (n + min_chunk - 1) / (min_chunk ? min_chunk : 1));
    if (parts <= 1) { fn((uint64_t)0, n); return; }
    std::vector<std::exception_ptr> err(parts);
    std::vector<std::thread> th;
    th.reserve(parts - 1);
    auto run = [&](uint64_t k) {
        try {
            fn(n * k / parts, n * (k + 1) / parts);
        } catch (...) {
            err[k] = std::current_exception();
        }
    };
    for (uint64_t k = 1; k < parts; ++k) th.emplace_back(run, k);
    run(0);
    for (auto& t : th) t.join();
    for (auto& e : err)
        if (e) std::rethrow_exception(e);
}

// a host array that is NOT zero-filled when it is made (std::vector::resize writes every byte once before the parser does).
// Large ones are 2 MB-aligned and advised into transparent huge pages: the parsers' worker threads first-touch 0.6 GB of
// fresh memory, and with 4 KB pages that is 150 000 page faults contending for one address-space lock - more than half of a
// parse (0.35 s of 0.49 s measured at the rs256 size on eight threads; 0.15 s on memory already faulted in).
template <class T>
struct RawArray {
    T* p = nullptr;
    size_t n = 0;
    RawArray() = default;
    RawArray(const RawArray&) = delete;
    RawArray& operator=(const RawArray&) = delete;
    RawArray(RawArray&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    RawArray& operator=(RawArray&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~RawArray() { release(); }
    void release() { free(p); p = nullptr; n = 0; }
    void alloc(size_t count) {
        static_assert(std::is_trivial<T>::value, "RawArray holds trivial types only");
        release();
        const size_t bytes = (count ? count : 1) * sizeof(T);
        void* q = nullptr;
        if (bytes >= (size_t)4 << 20) {
            const size_t huge = (size_t)2 << 20;
            if (posix_memalign(&q, huge, (bytes + huge - 1) / huge * huge) != 0) q = nullptr;
            if (q) (void)madvise(q, (bytes + huge - 1) / huge * huge, MADV_HUGEPAGE);      // advice only: failure is harmless
        } else {
            q = malloc(bytes);
        }
        if (!q) throw std::bad_alloc();
        p = (T*)q;
        n = count;
    }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
};

}  // namespace cg
