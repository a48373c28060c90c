// Micro-benchmark of the 29-bit-limb arithmetic on gfx950: throughput of field products and curve additions
// at a given number of resident waves.  ./f29_rates [blocks_per_cu]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../crescent-credentials_amd/csrc/curve29.hpp"
using namespace cg;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256) k_fq_mul(uint32_t* io, int iters) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    Fq29 a, b; for (int i = 0; i < 9; ++i) { a.l[i] = io[t * 18 + i] & M29; b.l[i] = io[t * 18 + 9 + i] & M29; }
    for (int i = 0; i < iters; ++i) { a = mul(a, b); b = mul(b, a); }
    for (int i = 0; i < 9; ++i) io[t * 18 + i] = a.l[i] + b.l[i];
}
__global__ void __launch_bounds__(256) k_fq2_mul(uint32_t* io, int iters) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    Fq2_29 a, b; for (int i = 0; i < 9; ++i) { a.c0.l[i] = io[t * 18 + i] & M29; a.c1.l[i] = (io[t * 18 + i] >> 3) & M29; b.c0.l[i] = io[t * 18 + 9 + i] & M29; b.c1.l[i] = (io[t*18+9+i] >> 2) & M29; }
    for (int i = 0; i < iters; ++i) { a = mul(a, b); b = mul(b, a); }
    for (int i = 0; i < 9; ++i) io[t * 18 + i] = a.c0.l[i] + b.c1.l[i];
}
template <class F>
__global__ void __launch_bounds__(256) k_madd(const uint32_t* table, uint32_t* out, int iters, int npts) {
    constexpr int ACC = Words29<F>::ACC;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ29<F> acc; bool inf = true;
    for (int i = 0; i < iters; ++i) {
        Affine29<F> p = load_table_point<F>(table, (uint32_t)((t * 7 + i * 13) % npts), (i & 1) != 0);
        madd29(acc, inf, p);
    }
    store_acc(out + (size_t)t * ACC, acc, inf);
}
template <class F>
__global__ void __launch_bounds__(256) k_add(const uint32_t* pts, uint32_t* out, int iters, int npts) {
    constexpr int ACC = Words29<F>::ACC;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ29<F> acc; bool inf = true;
    for (int i = 0; i < iters; ++i) {
        XYZZ29<F> q; bool qi = load_acc(pts + (size_t)((t * 7 + i * 13) % npts) * ACC, q);
        add29(acc, inf, q, qi);
    }
    store_acc(out + (size_t)t * ACC, acc, inf);
}

// the same two loops timed by every wave itself in shader cycles (clock64): cycles per operation per SIMD = T / (waves per SIMD x
// iterations), independent of the clock the chip holds.  ./f29_rates --cycles
__global__ void __launch_bounds__(256) k_fq_mul_cyc(uint32_t* io, unsigned long long* cyc, int iters) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    Fq29 a, b; for (int i = 0; i < 9; ++i) { a.l[i] = io[t * 18 + i] & M29; b.l[i] = io[t * 18 + 9 + i] & M29; }
    const unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) { a = mul(a, b); b = mul(b, a); }
    const unsigned long long c1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[t >> 6] = c1 - c0;
    for (int i = 0; i < 9; ++i) io[t * 18 + i] = a.l[i] + b.l[i];
}
__global__ void __launch_bounds__(256) k_madd_cyc(const uint32_t* table, uint32_t* out, unsigned long long* cyc, int iters, int npts) {
    constexpr int ACC = Words29<Fq29>::ACC;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ29<Fq29> acc; bool inf = true;
    const unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        Affine29<Fq29> p = load_table_point<Fq29>(table, (uint32_t)((t * 7 + i * 13) % npts), (i & 1) != 0);
        madd29(acc, inf, p);
    }
    const unsigned long long c1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[t >> 6] = c1 - c0;
    store_acc(out + (size_t)t * ACC, acc, inf);
}
static double mean_cycles(unsigned long long* d_cyc, int waves) {
    std::vector<unsigned long long> h(waves);
    CHECK(hipMemcpy(h.data(), d_cyc, (size_t)waves * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto c : h) s += (double)c;
    return s / waves;
}
static int cycles_mode() {
    const int NPTS = 4096;
    std::vector<uint32_t> h(NPTS * 72);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u + 12345u) & 0x0fffffffu;
    uint32_t *d_tab, *d_out, *d_io; unsigned long long* d_cyc;
    CHECK(hipMalloc(&d_tab, h.size() * 4)); CHECK(hipMemcpy(d_tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)256 * 8 * 256 * 36 * 4)); CHECK(hipMalloc(&d_io, (size_t)256 * 8 * 256 * 18 * 4));
    CHECK(hipMalloc(&d_cyc, (size_t)256 * 8 * 4 * 8));
    CHECK(hipMemset(d_io, 0x5a, (size_t)256 * 8 * 256 * 18 * 4));
    printf("{\"what\": \"shader cycles per operation per SIMD (clock64 around each wave's loop; L2-resident operands, no HBM)\", \"waves_per_simd\": [1, 2, 4, 8], ");
    double mulc[4], maddc[4];
    const int wps[4] = {1, 2, 4, 8};
    for (int i = 0; i < 4; ++i) {
        const int blocks = 256 * wps[i], waves = blocks * 4;
        k_fq_mul_cyc<<<blocks, 256>>>(d_io, d_cyc, 20); CHECK(hipDeviceSynchronize());
        k_fq_mul_cyc<<<blocks, 256>>>(d_io, d_cyc, 200); CHECK(hipDeviceSynchronize());
        mulc[i] = mean_cycles(d_cyc, waves) / (wps[i] * 400.0);
        k_madd_cyc<<<blocks, 256>>>(d_tab, d_out, d_cyc, 8, NPTS); CHECK(hipDeviceSynchronize());
        k_madd_cyc<<<blocks, 256>>>(d_tab, d_out, d_cyc, 64, NPTS); CHECK(hipDeviceSynchronize());
        maddc[i] = mean_cycles(d_cyc, waves) / (wps[i] * 64.0);
    }
    printf("\"fq_mul_cycles\": [%.1f, %.1f, %.1f, %.1f], \"g1_madd_cycles\": [%.1f, %.1f, %.1f, %.1f]}\n", mulc[0], mulc[1], mulc[2], mulc[3],
           maddc[0], maddc[1], maddc[2], maddc[3]);
    return 0;
}

template <class K, class... A>
static float timeit(K kern, dim3 g, dim3 b, A... args) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<<<g, b>>>(args...); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); kern<<<g, b>>>(args...); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

// random gathers from a table of `npts` points (64 B each): isolates the memory side of the accumulation kernel
__global__ void __launch_bounds__(256) k_madd_gather(const uint32_t* table, uint32_t* out, int iters, uint32_t npts, int do_math) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ29<Fq29> acc; bool inf = true;
    uint32_t h = t * 2654435761u + 12345u;
    uint32_t sink = 0;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        uint32_t idx = (uint32_t)(((uint64_t)h * npts) >> 32);
        Affine29<Fq29> p = load_table_point<Fq29>(table, idx, (i & 1) != 0);
        if (do_math) madd29(acc, inf, p); else sink += p.x.l[0] + p.y.l[3];
    }
    if (!do_math) { acc.x.l[0] = sink; inf = false; }
    store_acc(out + (size_t)t * 36, acc, inf);
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "--cycles")) return cycles_mode();
    {
        uint32_t* big; uint32_t *o2;
        const uint32_t NP = 27u * 1000 * 1000;   // 1.7 GB: the h-query table at c = 20
        CHECK(hipMalloc(&big, (size_t)NP * 64)); CHECK(hipMemset(big, 0x11, (size_t)NP * 64));
        CHECK(hipMalloc(&o2, (size_t)256 * 16 * 256 * 36 * 4));
        for (uint32_t np : {4096u, NP}) for (int math : {0, 1}) {
            int blocks = 256 * 8, it = 52;
            float ms = timeit(k_madd_gather, dim3(blocks), dim3(256), (const uint32_t*)big, o2, it, np, math);
            double n = (double)blocks * 256 * it;
            printf("gather table=%u pts math=%d: %.3f ms  %.2f G pts/s  (%.2f TB/s of 128-B lines)\n", np, math, ms, n / ms / 1e6, n * 128 / ms / 1e9);
        }
        CHECK(hipFree(big)); CHECK(hipFree(o2));
    }
    const int NPTS = 4096;
    // table of valid points is not needed for timing: use random canonical-looking data (formulas are branch-free except rare paths)
    std::vector<uint32_t> h(NPTS * 72);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u + 12345u) & 0x0fffffffu;
    uint32_t *d_tab, *d_out, *d_io;
    CHECK(hipMalloc(&d_tab, h.size() * 4)); CHECK(hipMemcpy(d_tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_out, (size_t)256 * 16 * 256 * 72 * 4)); CHECK(hipMalloc(&d_io, (size_t)256 * 16 * 256 * 18 * 4));
    CHECK(hipMemset(d_io, 0x5a, (size_t)256 * 16 * 256 * 18 * 4));
    printf("%-22s %8s %10s %12s %14s\n", "op", "blk/CU", "ms", "Gop/s", "ns/op/lane");
    for (int bpc : {1, 2, 4, 8}) {
        int blocks = 256 * bpc;
        double lanes = (double)blocks * 256;
        { int it = 400; float ms = timeit(k_fq_mul, dim3(blocks), dim3(256), d_io, it); double ops = lanes * it * 2; printf("%-22s %8d %10.3f %12.2f %14.1f\n", "Fq mul", bpc, ms, ops / ms / 1e6, ms * 1e6 / (it * 2)); }
        { int it = 200; float ms = timeit(k_fq2_mul, dim3(blocks), dim3(256), d_io, it); double ops = lanes * it * 2; printf("%-22s %8d %10.3f %12.2f %14.1f\n", "Fq2 mul", bpc, ms, ops / ms / 1e6, ms * 1e6 / (it * 2)); }
        { int it = 64; float ms = timeit(k_madd<Fq29>, dim3(blocks), dim3(256), (const uint32_t*)d_tab, d_out, it, NPTS); printf("%-22s %8d %10.3f %12.2f %14.1f\n", "G1 madd", bpc, ms, lanes * it / ms / 1e6, ms * 1e6 / it); }
        { int it = 32; float ms = timeit(k_madd<Fq2_29>, dim3(blocks), dim3(256), (const uint32_t*)d_tab, d_out, it, NPTS / 2); printf("%-22s %8d %10.3f %12.2f %14.1f\n", "G2 madd", bpc, ms, lanes * it / ms / 1e6, ms * 1e6 / it); }
        { int it = 32; float ms = timeit(k_add<Fq29>, dim3(blocks), dim3(256), (const uint32_t*)d_tab, d_out, it, NPTS / 4); printf("%-22s %8d %10.3f %12.2f %14.1f\n", "G1 add (xyzz)", bpc, ms, lanes * it / ms / 1e6, ms * 1e6 / it); }
        { int it = 16; float ms = timeit(k_add<Fq2_29>, dim3(blocks), dim3(256), (const uint32_t*)d_tab, d_out, it, NPTS / 8); printf("%-22s %8d %10.3f %12.2f %14.1f\n", "G2 add (xyzz)", bpc, ms, lanes * it / ms / 1e6, ms * 1e6 / it); }
    }
    return 0;
}
