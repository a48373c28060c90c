// Micro-benchmark: issue rate of the integer/FP64 VALU instructions that BN254 limb arithmetic
// is built from, on gfx950.  Prints wave-instructions per cycle per SIMD (1/cycles-per-instr).
// Usage: ./valu_rates          the round-5 table (wall-clock time, cycles at the nominal clock)
//        ./valu_rates --json   one JSON line for bench.py: cycles per wave-instruction per SIMD of the three instruction
//                              classes the prove path's roofline is priced in, measured in SHADER CYCLES on the device
//                              (clock64 around every wave's own loop), so the figure does not depend on the clock the chip
//                              happens to hold (needs a GPU; built by __graft_entry__.build())
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../crescent-credentials_amd/csrc/field.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 ^ 0x1234567;
    uint64_t d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    double f0 = a0, f1 = a1, f2 = a2, f3 = a3;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // v_mad_u64_u32, 4 independent chains
            REP64(asm volatile("v_mad_u64_u32 %0, s[10:11], %4, %5, %0\n v_mad_u64_u32 %1, s[10:11], %4, %6, %1\n"
                               "v_mad_u64_u32 %2, s[10:11], %5, %6, %2\n v_mad_u64_u32 %3, s[10:11], %6, %7, %3\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s10", "s11");)
        } else if (KIND == 1) {  // v_mul_lo_u32
            REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
        } else if (KIND == 2) {  // v_mul_hi_u32
            REP64(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
        } else if (KIND == 3) {  // v_add_co_u32 + v_addc_co_u32 chain
            REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) : "vcc");)
        } else if (KIND == 4) {  // v_lshl_add_u64
            REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
        } else if (KIND == 5) {  // v_mov_b32
            REP64(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 6) {  // v_mad_u32_u24
            REP64(asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
        } else if (KIND == 7) {  // v_mul_hi_u32_u24
            REP64(asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
        } else if (KIND == 8) {  // v_fma_f64
            REP64(asm volatile("v_fma_f64 %0, %0, %4, %1\n v_fma_f64 %1, %1, %4, %2\n v_fma_f64 %2, %2, %4, %3\n v_fma_f64 %3, %3, %4, %0\n"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0000001));)
        } else if (KIND == 9) {  // v_add3_u32
            REP64(asm volatile("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
        } else if (KIND == 10) {  // v_mad_u64_u32 with carry-out consumed by v_addc_co_u32
            REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n"
                               "v_mad_u64_u32 %1, vcc, %5, %4, %1\n v_addc_co_u32 %3, vcc, 0, %3, vcc\n"
                               : "+v"(d0), "+v"(d1), "+v"(a2), "+v"(a3) : "v"(a0), "v"(a1) : "vcc");)
        } else if (KIND == 11) {  // v_mul_u32_u24
            REP64(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
        } else if (KIND == 13) {  // v_mad_i64_i32 (the signed product of round 5's G1 accumulation)
            REP64(asm volatile("v_mad_i64_i32 %0, s[10:11], %4, %5, %0\n v_mad_i64_i32 %1, s[10:11], %4, %6, %1\n"
                               "v_mad_i64_i32 %2, s[10:11], %5, %6, %2\n v_mad_i64_i32 %3, s[10:11], %6, %7, %3\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s10", "s11");)
        } else if (KIND == 14) {  // v_ashrrev_i64
            REP64(asm volatile("v_ashrrev_i64 %0, 29, %0\n v_ashrrev_i64 %1, 29, %1\n v_ashrrev_i64 %2, 29, %2\n v_ashrrev_i64 %3, 29, %3\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
        } else if (KIND == 15) {  // v_lshrrev_b64
            REP64(asm volatile("v_lshrrev_b64 %0, 29, %0\n v_lshrrev_b64 %1, 29, %1\n v_lshrrev_b64 %2, 29, %2\n v_lshrrev_b64 %3, 29, %3\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
        } else if (KIND == 16) {  // v_xad_u32
            REP64(asm volatile("v_xad_u32 %0, %0, %4, %1\n v_xad_u32 %1, %1, %4, %2\n v_xad_u32 %2, %2, %4, %3\n v_xad_u32 %3, %3, %4, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
        } else if (KIND == 17) {  // v_and_b32 (VOP2)
            REP64(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 18) {  // v_mad_u64_u32 with an SGPR multiplier (the constant limbs of N)
            REP64(asm volatile("v_mad_u64_u32 %0, s[10:11], %4, s12, %0\n v_mad_u64_u32 %1, s[10:11], %5, s12, %1\n"
                               "v_mad_u64_u32 %2, s[10:11], %6, s12, %2\n v_mad_u64_u32 %3, s[10:11], %7, s12, %3\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s10", "s11", "s12");)
        } else if (KIND == 19) {  // a dependent chain: ONE accumulator, as in a product's column (4 waves per SIMD hide it or not)
            REP64(asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n"
                               "v_mad_u64_u32 %0, s[10:11], %3, %4, %0\n v_mad_u64_u32 %0, s[10:11], %4, %1, %0\n"
                               : "+v"(d0) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s10", "s11");)
        } else if (KIND == 20) {  // the same chain signed
            REP64(asm volatile("v_mad_i64_i32 %0, s[10:11], %1, %2, %0\n v_mad_i64_i32 %0, s[10:11], %2, %3, %0\n"
                               "v_mad_i64_i32 %0, s[10:11], %3, %4, %0\n v_mad_i64_i32 %0, s[10:11], %4, %1, %0\n"
                               : "+v"(d0) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s10", "s11");)
        } else if (KIND == 12) {  // v_cndmask_b32
            REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(d0 + d1 + d2 + d3) + (uint32_t)(f0 + f1 + f2 + f3);
}

// Fq Montgomery products per second with the library's own mul() (4 independent chains / lane)
template <class F>
__global__ void __launch_bounds__(256) k_fpmul(F* io, int iters) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    F a = io[t], b = io[t + gridDim.x * blockDim.x];
    F c = cg::add(a, b), d = cg::sub(a, b);
    for (int i = 0; i < iters; ++i) {
        a = cg::mul(a, b); b = cg::mul(b, c); c = cg::mul(c, d); d = cg::mul(d, a);
    }
    io[t] = cg::add(cg::add(a, b), cg::add(c, d));
}

// The same loops timed by every wave itself, in shader cycles.  1024 workgroups of four waves are one wave per SIMD slot
// four deep on 256 CUs x 4 SIMDs - all resident at once, all issuing the same instruction stream - so a SIMD issues
// (4 waves x N instructions) in the T cycles a wave sees: cycles per wave-instruction = T / (4 N).
template <int KIND>
__global__ void __launch_bounds__(256) k_rate_cycles(uint32_t* out, unsigned long long* cycles, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 ^ 0x1234567;
    uint64_t d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            REP64(asm volatile("v_mad_u64_u32 %0, s[10:11], %4, %5, %0\n v_mad_u64_u32 %1, s[10:11], %4, %6, %1\n"
                               "v_mad_u64_u32 %2, s[10:11], %5, %6, %2\n v_mad_u64_u32 %3, s[10:11], %6, %7, %3\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s10", "s11");)
        } else if (KIND == 1) {
            REP64(asm volatile("v_mad_i64_i32 %0, s[10:11], %4, %5, %0\n v_mad_i64_i32 %1, s[10:11], %4, %6, %1\n"
                               "v_mad_i64_i32 %2, s[10:11], %5, %6, %2\n v_mad_i64_i32 %3, s[10:11], %6, %7, %3\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s10", "s11");)
        } else if (KIND == 2) {
            REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) : "vcc");)
        } else if (KIND == 3) {
            REP64(asm volatile("v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
        } else if (KIND == 4) {
            REP64(asm volatile("v_lshrrev_b64 %0, 29, %0\n v_lshrrev_b64 %1, 29, %1\n v_lshrrev_b64 %2, 29, %2\n v_lshrrev_b64 %3, 29, %3\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
        } else if (KIND == 5) {
            REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
        } else if (KIND == 6) {
            REP64(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 7) {
            REP64(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 8) {
            REP64(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 9) {
            REP64(asm volatile("v_lshrrev_b32 %0, 3, %1\n v_lshrrev_b32 %1, 3, %2\n v_lshrrev_b32 %2, 3, %3\n v_lshrrev_b32 %3, 3, %0\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 10) {      // ONE dependent chain of multiply-adds: a product's column sum as field29.hpp writes it
            REP64(asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n"
                               "v_mad_u64_u32 %0, s[10:11], %3, %4, %0\n v_mad_u64_u32 %0, s[10:11], %4, %1, %0\n"
                               : "+v"(d0) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s10", "s11");)
        } else if (KIND == 11) {      // TWO interleaved chains
            REP64(asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %3, %0\n v_mad_u64_u32 %1, s[10:11], %3, %4, %1\n"
                               "v_mad_u64_u32 %0, s[10:11], %4, %5, %0\n v_mad_u64_u32 %1, s[10:11], %5, %2, %1\n"
                               : "+v"(d0), "+v"(d1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s10", "s11");)
        } else if (KIND == 12) {      // ONE dependent chain of simple 32-bit adds
            REP64(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %3\n v_add_u32 %0, %0, %1\n"
                               : "+v"(a0) : "v"(a1), "v"(a2), "v"(a3));)
        }
    }
    const unsigned long long c1 = clock64();
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = c1 - c0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(d0 + d1 + d2 + d3);
}

template <int KIND>
double cycles_per_instr(uint32_t* d_out, unsigned long long* d_cyc, double* wall_ms, int waves_per_simd = 4) {
    const int blocks = 256 * waves_per_simd, threads = 256, iters = 40, waves = blocks * threads / 64;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_rate_cycles<KIND><<<blocks, threads>>>(d_out, d_cyc, 4, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_rate_cycles<KIND><<<blocks, threads>>>(d_out, d_cyc, iters, 1);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (wall_ms) *wall_ms = ms;
    std::vector<unsigned long long> h(waves);
    CHECK(hipMemcpy(h.data(), d_cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto c : h) sum += (double)c;
    const double n_instr = (double)iters * 64.0 * 4.0;          // per wave
    return (sum / waves) / ((double)waves_per_simd * n_instr);
}

// The same multiply-add loop, long: ~60 ms of nothing but v_mad_u64_u32 on every SIMD, eight waves deep.  Its wall time gives
// the rate the chip SUSTAINS on this arithmetic, and its own cycle count over that wall time the clock it holds while doing
// so - the power limit under a dense integer-multiply load, which is what bounds the prove path (the clock measured during the
// proofs is higher only because their instruction mix is lighter).
static void sustained_mad(uint32_t* d_out, unsigned long long* d_cyc, double* g_per_s, double* clock_ghz, double* wall_ms_out) {
    const int blocks = 256 * 8, threads = 256, waves = blocks * threads / 64;
    uint32_t* out2; CHECK(hipMalloc(&out2, (size_t)blocks * threads * 4));
    unsigned long long* cyc2; CHECK(hipMalloc(&cyc2, (size_t)waves * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int iters = 2000;
    k_rate_cycles<0><<<blocks, threads>>>(out2, cyc2, 200, 1);          // warm: the clock settles under the load
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {                                  // the second, longer run is the one reported
        CHECK(hipEventRecord(e0));
        k_rate_cycles<0><<<blocks, threads>>>(out2, cyc2, iters, 1);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0 && ms < 40.f) iters = (int)(iters * 60.0 / (ms > 1e-3f ? ms : 1.f));
    }
    std::vector<unsigned long long> h(waves);
    CHECK(hipMemcpy(h.data(), cyc2, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto c : h) sum += (double)c;
    const double n_instr = (double)iters * 64.0 * 4.0;
    *g_per_s = (double)waves * n_instr / (ms * 1e-3) / 1e9;
    *clock_ghz = (sum / waves) / (ms * 1e-3) / 1e9;                       // a wave's own cycles over (nearly) the whole launch
    *wall_ms_out = ms;
    (void)d_out; (void)d_cyc;
    CHECK(hipFree(out2)); CHECK(hipFree(cyc2));
}

static int json_mode() {
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    uint32_t* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
    unsigned long long* d_cyc; CHECK(hipMalloc(&d_cyc, 256 * 8 * 4 * sizeof(unsigned long long)));
    double w_mad = 0;
    const double mad_u = cycles_per_instr<0>(d_out, d_cyc, &w_mad), mad_i = cycles_per_instr<1>(d_out, d_cyc, nullptr);
    const double addc = cycles_per_instr<2>(d_out, d_cyc, nullptr), add3 = cycles_per_instr<3>(d_out, d_cyc, nullptr);
    const double shr64 = cycles_per_instr<4>(d_out, d_cyc, nullptr), lshladd64 = cycles_per_instr<5>(d_out, d_cyc, nullptr);
    const double mov = cycles_per_instr<6>(d_out, d_cyc, nullptr), and32 = cycles_per_instr<7>(d_out, d_cyc, nullptr);
    const double add32 = cycles_per_instr<8>(d_out, d_cyc, nullptr), shr32 = cycles_per_instr<9>(d_out, d_cyc, nullptr);
    // how the issue rate depends on the waves a SIMD holds and on the independent chains a wave offers: a dependent instruction
    // cannot issue before its predecessor's result is there, so a wave with ONE chain issues once per latency, and a SIMD
    // needs latency / issue-cycles such waves (or chains) to stay busy
    double dep[3][4];
    const int wps[4] = {1, 2, 4, 8};
    for (int i = 0; i < 4; ++i) {
        dep[0][i] = cycles_per_instr<10>(d_out, d_cyc, nullptr, wps[i]);
        dep[1][i] = cycles_per_instr<11>(d_out, d_cyc, nullptr, wps[i]);
        dep[2][i] = cycles_per_instr<0>(d_out, d_cyc, nullptr, wps[i]);
    }
    const double add_chain_1 = cycles_per_instr<12>(d_out, d_cyc, nullptr, 1), add_chain_4 = cycles_per_instr<12>(d_out, d_cyc, nullptr, 4);
    double sus_g = 0, sus_clk = 0, sus_ms = 0;
    sustained_mad(d_out, d_cyc, &sus_g, &sus_clk, &sus_ms);
    (void)w_mad;
    printf("{\"cus\": %d, \"sustained_mad64\": {\"G_wave_instr_per_s\": %.1f, \"clock_ghz\": %.3f, \"wall_ms\": %.1f, "
           "\"cycles_per_wave_instr\": %.3f, \"what\": \"v_mad_u64_u32 only, eight waves per SIMD on every SIMD, one launch of that length: the rate "
           "the chip sustains on dense 32x32+64 multiply-adds and the shader clock (clock64 over wall time) it holds meanwhile\"}, ", ncu, sus_g,
           sus_clk, sus_ms, sus_clk * 1e9 * 1024.0 / (sus_g * 1e9));
    printf("\"mad64_cycles_by_chains_and_waves_per_simd\": {\"waves_per_simd\": [1, 2, 4, 8], \"one_chain\": [%.3f, %.3f, %.3f, %.3f], "
           "\"two_chains\": [%.3f, %.3f, %.3f, %.3f], \"four_chains\": [%.3f, %.3f, %.3f, %.3f], \"add32_one_chain_1_and_4_waves\": [%.3f, %.3f]}, ",
           dep[0][0], dep[0][1], dep[0][2], dep[0][3], dep[1][0], dep[1][1], dep[1][2], dep[1][3], dep[2][0], dep[2][1], dep[2][2], dep[2][3],
           add_chain_1, add_chain_4);
    printf("\"cycles_per_wave_instr\": {\"mad64\": %.3f, \"other\": %.3f, \"simple32\": %.3f}, "
           "\"per_instruction\": {\"v_mad_u64_u32\": %.3f, \"v_mad_i64_i32\": %.3f, \"v_add_co_u32+v_addc_co_u32\": %.3f, \"v_add3_u32\": %.3f, "
           "\"v_lshrrev_b64\": %.3f, \"v_lshl_add_u64\": %.3f, \"v_mov_b32\": %.3f, \"v_and_b32\": %.3f, \"v_add_u32\": %.3f, \"v_lshrrev_b32\": %.3f}, "
           "\"how\": \"shader cycles (clock64) around each wave's own loop of 10240 instructions, 4 waves per SIMD resident on every SIMD; "
           "mad64 = mean of the two multiply-adds, other = mean of the four VOP3 / 64-bit operations, simple32 = mean of the four "
           "32-bit VOP1 / VOP2 operations\"}\n",
           (mad_u + mad_i) / 2, (addc + add3 + shr64 + lshladd64) / 4, (mov + and32 + add32 + shr32) / 4, mad_u, mad_i, addc, add3, shr64,
           lshladd64, mov, and32, add32, shr32);
    return 0;
}

template <int KIND>
void run(const char* name, int per_iter, uint32_t* d_out, double clk_hz) {
    const int blocks = 256 * 8, threads = 256, iters = 200;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_rate<KIND><<<blocks, threads>>>(d_out, 10, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_rate<KIND><<<blocks, threads>>>(d_out, iters, 1);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double wave_instr = (double)blocks * (threads / 64) * iters * 64.0 * per_iter;
    double per_simd_per_s = wave_instr / (ms * 1e-3) / (256.0 * 4);
    printf("%-28s %8.3f ms  %7.2f G wave-instr/s/SIMD-sum  cycles/wave-instr/SIMD @%.2f GHz = %.2f\n",
           name, ms, wave_instr / (ms * 1e-3) / 1e9, clk_hz / 1e9, clk_hz / per_simd_per_s);
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "--json")) return json_mode();
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    double clk = p.clockRate * 1e3;
    printf("device %s CUs %d clock %.0f MHz\n", p.name, p.multiProcessorCount, clk / 1e6);
    uint32_t* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
    run<0>("v_mad_u64_u32", 4, d_out, clk);
    run<10>("v_mad_u64_u32+v_addc_co", 4, d_out, clk);
    run<1>("v_mul_lo_u32", 4, d_out, clk);
    run<2>("v_mul_hi_u32", 4, d_out, clk);
    run<3>("v_add_co/addc_co", 4, d_out, clk);
    run<4>("v_lshl_add_u64", 4, d_out, clk);
    run<5>("v_mov_b32", 4, d_out, clk);
    run<6>("v_mad_u32_u24", 4, d_out, clk);
    run<7>("v_mul_hi_u32_u24", 4, d_out, clk);
    run<11>("v_mul_u32_u24", 4, d_out, clk);
    run<8>("v_fma_f64", 4, d_out, clk);
    run<9>("v_add3_u32", 4, d_out, clk);
    run<12>("v_cndmask_b32", 4, d_out, clk);
    run<13>("v_mad_i64_i32", 4, d_out, clk);
    run<14>("v_ashrrev_i64", 4, d_out, clk);
    run<15>("v_lshrrev_b64", 4, d_out, clk);
    run<16>("v_xad_u32", 4, d_out, clk);
    run<17>("v_and_b32", 4, d_out, clk);
    run<18>("v_mad_u64_u32 sgpr operand", 4, d_out, clk);
    run<19>("v_mad_u64_u32 one chain", 4, d_out, clk);
    run<20>("v_mad_i64_i32 one chain", 4, d_out, clk);
    // field mul throughput
    {
        const int blocks = 256 * 8, threads = 256, iters = 2000;
        cg::Fq* io; CHECK(hipMalloc(&io, sizeof(cg::Fq) * blocks * threads * 2));
        std::vector<cg::Fq> h(blocks * threads * 2);
        for (size_t i = 0; i < h.size(); ++i) for (int k = 0; k < 8; ++k) h[i].l[k] = (uint32_t)(i * 2654435761u + k * 40503u) & (k == 7 ? 0x1fffffffu : 0xffffffffu);
        CHECK(hipMemcpy(io, h.data(), sizeof(cg::Fq) * h.size(), hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k_fpmul<cg::Fq><<<blocks, threads>>>(io, 10);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k_fpmul<cg::Fq><<<blocks, threads>>>(io, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double muls = (double)blocks * threads * iters * 4;
        printf("Fq mont mul: %.3f ms, %.2f G mul/s  (%.1f cycles/wave-mul/SIMD)\n", ms, muls / (ms * 1e-3) / 1e9,
               clk / (muls / 64 / (ms * 1e-3) / 1024));
    }
    return 0;
}
