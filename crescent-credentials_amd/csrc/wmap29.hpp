// R1CS -> QAP witness map (forks/groth16/src/r1cs_to_qap.rs:150-213) on the lazy 29-bit-limb field
// arithmetic: the transforms of cg_prove and of the unit-level cg_ntt_* entry points.  (ntt.hip keeps the
// saturated-limb domain constants these tables are converted from, and the sparse product of cg_setup.)
//
// Vector elements live in HBM as 32-byte packed canonical values of x·2^261 mod r ("R' form").
// Every transform is decimation-in-time (bit-reversed in, natural out): a DIT butterfly only ever
// multiplies one operand and adds/subtracts, so lazy values grow linearly (+2N per stage) instead of
// doubling.  The permutations DIT needs are free: the producer of each transform's input writes it
// bit-reversed (the sparse product scatters its rows; the last pass of a transform scatters its output).
#pragma once
#include "field29.hpp"
#include "ntt.hpp"

namespace cg {

struct Wm29Domain {
    int logn = 0;
    uint64_t n = 0;
    DevBuf<uint32_t> tw_fwd;     // ω^e   (e < n/2), R' form, nine 29-bit limbs in a 12-word record
    DevBuf<uint32_t> tw_inv;     // ω^-e
    DevBuf<uint32_t> coset;      // g^i / n   at natural index i, R' form      (r1cs_to_qap.rs:182-185 + the 1/n of :179-180)
    DevBuf<uint32_t> icoset;     // g^-i / n  at natural index i, PLAIN form (multiplying by it also leaves Montgomery form)
    uint32_t vinv[8];            // (g^n - 1)^-1, R' form (host copy; passed by value to the kernel)
    uint32_t vinv_plain[8];      // the same as a plain integer
    void build(const NttDomain& d, hipStream_t st);
};

// A shard that owns the coset points j ≡ rank (mod 2^logs) of the h MSM (one proof split over 2^logs GPUs).
// The values it needs, a(g·ω^j) for those j, are a's values on the smaller coset (g·ω^rank)·<ω^(2^logs)> of d = n / 2^logs
// points: a(x) mod (x^d − s^d) with s = g·ω^rank, evaluated there.  With the coefficients a_e in hand (one inverse
// transform of size n, as before) that is
//     a'_i = Σ_t a_{i + t·d} · s^{i + t·d} / n          (i < d; one pass over the vector: k_fold29)
// followed by a plain transform of size d — instead of the second full-size transform.  Per proof a shard then runs two
// transforms of size n and two of size d where a contiguous range of j costs four of size n.
struct Wm29Strided {
    int logs = 0;                // log2 of the shard count
    uint64_t d = 0;              // points owned: n >> logs
    Wm29Domain sub;              // twiddles of the size-d domain (ω^(2^logs))
    DevBuf<uint32_t> fold;       // s^e / n at natural index e < n, R' form, s = g·ω^rank
    void build(const NttDomain& big, int logs, int rank, hipStream_t st);
};

struct Csr29 {     // the matrices' coefficient dictionary in R' form (index arrays are shared with DevCsr)
    DevBuf<uint32_t> dict;
    void build(const DevCsr& m, hipStream_t st);
};

struct Wm29Buffers {     // per proof slot
    DevBuf<uint32_t> w29;            // witness in R' form, M x 8 words
    DevBuf<uint32_t> va, vb, vc, vt; // D x 8 words each (vt: ping-pong partner of the bit-reversing stores)
    DevBuf<uint32_t> sp_a, sp_b;     // partial sums of the sliced sparse product, levels alternating (sp_cap x 8 words)
    uint32_t sp_cap = 0;
    PinnedBuf<uint32_t> h_bad_input; // host memory; a kernel sets it to 1 when a witness element is not a canonical field element
    // sparse_scratch: the largest DevCsr::sell_scratch of the matrices this working set will serve
    void alloc(uint64_t M, uint64_t D, uint32_t sparse_scratch) {
        w29.alloc(M * 8); va.alloc(D * 8); vb.alloc(D * 8); vc.alloc(D * 8); vt.alloc(D * 8);
        sp_cap = sparse_scratch;
        sp_a.alloc((size_t)(sparse_scratch ? sparse_scratch : 1) * 8); sp_b.alloc((size_t)(sparse_scratch ? sparse_scratch : 1) * 8);
        h_bad_input.alloc(1);
    }
    uint64_t device_bytes() const { return w29.bytes() + va.bytes() + vb.bytes() + vc.bytes() + vt.bytes() + sp_a.bytes() + sp_b.bytes(); }
};

// Unit-level transform over the same kernels (cg_ntt_*): 2^logn canonical scalars on the device, natural order in
// and out, in place.  Replaces EvaluationDomain::{fft,ifft}_in_place and their coset forms (ark-poly; call sites
// r1cs_to_qap.rs:179-185,198-199,210).
struct Ntt29Unit {
    int logn = 0;
    uint64_t n = 0;
    Wm29Domain dom;
    DevBuf<uint32_t> gpow;       // g^i, R' form, natural index
    DevBuf<uint32_t> work;       // n x 8 words
    PinnedBuf<uint32_t> h_bad_input;
    uint32_t one_plain[8], ninv_plain[8];
    void build(int logn, hipStream_t st);
    // returns false when an input element was not a canonical field element (data is then unspecified); synchronises st
    bool run(Fr* data_dev, bool inverse, bool coset, hipStream_t st);
};

// w_canon: M canonical scalars on the device.  h_out: D canonical scalars (natural order), on the device:
// the coefficients of h (coset_values = false: the reference's result, seven transforms), or the coset values of the
// quotient's a∘b part, q_j = vinv·a(g·ω^j)·b(g·ω^j) (coset_values = true: four transforms; for a key whose h query is
// held in that basis and whose l query carries the C matrix, msm.hpp).  With `strided` (coset_values only): the d values
// q_j of the shard's points j = rank + k·2^logs, k < d, in the order of k.
void wm29_run(const Wm29Domain& dom, const DevCsr& A, const DevCsr& B, const DevCsr& C, const Csr29& dA, const Csr29& dB,
              const Csr29& dC, Wm29Buffers& buf, const Fr* w_canon, uint64_t M, uint64_t m, uint64_t l, Fr* h_out,
              hipStream_t st, bool coset_values, const Wm29Strided* strided = nullptr, int half = 0);
// half (coset_values only, no `strided`): 1 = the a side alone, vinv·a(g·ω^j); 2 = the b side alone, b(g·ω^j) - D plain
// canonical integers each, natural order, from ONE sparse product and TWO transforms: the two sides are independent until
// the pointwise product, so two GPUs can compute one each (cg_witness_map_coset_half).  q_j = side1_j · side2_j:
void fr_mul_plain29(const Fr* a, const Fr* b, Fr* out, uint64_t n, uint32_t* bad_input_dev, hipStream_t st);

}  // namespace cg
