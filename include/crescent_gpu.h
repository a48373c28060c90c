/*
 * crescent_gpu.h — C ABI of the MI355X-native Groth16 prover hot path for Crescent.
 *
 * This is the drop-in boundary for `forks/groth16` of microsoft/crescent-credentials: every entry
 * point names the reference interface it replaces (paths relative to the reference root).  The
 * reference has no FFI today (Rust generics only); the seams are
 *   - the `QAP: R1CSToQAP` type parameter      forks/groth16/src/lib.rs:55, r1cs_to_qap.rs:49-98
 *   - the three `msm_bigint` call sites         forks/groth16/src/prover.rs:66,74,266
 *   - the caller                                creds/src/lib.rs:283 (Groth16::<Bn254>::prove)
 * INTEGRATION.md shows the Rust `extern "C"` block + shim a maintainer would add.
 *
 * Conventions
 *   - All field elements cross the boundary as 32-byte little-endian integers.
 *     CG_FORM_CANONICAL : the plain integer in [0, p)   (what ark-serialize files contain and what
 *                         `into_bigint()` yields, prover.rs:64,71,86)
 *     CG_FORM_MONTGOMERY: x * 2^256 mod p, i.e. arkworks' in-memory Fp256<MontBackend<_,4>> limbs
 *                         (KAT: forks/circom-compat/src/zkey.rs:397-402)
 *   - G1 affine point  = x ‖ y                  (64 bytes);   identity = 64 zero bytes
 *   - G2 affine point  = x.c0 ‖ x.c1 ‖ y.c0 ‖ y.c1 (128 bytes); identity = 128 zero bytes
 *     (same component order as snarkjs/arkworks, zkey.rs:421-431)
 *   - Scalars (witness, r, s) are always CG_FORM_CANONICAL.
 *   - Every function returns CG_OK (0) or a negative cg_status; cg_last_error() returns a
 *     thread-local human-readable message for the last failure on the calling thread.
 *   - The caller owns every host buffer; the library copies what it keeps.
 *   - A cg_ctx is bound to one GPU.  Calls on different contexts are independent and may run
 *     concurrently from different threads; calls on one context take one of its `proof_slots`
 *     working sets each (blocking while none is free), so up to that many proofs overlap on the GPU.
 */
#ifndef CRESCENT_GPU_H
#define CRESCENT_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cg_status {
    CG_OK = 0,
    CG_ERR_INVALID_ARGUMENT = -1,
    CG_ERR_NO_DEVICE = -2,            /* no HIP device / HIP runtime failure at init */
    CG_ERR_HIP = -3,                  /* a HIP call failed; see cg_last_error() */
    CG_ERR_OUT_OF_MEMORY = -4,
    CG_ERR_POLY_DEGREE_TOO_LARGE = -5, /* SynthesisError::PolynomialDegreeTooLarge, r1cs_to_qap.rs:156-157 */
    CG_ERR_MALFORMED_KEY = -6,        /* query lengths inconsistent with the circuit (SynthesisError::MalformedVerifyingKey analogue) */
    CG_ERR_PARSE = -7,                /* .r1cs / serialized key parse failure */
    CG_ERR_UNSATISFIED = -8           /* witness does not satisfy the R1CS (debug check only) */
} cg_status;

enum { CG_FORM_CANONICAL = 0, CG_FORM_MONTGOMERY = 1 };

typedef struct cg_ctx cg_ctx;

/* Packed arrays of a Groth16 proving key.
 * Replaces: `ProvingKey<E>` / `VerifyingKey<E>` operands of the prover,
 *           forks/groth16/src/data_structures.rs:31-44,101-118.
 * Lengths must be (generator.rs:140,162,168,174-179,185):
 *   a_query, b_g1_query, b_g2_query : num_variables
 *   l_query                         : num_variables - num_inputs
 *   h_query                         : domain_size - 1   (domain_size = next pow2 >= m + num_inputs) */
typedef struct cg_proving_key {
    uint32_t coord_form;        /* CG_FORM_CANONICAL or CG_FORM_MONTGOMERY for every coordinate below */
    const uint8_t* alpha_g1;    /* vk.alpha_g1  64 B */
    const uint8_t* beta_g1;     /* pk.beta_g1   64 B */
    const uint8_t* delta_g1;    /* pk.delta_g1  64 B */
    const uint8_t* beta_g2;     /* vk.beta_g2  128 B */
    const uint8_t* delta_g2;    /* vk.delta_g2 128 B */
    const uint8_t* a_query;     uint64_t a_len;      /* G1 */
    const uint8_t* b_g1_query;  uint64_t b_g1_len;   /* G1 */
    const uint8_t* b_g2_query;  uint64_t b_g2_len;   /* G2 */
    const uint8_t* h_query;     uint64_t h_len;      /* G1 */
    const uint8_t* l_query;     uint64_t l_len;      /* G1 */
} cg_proving_key;

/* One R1CS matrix in CSR form.
 * Replaces: one of `ConstraintMatrices<F>::{a,b,c}` (Vec<Vec<(F, usize)>>),
 *           operand of forks/groth16/src/prover.rs:26-33 / r1cs_to_qap.rs:150-155.
 * Row i holds terms [row_ptr[i], row_ptr[i+1]); column = variable index (instance vars first,
 * forks/circom-compat/src/circom/circuit.rs:61-67). */
typedef struct cg_csr {
    const uint64_t* row_ptr;   /* num_constraints + 1 entries */
    const uint32_t* col;       /* nnz entries, each < num_variables */
    const uint8_t* coeff;      /* nnz x 32 B, CG_FORM_CANONICAL */
    uint64_t nnz;
} cg_csr;

typedef struct cg_options {
    int32_t device;        /* HIP device ordinal; -1 = current device */
    int32_t window_bits;   /* Pippenger window c; 0 = library default for the size */
    int32_t shard_rank;    /* multi-GPU MSM range sharding: this context's rank ... */
    int32_t shard_count;   /* ... of shard_count; 0 or 1 = unsharded.  A shard with proof_slots > 1 keeps that many
                              sharded proofs in flight (cg_prove_partial from as many threads) */
    int32_t proof_slots;   /* proofs that may be in flight on this context at once (each has its own working
                              set and streams; cg_prove* from different threads overlap on the GPU); 0 = 1, at
                              most 16.  1 = a latency context (five streams per proof); more = a throughput
                              context (one stream per proof; twelve reach the full rate at the rs256 size) */
    int32_t flags;         /* CG_FLAG_* (the library reads none of ITS OWN switches from the environment: what a host may choose is here) */
    int32_t hw_queues;     /* the hardware queues the host's HIP runtime maps its streams onto (the runtime's own GPU_MAX_HW_QUEUES,
                              read once when HIP initialises; default 4), if the host knows; 0 = the library looks at that variable:
                              cg_init sets it to 20 when the host has not - which only takes effect if HIP was not initialised before
                              cg_init - and a throughput context then shares four copy-only streams among its upload buffers.  A host that
                              initialises HIP first and cannot export the variable passes the real count here (4 unless it chose
                              otherwise): with fewer queues than proof_slots + 4 every upload buffer keeps a stream of its own */
    int32_t shard_span;    /* sharded contexts: 0 = this shard owns the shard_rank-th of shard_count EQUAL parts of every query.
                              Otherwise lo | hi << 16, both in 1/10000 of a query's length (0 <= lo < hi <= 10000): the shard owns the
                              entries [n·lo/10000, n·hi/10000) of each of the five queries (of the h query: that contiguous range
                              of its coset points).  The spans of a proof's shards must tile [0, 10000] - the host's business, as
                              the ranks are.  For hosts that give unequal shares: the ranks that also run (half of) the witness map
                              take a smaller share of the MSMs.  cg_h_scalars_slice then answers for the
                              context's own shard only */
} cg_options;

/* By default cg_circuit_load moves the h query into the evaluation basis of the coset and folds the C matrix into
 * the l query (two inverse DFTs over the h query's G1 points, ~1.6 s at 2^21), so that a proof needs four transforms
 * instead of seven and no sparse product with C: Σ h_i·H_i of prover.rs:63-66 is computed as Σ q_j·H'_j over the coset
 * values q_j = a·b/Z plus a per-wire term carried by the l query — the same group elements, hence the same proof bytes
 * for any assignment.  This flag keeps both queries as loaded and runs the reference's seven transforms
 * (r1cs_to_qap.rs:179-210) per proof. */
enum {
    CG_FLAG_H_COEFFICIENT_BASIS = 1,
    /* How a context arranges a proof's work on the GPU.  By default proof_slots decides: one slot = a LATENCY context (the
     * five MSMs and the witness map on five streams, short accumulation segments, tree reductions, callers spin while
     * they wait), several = a THROUGHPUT context (one stream per proof, long segments, fewest instructions, callers
     * sleep-poll).  A host may force either - e.g. throughput kernels for a single slot that shares the GPU with other
     * contexts, or the latency arrangement for two slots.  Both set = CG_ERR_INVALID_ARGUMENT. */
    CG_FLAG_LATENCY_MODE = 2,
    CG_FLAG_THROUGHPUT_MODE = 4,
    /* Calling threads spin in the HIP runtime while they wait for their proof instead of sleep-polling (a throughput
     * context's default): lowest wake-up latency, one busy CPU per proof in flight. */
    CG_FLAG_SPIN_WAIT = 8,
    /* Sharded contexts (shard_count a power of two): own a contiguous range of the h query's coset points instead of
     * the points j = shard_rank (mod shard_count); the shard then runs four full-size transforms per proof. */
    CG_FLAG_CONTIGUOUS_H_SHARDS = 16,
    /* Sharded contexts over a folded key (SURVEY 8e's other arrangement: "run the witness map on GPU 0 and scatter h"): this
     * shard never runs the witness map - its h scalars arrive with every proof (cg_prove_partial_q), computed once for all
     * shards by cg_witness_map_coset on a context loaded WITHOUT this flag.  The context then holds no witness-map vectors
     * and no matrices; cg_prove_partial / cg_witness_map on it are CG_ERR_INVALID_ARGUMENT. */
    CG_FLAG_H_SCALARS_EXTERNAL = 32,
    /* Staged load: cg_circuit_load returns as soon as the context can PROVE, and finishes loading behind the first proofs.
     * The reference's caller reads its two files and proves once (`create_client_state`, creds/src/lib.rs:255-301; the sample
     * client does so per credential, sample/client_helper/src/main.rs:177-216), so what it waits for is load + ONE proof, and
     * two thirds of a synchronous load are spent on tables that only pay off over many proofs (the change of basis of the
     * h query, ~1.5 s at 2^21, and the per-window tables, ~0.6 s).  With this flag the load copies the key and the matrices,
     * keeps every query as its row-0 table and returns; proofs then run in the WARM-UP arrangement - the reference's own
     * (seven transforms, plain h query, r1cs_to_qap.rs:179-210) over classic Pippenger with one bucket set per window - while
     * a worker thread of the library builds the final arrangement (folded key, per-window tables, windows chosen from the
     * first finished proof's digit statistics when there is one) and swaps it in between two proofs.  Proof bytes are the
     * same in both arrangements (they are the same group elements).  cg_ctx_wait_ready blocks until the swap; a host that
     * never calls it loses nothing but the first seconds' throughput.  Honoured for unsharded contexts over the folded key;
     * sharded contexts and CG_FLAG_H_COEFFICIENT_BASIS load synchronously as before. */
    CG_FLAG_STAGED_LOAD = 64,
    /* A THROUGHPUT context with more than one slot holds up to two slots more than proof_slots: the LONE slots, arranged as a
     * latency context's (five streams, short segments).  A proof that arrives when fewer than two proofs of the context are in
     * flight - a server between requests: the sample client proves one credential per task,
     * sample/client_helper/src/main.rs:177-216 - runs on one of them (at 2^21 from pageable memory: 7.7 ms instead of 10.9 alone,
     * 13.8 instead of 15.3-16.4 in a pair); proofs that find more in flight take the one-stream slots as before, so the steady
     * rate is untouched.  Costs a latency slot's memory each (cg_ctx_info.lone_slot_bytes is their sum; a device without room
     * for them does without) and no stream: they run on streams of the last one-stream slots.  A call that asks for cg_timings
     * always runs on a one-stream slot: its phases are then stand-alone durations that add up.  This flag leaves them out. */
    CG_FLAG_NO_LONE_SLOT = 128
};

/* Per-phase wall/GPU times of one cg_prove call, mirroring the reference's `print-trace` phases
 * (forks/groth16/src/prover.rs:35-36,62,93,103,115,123). Milliseconds. */
typedef struct cg_timings {
    float upload_ms;       /* host -> device copy of the assignment (0 for *_dev entry points) */
    float witness_map_ms;  /* "R1CS to QAP witness map" */
    float msm_h_ms;        /* "Compute C": h_query MSM */
    float msm_l_ms;        /* "Compute C": l_query MSM */
    float msm_a_ms;        /* "Compute A" */
    float msm_b1_ms;       /* "Compute B in G1" */
    float msm_b2_ms;       /* "Compute B in G2" */
    float finish_ms;       /* "Finish C" + affine normalisation + serialisation (host) */
    float total_ms;        /* "Groth16::Prover" */
    uint64_t msm_g1_pairs; /* (base, scalar) pairs consumed by the four G1 MSMs */
    uint64_t msm_g2_pairs;
    /* HIP-event durations of the dominant kernels, summed over this proof's launches */
    float accum_g1_ms;     /* bucket-accumulation kernel (k_accum_affine<Fq>), the four G1 MSMs */
    float accum_g2_ms;     /* same kernel over Fq2 (the G2 MSM) */
    float sort_ms;         /* grouping the digit entries by bucket (counting partition), all five MSMs */
    float reserved_ms;
    uint64_t entries_g1;   /* non-zero signed digits (= mixed additions) accumulated, G1 */
    uint64_t entries_g2;
    uint32_t accum_g1_launches;
    uint32_t accum_g2_launches;
} cg_timings;

/* Process-wide initialisation: checks that a HIP device is present.
 * n_devices/device_ids may be 0/NULL (use whatever is visible). */
int cg_init(int n_devices, const int* device_ids);

/* The GPU that entry points WITHOUT a device of their own run on when called from this thread afterwards: cg_setup,
 * cg_msm_g1/g2, cg_ntt, cg_fixed_base_*, and the loaders given device -1.  (Contexts carry their device and switch to
 * it on every call.)  A host with one process per GPU calls this once with its local rank; the reference has no
 * counterpart (it has no devices). */
int cg_set_device(int32_t device);

/* Thread-local description of the last error on this thread ("" if none). */
const char* cg_last_error(void);

/* Load a circuit: copy the proving key and the three constraint matrices to the GPU, build the
 * per-window base tables and the NTT tables.  One-time per circuit.
 * Replaces: `read_from_file::<ProverParams>` + `CircomConfig::new` products as consumed by the
 *           prover (creds/src/lib.rs:258,268), and `cs.to_matrices()` (r1cs_to_qap.rs:62).
 * num_inputs (ℓ) counts the constant-one variable; num_variables (M) = instance + witness. */
int cg_circuit_load(cg_ctx** out, const cg_proving_key* pk, const cg_csr abc[3],
                    uint64_t num_inputs, uint64_t num_constraints, uint64_t num_variables,
                    const cg_options* opt);

/* Waits for every call that is still inside the context (cg_prove*, cg_prove_partial, cg_assemble, cg_witness_map: their GPU
 * work AND their host tails) and then frees it.  No call may START on the context once this one has. */
void cg_circuit_free(cg_ctx* ctx);

/* Create one Groth16 proof.
 * Replaces: `Groth16::<E,QAP>::create_proof_with_reduction_and_matrices`,
 *           forks/groth16/src/prover.rs:26-51 (and through it :54-136, :256-274 and
 *           r1cs_to_qap.rs:150-213).
 * full_assignment: num_variables x 32 B canonical (instance ‖ witness, full_assignment[0] = 1).
 * r, s: 32 B canonical each (prover.rs:150-151 samples them; r = s = 0 gives the no-zk proof,
 *       prover.rs:160-173).
 * proof_out: 256 B = ark-serialize uncompressed a ‖ b ‖ c (data_structures.rs:7-14).
 * timings may be NULL.
 * A failed one-time window re-tune never fails the proof it follows; if it ran out of device memory while the proof
 * slots were being re-sized, every LATER proof on the context returns CG_ERR_OUT_OF_MEMORY (free and reload). */
int cg_prove(cg_ctx* ctx, const uint8_t* full_assignment, const uint8_t r[32], const uint8_t s[32],
             uint8_t proof_out[256], cg_timings* timings);

/* Same, with the assignment already resident in this context's GPU memory
 * (d_full_assignment is a device pointer, num_variables x 32 B canonical). */
int cg_prove_dev(cg_ctx* ctx, const void* d_full_assignment, const uint8_t r[32], const uint8_t s[32],
                 uint8_t proof_out[256], cg_timings* timings);

/* Page-locked host memory for assignments.
 * Replaces: the allocation behind `full_assignment: &[E::ScalarField]` (forks/groth16/src/prover.rs:33; built as
 *           `[instance_assignment, witness_assignment].concat()` at r1cs_to_qap.rs:68-72 from what the WASM witness
 *           calculator returned, forks/circom-compat/src/circom/builder.rs:71-98).  cg_prove accepts ANY
 *           host pointer; from pageable memory the 32·num_variables-byte upload is staged by the HIP runtime inside the call,
 *           from page-locked memory it is one asynchronous DMA that overlaps the other proofs in flight.  A host that
 *           can choose where the witness calculator writes its output takes the buffer from cg_host_alloc; a host that
 *           cannot may pin its own buffer once with cg_host_register (and must cg_host_unregister it before freeing it).
 * cg_host_alloc returns NULL on failure (cg_last_error says why). */
void* cg_host_alloc(uint64_t bytes);
void cg_host_free(void* p);
int cg_host_register(void* p, uint64_t bytes);
int cg_host_unregister(void* p);

/* What a loaded circuit occupies and how its MSMs are configured.  The reference's prover has no counterpart (its key is
 * a Vec in host memory, data_structures.rs:101-118); a host sizing `proof_slots`, or running several circuits on one GPU,
 * needs the numbers, and a host that relies on the one-time window re-tune (see cg_options.window_bits) needs to know
 * whether it happened. */
typedef struct cg_ctx_info {
    uint64_t table_bytes;        /* per-window base tables of the five queries + validity flags (shared by all slots) */
    uint64_t matrix_bytes;       /* resident constraint matrices + NTT / coset tables */
    uint64_t slot_bytes;         /* ONE proof slot's working set; the context holds proof_slots of them */
    uint64_t total_bytes;        /* table_bytes + matrix_bytes + proof_slots x slot_bytes + lone_slot_bytes */
    uint64_t device_free_bytes;  /* hipMemGetInfo at the time of the call */
    uint64_t device_total_bytes;
    int32_t proof_slots;
    int32_t window_bits[5];      /* current window of the h, l, a, b_g1, b_g2 MSMs */
    int32_t tuned;               /* 1 once the assignment-driven windows were re-chosen from a proof's digit statistics */
    int32_t retune_skipped_for_memory; /* queries whose re-tuned table did not fit beside the old one: they keep the size-based window */
    int32_t retune_attempts;     /* proofs inspected for the re-tune so far (it gives up after a few degenerate ones) */
    int32_t shard_rank;
    int32_t shard_count;
    int32_t latency_mode;        /* 1: short accumulation segments + tree reductions (one proof at a time); 0: throughput */
    int32_t warmup;              /* 1: a staged load (CG_FLAG_STAGED_LOAD) whose final arrangement is not in force yet */
    int32_t lone_slots;          /* extra slots for proofs that arrive (nearly) alone: 0-2 (CG_FLAG_NO_LONE_SLOT) */
    int32_t reserved[2];
    /* slot_bytes by kind (they add up to it).  In a throughput context the five MSMs of a proof run one after another and
     * share one set of entry lists and segment pieces, sized for the largest of them; a latency context (proof_slots = 1)
     * runs them concurrently and holds a set per MSM. */
    uint64_t slot_entry_bytes;     /* digit-entry lists: 8 B x bases x windows, two of them (grouped by the high / by the whole key) */
    uint64_t slot_piece_bytes;     /* first / last run of every accumulation segment, for the wave-combine levels */
    uint64_t slot_bucket_bytes;    /* bucket arrays, row / column sums of the reduction, partition histograms and cursors */
    uint64_t slot_transform_bytes; /* the witness map's vectors (assignment + four domain-sized vectors) and the h MSM's scalars */
    uint64_t slot_upload_bytes;    /* one device copy of an assignment arriving from the host (the context holds proof_slots + 2) */
    uint64_t lone_slot_bytes;      /* the lone slots' working sets together (each a set of entry lists and pieces per MSM) */
} cg_ctx_info;
int cg_ctx_get_info(cg_ctx* ctx, cg_ctx_info* out);

/* Where the time of cg_circuit_load went, by the reference's own timer names where it has them ("Reading ProverParams",
 * "Reading R1CS", creds/src/lib.rs:257,266 - there the cost is deserialisation; here it is the copy into HBM and the tables).
 * Milliseconds on the host clock.  With CG_FLAG_STAGED_LOAD, fold_ms / window_tables_ms / final_slots_ms belong to the
 * background part and are 0 until it is done (ready = 1). */
typedef struct cg_load_timings {
    float total_ms;           /* cg_circuit_load, call to return */
    float matrices_ms;        /* the three matrices: validation, coefficient dictionary, sliced layout, copy (host threads) */
    float domain_ms;          /* twiddle and coset tables of the evaluation domain */
    float key_copy_ms;        /* the five queries: host -> device copy, Montgomery import, row-0 table points */
    float fold_ms;            /* h query -> coset evaluation basis, C matrix folded into the l query (two DFTs over G1) */
    float window_tables_ms;   /* rows 1.. of the per-window tables of the five queries */
    float slots_ms;           /* proof slots and upload buffers (the warm-up slots of a staged load) */
    float final_slots_ms;     /* staged load: the proof slots of the final arrangement */
    float background_ms;      /* staged load: the worker's whole run, return of cg_circuit_load -> swap done */
    float swap_wait_ms;       /* staged load: how long the swap waited for the proofs in flight to drain */
    float ready_after_ms;     /* staged load: call of cg_circuit_load -> final arrangement in force */
    int32_t staged;           /* 1: the load was staged */
    int32_t ready;            /* 1: the final arrangement is in force (always 1 for a synchronous load) */
    int32_t windows_from_proof; /* staged load: 1 = the final windows were chosen from a warm-up proof's digit statistics (no re-tune follows) */
    int32_t warmup_proofs;    /* proofs finished in the warm-up arrangement */
    int32_t background_status; /* 0, or the cg_status the worker failed with (the context keeps proving in the warm-up arrangement) */
    int32_t reserved[3];
} cg_load_timings;
int cg_ctx_get_load_timings(cg_ctx* ctx, cg_load_timings* out);

/* Staged load: wait until the final arrangement is in force.  timeout_ms < 0 waits without limit.
 * Returns CG_OK when it is (at once for a synchronous load), 1 when the time ran out first, or the negative cg_status the
 * background part failed with (cg_last_error says why; the context keeps working in the warm-up arrangement). */
int cg_ctx_wait_ready(cg_ctx* ctx, int32_t timeout_ms);

/* Multi-GPU: a context loaded with shard_count > 1 owns a contiguous range of the l, a and b
 * queries and, of the h query, a contiguous range or — for a power-of-two shard_count — the coset points
 * j = shard_rank (mod shard_count), which lets two of its four transforms run at 1/shard_count of the size.  Which
 * points a shard owns is the library's business: the partial sums of all shards add up to the same five values.
 * cg_prove_partial computes this shard's five partial sums
 *   out = h ‖ l ‖ a ‖ b1 (4 x 64 B G1 affine canonical) ‖ b2 (128 B G2 affine canonical) = 384 B
 * (identity = zeros; b1 is all-zero and skipped when r == 0, prover.rs:102-112).
 * cg_assemble adds the gathered partials of all shards and finishes A, B, C exactly as
 * prover.rs:76-135.  partials: n_shards x 384 B. */
int cg_prove_partial(cg_ctx* ctx, const void* full_assignment, int assignment_on_device,
                     const uint8_t r[32], uint8_t out_partials[384], cg_timings* timings);
int cg_assemble(cg_ctx* ctx, const uint8_t* partials, uint32_t n_shards, const uint8_t r[32],
                const uint8_t s[32], uint8_t proof_out[256]);

/* SURVEY 8e's second arrangement of a sharded proof: the witness map runs ONCE (on one rank) and every shard receives the
 * scalars of its share of the h MSM, instead of every shard repeating the sparse products and the full-size transforms.
 * With the folded key (the default) those scalars are the coset values q_j = a(g w^j) b(g w^j) / Z(g), j < domain_size
 * (r1cs_to_qap.rs:187,201-208 without c's part, which the folded l query carries).
 *   cg_witness_map_coset : on a folded context WITHOUT CG_FLAG_H_SCALARS_EXTERNAL (whole or shard): all domain_size values,
 *       32 B canonical each, to host or device memory, laid out SHARD-MAJOR for this context's shard_count - shard p's
 *       scalars are the slice [offset_p, offset_p + count_p) that cg_h_scalars_slice reports (strided shards: q_{p + k·count},
 *       k < domain_size / count; contiguous shards and unsharded contexts: natural order) - so a scatter sends contiguous
 *       chunks.
 *   cg_prove_partial_q   : cg_prove_partial with this shard's slice supplied (host or device memory); the witness map is
 *       skipped.  Works on any shard context over a folded key; a context loaded with CG_FLAG_H_SCALARS_EXTERNAL can prove
 *       no other way.  The assignment is still checked for canonical elements (as the witness map would), the slice too. */
int cg_witness_map_coset(cg_ctx* ctx, const void* full_assignment, int assignment_on_device, void* q_out, int q_on_device);
int cg_h_scalars_slice(const cg_ctx* ctx, uint32_t shard, uint64_t* offset, uint64_t* count);
int cg_prove_partial_q(cg_ctx* ctx, const void* full_assignment, int assignment_on_device, const void* q_slice,
                       int q_on_device, const uint8_t r[32], uint8_t out_partials[384], cg_timings* timings);

/* The same in TWO CALLS, so that a shard's assignment-driven MSMs run WHILE the witness map and the scatter are still under way
 * somewhere else: of a sharded proof's critical path - witness map, scatter, partial sums - only the h share has to wait for
 * the slice (the l, a and b sums of prover.rs:74,266 need the assignment alone).
 *   cg_prove_partial_q_begin  : takes one of the context's proof slots, brings the assignment onto the GPU if it is not there,
 *       queues the l, a, b1 and b2 partial sums and RETURNS without waiting for them.  *out is the open proof.
 *   cg_partial_witness_map_coset : on the rank that runs the witness map (a context loaded WITHOUT CG_FLAG_H_SCALARS_EXTERNAL):
 *       cg_witness_map_coset for the open proof's assignment, on the open proof's own working set (a one-slot context has no other),
 *       next to the MSMs already queued.  Waits for the values.
 *   cg_prove_partial_q_finish : with this shard's slice (host or device memory): queues the h share, waits for everything, writes
 *       the 384-byte record, gives the slot back and destroys the handle - also when it fails.
 *   cg_prove_partial_q_abort  : for a caller that cannot deliver the slice: waits for what was queued, gives the slot back.
 * An open proof holds its slot: begin as many as the context has slots and no more, or begin blocks.  Calls on one handle are
 * the caller's to serialise; begin and finish may come from different threads.  Every open proof must be finished or aborted
 * before cg_circuit_free (which waits for the calls inside the context, open proofs included). */
/* The witness map in two HALVES (round 6).  Until their pointwise product the two sides of q_j = vinv·a(g w^j) · b(g w^j)
 * (r1cs_to_qap.rs:164-187) are independent - a sparse product with A (resp. B) and two transforms each - so two ranks can
 * compute one side each, at half the witness map's time, and every shard multiplies its two slices itself:
 *   cg_witness_map_coset_half / cg_partial_witness_map_coset_half : which = 0: the a side, vinv·a(g w^j); which = 1: the b side,
 *       b(g w^j); domain_size plain canonical values, laid out shard-major exactly as cg_witness_map_coset lays out q.
 *   cg_prove_partial_q_finish2 : cg_prove_partial_q_finish with the shard's slices of BOTH sides (both in host or both in device
 *       memory); the h scalars are their products mod r, formed on the GPU.  (The one-call form: begin, then finish2.) */
typedef struct cg_partial cg_partial;
int cg_witness_map_coset_half(cg_ctx* ctx, const void* full_assignment, int assignment_on_device, int which, void* out, int out_on_device);
int cg_partial_witness_map_coset_half(cg_partial* p, int which, void* out, int out_on_device);
int cg_prove_partial_q_finish2(cg_partial* p, const void* a_slice, const void* b_slice, int slices_on_device, uint8_t out_partials[384],
                               cg_timings* timings);
int cg_prove_partial_q_begin(cg_ctx* ctx, const void* full_assignment, int assignment_on_device, const uint8_t r[32], cg_partial** out);
int cg_partial_witness_map_coset(cg_partial* p, void* q_out, int q_on_device);
int cg_prove_partial_q_finish(cg_partial* p, const void* q_slice, int q_on_device, uint8_t out_partials[384], cg_timings* timings);
void cg_prove_partial_q_abort(cg_partial* p);

/* R1CS -> QAP witness map only: h coefficients, domain_size x 32 B canonical.
 * Replaces: `LibsnarkReduction::witness_map_from_matrices`, r1cs_to_qap.rs:150-213. */
int cg_witness_map(cg_ctx* ctx, const uint8_t* full_assignment, uint8_t* h_out);
uint64_t cg_domain_size(const cg_ctx* ctx);

/* The witness map over RESIDENT matrices, without a proving key: the reference's own plug point.
 * Replaces: an `impl R1CSToQAP` (forks/groth16/src/r1cs_to_qap.rs:49-98) whose `witness_map_from_matrices`
 *           (:150-213) is `Groth16<E, QAP>`'s type parameter (forks/groth16/src/lib.rs:55-57): a maintainer who only
 *           swaps the QAP type (INTEGRATION.md "GpuReduction") keeps arkworks' MSMs and moves the seven transforms
 *           and three sparse products here.  cg_qap_load copies the matrices once (the `cs.to_matrices()` product,
 *           r1cs_to_qap.rs:62); cg_qap_witness_map returns the reference's result, the coefficients of h
 *           (domain_size x 32 B canonical), for a full assignment in host or device memory.
 * Errors: CG_ERR_POLY_DEGREE_TOO_LARGE as r1cs_to_qap.rs:156-157; a non-canonical assignment element is
 *         CG_ERR_INVALID_ARGUMENT.  Calls on one handle serialise. */
typedef struct cg_qap_ctx cg_qap_ctx;
int cg_qap_load(cg_qap_ctx** out, const cg_csr abc[3], uint64_t num_inputs, uint64_t num_constraints,
                uint64_t num_variables, int32_t device /* -1 = current */);
int cg_qap_witness_map(cg_qap_ctx* ctx, const void* full_assignment, int assignment_on_device, void* h_out,
                       int h_on_device);
uint64_t cg_qap_domain_size(const cg_qap_ctx* ctx);
void cg_qap_free(cg_qap_ctx* ctx);

/* Unit level: Σ scalars[i]·bases[i] over BN254 G1 / G2 for caller-supplied bases.
 * Replaces: `<G as VariableBaseMSM>::msm_bigint(bases, scalars)` (ark-ec; call sites
 *           prover.rs:66,74,266).  bases in `coord_form`; scalars canonical; result affine
 *           canonical (64 / 128 B, zeros = identity).  Uses min(n_bases, n_scalars) pairs, as
 *           msm_bigint's zip does. */
int cg_msm_g1(const uint8_t* bases, uint32_t coord_form, uint64_t n_bases, const uint8_t* scalars,
              uint64_t n_scalars, int32_t window_bits, uint8_t out[64]);
int cg_msm_g2(const uint8_t* bases, uint32_t coord_form, uint64_t n_bases, const uint8_t* scalars,
              uint64_t n_scalars, int32_t window_bits, uint8_t out[128]);

/* Unit level: in-place radix-2 NTT over Fr, natural order in and out, 2^log_n x 32 B canonical.
 * Replaces: `EvaluationDomain::{fft,ifft}_in_place` and the coset variants obtained through
 *           `get_coset(F::GENERATOR)` (ark-poly; call sites r1cs_to_qap.rs:179-185,198-199,210).
 * inverse = 0: out[k] = Σ a[j] (g^j if coset) ω^{jk};  inverse = 1: the inverse map. */
int cg_ntt(uint8_t* data, uint32_t log_n, int inverse, int coset);

/* Unit level, resident operands: the handle forms of the two entry points above.  A caller that reuses a set
 * of bases (a query of the key, a commitment key) or a domain keeps the expanded window tables / twiddle tables
 * in HBM and may pass scalars / data that already live on the device; these run exactly the kernels cg_prove
 * runs and are what `tools/sweep.py` times (SURVEY 8d "unit sweeps").
 *   cg_msm_load_g1/g2 : opt may be NULL; opt->device and opt->window_bits are honoured (0 = size-based default).
 *   cg_msm_run        : Σ scalars[i]·bases[i] over min(n_scalars, n_bases) pairs; scalars canonical, host or
 *                       device memory; out = 64 B (G1) / 128 B (G2) affine canonical, zeros = identity.
 *                       timings (optional): the h (G1) or b2 (G2) MSM fields, accum_*, sort_ms, entries_*.
 *   cg_ntt_load/run   : in-place transform of 2^log_n canonical scalars in host or device memory, natural
 *                       order in and out; kernel_ms (optional) = HIP-event time of the transform's kernels.
 *                       A non-canonical element is CG_ERR_INVALID_ARGUMENT (device data is then unspecified).
 * Calls on one handle serialise; different handles are independent. */
typedef struct cg_msm_ctx cg_msm_ctx;
int cg_msm_load_g1(cg_msm_ctx** out, const uint8_t* bases, uint32_t coord_form, uint64_t n_bases, const cg_options* opt);
int cg_msm_load_g2(cg_msm_ctx** out, const uint8_t* bases, uint32_t coord_form, uint64_t n_bases, const cg_options* opt);
int cg_msm_run(cg_msm_ctx* ctx, const void* scalars, int scalars_on_device, uint64_t n_scalars, uint8_t* out,
               cg_timings* timings);
void cg_msm_free(cg_msm_ctx* ctx);
typedef struct cg_ntt_ctx cg_ntt_ctx;
int cg_ntt_load(cg_ntt_ctx** out, uint32_t log_n, int32_t device /* -1 = current */);
int cg_ntt_run(cg_ntt_ctx* ctx, void* data, int data_on_device, int inverse, int coset, float* kernel_ms);
void cg_ntt_free(cg_ntt_ctx* ctx);

/* out[i] = scalars[i]·G for the standard generator of G1 / G2 (canonical scalars in, canonical affine points
 * out, 64 / 128 B each, zeros = identity).
 * Replaces: `FixedBase::msm::<E::G1 | E::G2>(scalar_bits, window, &table, &scalars)` [ark-ec], the building block
 *           of the generator (forks/groth16/src/generator.rs:134-140,147-148,162-194). */
int cg_fixed_base_g1(const uint8_t* scalars, uint64_t n, uint8_t* out);
int cg_fixed_base_g2(const uint8_t* scalars, uint64_t n, uint8_t* out);

/* Trusted setup from explicit toxic waste, on the GPU.
 * Replaces: `generate_parameters_with_qap`, forks/groth16/src/generator.rs:50-228, with
 *           gamma = 1 and the standard generators as the fork fixes them (:28,:34-35).
 * Outputs are written canonical into caller buffers sized as in cg_proving_key;
 * gamma_abc_g1 gets num_inputs x 64 B (vk.gamma_abc_g1), vk_points gets
 * alpha_g1(64) ‖ beta_g1(64) ‖ delta_g1(64) ‖ beta_g2(128) ‖ gamma_g2(128) ‖ delta_g2(128). */
int cg_setup(const cg_csr abc[3], uint64_t num_inputs, uint64_t num_constraints,
             uint64_t num_variables, const uint8_t tau[32], const uint8_t alpha[32],
             const uint8_t beta[32], const uint8_t delta[32], uint8_t* a_query, uint8_t* b_g1_query,
             uint8_t* b_g2_query, uint8_t* h_query, uint8_t* l_query, uint8_t* gamma_abc_g1,
             uint8_t vk_points[576]);

/* .r1cs (iden3 binary) -> CSR.  Replaces: `R1CSFile::new` + `R1CS::from`,
 * forks/circom-compat/src/circom/r1cs_reader.rs:54-148,26-38.
 * The returned object owns its arrays; cg_r1cs_csr fills views valid until cg_r1cs_free. */
typedef struct cg_r1cs cg_r1cs;
typedef struct cg_r1cs_header {
    uint32_t field_size, n_wires, n_pub_out, n_pub_in, n_prv_in, n_constraints;
    uint64_t n_labels;
    uint64_t num_inputs;      /* 1 + n_pub_in + n_pub_out  (r1cs_reader.rs:28) */
    uint64_t num_variables;   /* n_wires */
} cg_r1cs_header;
int cg_r1cs_parse(const uint8_t* data, uint64_t len, cg_r1cs** out);
int cg_r1cs_get(const cg_r1cs* r, cg_r1cs_header* header, cg_csr abc[3], const uint64_t** wire_mapping);
void cg_r1cs_free(cg_r1cs* r);

/* ark-serialize (uncompressed) proving key <-> packed arrays.
 * Replaces: `read_from_file::<ProverParams>` / `write_to_file` for the `groth16_params: ProvingKey<Bn254>` that
 *           leads creds' prover_params.bin (creds/src/utils.rs:140-152,179-189; creds/src/lib.rs:58-63;
 *           field order forks/groth16/src/data_structures.rs:31-44,101-118).  Parsing is the reference's
 *           `deserialize_uncompressed_unchecked`: flags stripped, no curve / subgroup checks.
 * cg_pk_parse reads one ProvingKey from the start of `data` (bytes_consumed, optional, tells where the
 * PreparedVerifyingKey / config string that follow in prover_params.bin begin); cg_pk_get fills a view valid
 * until cg_pk_free, plus the verifying-key members the prover itself does not need. */
typedef struct cg_pk cg_pk;
int cg_pk_parse(const uint8_t* data, uint64_t len, cg_pk** out, uint64_t* bytes_consumed);
int cg_pk_get(const cg_pk* k, cg_proving_key* view, const uint8_t** gamma_g2, const uint8_t** gamma_abc_g1,
              uint64_t* gamma_abc_len);
void cg_pk_free(cg_pk* k);
uint64_t cg_pk_serialized_size(const cg_proving_key* pk, uint64_t gamma_abc_len);
int cg_pk_serialize(const cg_proving_key* pk, const uint8_t* gamma_g2, const uint8_t* gamma_abc_g1,
                    uint64_t gamma_abc_len, uint8_t* out, uint64_t out_len);

/* prover_params.bin: `ProverParams { groth16_params: ProvingKey, groth16_pvk: PreparedVerifyingKey,
 * config_str: String }`.
 * Replaces: `read_from_file::<ProverParams>` in `create_client_state` (creds/src/lib.rs:58-63,268;
 *           creds/src/utils.rs:179-189) and `write_to_file(&prover_params, ..)` in `run_zksetup` (creds/src/lib.rs:245-248).
 * The prover consumes the key; the serialized VerifyingKey, PreparedVerifyingKey (data_structures.rs:62-71: vk, an Fq12
 * and two bn::G2Prepared) and the configuration string are returned verbatim, because `create_client_state` copies
 * them into the ClientState (creds/src/lib.rs:292-299).  Views stay valid until cg_prover_params_free. */
typedef struct cg_prover_params cg_prover_params;
typedef struct cg_prover_params_view {
    cg_proving_key pk;               /* canonical packed arrays, as cg_pk_get */
    const uint8_t* gamma_g2;         /* vk.gamma_g2, 128 B */
    const uint8_t* gamma_abc_g1;     /* vk.gamma_abc_g1, gamma_abc_len x 64 B */
    uint64_t gamma_abc_len;
    const uint8_t* vk_bytes;  uint64_t vk_len;      /* groth16_params.vk as serialized */
    const uint8_t* pvk_bytes; uint64_t pvk_len;     /* groth16_pvk as serialized */
    const uint8_t* config_str; uint64_t config_len; /* UTF-8, no terminator */
} cg_prover_params_view;
int cg_prover_params_parse(const uint8_t* data, uint64_t len, cg_prover_params** out);
int cg_prover_params_get(const cg_prover_params* pp, cg_prover_params_view* view);
void cg_prover_params_free(cg_prover_params* pp);
uint64_t cg_prover_params_serialized_size(const cg_proving_key* pk, uint64_t gamma_abc_len, uint64_t pvk_len,
                                          uint64_t config_len);
int cg_prover_params_serialize(const cg_proving_key* pk, const uint8_t* gamma_g2, const uint8_t* gamma_abc_g1,
                               uint64_t gamma_abc_len, const uint8_t* pvk_bytes, uint64_t pvk_len,
                               const uint8_t* config_str, uint64_t config_len, uint8_t* out, uint64_t out_len);

/* client_state.bin: the hand-over from the prove step to the host-side `show` step.
 * Replaces: `ClientState::<E>::new` + `write_to_file` / `new_from_file` (creds/src/groth16rand.rs:23-35,60-98;
 *           creds/src/lib.rs:292-300), ark-serialize uncompressed, fields in declaration order:
 *           inputs: Vec<Fr> | aux: Option<String> | proof | vk | pvk | input_com_randomness: Option<Fr> |
 *           committed_input_openings: Vec<PedersenOpening<G1>> (creds/src/dlog.rs:24-29) | credtype | config_str.
 * The library writes the 256-byte proof cg_prove returned next to the vk / pvk / config bytes that came out of
 * prover_params.bin; openings (empty for a fresh state) travel as their serialized bytes. */
typedef struct cg_client_state cg_client_state;
typedef struct cg_client_state_view {
    const uint8_t* inputs; uint64_t n_inputs;        /* public inputs (wires 1 .. num_inputs-1), 32 B canonical each */
    const uint8_t* aux; uint64_t aux_len; int32_t has_aux;
    int32_t has_input_com_randomness;
    const uint8_t* proof;                            /* 256 B */
    const uint8_t* vk_bytes; uint64_t vk_len;
    const uint8_t* pvk_bytes; uint64_t pvk_len;
    const uint8_t* input_com_randomness;             /* 32 B when has_input_com_randomness */
    const uint8_t* openings_bytes; uint64_t openings_len; uint64_t n_openings;
    const uint8_t* credtype; uint64_t credtype_len;  /* "jwt" (groth16rand.rs:76) or "mdl" */
    const uint8_t* config_str; uint64_t config_len;
} cg_client_state_view;
uint64_t cg_client_state_serialized_size(const cg_client_state_view* v);
int cg_client_state_serialize(const cg_client_state_view* v, uint8_t* out, uint64_t out_len);
int cg_client_state_parse(const uint8_t* data, uint64_t len, cg_client_state** out);
int cg_client_state_get(const cg_client_state* cs, cg_client_state_view* view);
void cg_client_state_free(cg_client_state* cs);

/* Diagnostic: the shader clock (GHz) the GPU holds over the next `window_us` microseconds, measured on the device by
 * one wave that compares the shader-cycle counter with the constant-rate counter - callable from a second thread while
 * proofs run, which is how bench.py reports the clock its peaks should be scaled by.  device -1 = current.
 * (No counterpart in the reference: CPU provers do not report their clock either.) */
int cg_probe_shader_clock(int32_t device, uint32_t window_us, double* ghz_out);

/* Library / device description for logs ("crescent_gpu 0.1 gfx950 ..."). */
const char* cg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CRESCENT_GPU_H */
