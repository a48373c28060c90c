/*
 * crescent_throughput — Groth16 proofs per second through the C ABI from a plain C host with POSIX threads: no Python, no
 * torch, nothing in the process but libcrescent_gpu.so (and the build's own synthetic-circuit generator, libcg_synth.so,
 * which stands in for the circuit and witness a Crescent cache directory would supply: creds/src/lib.rs:255-283).
 *
 *   crescent_throughput [--shape l m M] [--bits f] [--slots T] [--proofs N] [--warmup W] [--pageable]
 *
 * What it does, in the reference's terms: one-time `zksetup` (cg_setup: forks/groth16/src/generator.rs:50-228) and circuit
 * load, then T + 2 host threads that each call `Groth16::prove` (cg_prove: forks/groth16/src/prover.rs:26-51) in a loop
 * with fresh (r, s), the witness arriving in HOST memory - page-locked (cg_host_alloc) unless --pageable - the way a
 * server proving credentials for many clients would (sample/client_helper/src/main.rs:177-216 runs one task per
 * credential).  It reports the steady-state rate between the W-th and the (W + N)-th completion and the CPU time the
 * process spent meanwhile.  bench.py measures the same thing from Python; this program is the check that neither the
 * rate nor the host cost is an artefact of that harness.
 */
#define _GNU_SOURCE
#include <crescent_gpu.h>

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* crescent-credentials_amd/synth/synth.cpp (host-only workload generator of the tests and the bench) */
typedef struct cgs_instance cgs_instance;
cgs_instance* cgs_generate_gates(uint64_t seed, uint64_t num_inputs, uint64_t num_constraints, uint64_t num_variables, double bit_fraction,
                                 uint32_t limb_terms);
void cgs_views(const cgs_instance* I, const uint64_t** row_ptr, const uint32_t** col, const uint8_t** coeff, uint64_t* nnz,
               const uint8_t** witness);
void cgs_free(cgs_instance* I);

static const uint8_t FR_MODULUS_LE[32] = { /* r1cs_reader.rs:183 */
    0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
    0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + (double)t.tv_nsec * 1e-9;
}
static double cpu_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &t);
    return (double)t.tv_sec + (double)t.tv_nsec * 1e-9;
}
static int below_modulus(const uint8_t x[32]) {
    for (int i = 31; i >= 0; --i) {
        if (x[i] < FR_MODULUS_LE[i]) return 1;
        if (x[i] > FR_MODULUS_LE[i]) return 0;
    }
    return 0;
}
/* splitmix64: a reproducible stream of (r, s); uniform below r by rejection */
static uint64_t mix(uint64_t* s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static void scalar_from(uint64_t* s, uint8_t out[32]) {
    do {
        for (int k = 0; k < 4; ++k) { uint64_t v = mix(s); memcpy(out + 8 * k, &v, 8); }
        out[31] &= 0x3f;
    } while (!below_modulus(out));
}

typedef struct {
    cg_ctx* ctx;
    uint8_t** witness;        /* n_witness buffers, taken in rotation */
    int n_witness;
    long total;               /* proofs to make in all */
    long next;                /* next proof index (under mu) */
    double* done_at;          /* completion time of proof k */
    pthread_mutex_t mu;
    int failed;
} shared_t;

static void* caller(void* arg) {
    shared_t* S = (shared_t*)arg;
    for (;;) {
        pthread_mutex_lock(&S->mu);
        const long k = S->next++;
        pthread_mutex_unlock(&S->mu);
        if (k >= S->total || S->failed) return NULL;
        uint64_t seed = 0xC5E5CE47ull + (uint64_t)k * 2654435761ull;
        uint8_t r[32], s[32], proof[256];
        scalar_from(&seed, r);
        scalar_from(&seed, s);
        if (cg_prove(S->ctx, S->witness[k % S->n_witness], r, s, proof, NULL) != CG_OK) {
            fprintf(stderr, "cg_prove: %s\n", cg_last_error());
            S->failed = 1;
            return NULL;
        }
        S->done_at[k] = now_s();
    }
}

static int cmp_double(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : x > y;
}

int main(int argc, char** argv) {
    uint64_t l = 26, m = 1480000, M = 1500000;          /* rs256-sd shape */
    double bits = 0.9;
    int slots = 16, pageable = 0;
    long proofs = 1200, warmup = 64;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--shape") && i + 3 < argc) { l = strtoull(argv[i + 1], NULL, 10); m = strtoull(argv[i + 2], NULL, 10); M = strtoull(argv[i + 3], NULL, 10); i += 3; }
        else if (!strcmp(argv[i], "--bits") && i + 1 < argc) bits = atof(argv[++i]);
        else if (!strcmp(argv[i], "--slots") && i + 1 < argc) slots = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--proofs") && i + 1 < argc) proofs = atol(argv[++i]);
        else if (!strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = atol(argv[++i]);
        else if (!strcmp(argv[i], "--pageable")) pageable = 1;
        else { fprintf(stderr, "usage: %s [--shape l m M] [--bits f] [--slots T] [--proofs N] [--warmup W] [--pageable]\n", argv[0]); return 2; }
    }
    if (slots < 1 || slots > 16 || proofs < 1 || warmup < 0) { fprintf(stderr, "bad arguments\n"); return 2; }
    if (cg_init(0, NULL) != CG_OK) { fprintf(stderr, "cg_init: %s\n", cg_last_error()); return 1; }
    fprintf(stderr, "%s\n", cg_version());

    /* the circuit and its witness (gate mix of the bench: 17 limbs a side in the bigint rows at bit_fraction 0.9) */
    double t0 = now_s();
    cgs_instance* inst = cgs_generate_gates(0xC5E5CE47ull + 3, l, m, M, bits, 17);
    if (!inst) { fprintf(stderr, "unsupported shape\n"); return 1; }
    const uint64_t* rp[3]; const uint32_t* col[3]; const uint8_t* coeff[3]; uint64_t nnz[3]; const uint8_t* wit;
    cgs_views(inst, rp, col, coeff, nnz, &wit);
    cg_csr abc[3];
    for (int k = 0; k < 3; ++k) { abc[k].row_ptr = rp[k]; abc[k].col = col[k]; abc[k].coeff = coeff[k]; abc[k].nnz = nnz[k]; }
    uint64_t D = 1;
    while (D < m + l) D <<= 1;

    /* zksetup from fixed toxic waste (generator.rs:50-228 with gamma = 1) */
    uint8_t tau[32] = {3}, alpha[32] = {5}, beta[32] = {7}, delta[32] = {11};
    tau[9] = 0x5a; alpha[11] = 0x33; beta[13] = 0x77; delta[17] = 0x19;
    uint8_t* a_q = malloc(M * 64); uint8_t* b1_q = malloc(M * 64); uint8_t* b2_q = malloc(M * 128);
    uint8_t* h_q = malloc((D - 1) * 64); uint8_t* l_q = malloc((M - l ? M - l : 1) * 64); uint8_t* gabc = malloc(l * 64);
    uint8_t vkp[576];
    if (!a_q || !b1_q || !b2_q || !h_q || !l_q || !gabc) { fprintf(stderr, "out of memory\n"); return 1; }
    if (cg_setup(abc, l, m, M, tau, alpha, beta, delta, a_q, b1_q, b2_q, h_q, l_q, gabc, vkp) != CG_OK) { fprintf(stderr, "cg_setup: %s\n", cg_last_error()); return 1; }
    cg_proving_key pk;
    memset(&pk, 0, sizeof pk);
    pk.coord_form = CG_FORM_CANONICAL;
    pk.alpha_g1 = vkp; pk.beta_g1 = vkp + 64; pk.delta_g1 = vkp + 128; pk.beta_g2 = vkp + 192; pk.delta_g2 = vkp + 448;
    pk.a_query = a_q; pk.a_len = M; pk.b_g1_query = b1_q; pk.b_g1_len = M; pk.b_g2_query = b2_q; pk.b_g2_len = M;
    pk.h_query = h_q; pk.h_len = D - 1; pk.l_query = l_q; pk.l_len = M - l;
    double t1 = now_s();

    cg_options opt;
    memset(&opt, 0, sizeof opt);
    opt.device = -1;
    opt.proof_slots = slots;
    cg_ctx* ctx = NULL;
    if (cg_circuit_load(&ctx, &pk, abc, l, m, M, &opt) != CG_OK) { fprintf(stderr, "cg_circuit_load: %s\n", cg_last_error()); return 1; }
    double t2 = now_s();
    fprintf(stderr, "circuit: m = %llu, M = %llu, l = %llu, nnz = %llu; generated + key in %.1f s, loaded in %.1f s\n", (unsigned long long)m,
            (unsigned long long)M, (unsigned long long)l, (unsigned long long)(nnz[0] + nnz[1] + nnz[2]), t1 - t0, t2 - t1);

    /* the witness in host memory, four copies taken in rotation */
    shared_t S;
    memset(&S, 0, sizeof S);
    S.ctx = ctx;
    S.n_witness = 4;
    S.witness = malloc(sizeof(uint8_t*) * (size_t)S.n_witness);
    for (int k = 0; k < S.n_witness; ++k) {
        S.witness[k] = pageable ? (uint8_t*)malloc(M * 32) : (uint8_t*)cg_host_alloc(M * 32);
        if (!S.witness[k]) { fprintf(stderr, "witness buffer: %s\n", cg_last_error()); return 1; }
        memcpy(S.witness[k], wit, M * 32);
    }
    /* the first proof re-tunes the windows of the assignment-driven MSMs: circuit loading, not proving */
    {
        uint64_t seed = 1;
        uint8_t r[32], s[32], p0[256], p1[256];
        scalar_from(&seed, r); scalar_from(&seed, s);
        if (cg_prove(ctx, S.witness[0], r, s, p0, NULL) != CG_OK || cg_prove(ctx, S.witness[1], r, s, p1, NULL) != CG_OK) { fprintf(stderr, "cg_prove: %s\n", cg_last_error()); return 1; }
        if (memcmp(p0, p1, 256) != 0) { fprintf(stderr, "the same statement and randomness gave two different proofs\n"); return 1; }
    }
    const int callers = slots + 2;                 /* a context holds two more upload buffers than proof slots */
    S.total = warmup + proofs + callers;           /* the tail keeps the timed window in steady state */
    S.done_at = calloc((size_t)S.total, sizeof(double));
    pthread_mutex_init(&S.mu, NULL);
    pthread_t* th = malloc(sizeof(pthread_t) * (size_t)callers);
    const double c0 = cpu_s(), w0 = now_s();
    for (int k = 0; k < callers; ++k) pthread_create(&th[k], NULL, caller, &S);
    for (int k = 0; k < callers; ++k) pthread_join(th[k], NULL);
    const double c1 = cpu_s(), w1 = now_s();
    if (S.failed) return 1;
    qsort(S.done_at, (size_t)S.total, sizeof(double), cmp_double);
    const double t_a = warmup > 0 ? S.done_at[warmup - 1] : w0, t_b = S.done_at[warmup + proofs - 1];
    cg_ctx_info info;
    if (cg_ctx_get_info(ctx, &info) != CG_OK) { fprintf(stderr, "cg_ctx_get_info: %s\n", cg_last_error()); return 1; }
    printf("{\"proofs_per_s\": %.3f, \"proofs\": %ld, \"warmup\": %ld, \"proof_slots\": %d, \"caller_threads\": %d, \"witness\": \"%s host memory\", "
           "\"host_cpus_busy\": %.2f, \"resident_GB\": %.2f, \"window_bits\": [%d, %d, %d, %d, %d], \"tuned\": %d}\n",
           (double)proofs / (t_b - t_a), proofs, warmup, slots, callers, pageable ? "pageable" : "page-locked", (c1 - c0) / (w1 - w0),
           (double)info.total_bytes / 1e9, (int)info.window_bits[0], (int)info.window_bits[1], (int)info.window_bits[2], (int)info.window_bits[3],
           (int)info.window_bits[4], (int)info.tuned);
    for (int k = 0; k < S.n_witness; ++k) { if (pageable) free(S.witness[k]); else cg_host_free(S.witness[k]); }
    cg_circuit_free(ctx);
    cgs_free(inst);
    free(a_q); free(b1_q); free(b2_q); free(h_q); free(l_q); free(gabc); free(S.witness); free(S.done_at); free(th);
    return 0;
}
