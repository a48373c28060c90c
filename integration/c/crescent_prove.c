/*
 * crescent_prove — the reference's `create_client_state` (creds/src/lib.rs:255-301) as a plain C program over
 * libcrescent_gpu's C ABI: no Python, no torch, nothing but <include/crescent_gpu.h>.
 *
 *   crescent_prove <main_c.r1cs> <prover_params.bin> <witness.bin> <client_state.bin>
 *                  [--rs r_hex s_hex] [--credtype jwt|mdl] [--aux prover_aux_json] [--sync-load] [--timings-json]
 *
 *   main_c.r1cs       the circuit, iden3 binary format     (lib.rs:257-258; r1cs_reader.rs:54-148)
 *   prover_params.bin ProverParams{groth16_params, groth16_pvk, config_str}, ark-serialize uncompressed (lib.rs:268)
 *   witness.bin       the full assignment, num_variables x 32-byte little-endian canonical scalars, wire 0 = 1 — what
 *                     the WASM witness calculator hands to `CircomBuilder::build` (lib.rs:270-279; witness generation
 *                     stays on the host and is not part of this library)
 *   client_state.bin  out: ClientState{inputs, aux, proof, vk, pvk, ...} (groth16rand.rs:23-35, lib.rs:292-300),
 *                     the hand-over to the host-side `show` step
 *   --rs r_hex s_hex  the proof randomness as hex integers (reproducible runs); default: drawn from /dev/urandom by
 *                     rejection (prover.rs:150-151 samples r, then s)
 *   --credtype        ClientState::credtype (lib.rs:300), default "jwt" (groth16rand.rs:76)
 *   --aux             ClientState::aux, the contents of prover_aux.json (lib.rs:254,294); default None
 *   --sync-load       load the circuit synchronously (all tables before the first proof) instead of CG_FLAG_STAGED_LOAD
 *   --timings-json    one JSON line on stdout with the phases of this run, named after the reference's timers
 *                     ("Reading R1CS" lib.rs:257, "Reading ProverParams" :266, "Groth16 prove" :281); with a staged load it then
 *                     waits for the final arrangement and adds the background part and a second proof's time
 *
 * The two files are mapped, not read (the parsers take them straight from the page cache), and parsed on two threads while
 * this one starts the GPU runtime: the three are independent, and each is a few hundred milliseconds at the rs256 size.
 *
 * This is the reference-side binding in its smallest form: the Rust shim (integration/rust/crescent-gpu) makes the
 * same calls in the same order.  It is also the proof that the header is valid C and the ABI needs nothing else.
 */
#include <crescent_gpu.h>

#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

static const uint8_t FR_MODULUS_LE[32] = { /* r1cs_reader.rs:183 */
    0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
    0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

static int die(const char* what) {
    fprintf(stderr, "crescent_prove: %s: %s\n", what, cg_last_error());
    return 1;
}

/* a whole file, read-only: mapped (no copy out of the page cache); *mapped says how to give it back */
static uint8_t* read_file(const char* path, uint64_t* len, int* mapped) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) { perror(path); return NULL; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 0) { perror(path); close(fd); return NULL; }
    *len = (uint64_t)st.st_size;
    *mapped = 0;
    if (st.st_size > 0) {
        void* m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) { close(fd); *mapped = 1; return (uint8_t*)m; }
    }
    uint8_t* buf = (uint8_t*)malloc(st.st_size ? (size_t)st.st_size : 1);          /* not mappable: read it */
    size_t got = 0;
    while (buf && got < (size_t)st.st_size) {
        ssize_t n = read(fd, buf + got, (size_t)st.st_size - got);
        if (n <= 0) { perror(path); free(buf); buf = NULL; break; }
        got += (size_t)n;
    }
    close(fd);
    return buf;
}
static void release_file(uint8_t* p, uint64_t len, int mapped) {
    if (!p) return;
    if (mapped) munmap(p, (size_t)len); else free(p);
}

/* the witness goes straight into page-locked memory (cg_host_alloc): where a host lets its witness calculator write,
 * so that cg_prove's upload is one asynchronous DMA */
static uint8_t* read_file_pinned(const char* path, uint64_t* len) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return NULL; }
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return NULL; }
    long n = ftell(f);
    if (n < 0) { fclose(f); return NULL; }
    rewind(f);
    uint8_t* buf = (uint8_t*)cg_host_alloc((uint64_t)n);
    if (!buf) { fprintf(stderr, "cg_host_alloc: %s\n", cg_last_error()); fclose(f); return NULL; }
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) { perror(path); cg_host_free(buf); fclose(f); return NULL; }
    fclose(f);
    *len = (uint64_t)n;
    return buf;
}

static int below_modulus(const uint8_t x[32]) {
    for (int i = 31; i >= 0; --i) {
        if (x[i] < FR_MODULUS_LE[i]) return 1;
        if (x[i] > FR_MODULUS_LE[i]) return 0;
    }
    return 0;
}

/* uniform in [0, r): 254 random bits, rejected while >= r (what `Fr::rand` does) */
static int random_scalar(uint8_t out[32]) {
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f) return 0;
    do {
        if (fread(out, 1, 32, f) != 32) { fclose(f); return 0; }
        out[31] &= 0x3f;
    } while (!below_modulus(out));
    fclose(f);
    return 1;
}

static int hex_scalar(const char* hex, uint8_t out[32]) {
    size_t n = strlen(hex);
    if (n >= 2 && hex[0] == '0' && (hex[1] == 'x' || hex[1] == 'X')) { hex += 2; n -= 2; }
    if (n == 0 || n > 64) return 0;
    memset(out, 0, 32);
    for (size_t i = 0; i < n; ++i) {              /* digit i from the right -> nibble i */
        char ch = hex[n - 1 - i];
        int v = (ch >= '0' && ch <= '9') ? ch - '0' : (ch >= 'a' && ch <= 'f') ? ch - 'a' + 10 : (ch >= 'A' && ch <= 'F') ? ch - 'A' + 10 : -1;
        if (v < 0) return 0;
        out[i / 2] |= (uint8_t)(v << (4 * (i & 1)));
    }
    return below_modulus(out);
}

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec / 1e6;
}

/* one input file -> its parsed form, on a thread of its own */
typedef struct {
    const char* path;
    int is_r1cs;
    uint8_t* bytes; uint64_t len; int mapped;
    cg_r1cs* r1cs; cg_prover_params* pp;
    double read_s, parse_s;
    int ok;
    char err[512];
} parse_job;

static void* parse_thread(void* arg) {
    parse_job* j = (parse_job*)arg;
    double t0 = now_ms();
    j->bytes = read_file(j->path, &j->len, &j->mapped);
    j->read_s = (now_ms() - t0) / 1e3;
    if (!j->bytes) { snprintf(j->err, sizeof j->err, "cannot read %s", j->path); return NULL; }
    t0 = now_ms();
    int rc = j->is_r1cs ? cg_r1cs_parse(j->bytes, j->len, &j->r1cs) : cg_prover_params_parse(j->bytes, j->len, &j->pp);
    j->parse_s = (now_ms() - t0) / 1e3;
    if (rc != CG_OK) { snprintf(j->err, sizeof j->err, "%s: %s", j->is_r1cs ? "cg_r1cs_parse" : "cg_prover_params_parse", cg_last_error()); return NULL; }
    j->ok = 1;
    return NULL;
}

int main(int argc, char** argv) {
    const char* credtype = "jwt";
    const char* aux = NULL;
    uint8_t r[32], s[32];
    int have_rs = 0, bad = argc < 5, sync_load = 0, timings_json = 0;
    for (int i = 5; i < argc && !bad; ++i) {
        if (!strcmp(argv[i], "--rs") && i + 2 < argc) {
            if (!hex_scalar(argv[i + 1], r) || !hex_scalar(argv[i + 2], s)) { fprintf(stderr, "--rs: need hex values below the scalar modulus\n"); return 2; }
            have_rs = 1; i += 2;
        } else if (!strcmp(argv[i], "--credtype") && i + 1 < argc) credtype = argv[++i];
        else if (!strcmp(argv[i], "--aux") && i + 1 < argc) aux = argv[++i];
        else if (!strcmp(argv[i], "--sync-load")) sync_load = 1;
        else if (!strcmp(argv[i], "--timings-json")) timings_json = 1;
        else bad = 1;
    }
    if (bad) {
        fprintf(stderr, "usage: %s main_c.r1cs prover_params.bin witness.bin client_state.bin [--rs r_hex s_hex] [--credtype jwt|mdl] [--aux json] "
                        "[--sync-load] [--timings-json]\n", argv[0]);
        return 2;
    }
    if (!have_rs && (!random_scalar(r) || !random_scalar(s))) {
        fprintf(stderr, "cannot read /dev/urandom\n");
        return 1;
    }
    const double t_start = now_ms();

    /* "Reading R1CS" (lib.rs:257-258: R1CSFile::new + R1CS::from) and "Reading ProverParams" (lib.rs:266-268:
     * read_from_file::<ProverParams>), each on its own thread ... */
    parse_job jr, jp;
    memset(&jr, 0, sizeof jr); memset(&jp, 0, sizeof jp);
    jr.path = argv[1]; jr.is_r1cs = 1;
    jp.path = argv[2]; jp.is_r1cs = 0;
    pthread_t tr, tp;
    if (pthread_create(&tr, NULL, parse_thread, &jr) != 0 || pthread_create(&tp, NULL, parse_thread, &jp) != 0) { perror("pthread_create"); return 1; }

    /* ... while this one starts the GPU runtime and reads the witness into page-locked memory */
    double t0 = now_ms();
    if (cg_init(0, NULL) != CG_OK) return die("cg_init");
    uint64_t w_len = 0;
    uint8_t* witness = read_file_pinned(argv[3], &w_len);
    const double gpu_init_s = (now_ms() - t0) / 1e3;
    if (!witness) return 1;
    fprintf(stderr, "%s\n", cg_version());

    pthread_join(tr, NULL);
    pthread_join(tp, NULL);
    if (!jr.ok) { fprintf(stderr, "crescent_prove: %s\n", jr.err); return 1; }
    if (!jp.ok) { fprintf(stderr, "crescent_prove: %s\n", jp.err); return 1; }
    const double t_parsed = now_ms();
    cg_r1cs* r1cs = jr.r1cs;
    cg_prover_params* pp = jp.pp;
    cg_r1cs_header hdr;
    cg_csr abc[3];
    if (cg_r1cs_get(r1cs, &hdr, abc, NULL) != CG_OK) return die("cg_r1cs_get");
    if (w_len != hdr.num_variables * 32) {
        fprintf(stderr, "witness.bin holds %llu bytes, the circuit has %llu wires (x 32 bytes)\n", (unsigned long long)w_len,
                (unsigned long long)hdr.num_variables);
        return 1;
    }
    cg_prover_params_view ppv;
    if (cg_prover_params_get(pp, &ppv) != CG_OK) return die("cg_prover_params_get");

    /* one-time: the key and the matrices into HBM.  Staged: the call returns as soon as the context can prove and the
     * library builds its tables behind the proof below (this program makes ONE proof, as create_client_state does) */
    t0 = now_ms();
    cg_ctx* ctx = NULL;
    cg_options opt;
    memset(&opt, 0, sizeof opt);
    opt.device = -1;
    opt.flags = sync_load ? 0 : CG_FLAG_STAGED_LOAD;
    if (cg_circuit_load(&ctx, &ppv.pk, abc, hdr.num_inputs, hdr.n_constraints, hdr.num_variables, &opt) != CG_OK) return die("cg_circuit_load");
    double t1 = now_ms();

    /* "Groth16 prove" (lib.rs:281-283): Groth16::prove(pk, circuit, rng) */
    uint8_t proof[256];
    cg_timings tm;
    if (cg_prove(ctx, witness, r, s, proof, &tm) != CG_OK) return die("cg_prove");
    double t2 = now_ms();
    cg_ctx_info info;
    if (cg_ctx_get_info(ctx, &info) != CG_OK) return die("cg_ctx_get_info");
    cg_load_timings lt;
    if (cg_ctx_get_load_timings(ctx, &lt) != CG_OK) return die("cg_ctx_get_load_timings");
    fprintf(stderr, "resident: %.2f GB (tables %.2f, matrices %.2f, per proof slot %.2f x %d); windows h/l/a/b1/b2 = %d/%d/%d/%d/%d%s%s; upload %.2f ms\n",
            (double)info.total_bytes / 1e9, (double)info.table_bytes / 1e9, (double)info.matrix_bytes / 1e9, (double)info.slot_bytes / 1e9,
            (int)info.proof_slots, (int)info.window_bits[0], (int)info.window_bits[1], (int)info.window_bits[2], (int)info.window_bits[3],
            (int)info.window_bits[4], info.tuned ? " (re-tuned from this proof)" : "", info.warmup ? " (warm-up arrangement)" : "", tm.upload_ms);
    fprintf(stderr, "circuit: %llu constraints, %llu wires, %llu public; load %.0f ms; prove %.2f ms (witness map %.2f, h %.2f, l %.2f, a %.2f, b1 %.2f, b2 %.2f)\n",
            (unsigned long long)hdr.n_constraints, (unsigned long long)hdr.num_variables, (unsigned long long)hdr.num_inputs, t1 - t0,
            t2 - t1, tm.witness_map_ms, tm.msm_h_ms, tm.msm_l_ms, tm.msm_a_ms, tm.msm_b1_ms, tm.msm_b2_ms);

    /* ClientState::new (lib.rs:292-299): inputs = the public wires after the constant one */
    cg_client_state_view cs;
    memset(&cs, 0, sizeof cs);
    cs.inputs = witness + 32;
    cs.n_inputs = hdr.num_inputs - 1;
    if (aux) { cs.has_aux = 1; cs.aux = (const uint8_t*)aux; cs.aux_len = strlen(aux); }
    cs.proof = proof;
    cs.vk_bytes = ppv.vk_bytes; cs.vk_len = ppv.vk_len;
    cs.pvk_bytes = ppv.pvk_bytes; cs.pvk_len = ppv.pvk_len;
    cs.credtype = (const uint8_t*)credtype; cs.credtype_len = strlen(credtype);
    cs.config_str = ppv.config_str; cs.config_len = ppv.config_len;
    uint64_t out_len = cg_client_state_serialized_size(&cs);
    uint8_t* out = (uint8_t*)malloc(out_len ? out_len : 1);
    if (!out) return 1;
    if (cg_client_state_serialize(&cs, out, out_len) != CG_OK) return die("cg_client_state_serialize");
    FILE* f = fopen(argv[4], "wb");
    if (!f || fwrite(out, 1, out_len, f) != out_len || fclose(f) != 0) { perror(argv[4]); return 1; }
    const double t_done = now_ms();
    fprintf(stderr, "wrote %s (%llu bytes); files -> client_state.bin in %.3f s\n", argv[4], (unsigned long long)out_len, (t_done - t_start) / 1e3);

    if (timings_json) {
        /* the line is about the run above; what follows it (the wait, a second proof) is diagnostic */
        double second_ms = -1.0;
        int ready_rc = cg_ctx_wait_ready(ctx, 120000);
        cg_load_timings lt2 = lt;
        if (ready_rc == CG_OK) {
            if (cg_ctx_get_load_timings(ctx, &lt2) != CG_OK) return die("cg_ctx_get_load_timings");
            double ta = now_ms();
            uint8_t proof2[256];
            if (cg_prove(ctx, witness, r, s, proof2, NULL) != CG_OK) return die("cg_prove (second)");
            second_ms = now_ms() - ta;
            if (memcmp(proof, proof2, 256) != 0) { fprintf(stderr, "crescent_prove: the two arrangements disagree on the proof bytes\n"); return 1; }
        }
        printf("{\"r1cs_bytes\": %llu, \"prover_params_bytes\": %llu, \"r1cs_read_s\": %.4f, \"r1cs_parse_s\": %.4f, \"prover_params_read_s\": %.4f, "
               "\"prover_params_parse_s\": %.4f, \"gpu_init_and_witness_s\": %.4f, \"files_parsed_after_s\": %.4f, \"circuit_load_s\": %.4f, "
               "\"circuit_load_split_ms\": {\"matrices\": %.1f, \"domain\": %.1f, \"key_copy\": %.1f, \"h_query_fold\": %.1f, \"window_tables\": %.1f, "
               "\"slots\": %.1f}, \"staged\": %d, \"first_proof_ms\": %.3f, \"first_proof_upload_ms\": %.3f, \"client_state_write_s\": %.4f, "
               "\"total_s\": %.4f, \"background\": {\"ready_rc\": %d, \"ready_after_load_call_ms\": %.1f, \"h_query_fold_ms\": %.1f, "
               "\"window_tables_ms\": %.1f, \"final_slots_ms\": %.1f, \"swap_wait_ms\": %.2f, \"windows_from_first_proof\": %d}, "
               "\"second_proof_ms\": %.3f, \"second_proof_bytes_identical\": %s}\n",
               (unsigned long long)jr.len, (unsigned long long)jp.len, jr.read_s, jr.parse_s, jp.read_s, jp.parse_s, gpu_init_s,
               (t_parsed - t_start) / 1e3, (t1 - t0) / 1e3, lt.matrices_ms, lt.domain_ms, lt.key_copy_ms, lt.staged ? 0.0 : lt.fold_ms,
               lt.staged ? 0.0 : lt.window_tables_ms, lt.slots_ms, (int)lt.staged, t2 - t1, tm.upload_ms, (t_done - t2) / 1e3, (t_done - t_start) / 1e3,
               ready_rc, lt2.ready_after_ms, lt.staged ? lt2.fold_ms : 0.0, lt.staged ? lt2.window_tables_ms : 0.0, lt2.final_slots_ms,
               lt2.swap_wait_ms, (int)lt2.windows_from_proof, second_ms, second_ms >= 0 ? "true" : "null");
        fflush(stdout);
    }

    free(out);
    cg_circuit_free(ctx);
    cg_prover_params_free(pp);
    cg_r1cs_free(r1cs);
    cg_host_free(witness);
    release_file(jp.bytes, jp.len, jp.mapped);
    release_file(jr.bytes, jr.len, jr.mapped);
    return 0;
}
