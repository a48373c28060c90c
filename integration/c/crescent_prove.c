/*
 * crescent_prove — the reference's `create_client_state` (creds/src/lib.rs:255-301) as a plain C program over
 * libcrescent_gpu's C ABI: no Python, no torch, nothing but <include/crescent_gpu.h>.
 *
 *   crescent_prove <main_c.r1cs> <prover_params.bin> <witness.bin> <client_state.bin>
 *                  [--rs r_hex s_hex] [--credtype jwt|mdl] [--aux prover_aux_json]
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
 *
 * This is the reference-side binding in its smallest form: the Rust shim (integration/rust/crescent-gpu) makes the
 * same calls in the same order.  It is also the proof that the header is valid C and the ABI needs nothing else.
 */
#include <crescent_gpu.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static const uint8_t FR_MODULUS_LE[32] = { /* r1cs_reader.rs:183 */
    0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
    0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

static int die(const char* what) {
    fprintf(stderr, "crescent_prove: %s: %s\n", what, cg_last_error());
    return 1;
}

static uint8_t* read_file(const char* path, uint64_t* len) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return NULL; }
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return NULL; }
    long n = ftell(f);
    if (n < 0) { fclose(f); return NULL; }
    rewind(f);
    uint8_t* buf = (uint8_t*)malloc(n ? (size_t)n : 1);
    if (!buf || fread(buf, 1, (size_t)n, f) != (size_t)n) { perror(path); free(buf); fclose(f); return NULL; }
    fclose(f);
    *len = (uint64_t)n;
    return buf;
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

int main(int argc, char** argv) {
    const char* credtype = "jwt";
    const char* aux = NULL;
    uint8_t r[32], s[32];
    int have_rs = 0, bad = argc < 5;
    for (int i = 5; i < argc && !bad; ++i) {
        if (!strcmp(argv[i], "--rs") && i + 2 < argc) {
            if (!hex_scalar(argv[i + 1], r) || !hex_scalar(argv[i + 2], s)) { fprintf(stderr, "--rs: need hex values below the scalar modulus\n"); return 2; }
            have_rs = 1; i += 2;
        } else if (!strcmp(argv[i], "--credtype") && i + 1 < argc) credtype = argv[++i];
        else if (!strcmp(argv[i], "--aux") && i + 1 < argc) aux = argv[++i];
        else bad = 1;
    }
    if (bad) {
        fprintf(stderr, "usage: %s main_c.r1cs prover_params.bin witness.bin client_state.bin [--rs r_hex s_hex] [--credtype jwt|mdl] [--aux json]\n", argv[0]);
        return 2;
    }
    if (!have_rs && (!random_scalar(r) || !random_scalar(s))) {
        fprintf(stderr, "cannot read /dev/urandom\n");
        return 1;
    }

    uint64_t r1cs_len = 0, pp_len = 0, w_len = 0;
    uint8_t* r1cs_bytes = read_file(argv[1], &r1cs_len);
    uint8_t* pp_bytes = read_file(argv[2], &pp_len);
    if (!r1cs_bytes || !pp_bytes) return 1;

    if (cg_init(0, NULL) != CG_OK) return die("cg_init");
    fprintf(stderr, "%s\n", cg_version());
    uint8_t* witness = read_file_pinned(argv[3], &w_len);
    if (!witness) return 1;

    /* the circuit: R1CSFile::new + R1CS::from */
    cg_r1cs* r1cs = NULL;
    if (cg_r1cs_parse(r1cs_bytes, r1cs_len, &r1cs) != CG_OK) return die("cg_r1cs_parse");
    cg_r1cs_header hdr;
    cg_csr abc[3];
    if (cg_r1cs_get(r1cs, &hdr, abc, NULL) != CG_OK) return die("cg_r1cs_get");
    if (w_len != hdr.num_variables * 32) {
        fprintf(stderr, "witness.bin holds %llu bytes, the circuit has %llu wires (x 32 bytes)\n", (unsigned long long)w_len,
                (unsigned long long)hdr.num_variables);
        return 1;
    }

    /* the parameters: read_from_file::<ProverParams> */
    cg_prover_params* pp = NULL;
    if (cg_prover_params_parse(pp_bytes, pp_len, &pp) != CG_OK) return die("cg_prover_params_parse");
    cg_prover_params_view ppv;
    if (cg_prover_params_get(pp, &ppv) != CG_OK) return die("cg_prover_params_get");

    /* one-time: key tables and matrices into HBM */
    double t0 = now_ms();
    cg_ctx* ctx = NULL;
    cg_options opt;
    memset(&opt, 0, sizeof opt);
    opt.device = -1;
    if (cg_circuit_load(&ctx, &ppv.pk, abc, hdr.num_inputs, hdr.n_constraints, hdr.num_variables, &opt) != CG_OK) return die("cg_circuit_load");
    double t1 = now_ms();

    /* Groth16::prove(pk, circuit, rng) */
    uint8_t proof[256];
    cg_timings tm;
    if (cg_prove(ctx, witness, r, s, proof, &tm) != CG_OK) return die("cg_prove");
    double t2 = now_ms();
    cg_ctx_info info;
    if (cg_ctx_get_info(ctx, &info) != CG_OK) return die("cg_ctx_get_info");
    fprintf(stderr, "resident: %.2f GB (tables %.2f, matrices %.2f, per proof slot %.2f x %d); windows h/l/a/b1/b2 = %d/%d/%d/%d/%d%s; upload %.2f ms\n",
            (double)info.total_bytes / 1e9, (double)info.table_bytes / 1e9, (double)info.matrix_bytes / 1e9, (double)info.slot_bytes / 1e9,
            (int)info.proof_slots, (int)info.window_bits[0], (int)info.window_bits[1], (int)info.window_bits[2], (int)info.window_bits[3],
            (int)info.window_bits[4], info.tuned ? " (re-tuned from this proof)" : "", tm.upload_ms);
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
    fprintf(stderr, "wrote %s (%llu bytes)\n", argv[4], (unsigned long long)out_len);

    free(out);
    cg_circuit_free(ctx);
    cg_prover_params_free(pp);
    cg_r1cs_free(r1cs);
    cg_host_free(witness); free(pp_bytes); free(r1cs_bytes);
    return 0;
}
