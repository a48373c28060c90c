// ark-serialize (uncompressed) reader/writer for the cache files either side of the prove step:
// the Groth16 proving key, the `ProverParams` wrapper of prover_params.bin and the `ClientState` of
// client_state.bin.  Host-only code.
//
// Restates what `read_from_file::<ProverParams>` / `write_to_file` (creds/src/utils.rs:140-152,179-189) do for
// the `groth16_params: ProvingKey<Bn254>` that leads `prover_params.bin` (creds/src/lib.rs:58-63): derive-order
// fields (forks/groth16/src/data_structures.rs:31-44,101-118), `Vec<T>` = u64 LE length + items, G1 = x ‖ y and
// G2 = x.c0 ‖ x.c1 ‖ y.c0 ‖ y.c1 as 32-byte LE canonical integers with the two spare top bits of the last byte
// carrying SWFlags (bit 7: y is the larger of {y, -y}; bit 6: infinity) [ark-mem, SURVEY Appendix B].  Reading is
// `deserialize_uncompressed_unchecked`: flags are stripped, no curve or subgroup checks (utils.rs:186).
#include <memory>

#include "common.hpp"
#include "host_parallel.hpp"

namespace cg {
int translate_current_exception();
}
using namespace cg;

struct cg_pk {
    std::vector<uint8_t> alpha_g1, beta_g2, gamma_g2, delta_g1, delta_g2, gamma_abc_g1, beta_g1, delta_g1_pk;
    RawArray<uint8_t> a_query, b_g1_query, b_g2_query, h_query, l_query;    // 0.6 GB at the rs256 size: filled by worker threads
};
struct cg_prover_params {
    cg_pk pk;
    std::vector<uint8_t> vk_bytes, pvk_bytes, config;
};
struct cg_client_state {
    std::vector<uint8_t> inputs, aux, proof, vk_bytes, pvk_bytes, randomness, openings, credtype, config;
    bool has_aux = false, has_randomness = false;
    uint64_t n_openings = 0;
};

namespace {

struct Rd {
    const uint8_t* p;
    uint64_t len, off;
    void need(uint64_t n) const {
        if (off + n > len || off + n < off) throw HipError(CG_ERR_PARSE, "unexpected end of serialized key");
    }
    uint64_t u64() {
        need(8);
        uint64_t v;
        memcpy(&v, p + off, 8);
        off += 8;
        return v;
    }
    // one uncompressed point of `sz` bytes -> packed canonical (flags stripped, infinity = zeros)
    void point(std::vector<uint8_t>& dst, uint64_t sz) {
        need(sz);
        size_t at = dst.size();
        dst.insert(dst.end(), p + off, p + off + sz);
        uint8_t flags = dst[at + sz - 1] & 0xC0;
        dst[at + sz - 1] &= 0x3F;
        if (flags & 0x40) memset(&dst[at], 0, sz);
        off += sz;
    }
    void raw(std::vector<uint8_t>& dst, uint64_t n) {
        need(n);
        dst.insert(dst.end(), p + off, p + off + n);
        off += n;
    }
    void skip(uint64_t n) { need(n); off += n; }
    uint64_t count(uint64_t item_bytes) {       // a Vec length that the remaining data can actually hold
        uint64_t n = u64();
        if (item_bytes && n > (len - off) / item_bytes) throw HipError(CG_ERR_PARSE, "vector length exceeds the remaining data");
        return n;
    }
    uint8_t byte() { need(1); return p[off++]; }
    // `String` / `Vec<u8>`: u64 length + bytes
    void bytes(std::vector<uint8_t>& dst) { uint64_t n = count(1); raw(dst, n); }
    // skip one serialized VerifyingKey (data_structures.rs:31-44); returns nothing, advances
    void skip_vk() { skip(64 + 128 + 128 + 64 + 128); uint64_t n = count(64); skip(n * 64); }
    // skip one G2Prepared (ark-ec bn::G2Prepared [ark-mem]: ell_coeffs: Vec<(Fq2, Fq2, Fq2)>, infinity: bool)
    void skip_g2_prepared() { uint64_t n = count(192); skip(n * 192); skip(1); }
    // skip one PreparedVerifyingKey (data_structures.rs:62-71): vk, alpha_g1_beta_g2: Fq12, two G2Prepared
    void skip_pvk() { skip_vk(); skip(384); skip_g2_prepared(); skip_g2_prepared(); }
    void points(std::vector<uint8_t>& dst, uint64_t sz) {
        uint64_t n = u64();
        if (n > (len - off) / sz) throw HipError(CG_ERR_PARSE, "vector length exceeds the remaining data");
        dst.reserve(dst.size() + n * sz);
        for (uint64_t i = 0; i < n; ++i) point(dst, sz);
    }
    // a query of the key: the same as points(), on all host threads (a straight copy, then the flag byte of every point)
    void query(RawArray<uint8_t>& dst, uint64_t sz) {
        const uint64_t n = u64();
        if (n > (len - off) / sz) throw HipError(CG_ERR_PARSE, "vector length exceeds the remaining data");
        dst.alloc(n * sz);
        const uint8_t* src = p + off;
        uint8_t* out = dst.p;
        parallel_ranges(n, 1u << 14, [&](uint64_t lo, uint64_t hi) {
            memcpy(out + lo * sz, src + lo * sz, (hi - lo) * sz);
            for (uint64_t i = lo; i < hi; ++i) {
                uint8_t* last = out + i * sz + sz - 1;
                const uint8_t flags = *last & 0xC0;
                *last &= 0x3F;
                if (flags & 0x40) memset(out + i * sz, 0, sz);
            }
        });
        off += n * sz;
    }
};

// y > -y for a canonical y, i.e. y > q - y, i.e. y > (q - 1) / 2 (q is odd; y = 0 is its own negative and not "greater"):
// one comparison with a constant instead of a modular negation per point (a key has 8 M of them)
bool gt_half_q(const uint8_t* y32) {
    static const Fq HALF = [] {
        Fq h;
        uint32_t carry = 0;
        for (int i = 7; i >= 0; --i) {          // (q - 1) >> 1; q - 1 only clears bit 0
            const uint32_t w = i == 0 ? FqP::N[0] - 1u : FqP::N[i];
            h.l[i] = (w >> 1) | (carry << 31);
            carry = w & 1u;
        }
        return h;
    }();
    const Fq y = fp_from_bytes<Fq>(y32);
    for (int i = 7; i >= 0; --i) {
        if (y.l[i] > HALF.l[i]) return true;
        if (y.l[i] < HALF.l[i]) return false;
    }
    return false;
}
bool all_zero(const uint8_t* p, int n) {
    for (int i = 0; i < n; ++i) if (p[i]) return false;
    return true;
}
// y of a packed canonical point -> SWFlags
uint8_t g1_flags(const uint8_t* pt) {
    if (all_zero(pt, 64)) return 0x40;
    return gt_half_q(pt + 32) ? 0x80 : 0x00;
}
uint8_t g2_flags(const uint8_t* pt) {
    if (all_zero(pt, 128)) return 0x40;
    // QuadExtField order: c1 first, then c0 [ark-mem]; c1 and -c1 differ unless c1 = 0
    const bool g = all_zero(pt + 96, 32) ? gt_half_q(pt + 64) : gt_half_q(pt + 96);
    return g ? 0x80 : 0x00;
}
struct Wr {
    uint8_t* p;
    uint64_t len, off;
    void need(uint64_t n) const {
        if (off + n > len) throw HipError(CG_ERR_INVALID_ARGUMENT, "output buffer too small");
    }
    void u64(uint64_t v) { need(8); memcpy(p + off, &v, 8); off += 8; }
    void g1(const uint8_t* pt) { need(64); memcpy(p + off, pt, 64); p[off + 63] |= g1_flags(pt); off += 64; }
    void g2(const uint8_t* pt) { need(128); memcpy(p + off, pt, 128); p[off + 127] |= g2_flags(pt); off += 128; }
    void raw(const uint8_t* b, uint64_t n) { need(n); if (n) memcpy(p + off, b, n); off += n; }
    void bytes(const uint8_t* b, uint64_t n) { u64(n); raw(b, n); }
    void byte(uint8_t v) { need(1); p[off++] = v; }
    // a query: the points are copied and flagged on all host threads
    void g1s(const uint8_t* pts, uint64_t n) {
        u64(n); need(64 * n);
        uint8_t* dst = p + off;
        parallel_ranges(n, 1u << 14, [&](uint64_t lo, uint64_t hi) {
            memcpy(dst + 64 * lo, pts + 64 * lo, 64 * (hi - lo));
            for (uint64_t i = lo; i < hi; ++i) dst[64 * i + 63] |= g1_flags(pts + 64 * i);
        });
        off += 64 * n;
    }
    void g2s(const uint8_t* pts, uint64_t n) {
        u64(n); need(128 * n);
        uint8_t* dst = p + off;
        parallel_ranges(n, 1u << 13, [&](uint64_t lo, uint64_t hi) {
            memcpy(dst + 128 * lo, pts + 128 * lo, 128 * (hi - lo));
            for (uint64_t i = lo; i < hi; ++i) dst[128 * i + 127] |= g2_flags(pts + 128 * i);
        });
        off += 128 * n;
    }
};

}  // namespace

static void read_pk(Rd& r, cg_pk* k) {
    // VerifyingKey (data_structures.rs:31-44)
    r.point(k->alpha_g1, 64);
    r.point(k->beta_g2, 128);
    r.point(k->gamma_g2, 128);
    r.point(k->delta_g1, 64);
    r.point(k->delta_g2, 128);
    r.points(k->gamma_abc_g1, 64);
    // ProvingKey (data_structures.rs:101-118)
    r.point(k->beta_g1, 64);
    r.point(k->delta_g1_pk, 64);
    r.query(k->a_query, 64);
    r.query(k->b_g1_query, 64);
    r.query(k->b_g2_query, 128);
    r.query(k->h_query, 64);
    r.query(k->l_query, 64);
}
static void pk_view(const cg_pk* k, cg_proving_key* view) {
    view->coord_form = CG_FORM_CANONICAL;
    view->alpha_g1 = k->alpha_g1.data();
    view->beta_g1 = k->beta_g1.data();
    view->delta_g1 = k->delta_g1_pk.data();
    view->beta_g2 = k->beta_g2.data();
    view->delta_g2 = k->delta_g2.data();
    view->a_query = k->a_query.data(); view->a_len = k->a_query.size() / 64;
    view->b_g1_query = k->b_g1_query.data(); view->b_g1_len = k->b_g1_query.size() / 64;
    view->b_g2_query = k->b_g2_query.data(); view->b_g2_len = k->b_g2_query.size() / 128;
    view->h_query = k->h_query.data(); view->h_len = k->h_query.size() / 64;
    view->l_query = k->l_query.data(); view->l_len = k->l_query.size() / 64;
}
static void write_pk(Wr& w, const cg_proving_key* pk, const uint8_t* gamma_g2, const uint8_t* gamma_abc_g1, uint64_t gamma_abc_len) {
    w.g1(pk->alpha_g1);            // vk.alpha_g1
    w.g2(pk->beta_g2);             // vk.beta_g2
    w.g2(gamma_g2);                // vk.gamma_g2
    w.g1(pk->delta_g1);            // vk.delta_g1 (the fork's extra field, data_structures.rs:38-39)
    w.g2(pk->delta_g2);            // vk.delta_g2
    w.g1s(gamma_abc_g1, gamma_abc_len);
    w.g1(pk->beta_g1);
    w.g1(pk->delta_g1);
    w.g1s(pk->a_query, pk->a_len);
    w.g1s(pk->b_g1_query, pk->b_g1_len);
    w.g2s(pk->b_g2_query, pk->b_g2_len);
    w.g1s(pk->h_query, pk->h_len);
    w.g1s(pk->l_query, pk->l_len);
}

extern "C" int cg_pk_parse(const uint8_t* data, uint64_t len, cg_pk** out, uint64_t* bytes_consumed) {
    if (!data || !out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    try {
        std::unique_ptr<cg_pk> k(new cg_pk());
        Rd r{data, len, 0};
        read_pk(r, k.get());
        if (bytes_consumed) *bytes_consumed = r.off;
        *out = k.release();
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}

extern "C" int cg_pk_get(const cg_pk* k, cg_proving_key* view, const uint8_t** gamma_g2, const uint8_t** gamma_abc_g1,
                         uint64_t* gamma_abc_len) {
    if (!k || !view) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    pk_view(k, view);
    if (gamma_g2) *gamma_g2 = k->gamma_g2.data();
    if (gamma_abc_g1) *gamma_abc_g1 = k->gamma_abc_g1.data();
    if (gamma_abc_len) *gamma_abc_len = k->gamma_abc_g1.size() / 64;
    return CG_OK;
}

extern "C" void cg_pk_free(cg_pk* k) { delete k; }

extern "C" uint64_t cg_pk_serialized_size(const cg_proving_key* pk, uint64_t gamma_abc_len) {
    if (!pk) return 0;
    return 64 + 128 + 128 + 64 + 128 + (8 + 64 * gamma_abc_len) + 64 + 64 + (8 + 64 * pk->a_len) + (8 + 64 * pk->b_g1_len) +
           (8 + 128 * pk->b_g2_len) + (8 + 64 * pk->h_len) + (8 + 64 * pk->l_len);
}

extern "C" int cg_pk_serialize(const cg_proving_key* pk, const uint8_t* gamma_g2, const uint8_t* gamma_abc_g1, uint64_t gamma_abc_len,
                               uint8_t* out, uint64_t out_len) {
    if (!pk || !gamma_g2 || (!gamma_abc_g1 && gamma_abc_len) || !out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    if (pk->coord_form != CG_FORM_CANONICAL) return fail(CG_ERR_INVALID_ARGUMENT, "serialisation takes canonical coordinates");
    try {
        Wr w{out, out_len, 0};
        write_pk(w, pk, gamma_g2, gamma_abc_g1, gamma_abc_len);
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}


// ---------------------------------------------------------------------------------------------
// prover_params.bin: ProverParams { groth16_params: ProvingKey, groth16_pvk: PreparedVerifyingKey, config_str: String }
// (creds/src/lib.rs:58-63, written by run_zksetup :245-248, read by create_client_state :268).  The prover needs the
// key; the prepared verifying key and the configuration string are what it must hand on to the ClientState
// (creds/src/lib.rs:292-299), so they are returned verbatim.
// ---------------------------------------------------------------------------------------------
extern "C" int cg_prover_params_parse(const uint8_t* data, uint64_t len, cg_prover_params** out) {
    if (!data || !out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    try {
        std::unique_ptr<cg_prover_params> pp(new cg_prover_params());
        Rd r{data, len, 0};
        { Rd v{data, len, 0}; v.skip_vk(); pp->vk_bytes.assign(data, data + v.off); }   // groth16_params.vk leads the file
        read_pk(r, &pp->pk);
        const uint64_t pvk_at = r.off;
        r.skip_pvk();
        pp->pvk_bytes.assign(data + pvk_at, data + r.off);
        r.bytes(pp->config);
        if (r.off != len) throw HipError(CG_ERR_PARSE, "trailing bytes after ProverParams");
        *out = pp.release();
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}
extern "C" int cg_prover_params_get(const cg_prover_params* pp, cg_prover_params_view* v) {
    if (!pp || !v) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    pk_view(&pp->pk, &v->pk);
    v->gamma_g2 = pp->pk.gamma_g2.data();
    v->gamma_abc_g1 = pp->pk.gamma_abc_g1.data();
    v->gamma_abc_len = pp->pk.gamma_abc_g1.size() / 64;
    v->vk_bytes = pp->vk_bytes.data(); v->vk_len = pp->vk_bytes.size();
    v->pvk_bytes = pp->pvk_bytes.data(); v->pvk_len = pp->pvk_bytes.size();
    v->config_str = pp->config.data(); v->config_len = pp->config.size();
    return CG_OK;
}
extern "C" void cg_prover_params_free(cg_prover_params* pp) { delete pp; }
extern "C" uint64_t cg_prover_params_serialized_size(const cg_proving_key* pk, uint64_t gamma_abc_len, uint64_t pvk_len, uint64_t config_len) {
    if (!pk) return 0;
    return cg_pk_serialized_size(pk, gamma_abc_len) + pvk_len + 8 + config_len;
}
extern "C" int cg_prover_params_serialize(const cg_proving_key* pk, const uint8_t* gamma_g2, const uint8_t* gamma_abc_g1,
                                          uint64_t gamma_abc_len, const uint8_t* pvk_bytes, uint64_t pvk_len,
                                          const uint8_t* config_str, uint64_t config_len, uint8_t* out, uint64_t out_len) {
    if (!pk || !gamma_g2 || (!gamma_abc_g1 && gamma_abc_len) || !pvk_bytes || (!config_str && config_len) || !out)
        return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    if (pk->coord_form != CG_FORM_CANONICAL) return fail(CG_ERR_INVALID_ARGUMENT, "serialisation takes canonical coordinates");
    try {
        { Rd chk{pvk_bytes, pvk_len, 0}; chk.skip_pvk(); if (chk.off != pvk_len) throw HipError(CG_ERR_PARSE, "pvk_bytes is not one PreparedVerifyingKey"); }
        Wr w{out, out_len, 0};
        write_pk(w, pk, gamma_g2, gamma_abc_g1, gamma_abc_len);
        w.raw(pvk_bytes, pvk_len);
        w.bytes(config_str, config_len);
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}

// ---------------------------------------------------------------------------------------------
// client_state.bin: ClientState (creds/src/groth16rand.rs:23-35), the hand-over from `prove` to the host-side `show`:
//   inputs: Vec<Fr> | aux: Option<String> | proof: Proof | vk: VerifyingKey | pvk: PreparedVerifyingKey |
//   input_com_randomness: Option<Fr> | committed_input_openings: Vec<PedersenOpening<G1>> | credtype: String |
//   config_str: String
// `Option<T>` = one tag byte (0 / 1) + T; PedersenOpening (creds/src/dlog.rs:24-29) = bases: Vec<G1Affine>, m: Fr,
// r: Fr, c: G1 (a projective point serialises as its affine form).  A state fresh out of `create_client_state`
// (creds/src/lib.rs:292-300) has no randomness and no openings; a parsed one returns both verbatim.
// ---------------------------------------------------------------------------------------------
static void skip_opening(Rd& r) { uint64_t n = r.count(64); r.skip(n * 64 + 32 + 32 + 64); }

extern "C" uint64_t cg_client_state_serialized_size(const cg_client_state_view* v) {
    if (!v) return 0;
    return 8 + 32 * v->n_inputs + 1 + (v->has_aux ? 8 + v->aux_len : 0) + 256 + v->vk_len + v->pvk_len + 1 +
           (v->has_input_com_randomness ? 32 : 0) + 8 + v->openings_len + 8 + v->credtype_len + 8 + v->config_len;
}
extern "C" int cg_client_state_serialize(const cg_client_state_view* v, uint8_t* out, uint64_t out_len) {
    if (!v || !out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    if ((v->n_inputs && !v->inputs) || !v->proof || !v->vk_bytes || !v->pvk_bytes || (v->has_aux && v->aux_len && !v->aux) ||
        (v->has_input_com_randomness && !v->input_com_randomness) || (v->openings_len && !v->openings_bytes) ||
        (v->credtype_len && !v->credtype) || (v->config_len && !v->config_str))
        return fail(CG_ERR_INVALID_ARGUMENT, "null member with a non-zero length");
    try {
        for (uint64_t i = 0; i < v->n_inputs; ++i)
            if (!scalar_is_canonical(v->inputs + 32 * i)) throw HipError(CG_ERR_INVALID_ARGUMENT, "input not a canonical field element");
        { Rd c{v->vk_bytes, v->vk_len, 0}; c.skip_vk(); if (c.off != v->vk_len) throw HipError(CG_ERR_PARSE, "vk_bytes is not one VerifyingKey"); }
        { Rd c{v->pvk_bytes, v->pvk_len, 0}; c.skip_pvk(); if (c.off != v->pvk_len) throw HipError(CG_ERR_PARSE, "pvk_bytes is not one PreparedVerifyingKey"); }
        { Rd c{v->openings_bytes, v->openings_len, 0}; for (uint64_t i = 0; i < v->n_openings; ++i) skip_opening(c);
          if (c.off != v->openings_len) throw HipError(CG_ERR_PARSE, "openings_bytes does not hold n_openings items"); }
        Wr w{out, out_len, 0};
        w.u64(v->n_inputs); w.raw(v->inputs, 32 * v->n_inputs);
        w.byte(v->has_aux ? 1 : 0);
        if (v->has_aux) w.bytes(v->aux, v->aux_len);
        w.raw(v->proof, 256);
        w.raw(v->vk_bytes, v->vk_len);
        w.raw(v->pvk_bytes, v->pvk_len);
        w.byte(v->has_input_com_randomness ? 1 : 0);
        if (v->has_input_com_randomness) w.raw(v->input_com_randomness, 32);
        w.u64(v->n_openings); w.raw(v->openings_bytes, v->openings_len);
        w.bytes(v->credtype, v->credtype_len);
        w.bytes(v->config_str, v->config_len);
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}
extern "C" int cg_client_state_parse(const uint8_t* data, uint64_t len, cg_client_state** out) {
    if (!data || !out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    try {
        std::unique_ptr<cg_client_state> cs(new cg_client_state());
        Rd r{data, len, 0};
        uint64_t n = r.count(32);
        r.raw(cs->inputs, 32 * n);
        uint8_t tag = r.byte();
        if (tag > 1) throw HipError(CG_ERR_PARSE, "bad Option tag (aux)");
        cs->has_aux = tag == 1;
        if (cs->has_aux) r.bytes(cs->aux);
        r.raw(cs->proof, 256);
        uint64_t at = r.off; r.skip_vk(); cs->vk_bytes.assign(data + at, data + r.off);
        at = r.off; r.skip_pvk(); cs->pvk_bytes.assign(data + at, data + r.off);
        tag = r.byte();
        if (tag > 1) throw HipError(CG_ERR_PARSE, "bad Option tag (input_com_randomness)");
        cs->has_randomness = tag == 1;
        if (cs->has_randomness) r.raw(cs->randomness, 32);
        cs->n_openings = r.count(64 + 32 + 32 + 8);
        at = r.off;
        for (uint64_t i = 0; i < cs->n_openings; ++i) skip_opening(r);
        cs->openings.assign(data + at, data + r.off);
        r.bytes(cs->credtype);
        r.bytes(cs->config);
        if (r.off != len) throw HipError(CG_ERR_PARSE, "trailing bytes after ClientState");
        *out = cs.release();
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}
extern "C" int cg_client_state_get(const cg_client_state* cs, cg_client_state_view* v) {
    if (!cs || !v) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    memset(v, 0, sizeof(*v));
    v->inputs = cs->inputs.data(); v->n_inputs = cs->inputs.size() / 32;
    v->has_aux = cs->has_aux; v->aux = cs->aux.data(); v->aux_len = cs->aux.size();
    v->proof = cs->proof.data();
    v->vk_bytes = cs->vk_bytes.data(); v->vk_len = cs->vk_bytes.size();
    v->pvk_bytes = cs->pvk_bytes.data(); v->pvk_len = cs->pvk_bytes.size();
    v->has_input_com_randomness = cs->has_randomness; v->input_com_randomness = cs->randomness.data();
    v->openings_bytes = cs->openings.data(); v->openings_len = cs->openings.size(); v->n_openings = cs->n_openings;
    v->credtype = cs->credtype.data(); v->credtype_len = cs->credtype.size();
    v->config_str = cs->config.data(); v->config_len = cs->config.size();
    return CG_OK;
}
extern "C" void cg_client_state_free(cg_client_state* cs) { delete cs; }
