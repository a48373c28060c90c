// ark-serialize (uncompressed) reader/writer for the Groth16 proving key.  Host-only code.
//
// Restates what `read_from_file::<ProverParams>` / `write_to_file` (creds/src/utils.rs:140-152,179-189) do for
// the `groth16_params: ProvingKey<Bn254>` that leads `prover_params.bin` (creds/src/lib.rs:58-63): derive-order
// fields (forks/groth16/src/data_structures.rs:31-44,101-118), `Vec<T>` = u64 LE length + items, G1 = x ‖ y and
// G2 = x.c0 ‖ x.c1 ‖ y.c0 ‖ y.c1 as 32-byte LE canonical integers with the two spare top bits of the last byte
// carrying SWFlags (bit 7: y is the larger of {y, -y}; bit 6: infinity) [ark-mem, SURVEY Appendix B].  Reading is
// `deserialize_uncompressed_unchecked`: flags are stripped, no curve or subgroup checks (utils.rs:186).
#include <memory>

#include "common.hpp"

namespace cg {
int translate_current_exception();
}
using namespace cg;

struct cg_pk {
    std::vector<uint8_t> alpha_g1, beta_g2, gamma_g2, delta_g1, delta_g2, gamma_abc_g1, beta_g1, delta_g1_pk;
    std::vector<uint8_t> a_query, b_g1_query, b_g2_query, h_query, l_query;
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
    void points(std::vector<uint8_t>& dst, uint64_t sz) {
        uint64_t n = u64();
        if (n > (len - off) / sz) throw HipError(CG_ERR_PARSE, "vector length exceeds the remaining data");
        dst.reserve(dst.size() + n * sz);
        for (uint64_t i = 0; i < n; ++i) point(dst, sz);
    }
};

bool gt(const Fq& a, const Fq& b) {   // canonical integers
    for (int i = 7; i >= 0; --i) {
        if (a.l[i] > b.l[i]) return true;
        if (a.l[i] < b.l[i]) return false;
    }
    return false;
}
// y of a packed canonical point -> SWFlags
uint8_t g1_flags(const uint8_t* pt) {
    bool zero = true;
    for (int i = 0; i < 64; ++i) if (pt[i]) { zero = false; break; }
    if (zero) return 0x40;
    Fq y = fp_from_bytes<Fq>(pt + 32);
    Fq ny = from_mont(neg(to_mont(y)));
    return gt(y, ny) ? 0x80 : 0x00;
}
uint8_t g2_flags(const uint8_t* pt) {
    bool zero = true;
    for (int i = 0; i < 128; ++i) if (pt[i]) { zero = false; break; }
    if (zero) return 0x40;
    Fq y0 = fp_from_bytes<Fq>(pt + 64), y1 = fp_from_bytes<Fq>(pt + 96);
    Fq n0 = from_mont(neg(to_mont(y0))), n1 = from_mont(neg(to_mont(y1)));
    bool g = gt(y1, n1) || (y1 == n1 && gt(y0, n0));   // QuadExtField order: c1 first, then c0 [ark-mem]
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
    void g1s(const uint8_t* pts, uint64_t n) { u64(n); for (uint64_t i = 0; i < n; ++i) g1(pts + 64 * i); }
    void g2s(const uint8_t* pts, uint64_t n) { u64(n); for (uint64_t i = 0; i < n; ++i) g2(pts + 128 * i); }
};

}  // namespace

extern "C" int cg_pk_parse(const uint8_t* data, uint64_t len, cg_pk** out, uint64_t* bytes_consumed) {
    if (!data || !out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    try {
        std::unique_ptr<cg_pk> k(new cg_pk());
        Rd r{data, len, 0};
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
        r.points(k->a_query, 64);
        r.points(k->b_g1_query, 64);
        r.points(k->b_g2_query, 128);
        r.points(k->h_query, 64);
        r.points(k->l_query, 64);
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
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}
