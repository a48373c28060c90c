// .r1cs (iden3 binary format) -> CSR matrices.  Host-only code.
//
// Restates `R1CSFile::new`, `Header::new`, `read_constraints`, `read_map` and `R1CS::from`
// (forks/circom-compat/src/circom/r1cs_reader.rs:54-148,162-202,205-236,238-256,26-38): same
// accepted inputs, same rejections.  Column index = wire id because Crescent disables the wire
// mapping (forks/circom-compat/src/circom/builder.rs:63-64; circuit.rs:61-67).
#include <memory>

#include "common.hpp"
#include "host_parallel.hpp"

namespace cg {
int translate_current_exception();
}
using namespace cg;

struct cg_r1cs {
    cg_r1cs_header header;
    std::vector<uint64_t> row_ptr[3];
    RawArray<uint32_t> col[3];          // filled by the parser's worker threads: not zeroed first
    RawArray<uint8_t> coeff[3];
    uint64_t nnz[3] = {0, 0, 0};
    std::vector<uint64_t> wire_mapping;
};

namespace {

struct Reader {
    const uint8_t* p;
    uint64_t len, off;
    void need(uint64_t n) const {
        if (off + n > len || off + n < off) throw HipError(CG_ERR_PARSE, "unexpected end of r1cs data");
    }
    uint32_t u32() {
        need(4);
        uint32_t v;
        memcpy(&v, p + off, 4);
        off += 4;
        return v;
    }
    uint64_t u64() {
        need(8);
        uint64_t v;
        memcpy(&v, p + off, 8);
        off += 8;
        return v;
    }
    void bytes(uint8_t* dst, uint64_t n) {
        need(n);
        memcpy(dst, p + off, n);
        off += n;
    }
};

// Fr modulus, little-endian (r1cs_reader.rs:183)
const uint8_t FR_MODULUS_LE[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
                                   0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

}  // namespace

extern "C" int cg_r1cs_parse(const uint8_t* data, uint64_t len, cg_r1cs** out) {
    if (!data || !out) return fail(CG_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    try {
        Reader rd{data, len, 0};
        uint8_t magic[4];
        rd.bytes(magic, 4);
        if (memcmp(magic, "\x72\x31\x63\x73", 4) != 0) throw HipError(CG_ERR_PARSE, "Invalid magic number");   // :57-62
        if (rd.u32() != 1) throw HipError(CG_ERR_PARSE, "Unsupported version");                                 // :64-70
        uint32_t nsec = rd.u32();
        bool have[4] = {false, false, false, false};
        uint64_t sec_off[4] = {0, 0, 0, 0}, sec_size[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < nsec; ++i) {                                                                  // :80-87
            uint32_t ty = rd.u32();
            uint64_t sz = rd.u64();
            if (ty < 4) { have[ty] = true; sec_off[ty] = rd.off; sec_size[ty] = sz; }
            rd.need(sz);
            rd.off += sz;
        }
        if (!have[1]) throw HipError(CG_ERR_PARSE, "No section offset for header type found");
        if (!have[2]) throw HipError(CG_ERR_PARSE, "No section offset for constraint type found");
        if (!have[3]) throw HipError(CG_ERR_PARSE, "No section offset for wire2label type found");
        std::unique_ptr<cg_r1cs> r(new cg_r1cs());
        // header (:162-202)
        rd.off = sec_off[1];
        cg_r1cs_header& h = r->header;
        h.field_size = rd.u32();
        if (h.field_size != 32) throw HipError(CG_ERR_PARSE, "This parser only supports 32-byte fields");
        if (sec_size[1] != 32 + (uint64_t)h.field_size) throw HipError(CG_ERR_PARSE, "Invalid header section size");
        uint8_t prime[32];
        rd.bytes(prime, 32);
        if (memcmp(prime, FR_MODULUS_LE, 32) != 0) throw HipError(CG_ERR_PARSE, "This parser only supports bn256");
        h.n_wires = rd.u32();
        h.n_pub_out = rd.u32();
        h.n_pub_in = rd.u32();
        h.n_prv_in = rd.u32();
        h.n_labels = rd.u64();
        h.n_constraints = rd.u32();
        h.num_inputs = 1ull + h.n_pub_in + h.n_pub_out;    // r1cs_reader.rs:28
        h.num_variables = h.n_wires;                       // :29
        if (h.num_inputs > h.num_variables) throw HipError(CG_ERR_PARSE, "more public signals than wires");
        // constraints (:205-236).  The reference sizes its buffer as off(section 3) - off(section 2)
        // (:125) and fails on a short read; bound the walk the same way.
        uint64_t cend = sec_off[3] > sec_off[2] ? sec_off[3] : sec_off[2] + sec_size[2];
        if (cend > len) cend = len;
        // Two passes.  A constraint is three blocks of `n` then n x (u32 wire, 32-byte coefficient): the block boundaries
        // are only known by walking, so pass 1 walks the counts alone (one 4-byte read per block: 4.4 M reads at the rs256
        // size) and fixes every block's place in the output; pass 2 copies and checks the terms - 36 bytes each, 17 M of
        // them, 0.6 GB - on all host threads, every thread a contiguous range of constraints.  (Rounds 2-5 pushed every
        // term onto a std::vector from one thread: ~0.25 GB/s.)
        const uint64_t nc = h.n_constraints;
        // (a constraint is at least its three counts: a header that promises more constraints than the section can hold is
        // refused before anything is sized by it)
        if (cend < sec_off[2] || nc > (cend - sec_off[2]) / 12) throw HipError(CG_ERR_PARSE, "unexpected end of r1cs data");
        for (int k = 0; k < 3; ++k) r->row_ptr[k].assign(nc + 1, 0);
        std::vector<uint64_t> block_at(nc + 1, 0);             // byte offset of constraint i's first block
        {
            Reader cr{data, cend, sec_off[2]};
            for (uint64_t i = 0; i < nc; ++i) {
                block_at[i] = cr.off;
                for (int k = 0; k < 3; ++k) {
                    const uint64_t n = cr.u32();
                    cr.need(n * 36);
                    cr.off += n * 36;
                    r->row_ptr[k][i + 1] = r->row_ptr[k][i] + n;
                }
            }
            block_at[nc] = cr.off;
        }
        for (int k = 0; k < 3; ++k) {
            r->nnz[k] = r->row_ptr[k][nc];
            r->col[k].alloc(r->nnz[k]);
            r->coeff[k].alloc(r->nnz[k] * 32);
        }
        // Fr modulus as four u64 (canonical check: most coefficients are decided by the top word)
        uint64_t mod[4];
        memcpy(mod, FR_MODULUS_LE, 32);
        const uint32_t n_wires = h.n_wires;
        cg_r1cs* rr = r.get();
        parallel_ranges(nc, 4096, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; ++i) {
                const uint8_t* p = data + block_at[i];
                for (int k = 0; k < 3; ++k) {
                    uint32_t n;
                    memcpy(&n, p, 4);
                    p += 4;
                    uint64_t t = rr->row_ptr[k][i];
                    uint32_t* col = rr->col[k].p + t;
                    uint8_t* co = rr->coeff[k].p + t * 32;
                    for (uint32_t j = 0; j < n; ++j, p += 36) {
                        uint32_t wire;
                        uint64_t c[4];
                        memcpy(&wire, p, 4);
                        memcpy(c, p + 4, 32);
                        bool lt = false;                       // c < r ?
                        for (int q = 3; q >= 0; --q) {
                            if (c[q] != mod[q]) { lt = c[q] < mod[q]; break; }
                        }
                        if (!lt) throw HipError(CG_ERR_PARSE, "non-canonical coefficient");  // deserialize_uncompressed rejects
                        if (wire >= n_wires) throw HipError(CG_ERR_PARSE, "wire id out of range");
                        col[j] = wire;
                        memcpy(co + 32 * (size_t)j, c, 32);
                    }
                }
            }
        });
        // wire map (:238-256)
        if (sec_size[3] != (uint64_t)h.n_wires * 8) throw HipError(CG_ERR_PARSE, "Invalid map section size");
        rd.off = sec_off[3];
        r->wire_mapping.resize(h.n_wires);
        for (uint32_t i = 0; i < h.n_wires; ++i) r->wire_mapping[i] = rd.u64();
        if (h.n_wires == 0 || r->wire_mapping[0] != 0) throw HipError(CG_ERR_PARSE, "Wire 0 should always be mapped to 0");
        *out = r.release();
        return CG_OK;
    } catch (...) {
        return translate_current_exception();
    }
}

extern "C" int cg_r1cs_get(const cg_r1cs* r, cg_r1cs_header* header, cg_csr abc[3], const uint64_t** wire_mapping) {
    if (!r) return fail(CG_ERR_INVALID_ARGUMENT, "null r1cs");
    if (header) *header = r->header;
    if (abc)
        for (int k = 0; k < 3; ++k) {
            abc[k].row_ptr = r->row_ptr[k].data();
            abc[k].col = r->col[k].data();
            abc[k].coeff = r->coeff[k].data();
            abc[k].nnz = r->nnz[k];
        }
    if (wire_mapping) *wire_mapping = r->wire_mapping.data();
    return CG_OK;
}

extern "C" void cg_r1cs_free(cg_r1cs* r) { delete r; }
