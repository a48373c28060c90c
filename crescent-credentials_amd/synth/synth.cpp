// Synthetic, satisfiable R1CS instances of a requested shape, for tests and bench.py.
//
// Real Crescent circuits cannot be produced in this environment (no circom, empty circomlib
// submodule, 0.6 GB artefacts: SURVEY.md 8d), so workloads are seeded stand-ins whose SHAPE follows
// the reference's circuits (creds/test-vectors/README.md:5-10, circuit_setup/inputs/*/config.json):
// rows are a mix of boolean rows b*(b-1)=0, as the SHA-256 / bit-decomposition gadgets emit, and
// product rows (Σ a_t w_t)*(Σ b_t w_t) = w_k over earlier wires; the witness is computed forward
// so every instance is satisfied.  Host-only C++; not part of the prover library.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../csrc/field.cuh"

using cg::Fr;

namespace {

struct Rng {  // xoshiro256**
    uint64_t s[4];
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    explicit Rng(uint64_t seed) {
        uint64_t z = seed;
        for (int i = 0; i < 4; ++i) {  // splitmix64
            z += 0x9e3779b97f4a7c15ull;
            uint64_t t = z;
            t = (t ^ (t >> 30)) * 0xbf58476d1ce4e5b9ull;
            t = (t ^ (t >> 27)) * 0x94d049bb133111ebull;
            s[i] = t ^ (t >> 31);
        }
    }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    uint64_t below(uint64_t n) { return next() % n; }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    Fr field() {  // uniform canonical element, returned in Montgomery form
        Fr a;
        for (;;) {
            for (int i = 0; i < 4; ++i) {
                uint64_t v = next();
                a.l[2 * i] = (uint32_t)v;
                a.l[2 * i + 1] = (uint32_t)(v >> 32);
            }
            a.l[7] &= 0x3fffffffu;
            bool lt = false;
            for (int i = 7; i >= 0; --i) {
                if (a.l[i] < cg::FrP::N[i]) { lt = true; break; }
                if (a.l[i] > cg::FrP::N[i]) break;
            }
            if (lt) return cg::to_mont(a);
        }
    }
};

struct Matrix {
    std::vector<uint64_t> row_ptr{0};
    std::vector<uint32_t> col;
    std::vector<uint8_t> coeff;
    void term(uint32_t c, const Fr& canonical) {
        col.push_back(c);
        const uint8_t* b = (const uint8_t*)canonical.l;
        coeff.insert(coeff.end(), b, b + 32);
    }
    void end_row() { row_ptr.push_back(col.size()); }
};

}  // namespace

struct cgs_instance {
    uint64_t l, m, M;
    Matrix mat[3];
    std::vector<uint8_t> witness;  // M x 32 canonical
};

extern "C" {

// bit_fraction: share of the aux wires that are boolean (0 = all wires uniform field elements,
// 0.9 = "circom-like" 45 % zero / 45 % one / 10 % uniform).  lc_terms: mean terms per LC of a product row.
cgs_instance* cgs_generate(uint64_t seed, uint64_t num_inputs, uint64_t num_constraints, uint64_t num_variables,
                           double bit_fraction, uint32_t lc_terms) {
    const uint64_t l = num_inputs, m = num_constraints, M = num_variables;
    if (l < 1 || M < l + 1 || M - l < m || m < 1) return nullptr;
    uint64_t n_aux = M - l;
    uint64_t n_bits = (uint64_t)(bit_fraction * (double)n_aux);
    if (n_bits > m) n_bits = m;
    uint64_t n_prod = m - n_bits;
    uint64_t n_free = n_aux - n_bits - n_prod;
    if (n_free + l < 2 && n_bits == 0) return nullptr;
    cgs_instance* I = new cgs_instance();
    I->l = l; I->m = m; I->M = M;
    Rng rng(seed);
    std::vector<Fr> w(M);  // Montgomery
    w[0] = Fr::one();
    for (uint64_t i = 1; i < l; ++i) w[i] = rng.field();
    uint64_t v = l;
    for (uint64_t i = 0; i < n_free; ++i) w[v++] = rng.field();
    const uint64_t first_bit = v;
    for (uint64_t i = 0; i < n_bits; ++i) w[v++] = (rng.next() & 1) ? Fr::one() : Fr::zero();
    const uint64_t first_prod = v;
    // coefficient palette (canonical): 1, -1, powers of two, occasionally a random element
    Fr one_c = Fr::zero(); one_c.l[0] = 1;
    Fr minus_one_c = cg::from_mont(cg::neg(Fr::one()));
    auto coeff = [&](Fr& canon, Fr& mont) {
        double u = rng.unit();
        if (u < 0.5) { canon = one_c; mont = Fr::one(); }
        else if (u < 0.65) { canon = minus_one_c; mont = cg::neg(Fr::one()); }
        else if (u < 0.9) {
            int k = 1 + (int)rng.below(200);
            canon = Fr::zero();
            canon.l[k >> 5] = 1u << (k & 31);
            mont = cg::to_mont(canon);
        } else { mont = rng.field(); canon = cg::from_mont(mont); }
    };
    // rows are emitted in an interleaved order so boolean and product rows mix as in a real circuit
    uint64_t bits_done = 0, prod_done = 0;
    for (uint64_t row = 0; row < m; ++row) {
        bool do_bit;
        if (bits_done == n_bits) do_bit = false;
        else if (prod_done == n_prod) do_bit = true;
        else do_bit = rng.unit() < (double)(n_bits - bits_done) / (double)(m - row);
        if (do_bit) {
            uint32_t b = (uint32_t)(first_bit + bits_done++);
            I->mat[0].term(b, one_c);                       // b
            I->mat[1].term(b, one_c);                       // b - 1
            I->mat[1].term(0, minus_one_c);
        } else {
            uint32_t k = (uint32_t)(first_prod + prod_done);
            uint64_t avail = first_prod + prod_done;        // wires defined so far
            ++prod_done;
            Fr lc[2];
            for (int side = 0; side < 2; ++side) {
                uint32_t terms = 1 + (lc_terms > 1 ? (uint32_t)rng.below(2 * lc_terms - 1) : 0);
                Fr acc = Fr::zero();
                for (uint32_t t = 0; t < terms; ++t) {
                    uint32_t c = (uint32_t)rng.below(avail);
                    Fr canon, mont;
                    coeff(canon, mont);
                    I->mat[side].term(c, canon);
                    acc = cg::add(acc, cg::mul(w[c], mont));
                }
                lc[side] = acc;
            }
            Fr prod = cg::mul(lc[0], lc[1]);
            // C row: w_k + Σ c_t w_t  (extra terms with probability 1/4)
            Fr extra = Fr::zero();
            if (lc_terms > 1 && rng.unit() < 0.25) {
                uint32_t terms = 1 + (uint32_t)rng.below(lc_terms);
                for (uint32_t t = 0; t < terms; ++t) {
                    uint32_t c = (uint32_t)rng.below(avail);
                    Fr canon, mont;
                    coeff(canon, mont);
                    I->mat[2].term(c, canon);
                    extra = cg::add(extra, cg::mul(w[c], mont));
                }
            }
            I->mat[2].term(k, one_c);
            w[k] = cg::sub(prod, extra);
        }
        for (int q = 0; q < 3; ++q) I->mat[q].end_row();
    }
    I->witness.resize(M * 32);
    for (uint64_t i = 0; i < M; ++i) {
        Fr c = cg::from_mont(w[i]);
        memcpy(&I->witness[32 * i], c.l, 32);
    }
    return I;
}

// A second satisfying witness for the same circuit is not derivable in general; fresh proofs use new (r, s).

void cgs_views(const cgs_instance* I, const uint64_t** row_ptr, const uint32_t** col, const uint8_t** coeff, uint64_t* nnz,
               const uint8_t** witness) {
    for (int k = 0; k < 3; ++k) {
        row_ptr[k] = I->mat[k].row_ptr.data();
        col[k] = I->mat[k].col.data();
        coeff[k] = I->mat[k].coeff.data();
        nnz[k] = I->mat[k].col.size();
    }
    *witness = I->witness.data();
}

void cgs_free(cgs_instance* I) { delete I; }

}  // extern "C"
