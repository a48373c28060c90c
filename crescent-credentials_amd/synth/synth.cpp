// Synthetic, satisfiable R1CS instances of a requested shape, for tests and bench.py.
//
// Real Crescent circuits cannot be produced in this environment (no circom, empty circomlib
// submodule, 0.6 GB artefacts: SURVEY.md 8d), so workloads are seeded stand-ins whose SHAPE follows
// the reference's circuits (creds/test-vectors/README.md:5-10, circuit_setup/inputs/*/config.json):
// two generators, both computing the witness forward so every instance is satisfied:
//   cgs_generate        (round-1 mix) boolean rows b*(b-1)=0 and product rows (Σ a_t w_t)*(Σ b_t w_t) = w_k;
//                       ≈3.4 terms per row over the three matrices - lighter than the real files;
//   cgs_generate_gates  the gate mix of the circuits' own sources (circuit_setup/circuits/utils/sha256general.circom,
//                       rsa.circom, bigint.circom over circomlib's gates.circom / binsum.circom / bitify.circom):
//                       XOR, AND, CH, XOR3 (= mid + out rows), MAJ, 32-bit multi-operand adders with their
//                       booleanity rows, and bigint-limb product rows, tuned to the ≈11.5 terms per row that the
//                       595 MB main_c.r1cs implies (creds/test-vectors/README.md:5-10; SURVEY.md 8d: nnz ≈ 17 M at S21).
// Host-only C++; not part of the prover library.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../csrc/field.hpp"

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


// The gate mix of a circom-compiled SHA-256 / RSA circuit (see the file header).  Wires are defined forward, one per
// row except for the adders' substituted low bit:
//   BOOL  b·(b − 1) = 0                                   (bitify.circom Num2Bits)                     3 terms
//   AND   a·b = out                                        (gates.circom AND)                           3
//   XOR   2a·b = a + b − out                               (gates.circom XOR)                           5
//   CH    a·(b − c) = out − c                              (sha256/ch.circom)                           5
//   XOR3  mid = b·c;  a·(1 − 2b − 2c + 4mid) = out − b − c + 2mid   (sha256/xor3.circom)              3 + 9
//   MAJ   mid = b·c;  a·(b + c − 2mid) = out − mid         (sha256/maj.circom)                        3 + 6
//   ADD   k-operand 32-bit adder (binsum.circom) after the compiler's linear substitution: 34 output bits with a
//         booleanity row each, and the low bit replaced by LC = Σ 2^i in_ji − Σ_{i>=1} 2^i out_i in its own
//         booleanity row LC·(LC − 1) = 0                                                               64k + 170 per 35 rows
//   MUL   (Σ a_t x_t)·(Σ b_t y_t) = w + Σ c_t z_t over ~limb_terms wires a side (bigint.circom limb products)
// Operands come from a window of recently defined wires three times out of four (gates consume their neighbours'
// outputs), else from anywhere earlier.
cgs_instance* cgs_generate_gates(uint64_t seed, uint64_t num_inputs, uint64_t num_constraints, uint64_t num_variables,
                                 double bit_fraction, uint32_t limb_terms) {
    const uint64_t l = num_inputs, m = num_constraints, M = num_variables;
    if (l < 1 || M < l + 1 || M - l < m || m < 1) return nullptr;
    if (m < 256 || limb_terms < 2) return cgs_generate(seed, l, m, M, bit_fraction, limb_terms ? limb_terms : 1);
    if (bit_fraction < 0) bit_fraction = 0;
    if (bit_fraction > 1) bit_fraction = 1;
    const uint64_t n_aux = M - l;
    const uint64_t ADD_ROWS = 35, ADD_WIRES = 34;
    uint64_t G = (uint64_t)(0.4 * bit_fraction * (double)m / (double)ADD_ROWS);
    const uint64_t wires_defined = m - G;
    const uint64_t n_free = n_aux - wires_defined;
    uint64_t n_bits_def = (uint64_t)(bit_fraction * (double)wires_defined + 0.5);
    if (n_bits_def < ADD_WIRES * G) n_bits_def = ADD_WIRES * G;
    uint64_t B_left = n_bits_def - ADD_WIRES * G;       // bit wires defined by the one- and two-row gates
    uint64_t P_left = wires_defined - n_bits_def;       // field-element wires defined by product rows
    uint64_t G_left = G;
    cgs_instance* I = new cgs_instance();
    I->l = l; I->m = m; I->M = M;
    Rng rng(seed);
    std::vector<Fr> w(M);
    std::vector<uint32_t> bits;                          // indices of the wires known to hold 0 or 1
    bits.reserve((size_t)(bit_fraction * (double)M) + 64);
    w[0] = Fr::one();
    bits.push_back(0);
    for (uint64_t i = 1; i < l; ++i) w[i] = rng.field();
    uint64_t v = l;
    for (uint64_t i = 0; i < n_free; ++i, ++v) {
        if (rng.unit() < bit_fraction) { w[v] = (rng.next() & 1) ? Fr::one() : Fr::zero(); bits.push_back((uint32_t)v); }
        else w[v] = rng.field();
    }
    auto canon_u = [](uint64_t x) { Fr c = Fr::zero(); c.l[0] = (uint32_t)x; c.l[1] = (uint32_t)(x >> 32); return c; };
    const Fr one_c = canon_u(1), two_c = canon_u(2), four_c = canon_u(4);
    const Fr one_m = Fr::one();
    const Fr m1_m = cg::neg(one_m), m1_c = cg::from_mont(m1_m);
    const Fr m2_c = cg::from_mont(cg::neg(cg::to_mont(two_c)));
    Fr pow2_c[36], npow2_c[36];
    for (int k = 0; k < 36; ++k) { pow2_c[k] = canon_u(1ull << k); npow2_c[k] = cg::from_mont(cg::neg(cg::to_mont(pow2_c[k]))); }
    auto bitval = [&](uint32_t i) { return w[i] == one_m ? 1u : 0u; };
    auto pick_bit = [&]() -> uint32_t {
        const uint64_t n = bits.size();
        const uint64_t win = n < 4096 ? n : 4096;
        return rng.unit() < 0.75 ? bits[n - 1 - rng.below(win)] : bits[rng.below(n)];
    };
    auto pick_any = [&]() -> uint32_t {
        const uint64_t win = v < 4096 ? v : 4096;
        return (uint32_t)(rng.unit() < 0.75 ? v - 1 - rng.below(win) : rng.below(v));
    };
    auto coeff = [&](Fr& canon, Fr& mont) {
        double u = rng.unit();
        if (u < 0.5) { canon = one_c; mont = one_m; }
        else if (u < 0.65) { canon = m1_c; mont = m1_m; }
        else if (u < 0.9) { int k = 1 + (int)rng.below(120); canon = Fr::zero(); canon.l[k >> 5] = 1u << (k & 31); mont = cg::to_mont(canon); }
        else { mont = rng.field(); canon = cg::from_mont(mont); }
    };
    Matrix& A = I->mat[0]; Matrix& B = I->mat[1]; Matrix& Cm = I->mat[2];
    uint64_t rows = 0;
    auto end_row = [&]() { A.end_row(); B.end_row(); Cm.end_row(); ++rows; };
    auto new_bit = [&](uint32_t val) -> uint32_t {
        w[v] = val ? one_m : Fr::zero();
        bits.push_back((uint32_t)v);
        return (uint32_t)v++;
    };
    auto gate_and = [&](uint32_t a, uint32_t b) -> uint32_t {       // a·b = out
        uint32_t o = new_bit(bitval(a) & bitval(b));
        A.term(a, one_c); B.term(b, one_c); Cm.term(o, one_c);
        end_row();
        return o;
    };
    while (rows < m) {
        const uint64_t add_rows_left = G_left * ADD_ROWS, left = add_rows_left + B_left + P_left;
        uint64_t pick = rng.below(left);
        if (pick < add_rows_left) {
            // ---- adder -------------------------------------------------------------------------------
            --G_left;
            const uint32_t k = 3 + (uint32_t)rng.below(4);          // 3..6 operands: sum < 6·2^32 < 2^35
            uint32_t ops[6][32];
            uint64_t sum = 0;
            for (uint32_t j = 0; j < k; ++j) {
                uint64_t val = 0;
                if (bits.size() >= 64) {                              // a word = 32 consecutive bit wires
                    const uint64_t n = bits.size() - 32;
                    const uint64_t win = n < 4096 ? n : 4096;
                    const uint64_t start = rng.unit() < 0.75 ? n - rng.below(win) : rng.below(n + 1);
                    for (int i = 0; i < 32; ++i) ops[j][i] = bits[start + i];
                } else {
                    for (int i = 0; i < 32; ++i) ops[j][i] = pick_bit();
                }
                for (int i = 0; i < 32; ++i) val |= (uint64_t)bitval(ops[j][i]) << i;
                sum += val;
            }
            uint32_t outs[35];
            for (int i = 1; i < 35; ++i) outs[i] = new_bit((uint32_t)((sum >> i) & 1));
            // low bit, substituted: LC·(LC − 1) = 0
            for (int side = 0; side < 2; ++side) {
                Matrix& X = side ? B : A;
                for (uint32_t j = 0; j < k; ++j)
                    for (int i = 0; i < 32; ++i) X.term(ops[j][i], pow2_c[i]);
                for (int i = 1; i < 35; ++i) X.term(outs[i], npow2_c[i]);
                if (side) X.term(0, m1_c);
            }
            end_row();
            for (int i = 1; i < 35; ++i) {                           // out_i·(out_i − 1) = 0
                A.term(outs[i], one_c); B.term(outs[i], one_c); B.term(0, m1_c);
                end_row();
            }
        } else if (pick < add_rows_left + B_left) {
            // ---- one- and two-row bit gates ----------------------------------------------------------
            double u = rng.unit();
            if (bits.size() < 8) u = 0.0;                            // nothing to combine yet: start from free bits
            if (B_left < 2 && u >= 0.47) u = 0.2;
            if (u < 0.10) {                                          // BOOL
                uint32_t b = new_bit((uint32_t)(rng.next() & 1));
                A.term(b, one_c); B.term(b, one_c); B.term(0, m1_c);
                end_row();
                B_left -= 1;
            } else if (u < 0.17) {                                   // AND
                gate_and(pick_bit(), pick_bit());
                B_left -= 1;
            } else if (u < 0.37) {                                   // XOR: 2a·b = a + b − out
                uint32_t a = pick_bit(), b = pick_bit();
                uint32_t o = new_bit(bitval(a) ^ bitval(b));
                A.term(a, two_c); B.term(b, one_c);
                Cm.term(a, one_c); Cm.term(b, one_c); Cm.term(o, m1_c);
                end_row();
                B_left -= 1;
            } else if (u < 0.47) {                                   // CH: a·(b − c) = out − c
                uint32_t a = pick_bit(), b = pick_bit(), c = pick_bit();
                uint32_t o = new_bit(bitval(a) ? bitval(b) : bitval(c));
                A.term(a, one_c); B.term(b, one_c); B.term(c, m1_c);
                Cm.term(o, one_c); Cm.term(c, m1_c);
                end_row();
                B_left -= 1;
            } else if (u < 0.82) {                                   // XOR3
                uint32_t a = pick_bit(), b = pick_bit(), c = pick_bit();
                uint32_t mid = gate_and(b, c);
                uint32_t o = new_bit(bitval(a) ^ bitval(b) ^ bitval(c));
                A.term(a, one_c);
                B.term(0, one_c); B.term(b, m2_c); B.term(c, m2_c); B.term(mid, four_c);
                Cm.term(o, one_c); Cm.term(b, m1_c); Cm.term(c, m1_c); Cm.term(mid, two_c);
                end_row();
                B_left -= 2;
            } else {                                                 // MAJ
                uint32_t a = pick_bit(), b = pick_bit(), c = pick_bit();
                uint32_t mid = gate_and(b, c);
                uint32_t o = new_bit((bitval(a) + bitval(b) + bitval(c)) >= 2 ? 1u : 0u);
                A.term(a, one_c);
                B.term(b, one_c); B.term(c, one_c); B.term(mid, m2_c);
                Cm.term(o, one_c); Cm.term(mid, m1_c);
                end_row();
                B_left -= 2;
            }
        } else {
            // ---- bigint-limb product row ---------------------------------------------------------------
            --P_left;
            Fr lc[2];
            for (int side = 0; side < 2; ++side) {
                const uint32_t terms = limb_terms / 2 + 1 + (uint32_t)rng.below(limb_terms);
                Fr acc = Fr::zero();
                for (uint32_t t = 0; t < terms; ++t) {
                    uint32_t c = pick_any();
                    Fr canon, mont;
                    coeff(canon, mont);
                    I->mat[side].term(c, canon);
                    acc = cg::add(acc, cg::mul(w[c], mont));
                }
                lc[side] = acc;
            }
            Fr extra = Fr::zero();
            if (rng.unit() < 0.5) {
                const uint32_t terms = 1 + (uint32_t)rng.below(limb_terms / 2 + 1);
                for (uint32_t t = 0; t < terms; ++t) {
                    uint32_t c = pick_any();
                    Fr canon, mont;
                    coeff(canon, mont);
                    Cm.term(c, canon);
                    extra = cg::add(extra, cg::mul(w[c], mont));
                }
            }
            Cm.term((uint32_t)v, one_c);
            w[v] = cg::sub(cg::mul(lc[0], lc[1]), extra);
            ++v;
            end_row();
        }
    }
    if (v != M || rows != m) { delete I; return nullptr; }          // the budgets are exact by construction
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
