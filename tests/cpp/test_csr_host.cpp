// Host-side check (and timing) of csrc/csr_host.hpp - the coefficient dictionary and the sliced layout a matrix is given
// for the GPU's sparse product - against a direct evaluation of the CSR rows.  Plain g++, no GPU.
//
// The layout is EXECUTED here the way k_sell29 (csrc/wmap29.hip) executes it - a piece per lane, term t of a slice's 64
// pieces side by side, partial sums handed to the next level - over the integers mod 2^61 - 1 instead of Fr (the layout does
// not care which ring it sums in), with the dictionary entries and the witness replaced by random residues.
//
//   test_csr_host                  : random matrices of every awkward shape; exits non-zero on a mismatch
//   test_csr_host bench <file>     : time csr_prepare_host on matrices dumped by tools/dump_matrices.py (full-size shapes)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "../../crescent-credentials_amd/csrc/csr_host.hpp"

using namespace cg;

static uint64_t rng_s = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }
static const uint64_t P61 = (1ull << 61) - 1;
static uint64_t mulm(uint64_t a, uint64_t b) { return (uint64_t)((unsigned __int128)a * b % P61); }
static uint64_t addm(uint64_t a, uint64_t b) { return (a + b) % P61; }
#define CHECK(c, ...) do { if (!(c)) { printf("FAIL line %d: ", __LINE__); printf(__VA_ARGS__); printf("\n"); exit(1); } } while (0)

struct Mat {
    std::vector<uint64_t> row_ptr;
    std::vector<uint32_t> col;
    std::vector<uint8_t> coeff;
    cg_csr view() const { return cg_csr{row_ptr.data(), col.data(), coeff.data(), (uint64_t)col.size()}; }
};

// a residue standing in for the coefficient with these 32 bytes (equal bytes -> equal residue)
static uint64_t residue_of(const uint8_t* c) {
    uint64_t w[4];
    memcpy(w, c, 32);
    if (w[0] == 1 && !(w[1] | w[2] | w[3])) return 1;
    return (w[0] * 0x9e3779b97f4a7c15ull ^ w[1] * 31 ^ w[2] * 131 ^ w[3] * 1031) % P61;
}

static void check_matrix(const Mat& m, uint64_t rows, uint64_t cols, bool reorder) {
    HostCsr h;
    csr_prepare_host(m.view(), rows, cols, true, reorder, h);
    const uint64_t nnz = m.col.size();
    // dictionary: index 0 is the literal one (and only it); every term's index names its own coefficient; no entry is
    // unused, none occurs twice
    CHECK(h.dict_mont.size() >= 1, "empty dictionary");
    std::vector<uint64_t> dict_res(h.dict_mont.size(), 0);
    std::vector<uint8_t> used(h.dict_mont.size(), 0);
    dict_res[0] = 1; used[0] = 1;
    for (uint64_t t = 0; t < nnz; ++t) {
        const uint32_t i = h.idx.p[t];
        CHECK(i < h.dict_mont.size(), "index out of the dictionary");
        const uint64_t r = residue_of(&m.coeff[32 * t]);
        CHECK((i == 0) == (r == 1), "index 0 and the literal one do not coincide");
        Fr c;
        memcpy(c.l, &m.coeff[32 * t], 32);
        CHECK(from_mont(h.dict_mont[i]) == c, "dictionary entry %u is not the coefficient", i);
        dict_res[i] = r; used[i] = 1;
    }
    for (size_t i = 0; i < used.size(); ++i) CHECK(used[i], "dictionary holds an entry no term uses");
    for (size_t i = 1; i < h.dict_mont.size(); ++i)
        for (size_t j = i + 1; j < h.dict_mont.size() && h.dict_mont.size() < 4000; ++j) CHECK(!(h.dict_mont[i] == h.dict_mont[j]), "duplicate dictionary entry");
    // direct evaluation
    std::vector<uint64_t> w(cols), want(rows, 0), got(rows, 0);
    for (auto& x : w) x = rnd() % P61;
    for (uint64_t i = 0; i < rows; ++i)
        for (uint64_t t = m.row_ptr[i]; t < m.row_ptr[i + 1]; ++t) want[i] = addm(want[i], mulm(w[m.col[t]], residue_of(&m.coeff[32 * t])));
    // the layout, level by level, as the kernel walks it
    std::vector<uint64_t> src = w, scratch;
    uint64_t terms_seen = 0;
    for (size_t lv = 0; lv < h.levels.size(); ++lv) {
        const HostSellLevel& L = h.levels[lv];
        scratch.assign(L.n_partials ? L.n_partials : 1, 0);
        CHECK(L.n_partials <= h.sell_scratch, "sell_scratch too small");
        uint32_t prev_products = SELL_PIECE + 1, prev_len = SELL_PIECE + 1;
        for (uint32_t p = 0; p < L.n_pieces; ++p) {
            const uint32_t s = p >> 6, lane = p & 63u;
            const uint32_t base = L.slice_ptr[s], len = (L.slice_ptr[s + 1] - base) >> 6;
            CHECK(len <= SELL_PIECE && L.slice_ptr[s + 1] <= L.n_slots, "slice out of shape");
            uint64_t acc = 0;
            uint32_t products = 0, plen = 0;
            bool past_products = false;
            for (uint32_t t = 0; t < len; ++t) {
                const uint32_t ci = L.cidx.p[base + t * 64 + lane];
                if (ci == SELL_PAD) continue;
                const uint32_t cl = L.col.p[base + t * 64 + lane];
                CHECK(cl < src.size(), "column beyond the source vector");
                uint64_t v = src[cl];
                if (ci != 0) { v = mulm(v, dict_res[ci]); ++products; if (reorder && lv == 0) CHECK(!past_products, "a product after a plain term inside a piece"); }
                else past_products = true;
                acc = addm(acc, v);
                ++plen; ++terms_seen;
            }
            if (reorder && lv == 0) {       // most products first, then longest first
                CHECK(products < prev_products || (products == prev_products && plen <= prev_len), "pieces not sorted (piece %u)", p);
                prev_products = products; prev_len = plen;
            }
            const uint32_t d = L.dst.p[p];
            if (d & SELL_FINAL) { CHECK((d & ~SELL_FINAL) < rows, "row out of range"); got[d & ~SELL_FINAL] = addm(got[d & ~SELL_FINAL], acc); }
            else { CHECK(d < L.n_partials, "partial out of range"); scratch[d] = acc; }
        }
        src.swap(scratch);
    }
    for (uint64_t i = 0; i < rows; ++i) CHECK(got[i] == want[i], "row %llu differs", (unsigned long long)i);
    CHECK(terms_seen >= nnz, "layout dropped terms");
    // long rows are listed for the generator's saturated product
    size_t nl = 0;
    for (uint64_t i = 0; i < rows; ++i) nl += m.row_ptr[i + 1] - m.row_ptr[i] > 4096;
    CHECK(nl == h.long_rows.size(), "long rows");
}

static Mat random_matrix(uint64_t rows, uint64_t cols, double mean_terms, int n_coeffs, uint32_t long_every, uint32_t long_len) {
    Mat m;
    m.row_ptr.push_back(0);
    std::vector<std::vector<uint8_t>> pool(n_coeffs, std::vector<uint8_t>(32, 0));
    for (int k = 0; k < n_coeffs; ++k) {
        for (int b = 0; b < 31; ++b) pool[k][b] = (uint8_t)rnd();
        pool[k][31] = (uint8_t)(rnd() & 0x1f);                  // below the modulus
    }
    for (uint64_t i = 0; i < rows; ++i) {
        uint32_t n = (uint32_t)(rnd() % (uint64_t)(2 * mean_terms + 1));
        if (long_every && i % long_every == long_every - 1) n = long_len;
        if (rnd() % 11 == 0) n = 0;                              // empty rows
        for (uint32_t t = 0; t < n; ++t) {
            m.col.push_back((uint32_t)(rnd() % cols));
            uint8_t c[32] = {0};
            const uint64_t pick = rnd() % 10;
            if (pick < 6) c[0] = 1;                               // the literal one
            else memcpy(c, pool[rnd() % n_coeffs].data(), 32);
            m.coeff.insert(m.coeff.end(), c, c + 32);
        }
        m.row_ptr.push_back(m.col.size());
    }
    if (m.col.empty()) { m.col.reserve(1); m.coeff.reserve(32); }
    return m;
}

static int bench(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    uint64_t hdr[3];
    if (fread(hdr, 8, 3, f) != 3) return 1;
    const uint64_t rows = hdr[0], cols = hdr[1], nmat = hdr[2];
    printf("%llu rows, %llu columns, %u host threads\n", (unsigned long long)rows, (unsigned long long)cols, host_threads());
    double total = 0;
    for (uint64_t k = 0; k < nmat; ++k) {
        uint64_t nnz;
        if (fread(&nnz, 8, 1, f) != 1) return 1;
        Mat m;
        m.row_ptr.resize(rows + 1); m.col.resize(nnz); m.coeff.resize(nnz * 32);
        if (fread(m.row_ptr.data(), 8, rows + 1, f) != rows + 1 || fread(m.col.data(), 4, nnz, f) != nnz || fread(m.coeff.data(), 32, nnz, f) != nnz) return 1;
        for (int rep = 0; rep < 3; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            HostCsr h;
            csr_prepare_host(m.view(), rows, cols, true, true, h);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("matrix %llu: nnz %llu, dictionary %zu, levels %zu: %.1f ms\n", (unsigned long long)k, (unsigned long long)nnz, h.dict_mont.size(),
                   h.levels.size(), ms);
            if (rep == 2) total += ms;
        }
    }
    printf("all matrices, one after another: %.1f ms\n", total);
    fclose(f);
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && !strcmp(argv[1], "bench")) return bench(argv[2]);
    // shapes: nothing, one row, rows shorter / equal / longer than a piece, rows of several levels (> 64, > 512 terms), long
    // rows (> 4096), few and many distinct coefficients, more rows than a thread's range
    check_matrix(random_matrix(1, 3, 0.0, 1, 0, 0), 1, 3, true);
    check_matrix(random_matrix(1, 3, 2.0, 1, 0, 0), 1, 3, true);
    check_matrix(random_matrix(50, 40, 3.0, 2, 0, 0), 50, 40, true);
    check_matrix(random_matrix(300, 200, 8.0, 5, 7, 9), 300, 200, true);
    check_matrix(random_matrix(3000, 2500, 5.0, 40, 100, 70), 3000, 2500, true);
    check_matrix(random_matrix(3000, 2500, 5.0, 40, 100, 70), 3000, 2500, false);
    check_matrix(random_matrix(2000, 6000, 4.0, 3000, 400, 600), 2000, 6000, true);
    check_matrix(random_matrix(40000, 30000, 6.0, 9, 9000, 5000), 40000, 30000, true);
    check_matrix(random_matrix(150000, 100000, 4.0, 200000, 0, 0), 150000, 100000, true);
    // rejected inputs
    {
        Mat m = random_matrix(100, 50, 3.0, 4, 0, 0);
        HostCsr h;
        Mat bad = m; bad.col[bad.col.size() / 2] = 50;
        bool threw = false;
        try { csr_prepare_host(bad.view(), 100, 50, true, true, h); } catch (const HipError& e) { threw = e.code == CG_ERR_INVALID_ARGUMENT; }
        CHECK(threw, "column out of range accepted");
        bad = m; memset(&bad.coeff[32 * (bad.col.size() / 3)], 0xff, 32);
        threw = false;
        try { csr_prepare_host(bad.view(), 100, 50, true, true, h); } catch (const HipError& e) { threw = e.code == CG_ERR_INVALID_ARGUMENT; }
        CHECK(threw, "non-canonical coefficient accepted");
        bad = m; bad.row_ptr[40] = bad.row_ptr[41] + 1;
        threw = false;
        try { csr_prepare_host(bad.view(), 100, 50, true, true, h); } catch (const HipError& e) { threw = e.code == CG_ERR_INVALID_ARGUMENT; }
        CHECK(threw, "non-monotone row_ptr accepted");
    }
    printf("ALL OK\n");
    return 0;
}
