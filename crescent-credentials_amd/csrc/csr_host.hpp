// Host-side preparation of one R1CS matrix for the GPU (pure host code: no HIP call; tests/cpp/bench_csr_host.cpp times it
// without a GPU): validation of the CSR view, the coefficient dictionary, and the sliced layout the sparse product runs on.
// Replaces what `cs.to_matrices()` hands the prover (forks/groth16/src/r1cs_to_qap.rs:62; operand of :150-155) by the form
// `evaluate_constraint` (:16-45) is executed in here.
//
// Every pass over the 17 M terms of a circuit of the rs256 size runs on all host threads (host_parallel.hpp): the load of a
// circuit is on a cold start's critical path, and until round 6 these passes ran on one thread per matrix (275 ms of a 380 ms
// cg_circuit_load).
#pragma once
#include <string.h>

#include <algorithm>
#include <unordered_map>
#include <vector>

#include "curve.hpp"
#include "errors.hpp"
#include "host_parallel.hpp"

namespace cg {

// One level of the sliced layout the prove path's sparse product runs on (wmap29.hip k_sell29).  A row is cut into
// PIECES of at most SELL_PIECE terms; a lane takes one piece, so no lane walks more than SELL_PIECE terms however long
// the row is (circom's substituted adder rows carry hundreds of terms).  Pieces are sorted by length and stored in
// slices of 64 with the terms of a slice interleaved (term t of the 64 pieces side by side): the index loads of a
// wave are contiguous.  A row of one piece is finished by that piece; the pieces of a longer row leave partial sums
// in a scratch vector and the row becomes a row of the next level, whose "terms" are those partial sums.
static constexpr uint32_t SELL_PIECE = 8;
static constexpr uint32_t SELL_FINAL = 0x80000000u;    // dst flag: the piece is its row's only piece; low bits = row
static constexpr uint32_t SELL_PAD = 0xffffffffu;      // coefficient index of a padding slot

struct HostSellLevel {
    uint32_t n_pieces = 0, n_partials = 0;
    std::vector<uint32_t> slice_ptr;
    RawArray<uint32_t> dst, col, cidx;
    size_t n_slots = 0;                  // entries of col / cidx
};
struct HostCsr {
    uint64_t rows = 0, nnz = 0;
    std::vector<uint32_t> rp;            // rows + 1
    RawArray<uint32_t> idx;              // nnz dictionary indices (0 = the literal one)
    std::vector<Fr> dict_mont;           // the dictionary, Montgomery form
    std::vector<uint32_t> long_rows;     // rows of more than 4096 terms
    std::vector<HostSellLevel> levels;
    uint32_t sell_scratch = 0;
};

// the ranges parallel_ranges would cut [0, n) into, so that two passes can share them (count, scan, fill)
inline std::vector<uint64_t> range_bounds(uint64_t n, uint64_t min_chunk) {
    uint64_t parts = std::min<uint64_t>(host_threads(), (n + min_chunk - 1) / (min_chunk ? min_chunk : 1));
    if (parts < 1) parts = 1;
    std::vector<uint64_t> b(parts + 1);
    for (uint64_t k = 0; k <= parts; ++k) b[k] = n * k / parts;
    return b;
}
// fn(k, lo, hi) for every range of `bounds`, one thread each
template <class Fn>
inline void for_ranges(const std::vector<uint64_t>& bounds, Fn fn) {
    const uint64_t parts = bounds.size() - 1;
    parallel_ranges(parts, 1, [&](uint64_t a, uint64_t b) { for (uint64_t k = a; k < b; ++k) fn(k, bounds[k], bounds[k + 1]); });
}

#ifdef CSR_HOST_TIMING
#include <chrono>
#include <stdio.h>
#define CSR_LAP(name) do { auto _n = std::chrono::steady_clock::now(); fprintf(stderr, "    %-28s %.1f ms\n", name, std::chrono::duration<double, std::milli>(_n - _lap).count()); _lap = _n; } while (0)
#define CSR_LAP_START auto _lap = std::chrono::steady_clock::now()
#else
#define CSR_LAP(name) do {} while (0)
#define CSR_LAP_START do {} while (0)
#endif
namespace csr_detail {
struct Key {
    uint64_t w[4];
    bool operator==(const Key& o) const { return !memcmp(w, o.w, 32); }
};
struct KeyHash {
    size_t operator()(const Key& k) const { return (size_t)(k.w[0] * 0x9e3779b97f4a7c15ull ^ k.w[1] ^ (k.w[2] << 1) ^ (k.w[3] << 7)); }
};
}  // namespace csr_detail

// terms_in_order = false keeps round 2's layout (terms as given, pieces by length): an A/B aid of tuning builds
inline void csr_build_sell_host(HostCsr& out, const uint32_t* col_h, bool reorder);

// Validates the view (row_ptr spans [0, nnz] and is monotone, columns < num_variables, coefficients canonical) and fills
// `out`.  Throws HipError(CG_ERR_INVALID_ARGUMENT).
inline void csr_prepare_host(const cg_csr& m, uint64_t rows, uint64_t num_variables, bool sliced, bool reorder, HostCsr& out) {
    using namespace csr_detail;
    out.rows = rows;
    out.nnz = m.nnz;
    const uint64_t nnz = m.nnz;
    if (nnz >= (1ull << 32)) throw HipError(CG_ERR_INVALID_ARGUMENT, "matrix with >= 2^32 non-zeros");
    if (m.row_ptr[0] != 0 || m.row_ptr[rows] != nnz) throw HipError(CG_ERR_INVALID_ARGUMENT, "row_ptr does not span [0, nnz]");
    CSR_LAP_START;
    out.rp.resize(rows + 1);
    parallel_ranges(rows + 1, 1u << 16, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) {
            if (i && m.row_ptr[i] < m.row_ptr[i - 1]) throw HipError(CG_ERR_INVALID_ARGUMENT, "row_ptr not monotone");
            out.rp[i] = (uint32_t)m.row_ptr[i];
        }
    });
    // ---- coefficient dictionary (circom matrices repeat a handful of constants millions of times).  Every range of terms
    // builds a local dictionary in first-occurrence order; the local dictionaries are merged range by range - which is
    // first-occurrence order over the whole matrix, whatever the thread count - and the indices remapped.  The literal one
    // (the reference's `coeff.is_one()` shortcut, r1cs_to_qap.rs:31-35: four fifths of a circom matrix) never reaches a map.
    CSR_LAP("row_ptr");
    out.idx.alloc(nnz);
    uint32_t* idx = out.idx.p;
    const std::vector<uint64_t> tb = range_bounds(nnz, 1u << 16);
    const uint64_t parts = tb.size() - 1;
    std::vector<std::vector<Key>> local_keys(parts);
    for (auto& v : local_keys) v.reserve(1024);      // (their control blocks sit side by side: keep the early growth off shared lines)
    for_ranges(tb, [&](uint64_t k, uint64_t lo, uint64_t hi) {
        std::unordered_map<Key, uint32_t, KeyHash> map;
        std::vector<Key>& keys = local_keys[k];
        Key last{};
        uint32_t last_id = 0;
        bool have_last = false;
        for (uint64_t t = lo; t < hi; ++t) {
            if (m.col[t] >= num_variables) throw HipError(CG_ERR_INVALID_ARGUMENT, "column index out of range");
            Key key;
            memcpy(key.w, m.coeff + 32 * t, 32);
            if (key.w[0] == 1 && !(key.w[1] | key.w[2] | key.w[3])) { idx[t] = 0; continue; }
            if (have_last && key == last) { idx[t] = last_id; continue; }
            auto it = map.find(key);
            if (it == map.end()) {
                it = map.emplace(key, (uint32_t)keys.size()).first;
                keys.push_back(key);
            }
            last = key; last_id = it->second | 0x80000000u; have_last = true;       // marked: a LOCAL index, remapped below
            idx[t] = last_id;
        }
    });
    CSR_LAP("local dictionaries");
    std::vector<Fr> dict_h;
    {
        Fr o = Fr::zero(); o.l[0] = 1;
        dict_h.push_back(o);
    }
    // The merge runs on MERGE_PARTS threads, each owning the keys of one residue class of the hash: it walks every range's
    // local keys in order and gives its own keys sub-indices in first-occurrence order; the dictionary is the classes one
    // after another.  (The order depends on neither the thread count nor the ranges; the coefficient of value one keeps
    // index 0.)  The synthetic circuits of bench.py carry 250 000 distinct coefficients per matrix - far more than a circom
    // circuit - and a single merging thread spent 60 ms per matrix on them.
    constexpr uint32_t MERGE_PARTS = 16;
    std::vector<std::vector<uint32_t>> remap(parts);
    for (uint64_t k = 0; k < parts; ++k) remap[k].resize(local_keys[k].size());
    std::vector<std::vector<Key>> class_keys(MERGE_PARTS);
    std::vector<std::exception_ptr> class_err(MERGE_PARTS);
    parallel_ranges(MERGE_PARTS, 1, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t cls = lo; cls < hi; ++cls) {
            std::unordered_map<Key, uint32_t, KeyHash> mine;
            std::vector<Key>& ck = class_keys[cls];
            for (uint64_t k = 0; k < parts; ++k)
                for (size_t j = 0; j < local_keys[k].size(); ++j) {
                    const Key& key = local_keys[k][j];
                    if ((KeyHash()(key) >> 7) % MERGE_PARTS != cls) continue;
                    auto it = mine.find(key);
                    if (it == mine.end()) {
                        bool lt = false;                     // key < r ?
                        for (int q = 3; q >= 0; --q) {
                            const uint64_t nq = (uint64_t)FrP::N[2 * q] | ((uint64_t)FrP::N[2 * q + 1] << 32);
                            if (key.w[q] != nq) { lt = key.w[q] < nq; break; }
                        }
                        if (!lt) throw HipError(CG_ERR_INVALID_ARGUMENT, "non-canonical matrix coefficient");
                        it = mine.emplace(key, (uint32_t)ck.size()).first;
                        ck.push_back(key);
                    }
                    remap[k][j] = it->second | ((uint32_t)cls << 27);      // sub-index (< 2^27) and class: resolved below
                }
        }
    });
    uint32_t class_off[MERGE_PARTS + 1];
    class_off[0] = 1;                                        // index 0 is the literal one
    for (uint32_t cls = 0; cls < MERGE_PARTS; ++cls) {
        if (class_keys[cls].size() >= (1u << 27)) throw HipError(CG_ERR_INVALID_ARGUMENT, "too many distinct matrix coefficients");
        class_off[cls + 1] = class_off[cls] + (uint32_t)class_keys[cls].size();
    }
    dict_h.resize(class_off[MERGE_PARTS]);
    parallel_ranges(MERGE_PARTS, 1, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t cls = lo; cls < hi; ++cls)
            for (size_t j = 0; j < class_keys[cls].size(); ++j) memcpy(dict_h[class_off[cls] + j].l, class_keys[cls][j].w, 32);
    });
    for (uint64_t k = 0; k < parts; ++k)
        for (uint32_t& v : remap[k]) v = class_off[v >> 27] + (v & ((1u << 27) - 1));
    CSR_LAP("merge");
    for_ranges(tb, [&](uint64_t k, uint64_t lo, uint64_t hi) {
        const uint32_t* rm = remap[k].data();
        for (uint64_t t = lo; t < hi; ++t)
            if (idx[t] & 0x80000000u) idx[t] = rm[idx[t] & 0x7fffffffu];
    });
    parallel_ranges(dict_h.size(), 1u << 12, [&](uint64_t lo, uint64_t hi) { for (uint64_t i = lo; i < hi; ++i) dict_h[i] = to_mont(dict_h[i]); });
    out.dict_mont.swap(dict_h);
    CSR_LAP("remap + to_mont");
    out.long_rows.clear();
    for (uint64_t i = 0; i < rows; ++i)
        if (out.rp[i + 1] - out.rp[i] > 4096u) out.long_rows.push_back((uint32_t)i);
    out.levels.clear();
    out.sell_scratch = 0;
    CSR_LAP("long rows");
    if (sliced) csr_build_sell_host(out, m.col, reorder);
    CSR_LAP("sliced layout");
}

// the sliced layout, level by level
inline void csr_build_sell_host(HostCsr& out, const uint32_t* col_h, bool reorder) {
    struct Item { uint32_t row, first, len; };              // a row of the current level: `len` terms from `first`
    struct Piece { uint32_t first, len, dst; };
    const uint64_t rows = out.rows, nnz = out.nnz;
    const std::vector<uint32_t>& rp = out.rp;
    CSR_LAP_START;
    RawArray<uint32_t> cur_col, cur_idx;
    cur_col.alloc(nnz);
    cur_idx.alloc(nnz);
    parallel_ranges(nnz, 1u << 18, [&](uint64_t lo, uint64_t hi) {
        memcpy(cur_col.p + lo, col_h + lo, (hi - lo) * 4);
        memcpy(cur_idx.p + lo, out.idx.p + lo, (hi - lo) * 4);
    });
    // Inside a row the terms with a coefficient other than one come first (a sum does not care), so a piece is "k
    // products, then plain additions", and the pieces are sorted by k before they are sliced: the 64 lanes of a slice
    // then agree on which of their steps multiply.  Circom rows are mostly unit coefficients with the powers of two
    // concentrated in the adder rows (gate mix: 82 % of A's rows carry no other coefficient at all), and a wave pays the
    // 207-instruction product at every step at which ANY of its lanes needs it.
    if (reorder)
        parallel_ranges(rows, 1u << 14, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; ++i) {
                uint32_t w = rp[i];
                for (uint32_t t = rp[i]; t < rp[i + 1]; ++t)
                    if (cur_idx.p[t] != 0) {
                        std::swap(cur_idx.p[t], cur_idx.p[w]);
                        std::swap(cur_col.p[t], cur_col.p[w]);
                        ++w;
                    }
            }
        });
    CSR_LAP("  copy + reorder");
    // the rows that have terms, in row order (count per range, scan, fill)
    RawArray<Item> items;
    size_t n_items = 0;
    {
        const std::vector<uint64_t> rb = range_bounds(rows, 1u << 14);
        std::vector<uint64_t> cnt(rb.size(), 0);
        for_ranges(rb, [&](uint64_t k, uint64_t lo, uint64_t hi) {
            uint64_t c = 0;
            for (uint64_t i = lo; i < hi; ++i) c += rp[i + 1] > rp[i];
            cnt[k + 1] = c;
        });
        for (size_t k = 1; k < cnt.size(); ++k) cnt[k] += cnt[k - 1];
        n_items = cnt.back();
        items.alloc(n_items);
        for_ranges(rb, [&](uint64_t k, uint64_t lo, uint64_t hi) {
            uint64_t at = cnt[k];
            for (uint64_t i = lo; i < hi; ++i)
                if (rp[i + 1] > rp[i]) items.p[at++] = {(uint32_t)i, rp[i], rp[i + 1] - rp[i]};
        });
    }
    CSR_LAP("  items");
    const uint32_t* ccol = cur_col.p;
    const uint32_t* cidx = cur_idx.p;
    RawArray<uint32_t> next_col_buf;
    while (n_items) {
        if (out.levels.size() >= 8) throw HipError(CG_ERR_INVALID_ARGUMENT, "matrix row too long for the sliced layout");
        out.levels.emplace_back();
        HostSellLevel& L = out.levels.back();
        // ---- cut the items into pieces: per range the numbers of pieces, of continued rows and of partial sums; scan; fill
        const std::vector<uint64_t> ib = range_bounds(n_items, 1u << 13);
        const size_t nr = ib.size() - 1;
        std::vector<uint64_t> c_pieces(nr + 1, 0), c_next(nr + 1, 0), c_part(nr + 1, 0);
        for_ranges(ib, [&](uint64_t k, uint64_t lo, uint64_t hi) {
            uint64_t p = 0, n = 0, q = 0;
            for (uint64_t i = lo; i < hi; ++i) {
                const uint32_t np = (items.p[i].len + SELL_PIECE - 1) / SELL_PIECE;
                p += np;
                if (np > 1) { n += 1; q += np; }
            }
            c_pieces[k + 1] = p; c_next[k + 1] = n; c_part[k + 1] = q;
        });
        for (size_t k = 1; k <= nr; ++k) { c_pieces[k] += c_pieces[k - 1]; c_next[k] += c_next[k - 1]; c_part[k] += c_part[k - 1]; }
        const uint64_t n_pieces = c_pieces[nr], partials = c_part[nr];
        if (n_pieces >= (1ull << 31) || partials >= (1ull << 31)) throw HipError(CG_ERR_INVALID_ARGUMENT, "matrix too large for the sliced layout");
        RawArray<Piece> pieces, sorted;
        RawArray<Item> next_items;
        pieces.alloc(n_pieces);
        sorted.alloc(n_pieces);
        next_items.alloc(c_next[nr]);
        RawArray<uint32_t> next_col;
        next_col.alloc(partials);
        for_ranges(ib, [&](uint64_t k, uint64_t lo, uint64_t hi) {
            uint64_t p = c_pieces[k], n = c_next[k], q = c_part[k];
            for (uint64_t i = lo; i < hi; ++i) {
                const Item& it = items.p[i];
                const uint32_t np = (it.len + SELL_PIECE - 1) / SELL_PIECE;
                if (np == 1) {
                    pieces.p[p++] = {it.first, it.len, it.row | SELL_FINAL};
                    continue;
                }
                next_items.p[n++] = {it.row, (uint32_t)q, np};
                for (uint32_t j = 0; j < np; ++j) {
                    const uint32_t b = it.first + j * SELL_PIECE;
                    const uint32_t l = it.len - j * SELL_PIECE < SELL_PIECE ? it.len - j * SELL_PIECE : SELL_PIECE;
                    pieces.p[p++] = {b, l, (uint32_t)q};
                    next_col.p[q] = (uint32_t)q;
                    ++q;
                }
            }
        });
        CSR_LAP("  pieces");
        // ---- most products first, then longest first: a slice holds pieces of (nearly) one shape.  A stable counting sort
        // over (SELL_PIECE + 1)^2 keys: keys per range, per-range histograms, scan key-major, scatter.
        constexpr uint32_t NK = (SELL_PIECE + 1) * (SELL_PIECE + 1);
        auto key_of = [&](const Piece& p) {
            uint32_t k = 0;
            if (reorder)
                for (uint32_t t = 0; t < p.len; ++t) k += cidx[p.first + t] != 0;
            return (SELL_PIECE - k) * (SELL_PIECE + 1) + (SELL_PIECE - p.len);
        };
        const std::vector<uint64_t> pb = range_bounds(n_pieces, 1u << 13);
        const size_t npr = pb.size() - 1;
        RawArray<uint8_t> keys;
        keys.alloc(n_pieces);
        std::vector<uint64_t> hist((npr + 1) * NK, 0);
        for_ranges(pb, [&](uint64_t k, uint64_t lo, uint64_t hi) {
            uint64_t* h = &hist[k * NK];
            for (uint64_t i = lo; i < hi; ++i) { keys.p[i] = (uint8_t)key_of(pieces.p[i]); h[keys.p[i]]++; }
        });
        {
            uint64_t run = 0;
            for (uint32_t key = 0; key < NK; ++key)
                for (size_t k = 0; k < npr; ++k) { const uint64_t c = hist[k * NK + key]; hist[k * NK + key] = run; run += c; }
        }
        for_ranges(pb, [&](uint64_t k, uint64_t lo, uint64_t hi) {
            uint64_t* h = &hist[k * NK];
            for (uint64_t i = lo; i < hi; ++i) sorted.p[h[keys.p[i]]++] = pieces.p[i];
        });
        CSR_LAP("  sort");
        const uint32_t np = (uint32_t)n_pieces, ns = (np + 63) / 64;
        L.slice_ptr.assign(ns + 1, 0);
        L.dst.alloc(np ? np : 1);
        if (!np) L.dst.p[0] = 0;
        parallel_ranges(ns, 1u << 10, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t s = lo; s < hi; ++s) {
                uint32_t longest = 0;
                for (uint32_t p = (uint32_t)s * 64; p < np && p < ((uint32_t)s + 1) * 64; ++p) longest = std::max(longest, sorted.p[p].len);
                L.slice_ptr[s + 1] = 64 * longest;
            }
        });
        for (uint32_t s = 0; s < ns; ++s) L.slice_ptr[s + 1] += L.slice_ptr[s];
        const size_t n_slots = L.slice_ptr[ns] ? L.slice_ptr[ns] : 1;
        L.n_slots = n_slots;
        L.col.alloc(n_slots);
        L.cidx.alloc(n_slots);
        // a slice at a time: its slots are filled (padding first), its pieces' terms interleaved
        const uint32_t* sp = L.slice_ptr.data();
        uint32_t* lc = L.col.p;
        uint32_t* li = L.cidx.p;
        if (!ns) { lc[0] = 0; li[0] = SELL_PAD; }
        parallel_ranges(ns, 1u << 8, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t s = lo; s < hi; ++s) {
                for (uint32_t t = sp[s]; t < sp[s + 1]; ++t) { lc[t] = 0; li[t] = SELL_PAD; }
                for (uint32_t p = (uint32_t)s * 64; p < np && p < ((uint32_t)s + 1) * 64; ++p) {
                    const Piece& pc = sorted.p[p];
                    L.dst.p[p] = pc.dst;
                    const uint32_t base = sp[s] + (p & 63);
                    for (uint32_t t = 0; t < pc.len; ++t) {
                        lc[base + t * 64] = ccol[pc.first + t];
                        li[base + t * 64] = cidx[pc.first + t];
                    }
                }
            }
        });
        CSR_LAP("  fill");
        L.n_pieces = np;
        L.n_partials = (uint32_t)partials;
        if (L.n_partials > out.sell_scratch) out.sell_scratch = L.n_partials;
        // the next level sums the partials: unit coefficients over this level's scratch vector
        n_items = c_next[nr];
        items = std::move(next_items);
        next_col_buf = std::move(next_col);
        ccol = next_col_buf.p;
        cur_idx.alloc(partials);
        parallel_ranges(partials, 1u << 18, [&](uint64_t lo, uint64_t hi) { memset(cur_idx.p + lo, 0, (hi - lo) * 4); });
        cidx = cur_idx.p;
    }
}

}  // namespace cg
