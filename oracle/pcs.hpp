// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).  PARITY UNPINNED vs Plonky3@bdd338d6.
//
// Restates the PCS the reference instantiates (TwoAdicFriPcs<..., Radix2Bowers, FieldMerkleTreeMmcs,
// ExtensionMmcs>, basic/tests/test_prover.rs:430-452) at the call sites Valida uses:
//   pcs.commit_batches / commit_shifted_batches   basic/src/lib.rs:199,223,258,599
//   pcs.get_ldes                                  basic/src/lib.rs:201,225,261
//   pcs.open_multi_batches                        basic/src/lib.rs:618-619
//   pcs.verify_multi_batches                      basic/src/lib.rs:825-837
// following SURVEY.md Appendix B3-B5, B9-B10, B12.
#pragma once
#include <algorithm>
#include <map>
#include <omp.h>
#include "hash.hpp"
#include "fast.hpp"

namespace oracle {

// ORACLE_TIMING=1 in the environment: wall-clock of every phase of prove() on stderr (where the CPU baseline's seconds go)
struct PhaseClock {
    bool on = getenv("ORACLE_TIMING") != nullptr;
    double t0 = omp_get_wtime();
    void lap(const char* what) { if (!on) return; double t = omp_get_wtime(); fprintf(stderr, "oracle phase %-28s %8.3f s\n", what, t - t0); t0 = t; }
};

// ---------------------------------------------------------------- DFT (App. B3)
// In-place iterative radix-2 on one column (natural in, natural out).
inline void dft_inplace(std::vector<Fp>& a, bool inverse, bool parallel = false) {
    size_t n = a.size();
    unsigned k = log2_strict(n);
    parallel = parallel && n >= (size_t(1) << 14);  // threads INSIDE one transform: for matrices with fewer columns than cores
    #pragma omp parallel for if (parallel) schedule(static)
    for (size_t i = 0; i < n; i++) { size_t j = reverse_bits_len(i, k); if (i < j) std::swap(a[i], a[j]); }
    for (unsigned s = 1; s <= k; s++) {
        size_t m = size_t(1) << s, half = m / 2;
        Fp wm = two_adic_generator(s);
        if (inverse) wm = wm.inv();
        std::vector<Fp> tw(half);
        tw[0] = Fp::one();
        for (size_t j = 1; j < half; j++) tw[j] = tw[j - 1] * wm;
        #pragma omp parallel for if (parallel) schedule(static)
        for (size_t t = 0; t < n / 2; t++) {  // butterfly t: block t / half, position t % half
            size_t j = t & (half - 1), lo = ((t >> (s - 1)) << s) + j, hi = lo + half;
            Fp x = tw[j] * a[hi], u = a[lo];
            a[lo] = u + x;
            a[hi] = u - x;
        }
    }
    if (inverse) {
        Fp ninv = Fp((uint32_t)(n % P)).inv();
        #pragma omp parallel for if (parallel) schedule(static)
        for (size_t i = 0; i < n; i++) a[i] *= ninv;
    }
}

// O(n^2) definition, used only by tests to pin dft_inplace.
inline std::vector<Fp> naive_dft(const std::vector<Fp>& a) {
    size_t n = a.size();
    Fp w = two_adic_generator(log2_strict(n));
    std::vector<Fp> out(n);
    for (size_t i = 0; i < n; i++) {
        Fp wi = w.pow(i), acc = Fp::zero(), pw = Fp::one();
        for (size_t j = 0; j < n; j++) { acc += a[j] * pw; pw *= wi; }
        out[i] = acc;
    }
    return out;
}

// TwoAdicSubgroupDft::coset_lde_batch(mat, added_bits, shift): iDFT over H_n, zero-extend, evaluate on
// shift*H_{n<<added_bits}; natural row order.
inline Matrix coset_lde_batch(const Matrix& m, unsigned added_bits, Fp shift) {
    size_t n = m.height, N = n << added_bits;
    Matrix out(N, m.width);
    auto one_column = [&](size_t c, bool inner_parallel) {
        std::vector<Fp> col(n);
        for (size_t r = 0; r < n; r++) col[r] = m.at(r, c);
        dft_inplace(col, true, inner_parallel);
        col.resize(N);
        Fp pw = Fp::one();
        for (size_t i = 0; i < n; i++) { col[i] *= pw; pw *= shift; }
        dft_inplace(col, false, inner_parallel);
        for (size_t r = 0; r < N; r++) out.at(r, c) = col[r];
    };
    // wide matrices: a thread per column; tall narrow ones (the memory chip: 14 columns of 2^22 rows): threads inside each transform
    if (m.width * 4 >= (size_t)omp_get_max_threads() || n < (size_t(1) << 14)) {
        #pragma omp parallel for schedule(dynamic)
        for (size_t c = 0; c < m.width; c++) one_column(c, false);
    } else {
        for (size_t c = 0; c < m.width; c++) one_column(c, true);
    }
    return out;
}

inline Matrix bit_reverse_rows(const Matrix& m) {
    Matrix out(m.height, m.width);
    unsigned k = log2_strict(m.height);
    #pragma omp parallel for schedule(static)
    for (size_t r = 0; r < m.height; r++) {
        size_t j = reverse_bits_len(r, k);
        std::copy(m.row(r), m.row(r) + m.width, &out.v[j * m.width]);
    }
    return out;
}

// ---------------------------------------------------------------- MMCS (App. B5)
struct MerkleTree {
    std::vector<Matrix> leaves;                      // committed matrices, commit order
    std::vector<std::vector<Digest>> digest_layers;  // [0] = leaf layer ... back() = {root}
    Digest root() const { return digest_layers.back()[0]; }
    size_t max_height() const { size_t h = 0; for (auto& m : leaves) h = std::max(h, m.height); return h; }
};

inline Digest hash_rows(const std::vector<const Matrix*>& mats, size_t r) {
    std::vector<Fp> buf;
    for (auto* m : mats) buf.insert(buf.end(), m->row(r), m->row(r) + m->width);
    return hash_elems(buf);
}

// FieldMerkleTree::new: stable sort by height descending; leaf = H(concat rows of the tallest);
// next[i] = C(prev[2i], prev[2i+1]); if matrices of height == len(next): next[i] = C(next[i], H(rows i)).
inline MerkleTree mmcs_commit(std::vector<Matrix> mats) {
    MerkleTree t;
    t.leaves = std::move(mats);
    std::vector<const Matrix*> order;
    for (auto& m : t.leaves) { log2_strict(m.height); order.push_back(&m); }
    std::stable_sort(order.begin(), order.end(), [](const Matrix* a, const Matrix* b) { return a->height > b->height; });
    size_t pos = 0, maxh = order[0]->height;
    std::vector<const Matrix*> group;
    while (pos < order.size() && order[pos]->height == maxh) group.push_back(order[pos++]);
    std::vector<Digest> layer(maxh);
    const bool fast_keccak = fast::enabled() && mmcs_hash().kind == 0;  // fast mode: unrolled permutation, rows absorbed in place (fast.hpp)
    if (fast_keccak && maxh >= 4) {  // four rows per call
        #pragma omp parallel for schedule(static)
        for (size_t r = 0; r < maxh; r += 4) fast::keccak_hash_rows_x4(group, r, &layer[r]);
    } else {
        #pragma omp parallel for
        for (size_t r = 0; r < maxh; r++) layer[r] = fast_keccak ? fast::keccak_hash_rows(group, r) : hash_rows(group, r);
    }
    t.digest_layers.push_back(std::move(layer));
    while (t.digest_layers.back().size() > 1) {
        const auto& prev = t.digest_layers.back();
        size_t len = prev.size() / 2;
        group.clear();
        while (pos < order.size() && order[pos]->height == len) group.push_back(order[pos++]);
        std::vector<Digest> next(len);
        if (fast_keccak && len >= 4) {  // four parents per call
            #pragma omp parallel for schedule(static)
            for (size_t i = 0; i < len; i += 4) {
                const Digest* l[4] = {&prev[2 * i], &prev[2 * i + 2], &prev[2 * i + 4], &prev[2 * i + 6]};
                const Digest* r[4] = {&prev[2 * i + 1], &prev[2 * i + 3], &prev[2 * i + 5], &prev[2 * i + 7]};
                if (group.empty()) { fast::keccak_compress_x4(l, r, &next[i]); continue; }
                Digest d[4], h[4];
                fast::keccak_compress_x4(l, r, d);
                fast::keccak_hash_rows_x4(group, i, h);
                const Digest* dp[4] = {&d[0], &d[1], &d[2], &d[3]};
                const Digest* hp[4] = {&h[0], &h[1], &h[2], &h[3]};
                fast::keccak_compress_x4(dp, hp, &next[i]);
            }
            t.digest_layers.push_back(std::move(next));
            continue;
        }
        #pragma omp parallel for
        for (size_t i = 0; i < len; i++) {
            if (fast_keccak) {
                Digest d = fast::keccak_compress(prev[2 * i], prev[2 * i + 1]);
                if (!group.empty()) d = fast::keccak_compress(d, fast::keccak_hash_rows(group, i));
                next[i] = d;
                continue;
            }
            Digest d = compress(prev[2 * i], prev[2 * i + 1]);
            if (!group.empty()) d = compress(d, hash_rows(group, i));
            next[i] = d;
        }
        t.digest_layers.push_back(std::move(next));
    }
    if (pos != order.size()) { fprintf(stderr, "oracle: mmcs: matrix shorter than 1 row?\n"); abort(); }
    return t;
}

struct BatchOpening {
    std::vector<std::vector<Fp>> opened_values;  // per matrix (commit order): row index>>bits_reduced
    std::vector<Digest> opening_proof;           // sibling path, leaf to root
};

inline BatchOpening mmcs_open(const MerkleTree& t, size_t index) {
    BatchOpening o;
    unsigned log_max = log2_ceil(t.max_height());
    for (auto& m : t.leaves) {
        unsigned lh = log2_ceil(m.height);
        size_t r = index >> (log_max - lh);
        o.opened_values.emplace_back(m.row(r), m.row(r) + m.width);
    }
    for (unsigned i = 0; i < log_max; i++) o.opening_proof.push_back(t.digest_layers[i][(index >> i) ^ 1]);
    return o;
}

// verify_batch: heights (not widths) drive the grouping.
inline bool mmcs_verify(const Digest& commit, const std::vector<size_t>& heights, size_t index,
                        const BatchOpening& o) {
    if (heights.size() != o.opened_values.size()) return false;
    std::vector<size_t> order(heights.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return heights[a] > heights[b]; });
    size_t pos = 0, cur = heights[order[0]];
    if (o.opening_proof.size() != log2_ceil(cur)) return false;
    auto take = [&](size_t h) {
        std::vector<Fp> buf;
        while (pos < order.size() && heights[order[pos]] == h) {
            auto& v = o.opened_values[order[pos++]];
            buf.insert(buf.end(), v.begin(), v.end());
        }
        return buf;
    };
    Digest root = hash_elems(take(cur));
    for (auto& sib : o.opening_proof) {
        root = (index & 1) ? compress(sib, root) : compress(root, sib);
        index >>= 1;
        cur >>= 1;
        if (pos < order.size() && heights[order[pos]] == cur) root = compress(root, hash_elems(take(cur)));
    }
    return pos == order.size() && root == commit;
}

// ---------------------------------------------------------------- PCS commit (App. B4)
struct FriConfig {
    unsigned log_blowup = 1, num_queries = 40, pow_bits = 8;
    bool observe_final_poly = false;  // convention switch (unpinned): absorb final_poly before grinding
};

inline Fp coset_shift() { return Fp(GENERATOR); }

// commit_shifted_batches: lde_i = coset_lde_batch(m_i, log_blowup, 31 / shift_i), rows bit-reversed.
inline MerkleTree pcs_commit(const std::vector<Matrix>& polys, const std::vector<Fp>& shifts, const FriConfig& cfg) {
    const double t_start = omp_get_wtime();
    std::vector<Matrix> ldes;
    for (size_t i = 0; i < polys.size(); i++) {
        Fp s = coset_shift() * shifts[i].inv();
        if (fast::enabled()) ldes.push_back(fast::coset_lde_bitrev(polys[i], cfg.log_blowup, s));  // fast mode: AVX2 Montgomery transforms (fast.hpp)
        else ldes.push_back(bit_reverse_rows(coset_lde_batch(polys[i], cfg.log_blowup, s)));
    }
    PhaseClock clk;
    clk.t0 = t_start;
    clk.lap("  lde");
    MerkleTree t = mmcs_commit(std::move(ldes));
    clk.lap("  merkle tree");
    return t;
}
inline MerkleTree pcs_commit(const std::vector<Matrix>& polys, const FriConfig& cfg) {
    return pcs_commit(polys, std::vector<Fp>(polys.size(), Fp::one()), cfg);
}
// get_ldes: natural-order view of a committed (bit-reversed) LDE.
struct LdeView {
    const Matrix* m;
    unsigned k;
    explicit LdeView(const Matrix* mm) : m(mm), k(log2_strict(mm->height)) {}
    size_t height() const { return m->height; }
    size_t width() const { return m->width; }
    Fp get(size_t r, size_t c) const { return m->at(reverse_bits_len(r, k), c); }
};

// ---------------------------------------------------------------- opening + FRI (App. B9, B10, B12)
struct CommitPhaseStep { Ext5 sibling_value; std::vector<Digest> opening_proof; };
struct QueryProof { std::vector<CommitPhaseStep> commit_phase_openings; };
struct FriProof {
    std::vector<Digest> commit_phase_commits;
    std::vector<QueryProof> query_proofs;
    Ext5 final_poly;
    Fp pow_witness;
};
struct PcsProof {
    FriProof fri;
    std::vector<std::vector<BatchOpening>> query_openings;  // [query][round]
};
// openings[round][matrix][point][column]
using OpenedValues = std::vector<std::vector<std::vector<std::vector<Ext5>>>>;

// interpolate_coset: evaluate every column of the degree<n interpolant of `evals` (given on
// shift*H_n, natural order) at `z`.
inline std::vector<Ext5> interpolate_coset(const Matrix& lde_bitrev, size_t n, Fp shift, const Ext5& z) {
    unsigned k = log2_strict(n);
    Fp g = two_adic_generator(k);
    std::vector<Ext5> w(n);  // w_i = g^i / (z - shift g^i), natural i
    #pragma omp parallel
    {
        // each thread walks a contiguous range of i with its own running power of g (the inversions dominate)
        #pragma omp for schedule(static)
        for (size_t blk = 0; blk < (n + 4095) / 4096; blk++) {
            size_t i0 = blk * 4096, i1 = std::min(n, i0 + 4096);
            Fp gi = g.pow(i0);
            for (size_t i = i0; i < i1; i++) { w[i] = (z - shift * gi).inv() * gi; gi *= g; }
        }
    }
    Ext5 zerofier = z.exp_power_of_2(k) - shift.exp_power_of_2(k);
    Fp denom = Fp((uint32_t)(n % P)) * shift.pow(n - 1);
    Ext5 scale = zerofier * denom.inv();
    // first n bit-reversed rows = shift*H_n.  Rows are split over the threads (a 14-column matrix must not leave 242 of 256
    // cores idle); exact field arithmetic makes the order of the partial sums irrelevant.
    const size_t W = lde_bitrev.width;
    std::vector<Ext5> ys(W);
    #pragma omp parallel
    {
        std::vector<Ext5> part(W);
        #pragma omp for schedule(static) nowait
        for (size_t i = 0; i < n; i++) {
            const Fp* row = lde_bitrev.row(reverse_bits_len(i, k));
            for (size_t c = 0; c < W; c++) part[c] += w[i] * row[c];
        }
        #pragma omp critical
        for (size_t c = 0; c < W; c++) ys[c] += part[c];
    }
    for (size_t c = 0; c < W; c++) ys[c] = ys[c] * scale;
    return ys;
}

inline std::vector<Ext5> fold_even_odd(const std::vector<Ext5>& f, const Ext5& beta) {
    size_t half = f.size() / 2;
    unsigned k = log2_strict(half);
    Fp g_inv = two_adic_generator(k + 1).inv();
    Fp one_half = Fp(2).inv();
    std::vector<Ext5> out(half);
    if (fast::enabled()) {  // the powers of 1 / g by running products scattered to their bit-reversed places, the products with reductions hoisted
        std::vector<Fp> xi(half);
        #pragma omp parallel for schedule(static)
        for (size_t c0 = 0; c0 < half; c0 += 4096) {
            Fp cur = g_inv.pow(c0);
            for (size_t i = c0; i < std::min(half, c0 + 4096); i++) { xi[reverse_bits_len(i, k)] = cur; cur *= g_inv; }
        }
        #pragma omp parallel for schedule(static)
        for (size_t i = 0; i < half; i++) {
            const Ext5 power = beta * (one_half * xi[i]);
            out[i] = fast::ext_mul(power + one_half, f[2 * i]) + fast::ext_mul(-power + one_half, f[2 * i + 1]);
        }
        return out;
    }
    #pragma omp parallel for
    for (size_t i = 0; i < half; i++) {
        Fp xinv = g_inv.pow(reverse_bits_len(i, k));
        Ext5 power = beta * (one_half * xinv);
        out[i] = (power + one_half) * f[2 * i] + (-power + one_half) * f[2 * i + 1];
    }
    return out;
}

// ExtensionMmcs::commit_matrix of a (len/2 x 2) Ext5 matrix = base MMCS over its 10-column flattening.
inline Matrix flatten_pairs(const std::vector<Ext5>& f) {
    Matrix m(f.size() / 2, 10);
    #pragma omp parallel for schedule(static) if (fast::enabled())
    for (size_t r = 0; r < m.height; r++)
        for (int e = 0; e < 2; e++)
            for (int c = 0; c < 5; c++) m.at(r, 5 * e + c) = f[2 * r + e].c[c];
    return m;
}

struct RoundData { const MerkleTree* tree; std::vector<std::vector<Ext5>> points; /* per matrix */ };

// ---------------------------------------------------------------- fast mode of the opening (fast.hpp): same values, computed once
// What the scalar code above does element by element — an Ext5 inversion per (matrix, point, LDE row), a power of the generator per row —
// is shared here: 1 / (z - x) over an LDE domain depends on (height, z) only and comes from ONE batch inversion; the trace-domain
// weights of interpolate_coset are a subset of it; a matrix's alpha-reduced rows are computed once for all its points.
namespace fast {
struct OpenTables {
    struct Key { unsigned lh; std::array<uint32_t, 5> z; bool operator<(const Key& o) const { return lh != o.lh ? lh < o.lh : z < o.z; } };
    std::map<Key, std::vector<Ext5>> dinv;       // [x] = 1 / (z - s g_L^bitrev(x)), x in committed (bit-reversed) order
    std::map<unsigned, std::vector<Fp>> gpow;    // [x] = g_L^bitrev(x)
    const std::vector<Fp>& powers(unsigned lh) {
        auto it = gpow.find(lh);
        if (it != gpow.end()) return it->second;
        const size_t L = size_t(1) << lh;
        std::vector<Fp> g(L);
        const Fp w = two_adic_generator(lh);
        #pragma omp parallel for schedule(static)
        for (size_t c0 = 0; c0 < L; c0 += 4096) {
            Fp cur = w.pow(c0);
            for (size_t i = c0; i < std::min(L, c0 + 4096); i++) { g[reverse_bits_len(i, lh)] = cur; cur *= w; }
        }
        return gpow.emplace(lh, std::move(g)).first->second;
    }
    const std::vector<Ext5>& inverses(unsigned lh, const Ext5& z) {
        Key key{lh, {z.c[0].v, z.c[1].v, z.c[2].v, z.c[3].v, z.c[4].v}};
        auto it = dinv.find(key);
        if (it != dinv.end()) return it->second;
        const std::vector<Fp>& g = powers(lh);
        const size_t L = size_t(1) << lh;
        std::vector<Ext5> d(L);
        const Fp s = Fp(GENERATOR);
        #pragma omp parallel for schedule(static)
        for (size_t x = 0; x < L; x++) d[x] = z - s * g[x];
        batch_inverse_ext(d.data(), L);
        return dinv.emplace(key, std::move(d)).first->second;
    }
};
// sum_r w_r * row_r[c] for Ext5 weights and base-field rows, five lazily reduced 64-bit accumulators per column
struct ExtAcc {
    std::vector<uint64_t> a;  // [c][5]
    unsigned pending = 0;
    explicit ExtAcc(size_t width) : a(5 * width, 0) {}
    void add_row(const Ext5& w, const Fp* row, size_t width) {
        const uint64_t w0 = w.c[0].v, w1 = w.c[1].v, w2 = w.c[2].v, w3 = w.c[3].v, w4 = w.c[4].v;
        for (size_t c = 0; c < width; c++) {
            const uint64_t v = row[c].v;
            uint64_t* p = &a[5 * c];
            p[0] += w0 * v; p[1] += w1 * v; p[2] += w2 * v; p[3] += w3 * v; p[4] += w4 * v;
        }
        if (++pending == 3) { for (auto& x : a) x %= P; pending = 0; }  // 2^31 + 3 (p - 1)^2 < 2^64
    }
    Ext5 get(size_t c) const { Ext5 r; for (int k = 0; k < 5; k++) r.c[k] = Fp::from_u64(a[5 * c + k]); return r; }
};
}  // namespace fast

inline std::pair<OpenedValues, PcsProof> pcs_open(const std::vector<RoundData>& rounds, Challenger& ch, const FriConfig& cfg) {
    const double t_open0 = omp_get_wtime();
    Ext5 alpha = ch.sample_ext();
    std::map<unsigned, std::vector<Ext5>> ro;  // log_height -> reduced openings (bit-reversed domain order)
    std::map<unsigned, size_t> num_reduced;
    OpenedValues all;
    fast::OpenTables ftab;
    for (auto& rd : rounds) {
        std::vector<std::vector<std::vector<Ext5>>> round_vals;
        for (size_t mi = 0; mi < rd.tree->leaves.size(); mi++) {
            const Matrix& mat = rd.tree->leaves[mi];
            unsigned lh = log2_strict(mat.height);
            auto& r = ro[lh];
            if (r.empty()) r.assign(mat.height, Ext5());
            // reduced row (independent of the point): sum_j alpha^j row_j(x)
            std::vector<Ext5> apow(mat.width);
            Ext5 ap = Ext5::one();
            for (size_t j = 0; j < mat.width; j++) { apow[j] = ap; ap *= alpha; }
            std::vector<std::vector<Ext5>> mat_vals;
            Fp gh = two_adic_generator(lh);
            if (fast::enabled()) {
                const size_t L = mat.height, n = L >> cfg.log_blowup, W = mat.width;
                const unsigned k = log2_strict(n);
                // the alpha-reduced rows, once for all points of this matrix
                std::vector<Ext5> rr(L);
                #pragma omp parallel for schedule(static)
                for (size_t x = 0; x < L; x++) {
                    uint64_t a[5] = {0, 0, 0, 0, 0};
                    const Fp* row = mat.row(x);
                    for (size_t j = 0; j < W; j++) {
                        const uint64_t v = row[j].v;
                        for (int q = 0; q < 5; q++) a[q] += (uint64_t)apow[j].c[q].v * v;
                        if (j % 3 == 2) for (int q = 0; q < 5; q++) a[q] %= P;
                    }
                    for (int q = 0; q < 5; q++) rr[x].c[q] = Fp::from_u64(a[q]);
                }
                const std::vector<Fp>& gn = ftab.powers(k);  // g_n^bitrev_k(r): the trace domain's points in committed order
                for (auto& z : rd.points[mi]) {
                    const std::vector<Ext5>& dinv = ftab.inverses(lh, z);
                    // interpolate_coset: the first n committed rows are the evaluations on s H_n; 1 / (z - s g_n^i) sits at the same row
                    std::vector<Ext5> ys(W);
                    #pragma omp parallel
                    {
                        fast::ExtAcc part(W);
                        #pragma omp for schedule(static) nowait
                        for (size_t r = 0; r < n; r++) part.add_row(dinv[r] * gn[r], mat.row(r), W);
                        #pragma omp critical
                        for (size_t c = 0; c < W; c++) ys[c] += part.get(c);
                    }
                    const Ext5 zerofier = z.exp_power_of_2(k) - coset_shift().exp_power_of_2(k);
                    const Ext5 scale = zerofier * (Fp((uint32_t)(n % P)) * coset_shift().pow(n - 1)).inv();
                    for (size_t c = 0; c < W; c++) ys[c] = ys[c] * scale;
                    const Ext5 off = alpha.pow(num_reduced[lh]);
                    Ext5 ysum;
                    for (size_t j = 0; j < W; j++) ysum += apow[j] * ys[j];
                    #pragma omp parallel for schedule(static)
                    for (size_t x = 0; x < L; x++) r[x] += fast::ext_mul(fast::ext_mul(off, ysum - rr[x]), dinv[x]);
                    num_reduced[lh] += W;
                    mat_vals.push_back(std::move(ys));
                }
                round_vals.push_back(std::move(mat_vals));
                continue;
            }
            for (auto& z : rd.points[mi]) {
                std::vector<Ext5> ys = interpolate_coset(mat, mat.height >> cfg.log_blowup, coset_shift(), z);
                Ext5 off = alpha.pow(num_reduced[lh]);
                Ext5 ysum;
                for (size_t j = 0; j < mat.width; j++) ysum += apow[j] * ys[j];
                #pragma omp parallel for
                for (size_t x = 0; x < mat.height; x++) {
                    Fp xv = coset_shift() * gh.pow(reverse_bits_len(x, lh));
                    Ext5 rr;
                    for (size_t j = 0; j < mat.width; j++) rr += apow[j] * mat.at(x, j);
                    r[x] += off * (ysum - rr) * (z - xv).inv();
                }
                num_reduced[lh] += mat.width;
                mat_vals.push_back(std::move(ys));
            }
            round_vals.push_back(std::move(mat_vals));
        }
        all.push_back(std::move(round_vals));
    }

    PhaseClock clk;
    clk.t0 = t_open0;
    clk.lap("  opened values + reduced openings");
    // FRI commit phase
    unsigned log_max = ro.rbegin()->first;
    std::vector<Ext5> cur = ro[log_max];
    PcsProof proof;
    std::vector<MerkleTree> layer_trees;
    for (unsigned lf = log_max; lf-- > cfg.log_blowup;) {
        layer_trees.push_back(mmcs_commit({flatten_pairs(cur)}));
        Digest root = layer_trees.back().root();
        ch.observe(root);
        proof.fri.commit_phase_commits.push_back(root);
        Ext5 beta = ch.sample_ext();
        cur = fold_even_odd(cur, beta);
        auto it = ro.find(lf);
        if (it != ro.end()) {
            #pragma omp parallel for schedule(static) if (fast::enabled())
            for (size_t i = 0; i < cur.size(); i++) cur[i] += it->second[i];
        }
    }
    clk.lap("  fri commit phase");
    if (cur.size() != (size_t(1) << cfg.log_blowup)) { fprintf(stderr, "oracle: fri: bad final length\n"); abort(); }
    for (auto& x : cur) if (x != cur[0]) { fprintf(stderr, "oracle: fri: final poly not constant\n"); abort(); }
    proof.fri.final_poly = cur[0];
    if (cfg.observe_final_poly) ch.observe_ext(cur[0]);
    proof.fri.pow_witness = ch.grind(cfg.pow_bits);
    std::vector<size_t> indices;
    for (unsigned q = 0; q < cfg.num_queries; q++) indices.push_back(ch.sample_bits(log_max));
    for (size_t index : indices) {
        QueryProof qp;
        for (size_t i = 0; i < layer_trees.size(); i++) {
            size_t idx_i = index >> i, sib = idx_i ^ 1, pair = idx_i >> 1;
            BatchOpening bo = mmcs_open(layer_trees[i], pair);
            CommitPhaseStep st;
            for (int c = 0; c < 5; c++) st.sibling_value.c[c] = bo.opened_values[0][5 * (sib % 2) + c];
            st.opening_proof = bo.opening_proof;
            qp.commit_phase_openings.push_back(std::move(st));
        }
        proof.fri.query_proofs.push_back(std::move(qp));
        std::vector<BatchOpening> per_round;
        // The input-round trees may be shorter than log_max only if a taller reduced vector exists in another round.
        for (auto& rd : rounds) {
            unsigned lt = log2_ceil(rd.tree->max_height());
            per_round.push_back(mmcs_open(*rd.tree, index >> (log_max - lt)));
        }
        proof.query_openings.push_back(std::move(per_round));
    }
    return {all, proof};
}

struct VerifyRound {
    Digest commit;
    std::vector<size_t> heights;  // trace heights (un-blown-up), per matrix
    std::vector<std::vector<Ext5>> points;
};

inline bool pcs_verify(const std::vector<VerifyRound>& rounds, const OpenedValues& values, const PcsProof& proof,
                       Challenger& ch, const FriConfig& cfg) {
    Ext5 alpha = ch.sample_ext();
    std::vector<Ext5> betas;
    for (auto& c : proof.fri.commit_phase_commits) { ch.observe(c); betas.push_back(ch.sample_ext()); }
    if (cfg.observe_final_poly) ch.observe_ext(proof.fri.final_poly);
    if (proof.fri.query_proofs.size() != cfg.num_queries || proof.query_openings.size() != cfg.num_queries) return false;
    if (!ch.check_witness(cfg.pow_bits, proof.fri.pow_witness)) return false;
    unsigned log_max = proof.fri.commit_phase_commits.size() + cfg.log_blowup;
    for (unsigned q = 0; q < cfg.num_queries; q++) {
        size_t index = ch.sample_bits(log_max);
        std::map<unsigned, Ext5> ro, apow;
        if (proof.query_openings[q].size() != rounds.size()) return false;
        for (size_t ri = 0; ri < rounds.size(); ri++) {
            auto& rd = rounds[ri];
            auto& bo = proof.query_openings[q][ri];
            std::vector<size_t> lde_heights;
            size_t maxh = 0;
            for (size_t h : rd.heights) { lde_heights.push_back(h << cfg.log_blowup); maxh = std::max(maxh, h << cfg.log_blowup); }
            unsigned lt = log2_ceil(maxh);
            if (!mmcs_verify(rd.commit, lde_heights, index >> (log_max - lt), bo)) return false;
            for (size_t mi = 0; mi < rd.heights.size(); mi++) {
                unsigned lh = log2_strict(lde_heights[mi]);
                size_t rev = reverse_bits_len(index >> (log_max - lh), lh);
                Fp x = coset_shift() * two_adic_generator(lh).pow(rev);
                if (!apow.count(lh)) { apow[lh] = Ext5::one(); ro[lh] = Ext5(); }
                auto& row = bo.opened_values[mi];
                for (size_t pi = 0; pi < rd.points[mi].size(); pi++) {
                    const Ext5& z = rd.points[mi][pi];
                    auto& ys = values[ri][mi][pi];
                    if (ys.size() != row.size()) return false;
                    Ext5 dinv = (-z + x).inv();
                    for (size_t j = 0; j < row.size(); j++) {
                        Ext5 quotient = (-ys[j] + row[j]) * dinv;
                        ro[lh] += apow[lh] * quotient;
                        apow[lh] *= alpha;
                    }
                }
            }
        }
        // verify_query
        auto& qp = proof.fri.query_proofs[q];
        if (qp.commit_phase_openings.size() != betas.size()) return false;
        Ext5 folded;
        Fp x = two_adic_generator(log_max).pow(reverse_bits_len(index, log_max));
        size_t idx = index;
        for (size_t i = 0; i < betas.size(); i++) {
            unsigned lf = log_max - 1 - i;
            if (ro.count(lf + 1)) folded += ro[lf + 1];
            size_t sib = idx ^ 1, pair = idx >> 1;
            Ext5 evals[2] = {folded, folded};
            evals[sib % 2] = qp.commit_phase_openings[i].sibling_value;
            BatchOpening bo;
            bo.opened_values.emplace_back();
            for (int e = 0; e < 2; e++) for (int c = 0; c < 5; c++) bo.opened_values[0].push_back(evals[e].c[c]);
            bo.opening_proof = qp.commit_phase_openings[i].opening_proof;
            if (!mmcs_verify(proof.fri.commit_phase_commits[i], {size_t(1) << lf}, pair, bo)) return false;
            Fp xs[2] = {x, x};
            xs[sib % 2] *= two_adic_generator(1);
            folded = evals[0] + (betas[i] - xs[0]) * (evals[1] - evals[0]) * (xs[1] - xs[0]).inv();
            idx = pair;
            x = x * x;
        }
        // Hardening: the prover adds the reduced opening of the SHORTEST LDE height (2^log_blowup:
        // the height-1 chips) after the last fold; it is identically zero for honest openings (constant columns), so a
        // verifier that stops at the loop above accepts proofs whose height-1 opened values were altered.  Adding the term
        // keeps every honest proof valid and binds those values too.
        if (ro.count(cfg.log_blowup)) folded += ro[cfg.log_blowup];
        if (folded != proof.fri.final_poly) return false;
    }
    return true;
}

}  // namespace oracle
