// Host side of the PCS: pcs.verify_multi_batches (basic/src/lib.rs:825-837 -> Plonky3 TwoAdicFriPcs::verify_multi_batches;
// conventions SURVEY.md App. B5, B6, B9, B10, B12).  Verification is O(queries x log n) hashing and field arithmetic and runs on
// the host in the reference too; it needs no device.  It completes the UnivariatePcsWithLde surface of INTEGRATION.md (commit /
// get_ldes / open on the device, verify here) and lets a host check what the device produced without any other tool.
//   * MMCS: FieldMerkleTreeMmcs::verify_batch over SerializingHasher32<Keccak256> + CompressionFunctionFromHasher, or the
//     Poseidon-16 sponge / truncated permutation of the north-star variant (cfg.hash_kind)
//   * opening reduction: ro[log_height] += alpha^k (row_j - y_j) / (x - z) over matrices / points / columns in order
//   * FRI: per query fold the reduced openings through the commit-phase layers against the sibling values and their Merkle
//     paths, compare with final_poly; proof-of-work witness checked on the transcript before the indices are sampled
#pragma once
#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <stdexcept>
#include "challenger.hpp"

namespace vhost {

using Digest8 = std::array<uint32_t, 8>;  // canonical words

// Keccak-f[1600] (FIPS 202 Keccak-p[1600, 24]) on 64-bit lanes — the host twin of kernels/keccak.hpp
inline void host_keccak_f1600(uint64_t (&a)[25]) {
    static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
                                    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
                                    0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
                                    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
                                    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    auto rotl = [](uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; };
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) {
            const uint64_t d = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
        }
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];
    }
}

struct HostMmcs {
    int hash_kind;             // vgpu_config.hash_kind
    const Poseidon16* perm;    // for hash_kind 1

    // H(elements): canonical words in, digest as 8 canonical words out
    Digest8 hash(const std::vector<uint32_t>& e) const {
        Digest8 d;
        if (hash_kind == 1) {  // PaddingFreeSponge<Perm16, 16, 8, 8>
            Fp st[16];
            for (auto& s : st) s = Fp::zero();
            for (size_t base = 0; base < e.size(); base += 8) {
                for (size_t k = 0; k < 8 && base + k < e.size(); k++) st[k] = Fp::from_canonical(e[base + k]);
                perm->permute(st);
            }
            for (int i = 0; i < 8; i++) d[i] = st[i].canonical();
            return d;
        }
        // SerializingHasher32<Keccak256Hash>: 4 LE bytes per canonical element, rate 136 bytes = 34 words, padding 0x01 .. 0x80
        uint64_t a[25] = {0};
        auto absorb_word = [&](size_t k, uint32_t w) { a[k >> 1] ^= (uint64_t)w << (32 * (k & 1)); };
        size_t pos = 0;
        for (uint32_t w : e) {
            absorb_word(pos++, w);
            if (pos == 34) { host_keccak_f1600(a); pos = 0; }
        }
        absorb_word(pos, 0x01u);
        absorb_word(33, 0x80000000u);
        host_keccak_f1600(a);
        for (int i = 0; i < 8; i++) {
            uint32_t w = (uint32_t)(a[i >> 1] >> (32 * (i & 1)));  // from_wrapped_u32
            d[i] = w % vg::P;
        }
        return d;
    }
    Digest8 compress(const Digest8& l, const Digest8& r) const {
        if (hash_kind == 1) {  // TruncatedPermutation<Perm16, 2, 8, 16>
            Fp st[16];
            for (int i = 0; i < 8; i++) { st[i] = Fp::from_canonical(l[i]); st[8 + i] = Fp::from_canonical(r[i]); }
            perm->permute(st);
            Digest8 d;
            for (int i = 0; i < 8; i++) d[i] = st[i].canonical();
            return d;
        }
        std::vector<uint32_t> both(l.begin(), l.end());
        both.insert(both.end(), r.begin(), r.end());
        return hash(both);
    }
    // FieldMerkleTreeMmcs::verify_batch: heights of the committed matrices (commit order), the opened rows, the sibling path
    bool verify_batch(const Digest8& commit, const std::vector<uint64_t>& heights, uint64_t index, const std::vector<std::vector<uint32_t>>& rows,
                      const std::vector<Digest8>& path) const {
        if (heights.empty() || heights.size() != rows.size()) return false;
        std::vector<size_t> order(heights.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return heights[a] > heights[b]; });
        uint64_t cur = heights[order[0]];
        if (cur == 0 || (cur & (cur - 1)) || path.size() != vg::log2_strict_u64(cur)) return false;
        size_t pos = 0;
        auto take = [&](uint64_t h) {
            std::vector<uint32_t> buf;
            while (pos < order.size() && heights[order[pos]] == h) { auto& v = rows[order[pos++]]; buf.insert(buf.end(), v.begin(), v.end()); }
            return buf;
        };
        Digest8 node = hash(take(cur));
        for (auto& sib : path) {
            node = (index & 1) ? compress(sib, node) : compress(node, sib);
            index >>= 1;
            cur >>= 1;
            if (pos < order.size() && heights[order[pos]] == cur) node = compress(node, hash(take(cur)));
        }
        return pos == order.size() && node == commit;
    }
};

struct VerifyRoundIn {
    Digest8 commit;
    std::vector<uint64_t> heights;                  // trace heights of the committed matrices (before the blowup)
    std::vector<uint32_t> widths;
    std::vector<std::vector<Ext5>> points;          // per matrix
    std::vector<std::vector<std::vector<Ext5>>> values;  // [matrix][point][column]
};

struct WordCursor {
    const uint32_t* p; size_t n, pos = 0;
    uint32_t u() { if (pos >= n) throw std::invalid_argument("verify: proof words end early"); return p[pos++]; }
    uint32_t len(size_t unit) { uint32_t l = u(); if ((uint64_t)l * unit > n - pos) throw std::invalid_argument("verify: length field exceeds the proof"); return l; }
    Fp f() { uint32_t x = u(); if (x >= vg::P) throw std::invalid_argument("verify: non-canonical field element"); return Fp::from_canonical(x); }
    Ext5 e() { Ext5 r; for (auto& c : r.c) c = f(); return r; }
    Digest8 d() { Digest8 r; for (auto& c : r) { c = u(); if (c >= vg::P) throw std::invalid_argument("verify: non-canonical digest word"); } return r; }
    std::vector<Digest8> dv() { uint32_t l = len(8); std::vector<Digest8> v(l); for (auto& x : v) x = d(); return v; }
};

// Throws std::invalid_argument with the reason when the proof is rejected.
inline void verify_multi_batches(const std::vector<VerifyRoundIn>& rounds, const uint32_t* proof_words, size_t n_words, Challenger& ch, unsigned log_blowup,
                                 unsigned num_queries, unsigned pow_bits, bool observe_final_poly, const HostMmcs& mmcs) {
    auto reject = [](const char* why) { throw std::invalid_argument(std::string("verify: ") + why); };
    // ---- parse TwoAdicFriPcsProof (App. B12; the tail of the "VPF1" layout)
    WordCursor r{proof_words, n_words};
    std::vector<Digest8> commits = r.dv();
    struct Step { Ext5 sibling; std::vector<Digest8> path; };
    std::vector<std::vector<Step>> qsteps(r.len(1));
    for (auto& q : qsteps) { q.resize(r.len(1)); for (auto& s : q) { s.sibling = r.e(); s.path = r.dv(); } }
    const Ext5 final_poly = r.e();
    const Fp pow_witness = r.f();
    struct Batch { std::vector<std::vector<uint32_t>> rows; std::vector<Digest8> path; };
    std::vector<std::vector<Batch>> qopen(r.len(1));
    for (auto& q : qopen) {
        q.resize(r.len(1));
        for (auto& b : q) {
            b.rows.resize(r.len(1));
            for (auto& row : b.rows) { row.resize(r.len(1)); for (auto& x : row) { x = r.u(); if (x >= vg::P) reject("non-canonical opened value"); } }
            b.path = r.dv();
        }
    }
    if (r.pos != n_words) reject("trailing words after the proof");
    if (qsteps.size() != num_queries || qopen.size() != num_queries) reject("wrong number of queries");

    // ---- transcript: batch challenge, one beta per commit-phase root, proof of work, query indices
    const Ext5 alpha = ch.sample_ext();
    std::vector<Ext5> betas;
    for (auto& cmt : commits) { ch.observe_digest(cmt.data()); betas.push_back(ch.sample_ext()); }
    if (observe_final_poly) ch.observe_ext(final_poly);
    if (!ch.check_witness(pow_bits, pow_witness)) reject("proof-of-work witness does not satisfy the transcript");
    const unsigned log_max = (unsigned)commits.size() + log_blowup;
    if (log_max > 27) reject("too many commit-phase layers");
    const Fp s = Fp::from_canonical(vg::GENERATOR);
    for (auto& rd : rounds)
        for (uint64_t h : rd.heights) {
            if (!h || (h & (h - 1))) reject("matrix heights must be powers of two");
            if (vg::log2_strict_u64(h) + log_blowup > log_max) reject("a matrix is taller than the first FRI layer");
        }

    for (unsigned q = 0; q < num_queries; q++) {
        const uint64_t index = ch.sample_bits(log_max);
        std::map<unsigned, Ext5> ro, apow;  // per log LDE height: reduced opening at this query's point, running alpha power
        if (qopen[q].size() != rounds.size()) reject("wrong number of rounds in a query opening");
        for (size_t ri = 0; ri < rounds.size(); ri++) {
            const VerifyRoundIn& rd = rounds[ri];
            const Batch& b = qopen[q][ri];
            std::vector<uint64_t> lde_h;
            uint64_t maxh = 0;
            for (uint64_t h : rd.heights) { lde_h.push_back(h << log_blowup); maxh = std::max(maxh, h << log_blowup); }
            const unsigned lt = vg::log2_strict_u64(maxh);
            if (b.rows.size() != rd.heights.size()) reject("wrong number of opened rows");
            for (size_t mi = 0; mi < b.rows.size(); mi++) if (b.rows[mi].size() != rd.widths[mi]) reject("opened row has the wrong width");
            if (!mmcs.verify_batch(rd.commit, lde_h, index >> (log_max - lt), b.rows, b.path)) reject("an input-round Merkle opening does not match its commitment");
            for (size_t mi = 0; mi < rd.heights.size(); mi++) {
                const unsigned lh = vg::log2_strict_u64(lde_h[mi]);
                const uint64_t rev = vg::reverse_bits_len((uint32_t)(index >> (log_max - lh)), lh);
                const Fp x = s * vg::two_adic_generator(lh).pow(rev);
                if (!apow.count(lh)) { apow[lh] = Ext5::one(); ro[lh] = Ext5::zero(); }
                for (size_t pi = 0; pi < rd.points[mi].size(); pi++) {
                    const Ext5& z = rd.points[mi][pi];
                    const auto& ys = rd.values[mi][pi];
                    if (ys.size() != b.rows[mi].size()) reject("wrong number of opened values");
                    const Ext5 dinv = (Ext5::from_base(x) - z).inv();
                    for (size_t j = 0; j < ys.size(); j++) {
                        ro[lh] += apow[lh] * ((Ext5::from_base(Fp::from_canonical(b.rows[mi][j])) - ys[j]) * dinv);
                        apow[lh] *= alpha;
                    }
                }
            }
        }
        // ---- FRI verify_query
        if (qsteps[q].size() != betas.size()) reject("wrong number of commit-phase openings");
        Ext5 folded = Ext5::zero();
        Fp x = vg::two_adic_generator(log_max).pow(vg::reverse_bits_len((uint32_t)index, log_max));
        uint64_t idx = index;
        const Fp minus_one = vg::two_adic_generator(1);
        for (size_t i = 0; i < betas.size(); i++) {
            const unsigned lf = log_max - 1 - (unsigned)i;
            if (ro.count(lf + 1)) folded += ro[lf + 1];
            const uint64_t sib = idx ^ 1, pair = idx >> 1;
            Ext5 evals[2] = {folded, folded};
            evals[sib & 1] = qsteps[q][i].sibling;
            std::vector<uint32_t> row;
            for (int e = 0; e < 2; e++) for (int c = 0; c < 5; c++) row.push_back(evals[e].c[c].canonical());
            if (!mmcs.verify_batch(commits[i], {1ull << lf}, pair, {row}, qsteps[q][i].path)) reject("a commit-phase Merkle opening does not match its commitment");
            Fp xs[2] = {x, x};
            xs[sib & 1] *= minus_one;
            // interpolate the pair at beta: e0 + (beta - x0) (e1 - e0) / (x1 - x0)
            folded = evals[0] + (betas[i] - xs[0]) * ((evals[1] - evals[0]) * (xs[1] - xs[0]).inv());
            idx = pair;
            x = x * x;
        }
        // the reduced opening of the SHORTEST LDE height (2^log_blowup: height-1 matrices) enters after the last fold: it binds their
        // opened values (identically zero for honest openings)
        if (ro.count(log_blowup)) folded += ro[log_blowup];
        if (folded != final_poly) reject("a query's folded value differs from the final polynomial");
    }
}

}  // namespace vhost
