// Machine::verify on the host (basic/src/lib.rs:677-1064; verify_constraints, machine/src/verify.rs:11-107; eval_permutation_constraints,
// machine/src/chip.rs:210-289): the consumer of what Machine::prove produces, so that a host of this library can check a proof without
// any other tool.  Host-only like the reference's verifier (and like verifier.hpp, whose pcs.verify_multi_batches it calls): no device.
//   * transcript: observe the preprocessed commitment (recomputed from the preprocessed traces as the reference does, lib.rs:791-804 —
//     host_commit_root below — or handed in), the three commitments of the proof, sample the permutation challenges, alpha, zeta;
//   * pcs.verify_multi_batches over the three rounds at (zeta, zeta g) / zeta^(2^lqd);
//   * per chip verify_constraints: the chip's AIR (its compiled register program, air/symbolic.hpp — the same program the device
//     interpreter runs, here over Ext5 openings) and the permutation constraints, folded with alpha (VerifierConstraintFolder: Horner),
//     against Z_H(zeta) * quotient(zeta) recomposed from the opened chunks;
//   * the chips' cumulative sums cancel (lib.rs:1052-1061).
// Preprocessed openings are not part of a proof (lib.rs:612-613, :641): an AIR or interaction that reads a preprocessed column cannot be
// verified out of domain — the reference's verifier would index an empty slice; here the proof is rejected with that reason.
#pragma once
#include "machine.hpp"
#include "verifier.hpp"

namespace vhost {

struct HostMatrixView { const uint32_t* data; uint64_t height, width; };  // canonical, row-major

// In-place radix-2 DIT NTT over the host field, natural order in and out (n a power of two): X[f] = sum_i a[i] w^(i f)
inline void host_ntt(std::vector<Fp>& a, bool inverse) {
    const size_t n = a.size();
    if (n <= 1) return;
    const unsigned k = vg::log2_strict_u64(n);
    for (size_t i = 0; i < n; i++) { size_t j = vg::reverse_bits_len((uint32_t)i, k); if (i < j) std::swap(a[i], a[j]); }
    for (unsigned s = 1; s <= k; s++) {
        Fp w = vg::two_adic_generator(s);
        if (inverse) w = w.inv();
        const size_t half = (size_t)1 << (s - 1);
        for (size_t base = 0; base < n; base += 2 * half) {
            Fp t = Fp::one();
            for (size_t j = 0; j < half; j++) {
                const Fp u = a[base + j], v = a[base + j + half] * t;
                a[base + j] = u + v;
                a[base + j + half] = u - v;
                t *= w;
            }
        }
    }
    if (inverse) { const Fp ninv = Fp::from_canonical((uint32_t)(n % vg::P)).inv(); for (auto& x : a) x *= ninv; }
}

// pcs.commit_batches on the host (App. B3-B5): per matrix the LDE on 31 H_{n 2^log_blowup} (or (31 / shift_i) H), rows bit-reversed, one
// mixed-height MMCS over all of them.  Meant for the SMALL matrices a verifier commits itself (the preprocessed traces); O(n log n) per column.
inline Digest8 host_commit_root(const std::vector<HostMatrixView>& mats, const uint32_t* coset_shifts, unsigned log_blowup, const HostMmcs& mmcs) {
    if (mats.empty()) throw std::invalid_argument("commit: no matrices");
    const Fp g = Fp::from_canonical(vg::GENERATOR);
    std::vector<std::vector<std::vector<uint32_t>>> lde(mats.size());  // [matrix][storage row][column], canonical
    std::vector<uint64_t> heights;
    for (size_t mi = 0; mi < mats.size(); mi++) {
        const HostMatrixView& m = mats[mi];
        if (!m.height || (m.height & (m.height - 1))) throw std::invalid_argument("commit: matrix heights must be powers of two");
        const unsigned kn = vg::log2_strict_u64(m.height), kl = kn + log_blowup;
        if (kl > 27) throw std::invalid_argument("commit: LDE height exceeds the field's two-adicity");
        const uint64_t L = m.height << log_blowup;
        const Fp shift = coset_shifts ? g * Fp::from_canonical(coset_shifts[mi]).inv() : g;
        lde[mi].assign(L, std::vector<uint32_t>(m.width));
        for (uint64_t col = 0; col < m.width; col++) {
            std::vector<Fp> a(m.height);
            for (uint64_t r = 0; r < m.height; r++) a[r] = Fp::from_canonical(m.data[r * m.width + col]);
            host_ntt(a, true);
            a.resize(L, Fp::zero());
            Fp p = Fp::one();
            for (uint64_t i = 0; i < m.height; i++) { a[i] *= p; p *= shift; }
            host_ntt(a, false);
            for (uint64_t j = 0; j < L; j++) lde[mi][j][col] = a[vg::reverse_bits_len((uint32_t)j, kl)].canonical();
        }
        heights.push_back(L);
    }
    std::vector<size_t> order(mats.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return heights[a] > heights[b]; });
    size_t pos = 0;
    auto leaves_of = [&](uint64_t h) {  // digests of the concatenated rows of every matrix of height h, in commit order
        std::vector<Digest8> out;
        const size_t first = pos;
        while (pos < order.size() && heights[order[pos]] == h) pos++;
        if (pos == first) return out;
        out.resize(h);
        for (uint64_t r = 0; r < h; r++) {
            std::vector<uint32_t> row;
            for (size_t q = first; q < pos; q++) { auto& v = lde[order[q]][r]; row.insert(row.end(), v.begin(), v.end()); }
            out[r] = mmcs.hash(row);
        }
        return out;
    };
    uint64_t cur = heights[order[0]];
    std::vector<Digest8> layer = leaves_of(cur);
    while (cur > 1) {
        cur >>= 1;
        std::vector<Digest8> next(cur);
        for (uint64_t i = 0; i < cur; i++) next[i] = mmcs.compress(layer[2 * i], layer[2 * i + 1]);
        std::vector<Digest8> inj = leaves_of(cur);
        if (!inj.empty()) for (uint64_t i = 0; i < cur; i++) next[i] = mmcs.compress(next[i], inj[i]);
        layer.swap(next);
    }
    if (pos != order.size()) throw std::invalid_argument("commit: matrix heights must be powers of two");
    return layer[0];
}

namespace detail {

inline Ext5 ext_monomial(int k) { Ext5 e = Ext5::zero(); e.c[k] = Fp::one(); return e; }

// the chip's compiled register program over Ext5 openings; the constraints are folded as they are asserted (acc = acc alpha + c)
inline void fold_air(const vair::Program& prog, const std::vector<Ext5>& main_l, const std::vector<Ext5>& main_n, const Ext5& first, const Ext5& last,
                     const Ext5& trans, const Ext5& alpha, Ext5& acc) {
    std::vector<Ext5> reg(prog.num_regs ? prog.num_regs : 1, Ext5::zero());
    for (const vair::Instr& in : prog.instrs) {
        switch (in.op) {
            case vair::OP_CONST: reg.at(in.dst) = Ext5::from_base(Fp::raw((uint32_t)in.a | ((uint32_t)in.b << 16))); break;
            case vair::OP_LOAD_MAIN: reg.at(in.dst) = (in.flag ? main_n : main_l).at(in.a); break;
            case vair::OP_LOAD_PREP: throw std::invalid_argument("verify: a constraint reads a preprocessed column, which a proof does not open");
            case vair::OP_SEL_FIRST: reg.at(in.dst) = first; break;
            case vair::OP_SEL_LAST: reg.at(in.dst) = last; break;
            case vair::OP_SEL_TRANS: reg.at(in.dst) = trans; break;
            case vair::OP_ADD: reg.at(in.dst) = reg.at(in.a) + reg.at(in.b); break;
            case vair::OP_SUB: reg.at(in.dst) = reg.at(in.a) - reg.at(in.b); break;
            case vair::OP_MUL: reg.at(in.dst) = reg.at(in.a) * reg.at(in.b); break;
            case vair::OP_NEG: reg.at(in.dst) = -reg.at(in.a); break;
            case vair::OP_ASSERT: acc = acc * alpha + reg.at(in.a); break;
            default: break;  // OP_NOP padding
        }
    }
}

inline Ext5 apply_vcol(const vair::VirtualCol& v, const std::vector<Ext5>& main_row) {
    Ext5 acc = Ext5::from_base(Fp::from_canonical(v.constant));
    for (auto& t : v.terms) {
        if (t.preprocessed) throw std::invalid_argument("verify: an interaction reads a preprocessed column, which a proof does not open");
        acc += main_row.at((size_t)t.col) * Fp::from_canonical(t.weight);
    }
    return acc;
}

}  // namespace detail

struct ChipOpenings {
    unsigned log_degree = 0;
    std::vector<Ext5> trace_local, trace_next, permutation_local, permutation_next, quotient_chunks;
    Ext5 cumulative_sum;
};

// verify_constraints (machine/src/verify.rs:11-107); throws std::invalid_argument with the reason on a mismatch
inline void verify_chip_constraints(const AirDesc& air, const ChipOpenings& o, const Ext5& zeta, const Ext5& alpha, const Ext5 rnd[3]) {
    const size_t M = air.interactions.size(), parts_n = (size_t)1 << air.log_quotient_degree;
    if (o.trace_local.size() != air.width || o.trace_next.size() != air.width) throw std::invalid_argument("verify: chip " + air.name + ": wrong number of trace openings");
    if (o.permutation_local.size() != 5 * (M + 1) || o.permutation_next.size() != 5 * (M + 1)) throw std::invalid_argument("verify: chip " + air.name + ": wrong number of permutation openings");
    if (o.quotient_chunks.size() != 5 * parts_n) throw std::invalid_argument("verify: chip " + air.name + ": wrong number of quotient chunk openings");
    const Fp g_inv = vg::two_adic_generator(o.log_degree).inv();
    const Ext5 z_h = zeta.exp_power_of_2(o.log_degree) - Fp::one();
    const Ext5 is_first = z_h * (zeta - Fp::one()).inv(), is_trans = zeta - g_inv, is_last = z_h * is_trans.inv();
    auto unflatten = [](const std::vector<Ext5>& v) {  // base coefficients opened in the extension -> extension elements
        std::vector<Ext5> out;
        for (size_t i = 0; i + 5 <= v.size(); i += 5) {
            Ext5 acc = Ext5::zero();
            for (int k = 0; k < 5; k++) acc += v[i + k] * detail::ext_monomial(k);
            out.push_back(acc);
        }
        return out;
    };
    const std::vector<Ext5> pl = unflatten(o.permutation_local), pn = unflatten(o.permutation_next);
    std::vector<Ext5> parts = unflatten(o.quotient_chunks);
    Ext5 acc = Ext5::zero();
    detail::fold_air(air.program, o.trace_local, o.trace_next, is_first, is_last, is_trans, alpha, acc);
    // eval_permutation_constraints (machine/src/chip.rs:210-289)
    const Ext5 phi_local = pl[M], phi_next = pn[M];
    Ext5 rhs = Ext5::zero(), phi_0 = Ext5::zero();
    for (size_t m = 0; m < M; m++) {
        const vair::Interaction& it = air.interactions[m];
        Ext5 rlc = Ext5::zero(), beta = Ext5::one();
        for (auto& f : it.fields) { rlc += beta * detail::apply_vcol(f, o.trace_local); beta *= rnd[2]; }
        rlc += (it.is_local() ? rnd[0] : rnd[1]).pow((uint64_t)it.bus_index + 1);  // generate_rlc_elements: powers().skip(1)
        acc = acc * alpha + (rlc * pl[m] - Fp::one());                              // assert_one_ext
        const Ext5 mult_local = detail::apply_vcol(it.count, o.trace_local), mult_next = detail::apply_vcol(it.count, o.trace_next);
        if (it.is_send()) { phi_0 += pl[m] * mult_local; rhs += pn[m] * mult_next; } else { phi_0 -= pl[m] * mult_local; rhs -= pn[m] * mult_next; }
    }
    acc = acc * alpha + ((phi_next - phi_local) - rhs) * is_trans;
    acc = acc * alpha + (phi_local - phi_0) * is_first;
    acc = acc * alpha + (phi_local - o.cumulative_sum) * is_last;
    // quotient(zeta) = sum_i zeta^i part_i, parts in bit-reversed chunk order (reverse_slice_index_bits)
    const unsigned lq = air.log_quotient_degree;
    for (size_t i = 0; i < parts.size(); i++) { const size_t j = vg::reverse_bits_len((uint32_t)i, lq); if (i < j) std::swap(parts[i], parts[j]); }
    Ext5 quotient = Ext5::zero(), zp = Ext5::one();
    for (auto& p : parts) { quotient += p * zp; zp *= zeta; }
    if (acc != z_h * quotient) throw std::invalid_argument("verify: chip " + air.name + ": out-of-domain constraint mismatch (folded constraints != Z_H(zeta) * quotient(zeta))");
}

// Machine::verify over the flat "VPF1" proof words (DESIGN.md "Proof wire format").  preprocessed_commit: 8 canonical words, or null for a
// machine without preprocessed traces.  Throws std::invalid_argument with the reason when the proof is rejected.
inline void verify_machine_proof(const MachineDesc& machine, const FriParams& fri, const Poseidon16& perm16, const uint32_t* preprocessed_commit, const uint32_t* words,
                                 size_t n_words) {
    const size_t NC = machine.airs.size();
    WordCursor r{words, n_words};
    if (r.u() != 0x31465056u) throw std::invalid_argument("verify: not a VPF1 proof");
    if (r.u() != NC) throw std::invalid_argument("verify: wrong number of chip proofs");
    const Digest8 main_commit = r.d(), perm_commit = r.d(), quot_commit = r.d();
    std::vector<ChipOpenings> chips(NC);
    auto vec = [&](std::vector<Ext5>& v) { v.resize(r.len(5)); for (auto& e : v) e = r.e(); };
    for (auto& c : chips) {
        c.log_degree = r.u();
        if (c.log_degree + fri.log_blowup > 27) throw std::invalid_argument("verify: bad log_degree");
        vec(c.trace_local); vec(c.trace_next); vec(c.permutation_local); vec(c.permutation_next); vec(c.quotient_chunks);
        c.cumulative_sum = r.e();
    }
    Challenger ch(&perm16);
    // The reference's verifier ALWAYS recomputes and observes the preprocessed commitment (basic/src/lib.rs:791-804).  A machine with
    // preprocessed traces verified without it would run another transcript: honest proofs rejected, and a proof made without the ROM /
    // range table bound into the transcript accepted.  Refuse the call instead.
    if (!preprocessed_commit)
        for (auto& a : machine.airs)
            if (a.prep_width > 0) throw std::invalid_argument("verify: chip " + a.name + " has a preprocessed trace: the preprocessed commitment is required (vgpu_host_commit_root)");
    if (preprocessed_commit) ch.observe_digest(preprocessed_commit);
    ch.observe_digest(main_commit.data());
    Ext5 rnd[3];
    for (auto& x : rnd) x = ch.sample_ext();
    ch.observe_digest(perm_commit.data());
    const Ext5 alpha = ch.sample_ext();
    ch.observe_digest(quot_commit.data());
    const Ext5 zeta = ch.sample_ext();
    std::vector<VerifyRoundIn> rounds(3);
    rounds[0].commit = main_commit; rounds[1].commit = perm_commit; rounds[2].commit = quot_commit;
    for (size_t i = 0; i < NC; i++) {
        const AirDesc& air = machine.airs[i];
        const ChipOpenings& c = chips[i];
        const Fp g = vg::two_adic_generator(c.log_degree);
        const uint64_t h = 1ull << c.log_degree;
        const uint32_t widths[3] = {air.width, (uint32_t)(5 * (air.interactions.size() + 1)), (uint32_t)(5u << air.log_quotient_degree)};
        for (int q = 0; q < 3; q++) { rounds[q].heights.push_back(h); rounds[q].widths.push_back(widths[q]); }
        rounds[0].points.push_back({zeta, zeta * g});
        rounds[1].points.push_back({zeta, zeta * g});
        rounds[2].points.push_back({zeta.exp_power_of_2(air.log_quotient_degree)});
        rounds[0].values.push_back({c.trace_local, c.trace_next});
        rounds[1].values.push_back({c.permutation_local, c.permutation_next});
        rounds[2].values.push_back({c.quotient_chunks});
        // verify_multi_batches compares these against the opened rows' widths: a wrong count must be its rejection, not an exception type of ours
    }
    HostMmcs mmcs{fri.hash_kind, &perm16};
    verify_multi_batches(rounds, words + r.pos, n_words - r.pos, ch, fri.log_blowup, fri.num_queries, fri.pow_bits, fri.observe_final_poly, mmcs);
    for (size_t i = 0; i < NC; i++) verify_chip_constraints(machine.airs[i], chips[i], zeta, alpha, rnd);
    Ext5 sum = Ext5::zero();
    for (auto& c : chips) sum += c.cumulative_sum;
    if (!sum.is_zero()) throw std::invalid_argument("verify: the chips' cumulative sums do not cancel");  // lib.rs:1052-1061
}

}  // namespace vhost
