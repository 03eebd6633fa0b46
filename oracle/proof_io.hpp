// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).
//
// Flat wire format "VPF1" for MachineProof (machine/src/proof.rs:13-44 + TwoAdicFriPcsProof, App. B12):
// a sequence of little-endian u32 words, field elements canonical (< p), Ext5 = 5 words, digest = 8
// words.  This is NOT the reference's CBOR encoding (ciborium; SURVEY.md §8(f) item 2, next tier) — it
// is the byte string on which "GPU proof bytes == oracle proof bytes" is asserted.  Layout:
//   magic 0x31465056, num_chips,
//   main_commit[8], perm_commit[8], quotient_commit[8],
//   per chip: log_degree, then five length-prefixed Ext5 vectors (trace_local, trace_next,
//             permutation_local, permutation_next, quotient_chunks), cumulative_sum[5]
//   fri: n_commits, commits; n_queries, per query: n_steps, per step: sibling[5], n_path, path;
//        final_poly[5], pow_witness
//   query_openings: n_queries, per query: n_rounds, per round: n_mats, per mat: width, values;
//        n_path, path
#pragma once
#include "stark.hpp"

namespace oracle {

constexpr uint32_t PROOF_MAGIC = 0x31465056u;

struct WordWriter {
    std::vector<uint32_t> w;
    void u(uint32_t x) { w.push_back(x); }
    void f(const Fp& x) { w.push_back(x.v); }
    void e(const Ext5& x) { for (auto& c : x.c) f(c); }
    void d(const Digest& x) { for (auto& c : x) f(c); }
    void ev(const std::vector<Ext5>& v) { u((uint32_t)v.size()); for (auto& x : v) e(x); }
    void dv(const std::vector<Digest>& v) { u((uint32_t)v.size()); for (auto& x : v) d(x); }
};

// TwoAdicFriPcsProof: the tail of the layout above, also what pcs.open_multi_batches returns on its own
inline void serialize_pcs_proof(WordWriter& o, const PcsProof& pp) {
    auto& fri = pp.fri;
    o.dv(fri.commit_phase_commits);
    o.u((uint32_t)fri.query_proofs.size());
    for (auto& q : fri.query_proofs) {
        o.u((uint32_t)q.commit_phase_openings.size());
        for (auto& s : q.commit_phase_openings) { o.e(s.sibling_value); o.dv(s.opening_proof); }
    }
    o.e(fri.final_poly);
    o.f(fri.pow_witness);
    o.u((uint32_t)pp.query_openings.size());
    for (auto& q : pp.query_openings) {
        o.u((uint32_t)q.size());
        for (auto& bo : q) {
            o.u((uint32_t)bo.opened_values.size());
            for (auto& row : bo.opened_values) { o.u((uint32_t)row.size()); for (auto& x : row) o.f(x); }
            o.dv(bo.opening_proof);
        }
    }
}

inline std::vector<uint32_t> serialize_proof(const MachineProof& p) {
    WordWriter o;
    o.u(PROOF_MAGIC);
    o.u((uint32_t)p.chip_proofs.size());
    o.d(p.main_commit); o.d(p.perm_commit); o.d(p.quotient_commit);
    for (auto& c : p.chip_proofs) {
        o.u(c.log_degree);
        o.ev(c.trace_local); o.ev(c.trace_next); o.ev(c.permutation_local); o.ev(c.permutation_next); o.ev(c.quotient_chunks);
        o.e(c.cumulative_sum);
    }
    serialize_pcs_proof(o, p.opening_proof);
    return o.w;
}

struct WordReader {
    const uint32_t* p; size_t n, pos = 0; bool ok = true;
    uint32_t u() { if (pos >= n) { ok = false; return 0; } return p[pos++]; }
    uint32_t len(size_t unit) { uint32_t l = u(); if ((uint64_t)l * unit > n - std::min(n, pos)) { ok = false; return 0; } return l; }
    Fp f() { uint32_t x = u(); if (x >= P) ok = false; Fp r; r.v = x % P; return r; }
    Ext5 e() { Ext5 r; for (auto& c : r.c) c = f(); return r; }
    Digest d() { Digest r; for (auto& c : r) c = f(); return r; }
    std::vector<Ext5> ev() { uint32_t l = len(5); std::vector<Ext5> v(l); for (auto& x : v) x = e(); return v; }
    std::vector<Digest> dv() { uint32_t l = len(8); std::vector<Digest> v(l); for (auto& x : v) x = d(); return v; }
};

inline bool deserialize_proof(const uint32_t* words, size_t n, MachineProof& p) {
    WordReader r{words, n};
    if (r.u() != PROOF_MAGIC) return false;
    uint32_t nc = r.len(1);
    p.main_commit = r.d(); p.perm_commit = r.d(); p.quotient_commit = r.d();
    p.chip_proofs.resize(nc);
    for (auto& c : p.chip_proofs) {
        c.log_degree = r.u();
        c.trace_local = r.ev(); c.trace_next = r.ev(); c.permutation_local = r.ev(); c.permutation_next = r.ev(); c.quotient_chunks = r.ev();
        c.cumulative_sum = r.e();
        if (!r.ok) return false;
    }
    auto& fri = p.opening_proof.fri;
    fri.commit_phase_commits = r.dv();
    fri.query_proofs.resize(r.len(1));
    for (auto& q : fri.query_proofs) {
        q.commit_phase_openings.resize(r.len(1));
        for (auto& s : q.commit_phase_openings) { s.sibling_value = r.e(); s.opening_proof = r.dv(); }
        if (!r.ok) return false;
    }
    fri.final_poly = r.e();
    fri.pow_witness = r.f();
    p.opening_proof.query_openings.resize(r.len(1));
    for (auto& q : p.opening_proof.query_openings) {
        q.resize(r.len(1));
        for (auto& bo : q) {
            bo.opened_values.resize(r.len(1));
            for (auto& row : bo.opened_values) { row.resize(r.len(1)); for (auto& x : row) x = r.f(); }
            bo.opening_proof = r.dv();
            if (!r.ok) return false;
        }
    }
    return r.ok && r.pos == n;
}

}  // namespace oracle
