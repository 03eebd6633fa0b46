// CBOR image of MachineProof: what `ciborium::into_writer(&proof, ..)` produces for the serde-derived
// proof types (basic/src/bin/valida.rs:425-427), generated from the flat "VPF1" proof words.
//
// In-tree, authoritative field names and order (machine/src/proof.rs:13-44):
//   MachineProof { commitments, opening_proof, chip_proofs }      Commitments { main_trace, perm_trace, quotient_chunks }
//   ChipProof { log_degree, opened_values, cumulative_sum }
//   OpenedValues { preprocessed_local, preprocessed_next, trace_local, trace_next, permutation_local, permutation_next,
//                  quotient_chunks }      (preprocessed openings always empty, basic/src/lib.rs:641)
// Out-of-tree (valida-xyz/Plonky3 @ bdd338d6, absent here — names as recalled in SURVEY.md Appendix B12, UNPINNED):
//   TwoAdicFriPcsProof { fri_proof, query_openings }   FriProof { commit_phase_commits, query_proofs, final_poly, pow_witness }
//   QueryProof { commit_phase_openings }   CommitPhaseProofStep { sibling_value, opening_proof }
//   BatchOpening { opened_values, opening_proof }
// serde / ciborium conventions used: struct -> definite-length map keyed by the field names in declaration order;
// Vec<T> and [T; N] -> definite-length array; usize / u32 -> shortest-form unsigned integer; PhantomData -> null.
// Switches for what the absent crates decide:
//   CBOR_CANONICAL_FIELDS   BabyBear as its canonical u32 (default: the derive on `struct BabyBear { value: u32 }`, i.e. a
//                           one-entry map holding the raw Montgomery word; BinomialExtensionField likewise {"value": [..5]})
//   CBOR_PLAIN_DIGESTS      commitments / path nodes as bare [Val; 8] (default: Hash { value: [Val; 8], _marker: PhantomData })
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "../field.hpp"

namespace vhost {

constexpr uint32_t CBOR_CANONICAL_FIELDS = 1, CBOR_PLAIN_DIGESTS = 2;

struct CborWriter {
    std::vector<uint8_t> out;
    void head(unsigned major, uint64_t v) {
        const uint8_t m = (uint8_t)(major << 5);
        if (v < 24) out.push_back(m | (uint8_t)v);
        else if (v <= 0xff) { out.push_back(m | 24); out.push_back((uint8_t)v); }
        else if (v <= 0xffff) { out.push_back(m | 25); out.push_back((uint8_t)(v >> 8)); out.push_back((uint8_t)v); }
        else if (v <= 0xffffffffull) { out.push_back(m | 26); for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(v >> s)); }
        else { out.push_back(m | 27); for (int s = 56; s >= 0; s -= 8) out.push_back((uint8_t)(v >> s)); }
    }
    void uint(uint64_t v) { head(0, v); }
    void text(const char* s) { size_t n = std::char_traits<char>::length(s); head(3, n); out.insert(out.end(), s, s + n); }
    void array(uint64_t n) { head(4, n); }
    void map(uint64_t n) { head(5, n); }
    void null() { out.push_back(0xf6); }
};

class ProofCborEncoder {
  public:
    ProofCborEncoder(const uint32_t* words, size_t n, uint32_t flags) : w_(words), n_(n), flags_(flags) {}
    std::vector<uint8_t> encode() {
        if (take() != 0x31465056u) throw std::invalid_argument("cbor: not a VPF1 proof");
        const uint32_t nc = take();
        const size_t roots = pos_;
        skip(24);
        const size_t chips = pos_;
        for (uint32_t c = 0; c < nc; c++) { skip(1); for (int v = 0; v < 5; v++) skip(5 * (size_t)take()); skip(5); }
        const size_t pcs = pos_;
        c_.map(3);
        c_.text("commitments");
        c_.map(3);
        const char* names[3] = {"main_trace", "perm_trace", "quotient_chunks"};
        for (int r = 0; r < 3; r++) { c_.text(names[r]); digest(w_ + roots + 8 * r); }
        c_.text("opening_proof");
        pos_ = pcs;
        pcs_proof();
        if (pos_ != n_) throw std::invalid_argument("cbor: trailing words in the proof");
        c_.text("chip_proofs");
        pos_ = chips;
        c_.array(nc);
        for (uint32_t c = 0; c < nc; c++) chip_proof();
        return std::move(c_.out);
    }

  private:
    const uint32_t* w_;
    size_t n_, pos_ = 0;
    uint32_t flags_;
    CborWriter c_;
    uint32_t take() { if (pos_ >= n_) throw std::invalid_argument("cbor: truncated proof"); return w_[pos_++]; }
    void skip(size_t k) { if (pos_ + k > n_) throw std::invalid_argument("cbor: truncated proof"); pos_ += k; }
    void val(uint32_t canonical) {
        if (flags_ & CBOR_CANONICAL_FIELDS) { c_.uint(canonical); return; }
        c_.map(1); c_.text("value"); c_.uint(vg::Fp::from_canonical(canonical).v);
    }
    void ext(const uint32_t* e) {
        if (!(flags_ & CBOR_CANONICAL_FIELDS)) { c_.map(1); c_.text("value"); }
        c_.array(5);
        for (int k = 0; k < 5; k++) val(e[k]);
    }
    void ext_take() { const size_t p = pos_; skip(5); ext(w_ + p); }
    void digest(const uint32_t* d) {
        if (!(flags_ & CBOR_PLAIN_DIGESTS)) { c_.map(2); c_.text("value"); }
        c_.array(8);
        for (int k = 0; k < 8; k++) val(d[k]);
        if (!(flags_ & CBOR_PLAIN_DIGESTS)) { c_.text("_marker"); c_.null(); }
    }
    void path() {  // Vec<[Val; 8]>: bare arrays in every variant (the Mmcs proof type)
        const uint32_t len = take();
        c_.array(len);
        for (uint32_t i = 0; i < len; i++) { const size_t p = pos_; skip(8); c_.array(8); for (int k = 0; k < 8; k++) val(w_[p + k]); }
    }
    void ext_vec() { const uint32_t len = take(); c_.array(len); for (uint32_t i = 0; i < len; i++) ext_take(); }
    void chip_proof() {
        c_.map(3);
        c_.text("log_degree"); c_.uint(take());
        c_.text("opened_values");
        c_.map(7);
        c_.text("preprocessed_local"); c_.array(0);
        c_.text("preprocessed_next"); c_.array(0);
        const char* names[5] = {"trace_local", "trace_next", "permutation_local", "permutation_next", "quotient_chunks"};
        for (int v = 0; v < 5; v++) { c_.text(names[v]); ext_vec(); }
        c_.text("cumulative_sum"); ext_take();
    }
    void pcs_proof() {
        c_.map(2);
        c_.text("fri_proof");
        c_.map(4);
        c_.text("commit_phase_commits");
        const uint32_t n_commits = take();
        c_.array(n_commits);
        for (uint32_t i = 0; i < n_commits; i++) { const size_t p = pos_; skip(8); digest(w_ + p); }
        c_.text("query_proofs");
        const uint32_t nq = take();
        c_.array(nq);
        for (uint32_t q = 0; q < nq; q++) {
            c_.map(1);
            c_.text("commit_phase_openings");
            const uint32_t nl = take();
            c_.array(nl);
            for (uint32_t l = 0; l < nl; l++) {
                c_.map(2);
                c_.text("sibling_value"); ext_take();
                c_.text("opening_proof"); path();
            }
        }
        c_.text("final_poly"); ext_take();
        c_.text("pow_witness"); val(take());
        c_.text("query_openings");
        const uint32_t nq2 = take();
        c_.array(nq2);
        for (uint32_t q = 0; q < nq2; q++) {
            const uint32_t nr = take();
            c_.array(nr);
            for (uint32_t r = 0; r < nr; r++) {
                c_.map(2);
                c_.text("opened_values");
                const uint32_t nm = take();
                c_.array(nm);
                for (uint32_t m = 0; m < nm; m++) { const uint32_t wd = take(); c_.array(wd); for (uint32_t k = 0; k < wd; k++) val(take()); }
                c_.text("opening_proof"); path();
            }
        }
    }
};

inline std::vector<uint8_t> proof_to_cbor(const uint32_t* words, size_t n, uint32_t flags) { return ProofCborEncoder(words, n, flags).encode(); }

}  // namespace vhost
