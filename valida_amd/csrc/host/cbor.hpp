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
#include <cstring>
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

// The way back (`ciborium::from_reader`, basic/src/bin/valida.rs verify path; the reference's tests verify a proof after exactly this round
// trip, basic/tests/test_prover.rs:456-469): the CBOR image -> "VPF1" proof words.  Strict about structure (the maps' keys in declaration
// order, definite lengths, shortest-form integers are not required on input), and it accepts either setting of the two encoding switches
// per value (a field element is a bare integer or {"value": <Montgomery word>}; a digest a bare array or {value, _marker}).
class ProofCborDecoder {
  public:
    // bare_is_montgomery: read a bare integer field element as the raw Montgomery word (a serializer that writes `value` without the struct
    // wrapper) instead of the canonical value.  seen(): which forms occurred — 1 field as {"value": m}, 2 field as a bare integer,
    // 4 digest as {value, _marker}, 8 digest as a bare array — so a first contact with a real proof file can say what it met.
    ProofCborDecoder(const uint8_t* bytes, size_t n, bool bare_is_montgomery = false) : b_(bytes), n_(n), bare_monty_(bare_is_montgomery) {}
    unsigned seen() const { return seen_; }
    std::vector<uint32_t> decode() {
        std::vector<uint32_t> roots, chips, pcs;
        map(3);
        key("commitments");
        map(3);
        const char* names[3] = {"main_trace", "perm_trace", "quotient_chunks"};
        for (int r = 0; r < 3; r++) { key(names[r]); digest(roots); }
        key("opening_proof");
        pcs_proof(pcs);
        key("chip_proofs");
        const uint64_t nc = array();
        if (nc > 4096) bad("implausible number of chip proofs");
        for (uint64_t c = 0; c < nc; c++) chip_proof(chips);
        if (pos_ != n_) bad("trailing bytes after the proof");
        std::vector<uint32_t> w;
        w.reserve(2 + roots.size() + chips.size() + pcs.size());
        w.push_back(0x31465056u);
        w.push_back((uint32_t)nc);
        w.insert(w.end(), roots.begin(), roots.end());
        w.insert(w.end(), chips.begin(), chips.end());
        w.insert(w.end(), pcs.begin(), pcs.end());
        return w;
    }

  private:
    const uint8_t* b_;
    size_t n_, pos_ = 0;
    bool bare_monty_ = false;
    unsigned seen_ = 0;
    [[noreturn]] static void bad(const char* why) { throw std::invalid_argument(std::string("cbor: ") + why); }
    uint8_t byte() { if (pos_ >= n_) bad("truncated input"); return b_[pos_++]; }
    // (major type, argument) of the next item; only the definite-length forms a serde / ciborium writer produces
    std::pair<unsigned, uint64_t> head() {
        const uint8_t ib = byte();
        const unsigned major = ib >> 5, info = ib & 31;
        uint64_t v = info;
        if (info >= 24) {
            if (info > 27) bad("indefinite lengths / reserved forms are not part of a proof");
            const int nbytes = 1 << (info - 24);
            v = 0;
            for (int i = 0; i < nbytes; i++) v = (v << 8) | byte();
        }
        return {major, v};
    }
    uint64_t expect(unsigned major, const char* what) { auto h = head(); if (h.first != major) bad(what); return h.second; }
    void map(uint64_t n) { if (expect(5, "expected a map") != n) bad("a struct has the wrong number of fields"); }
    uint64_t array() { return expect(4, "expected an array"); }
    void array(uint64_t n) { if (array() != n) bad("an array has the wrong length"); }
    void key(const char* name) {
        const uint64_t len = expect(3, "expected a field name");
        const size_t want = std::char_traits<char>::length(name);
        if (len != want || pos_ + len > n_ || memcmp(b_ + pos_, name, want) != 0) bad("unexpected field name (fields come in declaration order)");
        pos_ += len;
    }
    uint32_t word(const char* what) { const uint64_t v = expect(0, what); if (v > 0xffffffffull) bad("integer out of range"); return (uint32_t)v; }
    // compared by division: l * unit may wrap 64 bits (l = 0x1C71C71C71C71C72, unit 9 gives 2) and pass the bound on hostile input
    uint32_t len(uint64_t unit) { const uint64_t l = array(); if (l > 0xffffffffull || l > (uint64_t)(n_ - pos_) / (unit ? unit : 1)) bad("a length exceeds the input"); return (uint32_t)l; }
    uint32_t val() {  // canonical field element
        if (pos_ < n_ && (b_[pos_] >> 5) == 5) {
            map(1); key("value");
            const uint32_t m = word("expected a field element");
            if (m >= vg::P) bad("field element out of range");
            seen_ |= 1u;
            return vg::Fp::raw(m).canonical();
        }
        const uint32_t c = word("expected a field element");
        if (c >= vg::P) bad("field element out of range");
        seen_ |= 2u;
        return bare_monty_ ? vg::Fp::raw(c).canonical() : c;
    }
    void ext(std::vector<uint32_t>& out) {
        if (pos_ < n_ && (b_[pos_] >> 5) == 5) { map(1); key("value"); }
        array(5);
        for (int k = 0; k < 5; k++) out.push_back(val());
    }
    void digest(std::vector<uint32_t>& out) {
        const bool wrapped = pos_ < n_ && (b_[pos_] >> 5) == 5;
        if (wrapped) { map(2); key("value"); }
        seen_ |= wrapped ? 4u : 8u;
        array(8);
        for (int k = 0; k < 8; k++) out.push_back(val());
        if (wrapped) { key("_marker"); if (byte() != 0xf6) bad("expected null for PhantomData"); }
    }
    void path(std::vector<uint32_t>& out) {
        const uint32_t l = len(9);
        out.push_back(l);
        for (uint32_t i = 0; i < l; i++) { array(8); for (int k = 0; k < 8; k++) out.push_back(val()); }
    }
    void ext_vec(std::vector<uint32_t>& out) { const uint32_t l = len(6); out.push_back(l); for (uint32_t i = 0; i < l; i++) ext(out); }
    void chip_proof(std::vector<uint32_t>& out) {
        map(3);
        key("log_degree"); out.push_back(word("expected log_degree"));
        key("opened_values");
        map(7);
        key("preprocessed_local"); array(0);
        key("preprocessed_next"); array(0);
        const char* names[5] = {"trace_local", "trace_next", "permutation_local", "permutation_next", "quotient_chunks"};
        for (int v = 0; v < 5; v++) { key(names[v]); ext_vec(out); }
        key("cumulative_sum"); ext(out);
    }
    void pcs_proof(std::vector<uint32_t>& out) {
        map(2);
        key("fri_proof");
        map(4);
        key("commit_phase_commits");
        const uint32_t n_commits = len(9);
        out.push_back(n_commits);
        for (uint32_t i = 0; i < n_commits; i++) digest(out);
        key("query_proofs");
        const uint32_t nq = len(1);
        out.push_back(nq);
        for (uint32_t q = 0; q < nq; q++) {
            map(1);
            key("commit_phase_openings");
            const uint32_t nl = len(1);
            out.push_back(nl);
            for (uint32_t l = 0; l < nl; l++) { map(2); key("sibling_value"); ext(out); key("opening_proof"); path(out); }
        }
        key("final_poly"); ext(out);
        key("pow_witness"); out.push_back(val());
        key("query_openings");
        const uint32_t nq2 = len(1);
        out.push_back(nq2);
        for (uint32_t q = 0; q < nq2; q++) {
            const uint32_t nr = len(1);
            out.push_back(nr);
            for (uint32_t r = 0; r < nr; r++) {
                map(2);
                key("opened_values");
                const uint32_t nm = len(1);
                out.push_back(nm);
                for (uint32_t m = 0; m < nm; m++) { const uint32_t wd = len(1); out.push_back(wd); for (uint32_t k = 0; k < wd; k++) out.push_back(val()); }
                key("opening_proof"); path(out);
            }
        }
    }
};

inline std::vector<uint32_t> proof_from_cbor(const uint8_t* bytes, size_t n) { return ProofCborDecoder(bytes, n).decode(); }

}  // namespace vhost
