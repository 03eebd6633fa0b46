// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points for tests/ (ctypes), __graft_entry__.smoke() and
// bench.py's cpu_baseline leg.  PARITY UNPINNED vs Plonky3@bdd338d6 (see field.hpp).
#include <malloc.h>
#include <chrono>
#include <cstring>
#include <memory>
#include "proof_io.hpp"

using namespace oracle;

namespace {
Matrix to_matrix(const uint32_t* v, size_t h, size_t w) {
    Matrix m(h, w);
    for (size_t i = 0; i < h * w; i++) m.v[i] = Fp(v[i]);
    return m;
}
std::vector<Ext5> to_ext(const uint32_t* v, size_t n) {
    std::vector<Ext5> out(n);
    for (size_t i = 0; i < n; i++) for (int k = 0; k < 5; k++) out[i].c[k] = Fp(v[5 * i + k]);
    return out;
}
struct ProveResult {
    std::vector<uint32_t> words;
    ProveDebug dbg;
    double seconds = 0;
};
bool g_observe_final_poly = false;  // convention switch, set by oracle_set_observe_final_poly
StarkConfig make_cfg(const uint32_t* rc, uint32_t log_blowup, uint32_t num_queries, uint32_t pow_bits) {
    StarkConfig cfg;
    cfg.fri.observe_final_poly = g_observe_final_poly;
    cfg.poseidon_constants.assign(rc, rc + 480);
    cfg.fri.log_blowup = log_blowup;
    cfg.fri.num_queries = num_queries;
    cfg.fri.pow_bits = pow_bits;
    return cfg;
}
}  // namespace

extern "C" {
void oracle_set_observe_final_poly(int on) { g_observe_final_poly = on != 0; }
// Fast mode (fast.hpp): the same proof words computed with AVX2 Montgomery transforms, an unrolled Keccak, batch inversions.  Process-wide.
void oracle_set_fast(int on) { oracle::fast::enabled() = on != 0; }
// For a process that proves ONCE and exits (oracle/cpu_baseline.py, the bench's CPU leg): a proof allocates and frees tens of GB in blocks of
// hundreds of MB; glibc hands such blocks back to the kernel at once and every new one is page-faulted in again (single-threaded, inside the
// vector constructors).  Keeping freed memory in the heap lets the later phases of the proof reuse pages the earlier ones faulted in: 4.1 s
// against 5.5 s for the headline segment on the MI355X box's 16 cores.  Process-wide and permanent — NOT for long-lived test processes (the
// CPU suite slowed down several-fold under it).
void oracle_keep_heap() {
    mallopt(M_MMAP_MAX, 0);
    mallopt(M_TRIM_THRESHOLD, -1);
}
int oracle_get_fast() { return oracle::fast::enabled() ? 1 : 0; }
// MMCS hash of every commitment made after the call: 0 = Keccak (reference), 1 = Poseidon-16 sponge / truncated permutation with
// the given round constants (hash.hpp "MMCS hash selection")
void oracle_set_mmcs_hash(int kind, const uint32_t* rc480) {
    static std::unique_ptr<Poseidon16> keep;
    if (kind == 1) { keep.reset(new Poseidon16(rc480)); mmcs_hash().poseidon = keep.get(); }
    mmcs_hash().kind = kind == 1 ? 1 : 0;
}

uint32_t oracle_two_adic_generator(uint32_t bits) { return two_adic_generator(bits).v; }
uint32_t oracle_fp_mul(uint32_t a, uint32_t b) { return (Fp(a) * Fp(b)).v; }
uint32_t oracle_fp_inv(uint32_t a) { return Fp(a).inv().v; }
void oracle_ext5_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    Ext5 r = to_ext(a, 1)[0] * to_ext(b, 1)[0];
    for (int k = 0; k < 5; k++) out[k] = r.c[k].v;
}
void oracle_ext5_inv(const uint32_t* a, uint32_t* out) {
    Ext5 r = to_ext(a, 1)[0].inv();
    for (int k = 0; k < 5; k++) out[k] = r.c[k].v;
}
void oracle_keccak256(const uint8_t* data, uint64_t len, uint8_t* out32, uint32_t pad) { keccak256(data, len, out32, (uint8_t)pad); }
void oracle_keccak_f1600(uint64_t* state25) { keccak_f1600(state25); }
void oracle_hash_elems(const uint32_t* e, uint64_t n, uint32_t* out8) {
    std::vector<Fp> v(n);
    for (size_t i = 0; i < n; i++) v[i] = Fp(e[i]);
    Digest d = hash_elems(v);
    for (int i = 0; i < 8; i++) out8[i] = d[i].v;
}
// C(a, b) of the MMCS (CompressionFunctionFromHasher / TruncatedPermutation, by the current hash selection)
void oracle_compress(const uint32_t* a8, const uint32_t* b8, uint32_t* out8) {
    Digest a, b;
    for (int i = 0; i < 8; i++) { a[i] = Fp(a8[i]); b[i] = Fp(b8[i]); }
    Digest d = compress(a, b);
    for (int i = 0; i < 8; i++) out8[i] = d[i].v;
}
void oracle_poseidon_permute(const uint32_t* rc480, uint32_t* state16) {
    Poseidon16 p(rc480);
    Fp st[16];
    for (int i = 0; i < 16; i++) st[i] = Fp(state16[i]);
    p.permute(st);
    for (int i = 0; i < 16; i++) state16[i] = st[i].v;
}
// Transcript probe: observe `n_obs` values, then sample `n_samp` values.
void oracle_challenger_probe(const uint32_t* rc480, const uint32_t* obs, uint64_t n_obs, uint32_t* samp, uint64_t n_samp) {
    Poseidon16 p(rc480);
    Challenger ch(&p);
    for (size_t i = 0; i < n_obs; i++) ch.observe(Fp(obs[i]));
    for (size_t i = 0; i < n_samp; i++) samp[i] = ch.sample().v;
}
uint32_t oracle_grind(const uint32_t* rc480, const uint32_t* obs, uint64_t n_obs, uint32_t bits) {
    Poseidon16 p(rc480);
    Challenger ch(&p);
    for (size_t i = 0; i < n_obs; i++) ch.observe(Fp(obs[i]));
    return ch.grind(bits).v;
}
void oracle_dft(uint32_t* col, uint64_t n, int inverse) {
    std::vector<Fp> a(n);
    for (size_t i = 0; i < n; i++) a[i] = Fp(col[i]);
    dft_inplace(a, inverse != 0);
    for (size_t i = 0; i < n; i++) col[i] = a[i].v;
}
void oracle_naive_dft(const uint32_t* col, uint64_t n, uint32_t* out) {
    std::vector<Fp> a(n);
    for (size_t i = 0; i < n; i++) a[i] = Fp(col[i]);
    auto r = naive_dft(a);
    for (size_t i = 0; i < n; i++) out[i] = r[i].v;
}
// Committed LDE of one matrix: rows bit-reversed, row-major, (h << added_bits) x w.
void oracle_committed_lde(const uint32_t* m, uint64_t h, uint64_t w, uint32_t added_bits, uint32_t lde_shift, uint32_t* out) {
    Matrix r = bit_reverse_rows(coset_lde_batch(to_matrix(m, h, w), added_bits, Fp(lde_shift)));
    for (size_t i = 0; i < r.v.size(); i++) out[i] = r.v[i].v;
}
// pcs.commit_shifted_batches root for a batch of row-major matrices (shifts may be null = all ones).
void oracle_commit_root(uint64_t n_mats, const uint32_t* const* mats, const uint64_t* heights, const uint64_t* widths,
                        const uint32_t* shifts, uint32_t log_blowup, uint32_t* root8) {
    std::vector<Matrix> ms;
    std::vector<Fp> sh;
    for (size_t i = 0; i < n_mats; i++) { ms.push_back(to_matrix(mats[i], heights[i], widths[i])); sh.push_back(shifts ? Fp(shifts[i]) : Fp::one()); }
    FriConfig fc; fc.log_blowup = log_blowup;
    MerkleTree t = pcs_commit(ms, sh, fc);
    for (int i = 0; i < 8; i++) root8[i] = t.root()[i].v;
}
// Plain MMCS root over already-extended matrices (no LDE).
void oracle_mmcs_root(uint64_t n_mats, const uint32_t* const* mats, const uint64_t* heights, const uint64_t* widths, uint32_t* root8) {
    std::vector<Matrix> ms;
    for (size_t i = 0; i < n_mats; i++) ms.push_back(to_matrix(mats[i], heights[i], widths[i]));
    MerkleTree t = mmcs_commit(std::move(ms));
    for (int i = 0; i < 8; i++) root8[i] = t.root()[i].v;
}
// Direct instantiation of chip `chip`'s Air::eval on one row pair: the asserted values, in order.
// Returns the number of constraints (out may be null to query it).
uint32_t oracle_eval_constraints(uint32_t chip, const uint32_t* local, const uint32_t* next, const uint32_t* prep_local, const uint32_t* prep_next,
                                 uint32_t is_first, uint32_t is_last, uint32_t is_transition, uint32_t* out, uint32_t cap) {
    (void)prep_local; (void)prep_next;  // no BasicMachine chip reads its preprocessed columns in eval
    const size_t w = chips::chip_shape((int)chip).width;
    struct Rec {
        using Expr = Fp;
        std::vector<Fp> l, n;
        Fp first, last, trans;
        std::vector<Fp> vals;
        Fp from_u32(uint32_t k) const { return Fp(k); }
        const Fp* main_local() const { return l.data(); }
        const Fp* main_next() const { return n.data(); }
        Fp is_first_row() const { return first; }
        Fp is_last_row() const { return last; }
        Fp is_transition() const { return trans; }
        void assert_zero(const Fp& x) { vals.push_back(x); }
    } b{std::vector<Fp>(w), std::vector<Fp>(w), Fp(is_first), Fp(is_last), Fp(is_transition), {}};
    for (size_t c = 0; c < w; c++) { b.l[c] = Fp(local[c]); b.n[c] = Fp(next[c]); }
    chips::eval((int)chip, b);
    if (out) for (size_t i = 0; i < b.vals.size() && i < cap; i++) out[i] = b.vals[i].v;
    return (uint32_t)b.vals.size();
}
// Neutral word image of chip `chip`'s all_interactions (tests compare it with the product's):
//   [n] then per interaction: [is_send] [is_global] [bus_index] [n_fields] count_vcol field_vcols..;  vcol = [n_terms] [constant] n_terms x ([is_prep] [col] [weight])
uint32_t oracle_interaction_words(uint32_t chip, uint32_t* out, uint32_t cap) {
    std::vector<uint32_t> w;
    auto its = chips::all_interactions((int)chip);
    auto vcol = [&](const chips::VirtualPairCol& v) {
        w.push_back((uint32_t)v.terms.size()); w.push_back(v.constant);
        for (auto& t : v.terms) { w.push_back(t.preprocessed ? 1 : 0); w.push_back((uint32_t)t.col); w.push_back(t.weight); }
    };
    w.push_back((uint32_t)its.size());
    for (auto& it : its) {
        w.push_back(it.is_send() ? 1 : 0); w.push_back(it.global ? 1 : 0); w.push_back((uint32_t)it.bus_index); w.push_back((uint32_t)it.fields.size());
        vcol(it.count);
        for (auto& f : it.fields) vcol(f);
    }
    if (out) for (size_t i = 0; i < w.size() && i < cap; i++) out[i] = w[i];
    return (uint32_t)w.size();
}
// [width, preprocessed width] of chip `chip` from the oracle's column structs
void oracle_chip_shape(uint32_t chip, uint32_t out[2]) { auto s = chips::chip_shape((int)chip); out[0] = (uint32_t)s.width; out[1] = (uint32_t)s.preprocessed_width; }
uint32_t oracle_log_quotient_degree(uint32_t chip) { return log_quotient_degree(MachineDesc::basic().chips[chip]); }
uint32_t oracle_num_interactions(uint32_t chip) { return (uint32_t)MachineDesc::basic().chips[chip].interactions.size(); }
// generate_permutation_trace of BasicMachine chip `chip`; out is height x 5(M+1), row-major (flatten_to_base).
void oracle_perm_trace(uint32_t chip, const uint32_t* main, uint64_t h, const uint32_t* rnd15, uint32_t* out) {
    MachineDesc md = MachineDesc::basic();
    auto& c = md.chips[chip];
    auto t = generate_permutation_trace(c, to_matrix(main, h, c.width), nullptr, to_ext(rnd15, 3));
    for (size_t i = 0; i < t.size(); i++) for (int k = 0; k < 5; k++) out[5 * i + k] = t[i].c[k].v;
}
// fold_even_odd on an Ext5 vector (n x 5 words), out has n/2 x 5 words.
void oracle_fri_fold(const uint32_t* f, uint64_t n, const uint32_t* beta5, uint32_t* out) {
    auto r = fold_even_odd(to_ext(f, n), to_ext(beta5, 1)[0]);
    for (size_t i = 0; i < r.size(); i++) for (int k = 0; k < 5; k++) out[5 * i + k] = r[i].c[k].v;
}

// ---- pcs.commit_batches + pcs.open_multi_batches on arbitrary rounds (generic shapes: any widths, heights, points per matrix)
// mats / heights / widths / n_points: one entry per (round, matrix), rounds concatenated; points: 5 words per point, concatenated.
// obs: values the transcript observed before the call.  The result's words = [8 words root per round][opened values, flattened
// [round][matrix][point][column] x 5][TwoAdicFriPcsProof words]; oracle_pcs_open_split gives the two boundaries.
struct PcsOpenResult { std::vector<uint32_t> words; uint64_t n_root_words = 0, n_value_words = 0; };
void* oracle_pcs_open(uint32_t n_rounds, const uint32_t* n_mats, const uint32_t* const* mats, const uint64_t* heights, const uint64_t* widths,
                      const uint32_t* n_points, const uint32_t* points, const uint32_t* rc480, uint32_t log_blowup, uint32_t num_queries, uint32_t pow_bits,
                      const uint32_t* obs, uint64_t n_obs) {
    StarkConfig cfg = make_cfg(rc480, log_blowup, num_queries, pow_bits);
    Poseidon16 perm16(cfg.poseidon_constants.data());
    Challenger ch(&perm16);
    for (size_t i = 0; i < n_obs; i++) ch.observe(Fp(obs[i]));
    std::vector<MerkleTree> trees(n_rounds);
    std::vector<RoundData> rounds(n_rounds);
    size_t k = 0, w = 0;
    for (uint32_t r = 0; r < n_rounds; r++) {
        std::vector<Matrix> ms;
        std::vector<std::vector<Ext5>> pts;
        for (uint32_t i = 0; i < n_mats[r]; i++, k++) {
            ms.push_back(to_matrix(mats[k], heights[k], widths[k]));
            pts.push_back(to_ext(points + w, n_points[k]));
            w += 5 * (size_t)n_points[k];
        }
        trees[r] = pcs_commit(ms, cfg.fri);
        rounds[r].points = std::move(pts);
    }
    for (uint32_t r = 0; r < n_rounds; r++) rounds[r].tree = &trees[r];
    auto opened = pcs_open(rounds, ch, cfg.fri);
    auto* res = new PcsOpenResult();
    WordWriter o;
    for (auto& t : trees) o.d(t.root());
    res->n_root_words = o.w.size();
    for (auto& round : opened.first) for (auto& mat : round) for (auto& pt : mat) for (auto& e : pt) o.e(e);
    res->n_value_words = o.w.size() - res->n_root_words;
    serialize_pcs_proof(o, opened.second);
    res->words = std::move(o.w);
    return res;
}
uint64_t oracle_pcs_open_len(void* r) { return ((PcsOpenResult*)r)->words.size(); }
const uint32_t* oracle_pcs_open_words(void* r) { return ((PcsOpenResult*)r)->words.data(); }
void oracle_pcs_open_split(void* r, uint64_t out[2]) { out[0] = ((PcsOpenResult*)r)->n_root_words; out[1] = ((PcsOpenResult*)r)->n_value_words; }
void oracle_pcs_open_free(void* r) { delete (PcsOpenResult*)r; }

// ---- full prover / verifier for BasicMachine ---------------------------------------------------
// main[i]: row-major heights[i] x chip width; prep_program: hp x 7; prep_range: 256 x 1.
void* oracle_prove_basic(const uint32_t* const* main, const uint64_t* heights, const uint32_t* prep_program, uint64_t hp,
                         const uint32_t* prep_range, const uint32_t* rc480, uint32_t log_blowup, uint32_t num_queries,
                         uint32_t pow_bits, int debug_check) {
    MachineDesc md = MachineDesc::basic();
    MachineInput in;
    for (int i = 0; i < chips::NUM_CHIPS; i++) in.main_traces.push_back(to_matrix(main[i], heights[i], md.chips[i].width));
    in.preprocessed.push_back({chips::PROGRAM, to_matrix(prep_program, hp, 7)});
    in.preprocessed.push_back({chips::RANGE, to_matrix(prep_range, 256, 1)});
    auto* res = new ProveResult();
    auto t0 = std::chrono::steady_clock::now();
    MachineProof p = prove(md, in, make_cfg(rc480, log_blowup, num_queries, pow_bits), &res->dbg, debug_check != 0);
    res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    res->words = serialize_proof(p);
    return res;
}
// Any machine made of chips::ChipIndex / test-AIR ids without preprocessed traces (the general log_quotient_degree tests)
void* oracle_prove_machine(const uint32_t* chip_ids, uint32_t n_chips, const uint32_t* const* main, const uint64_t* heights, const uint32_t* rc480,
                           uint32_t log_blowup, uint32_t num_queries, uint32_t pow_bits, int debug_check) {
    std::vector<int> ids(chip_ids, chip_ids + n_chips);
    MachineDesc md = MachineDesc::of(ids);
    MachineInput in;
    for (uint32_t i = 0; i < n_chips; i++) in.main_traces.push_back(to_matrix(main[i], heights[i], md.chips[i].width));
    auto* res = new ProveResult();
    auto t0 = std::chrono::steady_clock::now();
    MachineProof p = prove(md, in, make_cfg(rc480, log_blowup, num_queries, pow_bits), &res->dbg, debug_check != 0);
    res->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    res->words = serialize_proof(p);
    return res;
}
int oracle_verify_machine(const uint32_t* chip_ids, uint32_t n_chips, const uint32_t* proof, uint64_t n_words, const uint32_t* rc480, uint32_t log_blowup,
                          uint32_t num_queries, uint32_t pow_bits, char* msg, uint64_t msg_cap) {
    MachineProof p;
    const char* err = nullptr;
    if (!deserialize_proof(proof, n_words, p)) err = "malformed proof";
    if (!err) err = verify(MachineDesc::of(std::vector<int>(chip_ids, chip_ids + n_chips)), {}, p, make_cfg(rc480, log_blowup, num_queries, pow_bits));
    if (err && msg && msg_cap) { strncpy(msg, err, msg_cap - 1); msg[msg_cap - 1] = 0; }
    return err ? 1 : 0;
}
uint64_t oracle_result_len(void* r) { return ((ProveResult*)r)->words.size(); }
const uint32_t* oracle_result_words(void* r) { return ((ProveResult*)r)->words.data(); }
double oracle_result_seconds(void* r) { return ((ProveResult*)r)->seconds; }
// 8 (prep root) + 15 (perm challenges) + 5 (alpha) + 5 (zeta) words
void oracle_result_transcript(void* r, uint32_t* out33) {
    auto& d = ((ProveResult*)r)->dbg;
    int k = 0;
    for (auto& x : d.preprocessed_commit) out33[k++] = x.v;
    for (auto& e : d.perm_challenges) for (auto& c : e.c) out33[k++] = c.v;
    for (auto& c : d.alpha.c) out33[k++] = c.v;
    for (auto& c : d.zeta.c) out33[k++] = c.v;
}
// permutation trace of chip i (height x 5(M+1) words, row-major flattened)
uint64_t oracle_result_perm_trace(void* r, uint32_t chip, uint32_t* out, uint64_t cap) {
    auto& t = ((ProveResult*)r)->dbg.perm_traces[chip];
    if (out && cap >= 5 * t.size()) for (size_t i = 0; i < t.size(); i++) for (int k = 0; k < 5; k++) out[5 * i + k] = t[i].c[k].v;
    return 5 * t.size();
}
// quotient chunk matrix of chip i (height x 10, row-major)
uint64_t oracle_result_quotient(void* r, uint32_t chip, uint32_t* out, uint64_t cap) {
    auto& m = ((ProveResult*)r)->dbg.quotient_chunks[chip];
    if (out && cap >= m.v.size()) for (size_t i = 0; i < m.v.size(); i++) out[i] = m.v[i].v;
    return m.v.size();
}
void oracle_result_free(void* r) { delete (ProveResult*)r; }

// 0 = accepted; otherwise a nonzero code, message in `msg` (may be null).
int oracle_verify_basic(const uint32_t* prep_program, uint64_t hp, const uint32_t* prep_range, const uint32_t* proof, uint64_t n_words,
                        const uint32_t* rc480, uint32_t log_blowup, uint32_t num_queries, uint32_t pow_bits, char* msg, uint64_t msg_cap) {
    MachineProof p;
    const char* err = nullptr;
    if (!deserialize_proof(proof, n_words, p)) err = "malformed proof";
    if (!err) {
        std::vector<std::pair<int, Matrix>> prep;
        prep.push_back({chips::PROGRAM, to_matrix(prep_program, hp, 7)});
        prep.push_back({chips::RANGE, to_matrix(prep_range, 256, 1)});
        err = verify(MachineDesc::basic(), prep, p, make_cfg(rc480, log_blowup, num_queries, pow_bits));
    }
    if (err && msg && msg_cap) { strncpy(msg, err, msg_cap - 1); msg[msg_cap - 1] = 0; }
    return err ? 1 : 0;
}

}  // extern "C"
