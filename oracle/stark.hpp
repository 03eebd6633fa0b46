// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).  PARITY UNPINNED vs Plonky3@bdd338d6.
//
// CPU restatement of the hot path `Machine::prove` and of its acceptance check `Machine::verify`:
//   prove          basic/src/lib.rs:147-675 (transcript order: SURVEY.md Appendix C)
//   verify         basic/src/lib.rs:677-1064 + machine/src/verify.rs:11-107
//   perm trace     machine/src/chip.rs:121-208, :291-352; util/src/lib.rs:21-43
//   perm constr.   machine/src/chip.rs:210-289
//   quotient       machine/src/quotient.rs:18-238, folder machine/src/folding_builder.rs:32-125
//   decompose      p3_uni_stark::decompose_and_flatten (App. B11), zerofier (App. B11)
//   debug check    machine/src/check_constraints.rs:14-93
// The chips' AIR definitions and interactions are INPUT to the prover (Chip::eval / all_interactions): the oracle has its
// OWN transcription of them from the Rust sources (chips.hpp) — nothing under valida_amd/ is included here.
#pragma once
#include <memory>
#include <string>
#include "chips.hpp"
#include "pcs.hpp"

namespace oracle {
using chips::Interaction;
using VirtualCol = chips::VirtualPairCol;

// ---------------------------------------------------------------- folders
inline Ext5 to_ext(const Fp& x) { return Ext5(x); }
inline Ext5 to_ext(const Ext5& x) { return x; }

template <class T> struct TwoRows { const T* local = nullptr; const T* next = nullptr; };

// ProverConstraintFolder (T = Fp) and VerifierConstraintFolder (T = Ext5).
template <class T> struct Folder {
    using Expr = T;
    TwoRows<T> main_, prep_;
    TwoRows<Ext5> perm_;
    size_t perm_width = 0;
    T first, last, trans;
    Ext5 alpha, acc;
    size_t num_constraints = 0;
    T from_u32(uint32_t k) const { return T(Fp(k)); }
    const T* main_local() const { return main_.local; }
    const T* main_next() const { return main_.next; }
    T is_first_row() const { return first; }
    T is_last_row() const { return last; }
    T is_transition() const { return trans; }
    void assert_zero(const T& x) { acc = acc * alpha + to_ext(x); num_constraints++; }
    void assert_zero_ext(const Ext5& x) { acc = acc * alpha + x; num_constraints++; }
};

// Fast mode's ProverConstraintFolder: the same sum  acc = sum_k alpha^(K-1-k) c_k  with the K powers of alpha precomputed (K = the chip's
// constraint count, taken from one evaluation with the folder above): a base-field constraint costs five products instead of the
// twenty-five of acc * alpha, accumulated in five lazily reduced 64-bit sums.
struct FastFolder {
    using Expr = Fp;
    TwoRows<Fp> main_, prep_;
    TwoRows<Ext5> perm_;
    size_t perm_width = 0;
    Fp first, last, trans;
    const Ext5* apow = nullptr;  // apow[k] = alpha^(K-1-k)
    uint64_t a[5] = {0, 0, 0, 0, 0};
    unsigned pending = 0;
    size_t num_constraints = 0;
    Fp from_u32(uint32_t k) const { return Fp(k); }
    const Fp* main_local() const { return main_.local; }
    const Fp* main_next() const { return main_.next; }
    Fp is_first_row() const { return first; }
    Fp is_last_row() const { return last; }
    Fp is_transition() const { return trans; }
    void assert_zero(const Fp& x) {
        const Ext5& w = apow[num_constraints++];
        const uint64_t v = x.v;
        for (int k = 0; k < 5; k++) a[k] += (uint64_t)w.c[k].v * v;
        if (++pending == 3) { for (int k = 0; k < 5; k++) a[k] %= P; pending = 0; }
    }
    void assert_zero_ext(const Ext5& x) {
        const Ext5 t = fast::ext_mul(apow[num_constraints++], x);
        for (int k = 0; k < 5; k++) a[k] = (a[k] + t.c[k].v) % P;
        pending = 0;
    }
    Ext5 acc() const { Ext5 r; for (int k = 0; k < 5; k++) r.c[k] = Fp::from_u64(a[k]); return r; }
};

// DebugConstraintBuilder (machine/src/debug_builder.rs): records the first failing constraint index.
struct DebugBuilder {
    using Expr = Fp;
    TwoRows<Fp> main_, prep_;
    TwoRows<Ext5> perm_;
    size_t perm_width = 0;
    Fp first, last, trans;
    long failed = -1;
    size_t num_constraints = 0;
    Fp from_u32(uint32_t k) const { return Fp(k); }
    const Fp* main_local() const { return main_.local; }
    const Fp* main_next() const { return main_.next; }
    Fp is_first_row() const { return first; }
    Fp is_last_row() const { return last; }
    Fp is_transition() const { return trans; }
    void assert_zero(const Fp& x) { if (!x.is_zero() && failed < 0) failed = (long)num_constraints; num_constraints++; }
    void assert_zero_ext(const Ext5& x) { if (!x.is_zero() && failed < 0) failed = (long)num_constraints; num_constraints++; }
};

// SymbolicAirBuilder restricted to what get_log_quotient_degree needs: degree multiples
// (machine/src/symbolic/symbolic_expression.rs:36-62).
struct DegreeBuilder {
    struct Expr {
        int d = 0;
        Expr operator+(const Expr& o) const { return Expr{std::max(d, o.d)}; }
        Expr operator-(const Expr& o) const { return Expr{std::max(d, o.d)}; }
        Expr operator-() const { return *this; }
        Expr operator*(const Expr& o) const { return Expr{d + o.d}; }
    };
    int max_degree = 0;
    size_t num_constraints = 0;
    std::vector<Expr> row = std::vector<Expr>(128, Expr{1});  // every trace variable has degree 1
    Expr from_u32(uint32_t) const { return Expr{0}; }
    const Expr* main_local() const { return row.data(); }
    const Expr* main_next() const { return row.data(); }
    Expr is_first_row() const { return Expr{1}; }
    Expr is_last_row() const { return Expr{1}; }
    Expr is_transition() const { return Expr{0}; }
    void assert_zero(const Expr& x) { max_degree = std::max(max_degree, x.d); num_constraints++; }
};

// ---------------------------------------------------------------- machine description
struct ChipDesc {
    int id;  // chips::ChipIndex — selects eval and interactions
    std::vector<Interaction> interactions;
    size_t width, prep_width;
};
struct MachineDesc {
    std::vector<ChipDesc> chips;
    // a machine of the given chips::ChipIndex / test-AIR ids (tests of the general log_quotient_degree path)
    static MachineDesc of(const std::vector<int>& ids) {
        MachineDesc m;
        for (int id : ids) {
            ChipDesc c;
            c.id = id;
            c.interactions = chips::all_interactions(id);
            c.width = chips::chip_shape(id).width;
            c.prep_width = chips::chip_shape(id).preprocessed_width;
            m.chips.push_back(c);
        }
        return m;
    }
    static MachineDesc basic() {
        MachineDesc m;
        for (int i = 0; i < chips::NUM_CHIPS; i++) {
            ChipDesc c;
            c.id = i;
            c.interactions = chips::all_interactions(i);
            c.width = chips::chip_shape(i).width;
            c.prep_width = chips::chip_shape(i).preprocessed_width;
            m.chips.push_back(c);
        }
        return m;
    }
};

// get_log_quotient_degree (machine/src/symbolic/symbolic_builder.rs:17-30)
inline unsigned log_quotient_degree(const ChipDesc& c) {
    DegreeBuilder db;
    chips::eval(c.id, db);
    int deg = std::max(db.max_degree, 3);
    return log2_ceil((size_t)(deg - 1));
}

// ---------------------------------------------------------------- permutation argument
template <class T> T apply_vcol(const VirtualCol& v, const T* prep, const T* main) {
    T acc = T(Fp(v.constant));
    for (auto& t : v.terms) acc = acc + (t.preprocessed ? prep[t.col] : main[t.col]) * T(Fp(t.weight));
    return acc;
}

// generate_rlc_elements (machine/src/chip.rs:291-331): alpha_bus = r^(index+1), r = r0 (local) / r1 (global)
inline Ext5 bus_alpha(const Interaction& it, const std::vector<Ext5>& rnd) {
    const Ext5& r = it.is_local() ? rnd[0] : rnd[1];
    return r.pow((uint64_t)it.bus_index + 1);
}

// generate_permutation_trace (machine/src/chip.rs:121-208) -> height x (M+1) Ext5, row-major.
inline std::vector<Ext5> generate_permutation_trace(const ChipDesc& chip, const Matrix& main, const Matrix* prep,
                                                    const std::vector<Ext5>& rnd) {
    size_t M = chip.interactions.size(), W = M + 1, n = main.height;
    std::vector<Ext5> perm(n * W);
    std::vector<Ext5> alphas(M), betas;
    size_t max_fields = 0;
    for (size_t m = 0; m < M; m++) { alphas[m] = bus_alpha(chip.interactions[m], rnd); max_fields = std::max(max_fields, chip.interactions[m].fields.size()); }
    Ext5 bp = Ext5::one();
    for (size_t j = 0; j < max_fields; j++) { betas.push_back(bp); bp *= rnd[2]; }
    const bool fast_mode = fast::enabled();  // fast mode: ONE batch inversion (the reference's own batch_multiplicative_inverse_allowing_zero) instead of one per element
    #pragma omp parallel for
    for (size_t r = 0; r < n; r++) {
        const Fp* prow = prep ? prep->row(r) : nullptr;
        for (size_t m = 0; m < M; m++) {
            Ext5 rlc;
            auto& f = chip.interactions[m].fields;
            for (size_t j = 0; j < f.size(); j++) rlc += betas[j] * apply_vcol<Fp>(f[j], prow, main.row(r));
            rlc += alphas[m];
            // batch_multiplicative_inverse_allowing_zero: zeros stay zero (util/src/lib.rs:21-43)
            perm[r * W + m] = (fast_mode || rlc.is_zero()) ? rlc : rlc.inv();
        }
    }
    if (fast_mode && M) {
        std::vector<Ext5> tmp(n * M);
        #pragma omp parallel for schedule(static)
        for (size_t r = 0; r < n; r++) for (size_t m = 0; m < M; m++) tmp[r * M + m] = perm[r * W + m];
        fast::batch_inverse_ext(tmp.data(), tmp.size());
        #pragma omp parallel for schedule(static)
        for (size_t r = 0; r < n; r++) for (size_t m = 0; m < M; m++) perm[r * W + m] = tmp[r * M + m];
    }
    if (fast_mode) {
        // the running sum as a parallel prefix sum: per-row terms, per-chunk totals, a serial scan over the chunk totals, chunk-local sums —
        // exact arithmetic: the same values as the serial loop below
        const size_t CH = 8192, nch = (n + CH - 1) / CH;
        std::vector<Ext5> total(nch);
        #pragma omp parallel for schedule(static)
        for (size_t c = 0; c < nch; c++) {
            Ext5 acc;
            for (size_t r = c * CH; r < std::min(n, (c + 1) * CH); r++) {
                const Fp* prow = prep ? prep->row(r) : nullptr;
                for (size_t m = 0; m < M; m++) {
                    Fp mult = apply_vcol<Fp>(chip.interactions[m].count, prow, main.row(r));
                    if (chip.interactions[m].is_send()) acc += perm[r * W + m] * mult; else acc -= perm[r * W + m] * mult;
                }
                perm[r * W + M] = acc;  // chunk-local running sum
            }
            total[c] = acc;
        }
        std::vector<Ext5> before(nch);
        Ext5 run;
        for (size_t c = 0; c < nch; c++) { before[c] = run; run += total[c]; }
        #pragma omp parallel for schedule(static)
        for (size_t c = 1; c < nch; c++)
            for (size_t r = c * CH; r < std::min(n, (c + 1) * CH); r++) perm[r * W + M] += before[c];
        return perm;
    }
    Ext5 phi;
    for (size_t r = 0; r < n; r++) {  // serial running sum (chip.rs:178-201)
        const Fp* prow = prep ? prep->row(r) : nullptr;
        for (size_t m = 0; m < M; m++) {
            Fp mult = apply_vcol<Fp>(chip.interactions[m].count, prow, main.row(r));
            if (chip.interactions[m].is_send()) phi += perm[r * W + m] * mult; else phi -= perm[r * W + m] * mult;
        }
        perm[r * W + M] = phi;
    }
    return perm;
}

// eval_permutation_constraints (machine/src/chip.rs:210-289)
template <class B, class T = typename B::Expr>
void eval_permutation_constraints(const ChipDesc& chip, B& b, const std::vector<Ext5>& rnd, const Ext5& cumulative_sum) {
    size_t M = chip.interactions.size();
    const Ext5* pl = b.perm_.local; const Ext5* pn = b.perm_.next;
    Ext5 phi_local = pl[M], phi_next = pn[M];
    Ext5 lhs = phi_next - phi_local, rhs, phi_0;
    for (size_t m = 0; m < M; m++) {
        auto& it = chip.interactions[m];
        Ext5 rlc, beta = Ext5::one();
        for (auto& f : it.fields) {
            T elem = apply_vcol<T>(f, b.prep_.local, b.main_.local);
            rlc += beta * to_ext(elem);
            beta *= rnd[2];
        }
        rlc = rlc + bus_alpha(it, rnd);
        b.assert_zero_ext(rlc * pl[m] - Ext5::one());  // assert_one_ext
        Ext5 mult_local = to_ext(apply_vcol<T>(it.count, b.prep_.local, b.main_.local));
        Ext5 mult_next = to_ext(apply_vcol<T>(it.count, b.prep_.next, b.main_.next));
        if (it.is_send()) { phi_0 += pl[m] * mult_local; rhs += pn[m] * mult_next; }
        else { phi_0 -= pl[m] * mult_local; rhs -= pn[m] * mult_next; }
    }
    b.assert_zero_ext(to_ext(b.is_transition()) * (lhs - rhs));
    b.assert_zero_ext(to_ext(b.is_first_row()) * (pl[M] - phi_0));
    b.assert_zero_ext(to_ext(b.is_last_row()) * (pl[M] - cumulative_sum));
}

// check_constraints (machine/src/check_constraints.rs:14-84).  Returns "" or a description of the
// first violated (row, constraint).
inline std::string check_constraints(const ChipDesc& chip, const Matrix& main, const Matrix* prep,
                                     const std::vector<Ext5>& perm, const std::vector<Ext5>& rnd) {
    size_t n = main.height, W = chip.interactions.size() + 1;
    Ext5 cumulative_sum = perm[(n - 1) * W + W - 1];
    std::string err;
    for (size_t i = 0; i < n && err.empty(); i++) {
        size_t j = (i + 1) % n;
        DebugBuilder b;
        b.main_ = {main.row(i), main.row(j)};
        if (prep) b.prep_ = {prep->row(i), prep->row(j)};
        b.perm_ = {&perm[i * W], &perm[j * W]};
        b.first = i == 0 ? Fp::one() : Fp::zero();
        b.last = i == n - 1 ? Fp::one() : Fp::zero();
        b.trans = i == n - 1 ? Fp::zero() : Fp::one();
        chips::eval(chip.id, b);
        eval_permutation_constraints(chip, b, rnd, cumulative_sum);
        if (b.failed >= 0) err = std::string(chips::chip_shape(chip.id).name) + ": row " + std::to_string(i) + " constraint " + std::to_string(b.failed);
    }
    return err;
}

// ---------------------------------------------------------------- quotient
// quotient_values (machine/src/quotient.rs:70-238): one Ext5 per point of the quotient domain, natural order.
inline std::vector<Ext5> quotient_values(const ChipDesc& chip, unsigned log_degree, unsigned lqd, const LdeView* prep_lde,
                                         const LdeView& main_lde, const LdeView& perm_lde, unsigned log_blowup,
                                         const Ext5& cumulative_sum, const std::vector<Ext5>& rnd, const Ext5& alpha) {
    size_t qsize = size_t(1) << (log_degree + lqd);
    size_t stride = size_t(1) << (log_blowup - lqd);  // vertically_strided (quotient.rs:41-47)
    Fp g_sub = two_adic_generator(log_degree), g_ext = two_adic_generator(log_degree + lqd);
    Fp subgroup_last = g_sub.inv(), s = coset_shift();
    size_t next_step = size_t(1) << lqd;
    // ZerofierOnCoset::new(log_degree, lqd, s): evals[i] = s^n * w_{2^lqd}^i - 1
    std::vector<Fp> zh(next_step), zh_inv(next_step);
    Fp s_pow_n = s.exp_power_of_2(log_degree), wq = two_adic_generator(lqd), wp = Fp::one();
    for (size_t i = 0; i < next_step; i++) { zh[i] = s_pow_n * wp - Fp::one(); zh_inv[i] = zh[i].inv(); wp *= wq; }
    size_t mw = main_lde.width(), pw = prep_lde ? prep_lde->width() : 0, ew = perm_lde.width() / 5;
    std::vector<Ext5> out(qsize);
    if (fast::enabled()) {
        // fast mode: the points by running products, the two selector denominators by batch inversion, rows read in place (a committed
        // row IS a contiguous row-major row; five consecutive base elements of the flattened permutation LDE ARE one Ext5), powers of alpha
        static_assert(sizeof(Ext5) == 5 * sizeof(Fp), "Ext5 must be five packed base elements");
        std::vector<Fp> xs(qsize), inv_first(qsize), inv_last(qsize);
        #pragma omp parallel for schedule(static)
        for (size_t c0 = 0; c0 < qsize; c0 += 4096) {
            Fp cur = s * g_ext.pow(c0);
            for (size_t i = c0; i < std::min(qsize, c0 + 4096); i++) { xs[i] = cur; inv_first[i] = cur - Fp::one(); inv_last[i] = cur - subgroup_last; cur *= g_ext; }
        }
        fast::batch_inverse(inv_first.data(), qsize);
        fast::batch_inverse(inv_last.data(), qsize);
        size_t K = 0;
        {   // the chip's constraint count: one evaluation with the plain folder
            std::vector<Fp> zm(mw), zp(pw);
            std::vector<Ext5> ze(ew);
            Folder<Fp> f;
            f.main_ = {zm.data(), zm.data()}; f.prep_ = {zp.data(), zp.data()}; f.perm_ = {ze.data(), ze.data()};
            f.alpha = alpha;
            chips::eval(chip.id, f);
            eval_permutation_constraints(chip, f, rnd, cumulative_sum);
            K = f.num_constraints;
        }
        std::vector<Ext5> apow(K ? K : 1);
        Ext5 ap = Ext5::one();
        for (size_t k = K; k-- > 0;) { apow[k] = ap; ap = fast::ext_mul(ap, alpha); }
        const unsigned kb = main_lde.k;
        #pragma omp parallel for schedule(static)
        for (size_t i = 0; i < qsize; i++) {
            const size_t inext = (i + next_step) % qsize;
            const size_t r0 = reverse_bits_len(i * stride, kb), r1 = reverse_bits_len(inext * stride, kb);
            FastFolder f;
            f.main_ = {main_lde.m->row(r0), main_lde.m->row(r1)};
            if (prep_lde) f.prep_ = {prep_lde->m->row(r0), prep_lde->m->row(r1)};
            f.perm_ = {reinterpret_cast<const Ext5*>(perm_lde.m->row(r0)), reinterpret_cast<const Ext5*>(perm_lde.m->row(r1))};
            f.trans = xs[i] - subgroup_last;
            const Fp z = zh[i % next_step];
            f.first = z * inv_first[i];
            f.last = z * inv_last[i];
            f.apow = apow.data();
            chips::eval(chip.id, f);
            eval_permutation_constraints(chip, f, rnd, cumulative_sum);
            out[i] = f.acc() * zh_inv[i % next_step];
        }
        return out;
    }
    #pragma omp parallel
    {
        std::vector<Fp> ml(mw), mn(mw), pl(pw), pn(pw);
        std::vector<Ext5> el(ew), en(ew);
        #pragma omp for schedule(static)
        for (size_t i = 0; i < qsize; i++) {
            size_t inext = (i + next_step) % qsize;
            Fp x = s * g_ext.pow(i);
            for (size_t c = 0; c < mw; c++) { ml[c] = main_lde.get(i * stride, c); mn[c] = main_lde.get(inext * stride, c); }
            for (size_t c = 0; c < pw; c++) { pl[c] = prep_lde->get(i * stride, c); pn[c] = prep_lde->get(inext * stride, c); }
            for (size_t c = 0; c < ew; c++)
                for (int k = 0; k < 5; k++) { el[c].c[k] = perm_lde.get(i * stride, 5 * c + k); en[c].c[k] = perm_lde.get(inext * stride, 5 * c + k); }
            Folder<Fp> f;
            f.main_ = {ml.data(), mn.data()};
            f.prep_ = {pl.data(), pn.data()};
            f.perm_ = {el.data(), en.data()};
            f.trans = x - subgroup_last;
            Fp z = zh[i % next_step];
            f.first = z * (x - Fp::one()).inv();           // lagrange_basis_unnormalized(0)
            f.last = z * (x - subgroup_last).inv();         // lagrange_basis_unnormalized(degree-1)
            f.alpha = alpha;
            chips::eval(chip.id, f);
            eval_permutation_constraints(chip, f, rnd, cumulative_sum);
            out[i] = f.acc * zh_inv[i % next_step];
        }
    }
    return out;
}

// p3_uni_stark::decompose (App. B11), recursive even/odd split; chunk order as produced by the recursion.
inline std::vector<std::vector<Ext5>> decompose(const std::vector<Ext5>& poly, Fp shift, unsigned log_chunks) {
    if (log_chunks == 0) return {poly};
    size_t n = poly.size(), half = n / 2;
    Fp g_inv = two_adic_generator(log2_strict(n)).inv(), one_half = Fp(2).inv();
    std::vector<Ext5> even(half), odd(half);
    if (fast::enabled()) {  // the same loop in chunks with their own starting power
        const Fp sinv = shift.inv();
        #pragma omp parallel for schedule(static)
        for (size_t c0 = 0; c0 < half; c0 += 4096) {
            Fp gp = sinv * g_inv.pow(c0);
            for (size_t i = c0; i < std::min(half, c0 + 4096); i++) {
                Ext5 a = poly[i], bb = poly[half + i];
                even[i] = (a + bb) * one_half;
                odd[i] = (a - bb) * (one_half * gp);
                gp *= g_inv;
            }
        }
        auto combined = decompose(even, shift * shift, log_chunks - 1);
        auto rest = decompose(odd, shift * shift, log_chunks - 1);
        combined.insert(combined.end(), rest.begin(), rest.end());
        return combined;
    }
    Fp gp = shift.inv();  // g_inv.shifted_powers(shift.inverse())
    for (size_t i = 0; i < half; i++) {
        Ext5 a = poly[i], bb = poly[half + i];
        even[i] = (a + bb) * one_half;
        odd[i] = (a - bb) * (one_half * gp);
        gp *= g_inv;
    }
    auto combined = decompose(even, shift * shift, log_chunks - 1);
    auto rest = decompose(odd, shift * shift, log_chunks - 1);
    combined.insert(combined.end(), rest.begin(), rest.end());
    return combined;
}
inline Matrix decompose_and_flatten(const std::vector<Ext5>& q, Fp shift, unsigned log_chunks) {
    auto chunks = decompose(q, shift, log_chunks);
    size_t degree = chunks[0].size();
    Matrix m(degree, 5 * chunks.size());
    #pragma omp parallel for schedule(static) if (fast::enabled())
    for (size_t r = 0; r < degree; r++)
        for (size_t ch = 0; ch < chunks.size(); ch++)
            for (int k = 0; k < 5; k++) m.at(r, 5 * ch + k) = chunks[ch][r].c[k];
    return m;
}

// ---------------------------------------------------------------- proof
struct ChipProof {
    unsigned log_degree;
    std::vector<Ext5> trace_local, trace_next, permutation_local, permutation_next, quotient_chunks;  // preprocessed_* always empty (basic/src/lib.rs:641)
    Ext5 cumulative_sum;
};
struct MachineProof {
    Digest main_commit, perm_commit, quotient_commit;
    PcsProof opening_proof;
    std::vector<ChipProof> chip_proofs;
};

struct StarkConfig {
    std::vector<uint32_t> poseidon_constants;  // 480 canonical values
    FriConfig fri;
};

struct MachineInput {
    std::vector<Matrix> main_traces;                 // one per chip, chip order
    std::vector<std::pair<int, Matrix>> preprocessed;  // (chip index, trace), chip order — [program, range] for BasicMachine
};

struct ProveDebug {  // intermediate values, exposed for stage-parity tests
    Digest preprocessed_commit;
    std::vector<Ext5> perm_challenges;
    Ext5 alpha, zeta;
    std::vector<std::vector<Ext5>> perm_traces;
    std::vector<Matrix> quotient_chunks;
};

inline MachineProof prove(const MachineDesc& machine, const MachineInput& in, const StarkConfig& cfg, ProveDebug* dbg = nullptr,
                          bool debug_check = false) {
    PhaseClock clk;
    size_t NC = machine.chips.size();
    Poseidon16 perm16(cfg.poseidon_constants.data());
    Challenger ch(&perm16);
    std::vector<unsigned> lqd(NC), log_degrees(NC);
    for (size_t i = 0; i < NC; i++) { lqd[i] = log_quotient_degree(machine.chips[i]); log_degrees[i] = log2_strict(in.main_traces[i].height); }

    // preprocessed (lib.rs:189-201)
    std::vector<Matrix> prep_traces;
    std::vector<const Matrix*> prep_of(NC, nullptr);
    std::vector<int> prep_slot(NC, -1);
    for (auto& p : in.preprocessed) prep_traces.push_back(p.second);
    MerkleTree prep_tree;
    if (!prep_traces.empty()) {  // a machine without preprocessed traces commits and observes nothing here
        prep_tree = pcs_commit(prep_traces, cfg.fri);
        ch.observe(prep_tree.root());
    }
    for (size_t k = 0; k < in.preprocessed.size(); k++) { prep_of[in.preprocessed[k].first] = &in.preprocessed[k].second; prep_slot[in.preprocessed[k].first] = (int)k; }

    clk.lap("commit preprocessed");
    // main (lib.rs:203-225)
    MerkleTree main_tree = pcs_commit(in.main_traces, cfg.fri);
    ch.observe(main_tree.root());
    clk.lap("commit main");
    std::vector<Ext5> rnd;
    for (int i = 0; i < 3; i++) rnd.push_back(ch.sample_ext());

    // permutation traces (lib.rs:232-261)
    std::vector<std::vector<Ext5>> perm_traces(NC);
    std::vector<Matrix> perm_flat(NC);
    std::vector<Ext5> cumulative_sums(NC);
    for (size_t i = 0; i < NC; i++) {
        perm_traces[i] = generate_permutation_trace(machine.chips[i], in.main_traces[i], prep_of[i], rnd);
        size_t W = machine.chips[i].interactions.size() + 1, n = in.main_traces[i].height;
        cumulative_sums[i] = perm_traces[i][(n - 1) * W + W - 1];
        perm_flat[i] = Matrix(n, 5 * W);  // flatten_to_base
        #pragma omp parallel for schedule(static) if (fast::enabled())
        for (size_t r = 0; r < n; r++) for (size_t c = 0; c < W; c++) for (int k = 0; k < 5; k++) perm_flat[i].at(r, 5 * c + k) = perm_traces[i][r * W + c].c[k];
    }
    clk.lap("permutation traces");
    MerkleTree perm_tree = pcs_commit(perm_flat, cfg.fri);
    ch.observe(perm_tree.root());
    Ext5 alpha = ch.sample_ext();
    clk.lap("commit permutation");

    if (debug_check) {  // #[cfg(debug_assertions)] check_constraints + check_cumulative_sums
        for (size_t i = 0; i < NC; i++) {
            std::string e = check_constraints(machine.chips[i], in.main_traces[i], prep_of[i], perm_traces[i], rnd);
            if (!e.empty()) { fprintf(stderr, "oracle: constraint check failed: %s\n", e.c_str()); abort(); }
        }
        Ext5 sum;
        for (auto& c : cumulative_sums) sum += c;
        if (!sum.is_zero()) { fprintf(stderr, "oracle: cumulative sums do not cancel\n"); abort(); }
    }

    // quotients (lib.rs:265-599)
    std::vector<Matrix> quotients(NC);
    std::vector<Fp> coset_shifts(NC);
    for (size_t i = 0; i < NC; i++) {
        LdeView main_lde(&main_tree.leaves[i]), perm_lde(&perm_tree.leaves[i]);
        std::unique_ptr<LdeView> pl;
        if (prep_slot[i] >= 0) pl.reset(new LdeView(&prep_tree.leaves[prep_slot[i]]));
        auto qv = quotient_values(machine.chips[i], log_degrees[i], lqd[i], pl.get(), main_lde, perm_lde, cfg.fri.log_blowup,
                                  cumulative_sums[i], rnd, alpha);
        if (i == 0 || i == 2) clk.lap(i == 0 ? "  quotient values (cpu)" : "  quotient values (.. mem)");
        quotients[i] = decompose_and_flatten(qv, coset_shift(), lqd[i]);
        if (i == 0 || i == 2) clk.lap(i == 0 ? "  decompose (cpu)" : "  decompose (mem)");
        coset_shifts[i] = coset_shift().exp_power_of_2(lqd[i]);
    }
    clk.lap("quotient values + decompose");
    MerkleTree quot_tree = pcs_commit(quotients, coset_shifts, cfg.fri);
    ch.observe(quot_tree.root());
    clk.lap("commit quotient");

    // opening (lib.rs:606-619)
    Ext5 zeta = ch.sample_ext();
    RoundData rmain{&main_tree, {}}, rperm{&perm_tree, {}}, rquot{&quot_tree, {}};
    for (size_t i = 0; i < NC; i++) {
        Fp g = two_adic_generator(log_degrees[i]);
        rmain.points.push_back({zeta, zeta * g});
        rperm.points.push_back({zeta, zeta * g});
        rquot.points.push_back({zeta.exp_power_of_2(lqd[i])});
    }
    auto opened = pcs_open({rmain, rperm, rquot}, ch, cfg.fri);
    clk.lap("open (values, reduce, FRI)");

    MachineProof proof;
    proof.main_commit = main_tree.root();
    proof.perm_commit = perm_tree.root();
    proof.quotient_commit = quot_tree.root();
    proof.opening_proof = std::move(opened.second);
    for (size_t i = 0; i < NC; i++) {
        ChipProof cp;
        cp.log_degree = log_degrees[i];
        cp.trace_local = opened.first[0][i][0];
        cp.trace_next = opened.first[0][i][1];
        cp.permutation_local = opened.first[1][i][0];
        cp.permutation_next = opened.first[1][i][1];
        cp.quotient_chunks = opened.first[2][i][0];
        cp.cumulative_sum = cumulative_sums[i];
        proof.chip_proofs.push_back(std::move(cp));
    }
    if (dbg) {
        if (!prep_traces.empty()) dbg->preprocessed_commit = prep_tree.root();
        dbg->perm_challenges = rnd;
        dbg->alpha = alpha;
        dbg->zeta = zeta;
        dbg->perm_traces = std::move(perm_traces);
        dbg->quotient_chunks = std::move(quotients);
    }
    return proof;
}

// verify_constraints (machine/src/verify.rs:11-107)
inline bool verify_constraints(const ChipDesc& chip, const ChipProof& cp, unsigned lqd, const Ext5& zeta, const Ext5& alpha,
                               const std::vector<Ext5>& rnd) {
    Fp g = two_adic_generator(cp.log_degree);
    Ext5 z_h = zeta.exp_power_of_2(cp.log_degree) - Fp::one();
    Ext5 is_first = z_h * (zeta - Fp::one()).inv();
    Ext5 is_last = z_h * (zeta - g.inv()).inv();
    Ext5 is_trans = zeta - g.inv();
    auto unflatten = [](const std::vector<Ext5>& v) {
        std::vector<Ext5> out;
        for (size_t i = 0; i + 5 <= v.size(); i += 5) {
            Ext5 acc;
            for (int k = 0; k < 5; k++) acc += v[i + k] * Ext5::monomial(k);
            out.push_back(acc);
        }
        return out;
    };
    if (cp.trace_local.size() != chip.width || cp.trace_next.size() != chip.width) return false;
    size_t W = chip.interactions.size() + 1;
    if (cp.permutation_local.size() != 5 * W || cp.permutation_next.size() != 5 * W) return false;
    if (cp.quotient_chunks.size() != (size_t(5) << lqd)) return false;
    std::vector<Ext5> pl = unflatten(cp.permutation_local), pn = unflatten(cp.permutation_next), parts = unflatten(cp.quotient_chunks);
    // preprocessed openings are never produced (basic/src/lib.rs:612-613, :641); no in-tree chip reads them.
    std::vector<Ext5> empty_prep(chip.prep_width);
    Folder<Ext5> f;
    f.main_ = {cp.trace_local.data(), cp.trace_next.data()};
    f.prep_ = {empty_prep.data(), empty_prep.data()};
    f.perm_ = {pl.data(), pn.data()};
    f.first = is_first; f.last = is_last; f.trans = is_trans;
    f.alpha = alpha;
    chips::eval(chip.id, f);
    eval_permutation_constraints(chip, f, rnd, cp.cumulative_sum);
    reverse_slice_index_bits(parts);
    Ext5 quotient, zp = Ext5::one();
    for (auto& p : parts) { quotient += p * zp; zp *= zeta; }
    return f.acc == z_h * quotient;
}

inline const char* verify(const MachineDesc& machine, const std::vector<std::pair<int, Matrix>>& preprocessed, const MachineProof& proof,
                          const StarkConfig& cfg) {
    size_t NC = machine.chips.size();
    if (proof.chip_proofs.size() != NC) return "wrong number of chip proofs";
    Poseidon16 perm16(cfg.poseidon_constants.data());
    Challenger ch(&perm16);
    std::vector<unsigned> lqd(NC);
    for (size_t i = 0; i < NC; i++) lqd[i] = log_quotient_degree(machine.chips[i]);
    std::vector<Matrix> prep_traces;
    for (auto& p : preprocessed) prep_traces.push_back(p.second);
    if (!prep_traces.empty()) {
        MerkleTree prep_tree = pcs_commit(prep_traces, cfg.fri);  // recomputed (lib.rs:791-804)
        ch.observe(prep_tree.root());
    }
    ch.observe(proof.main_commit);
    std::vector<Ext5> rnd;
    for (int i = 0; i < 3; i++) rnd.push_back(ch.sample_ext());
    ch.observe(proof.perm_commit);
    Ext5 alpha = ch.sample_ext();
    ch.observe(proof.quotient_commit);
    Ext5 zeta = ch.sample_ext();
    VerifyRound rm{proof.main_commit, {}, {}}, rp{proof.perm_commit, {}, {}}, rq{proof.quotient_commit, {}, {}};
    OpenedValues values(3);
    for (size_t i = 0; i < NC; i++) {
        auto& cp = proof.chip_proofs[i];
        if (cp.log_degree > 27) return "bad log_degree";
        Fp g = two_adic_generator(cp.log_degree);
        size_t h = size_t(1) << cp.log_degree;
        rm.heights.push_back(h); rp.heights.push_back(h); rq.heights.push_back(h);
        rm.points.push_back({zeta, zeta * g});
        rp.points.push_back({zeta, zeta * g});
        rq.points.push_back({zeta.exp_power_of_2(lqd[i])});
        values[0].push_back({cp.trace_local, cp.trace_next});
        values[1].push_back({cp.permutation_local, cp.permutation_next});
        values[2].push_back({cp.quotient_chunks});
    }
    if (!pcs_verify({rm, rp, rq}, values, proof.opening_proof, ch, cfg.fri)) return "PCS opening proof rejected";
    for (size_t i = 0; i < NC; i++)
        if (!verify_constraints(machine.chips[i], proof.chip_proofs[i], lqd[i], zeta, alpha, rnd)) return "out-of-domain constraint mismatch";
    Ext5 sum;
    for (auto& cp : proof.chip_proofs) sum += cp.cumulative_sum;
    if (!sum.is_zero()) return "cumulative sums do not cancel";  // lib.rs:1052-1061
    return nullptr;
}

}  // namespace oracle
