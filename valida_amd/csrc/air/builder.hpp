// AirBuilder surface — the C++ mirror of the traits Valida's chips are written against
// (p3_air::{AirBuilder, PairBuilder, PermutationAirBuilder} as used by machine/src/chip.rs:15-20 and
// every chip's `impl Air<AB>`; reference folders: machine/src/folding_builder.rs:6-125,
// machine/src/symbolic/symbolic_builder.rs:57-154, machine/src/debug_builder.rs:7-114).
//
// A chip's `eval` is a function template over a builder type AB.  AB must provide:
//     using Expr;                               value type with + - * and unary -
//     Expr constant(uint32_t canonical);        AbstractField::from_canonical_u32
//     Expr main(int col, bool next);            builder.main().row_slice(next)[col]
//     Expr preprocessed(int col, bool next);    builder.preprocessed().row_slice(next)[col]
//     Expr is_first_row(), is_last_row(), is_transition();
//     void assert_zero(const Expr&);
// The sugar below restates p3-air's default methods (last paragraph):
//     assert_eq(a,b) = assert_zero(a-b); assert_one(a) = assert_zero(a-1);
//     assert_bool(a) = assert_zero(a*(a-1)); when(c).assert_zero(x) = assert_zero(c*x);
//     when_ne(a,b) = when(a-b); nested whens multiply.
// Only the VALUE and the ORDER of the asserted polynomials matter (exact field arithmetic); the
// order of assert_* calls below must match the Rust source because it fixes the alpha powers.
#pragma once
#include <cstdint>
#include <vector>

// Chip `eval` templates and the sugar below are compiled for the host (symbolic capture, the CPU checker's
// folders) AND, unchanged, for gfx950 (kernels/quotient.hip instantiates them over a device folder).
#if defined(__HIPCC__)
#define VAIR_HD __host__ __device__
#else
#define VAIR_HD
#endif

namespace vair {

template <class AB>
struct When {
    AB& b;
    typename AB::Expr cond;
    VAIR_HD When<AB> when(const typename AB::Expr& c) const { return When<AB>{b, cond * c}; }
    VAIR_HD When<AB> when_ne(const typename AB::Expr& x, const typename AB::Expr& y) const { return when(x - y); }
    VAIR_HD void assert_zero(const typename AB::Expr& x) const { b.assert_zero(cond * x); }
    VAIR_HD void assert_eq(const typename AB::Expr& x, const typename AB::Expr& y) const { assert_zero(x - y); }
    VAIR_HD void assert_one(const typename AB::Expr& x) const { assert_zero(x - b.constant(1)); }
};

template <class AB> VAIR_HD When<AB> when(AB& b, const typename AB::Expr& c) { return When<AB>{b, c}; }
template <class AB> VAIR_HD When<AB> when_ne(AB& b, const typename AB::Expr& x, const typename AB::Expr& y) { return When<AB>{b, x - y}; }
template <class AB> VAIR_HD When<AB> when_first_row(AB& b) { return When<AB>{b, b.is_first_row()}; }
template <class AB> VAIR_HD When<AB> when_last_row(AB& b) { return When<AB>{b, b.is_last_row()}; }
template <class AB> VAIR_HD When<AB> when_transition(AB& b) { return When<AB>{b, b.is_transition()}; }
template <class AB> VAIR_HD void assert_eq(AB& b, const typename AB::Expr& x, const typename AB::Expr& y) { b.assert_zero(x - y); }
template <class AB> VAIR_HD void assert_one(AB& b, const typename AB::Expr& x) { b.assert_zero(x - b.constant(1)); }
template <class AB> VAIR_HD void assert_bool(AB& b, const typename AB::Expr& x) { b.assert_zero(x * (x - b.constant(1))); }

// VirtualPairCol (p3_air) restricted to what Valida uses: an affine form over one row of
// (preprocessed, main) columns with canonical u32 weights.
struct VirtualCol {
    struct Term { bool preprocessed; int col; uint32_t weight; };
    std::vector<Term> terms;
    uint32_t constant = 0;
    static VirtualCol single_main(int c) { VirtualCol v; v.terms.push_back({false, c, 1}); return v; }
    static VirtualCol single_preprocessed(int c) { VirtualCol v; v.terms.push_back({true, c, 1}); return v; }
    static VirtualCol constant_(uint32_t k) { VirtualCol v; v.constant = k; return v; }
    static VirtualCol sum_main(std::vector<int> cols) { VirtualCol v; for (int c : cols) v.terms.push_back({false, c, 1}); return v; }
    static VirtualCol new_main(std::vector<std::pair<int, uint32_t>> cw, uint32_t k) {
        VirtualCol v; for (auto& p : cw) v.terms.push_back({false, p.first, p.second}); v.constant = k; return v;
    }
};

// machine/src/chip.rs:76-94
enum class BusKind { Local, Global };
enum class InteractionType { LocalSend, LocalReceive, GlobalSend, GlobalReceive };
struct Interaction {
    std::vector<VirtualCol> fields;
    VirtualCol count;
    BusKind bus_kind;
    int bus_index;
    InteractionType type;
    bool is_send() const { return type == InteractionType::LocalSend || type == InteractionType::GlobalSend; }
    bool is_local() const { return bus_kind == BusKind::Local; }
};

}  // namespace vair
