// Symbolic capture of a chip's constraints and its compilation to the device constraint program.
//
// Mirrors SymbolicAirBuilder / SymbolicExpression (machine/src/symbolic/symbolic_builder.rs:57-154,
// symbolic_expression.rs:12-62): `Air::eval` is run once against this builder; every assert_zero
// records a DAG node.  The DAG (hash-consed, constants folded — value-preserving only) is then
// lowered to a linear register program that the quotient kernel interprets once per LDE row
// (kernels/quotient.hip).  get_log_quotient_degree (symbolic_builder.rs:17-30) falls out of the same
// capture via degree multiples.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <tuple>
#include <vector>
#include "../field.hpp"
#include "builder.hpp"

namespace vair {

enum NodeKind : uint8_t { N_CONST = 0, N_MAIN, N_PREP, N_FIRST, N_LAST, N_TRANS, N_ADD, N_SUB, N_MUL, N_NEG };

struct Node {
    NodeKind kind;
    uint32_t a, b;  // CONST: a = canonical value.  MAIN/PREP: a = column, b = is_next.  ops: operand node ids.
    int degree;     // degree multiple (symbolic_expression.rs:36-62)
};

struct Dag {
    std::vector<Node> nodes;
    std::vector<uint32_t> constraints;  // node ids, in assert order
    std::map<std::tuple<int, uint32_t, uint32_t>, uint32_t> memo;
    int width = 0, prep_width = 0;

    uint32_t intern(NodeKind k, uint32_t a, uint32_t b, int degree) {
        auto key = std::make_tuple((int)k, a, b);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        nodes.push_back({k, a, b, degree});
        memo[key] = (uint32_t)nodes.size() - 1;
        return (uint32_t)nodes.size() - 1;
    }
    bool is_const(uint32_t id, uint32_t* v = nullptr) const {
        if (nodes[id].kind != N_CONST) return false;
        if (v) *v = nodes[id].a;
        return true;
    }
    uint32_t constant(uint32_t canonical) { return intern(N_CONST, canonical % vg::P, 0, 0); }
    uint32_t var(bool prep, int col, bool next) { return intern(prep ? N_PREP : N_MAIN, (uint32_t)col, next ? 1 : 0, 1); }
    uint32_t selector(NodeKind k) { return intern(k, 0, 0, k == N_TRANS ? 0 : 1); }
    uint32_t add(uint32_t x, uint32_t y) {
        uint32_t cx, cy;
        if (is_const(x, &cx) && is_const(y, &cy)) return constant((uint32_t)(((uint64_t)cx + cy) % vg::P));
        if (is_const(x, &cx) && cx == 0) return y;
        if (is_const(y, &cy) && cy == 0) return x;
        if (x > y) std::swap(x, y);
        return intern(N_ADD, x, y, std::max(nodes[x].degree, nodes[y].degree));
    }
    uint32_t sub(uint32_t x, uint32_t y) {
        uint32_t cx, cy;
        if (is_const(x, &cx) && is_const(y, &cy)) return constant((uint32_t)(((uint64_t)cx + vg::P - cy) % vg::P));
        if (is_const(y, &cy) && cy == 0) return x;
        if (x == y) return constant(0);
        if (is_const(x, &cx) && cx == 0) return neg(y);
        return intern(N_SUB, x, y, std::max(nodes[x].degree, nodes[y].degree));
    }
    uint32_t neg(uint32_t x) {
        uint32_t cx;
        if (is_const(x, &cx)) return constant(cx ? vg::P - cx : 0);
        return intern(N_NEG, x, 0, nodes[x].degree);
    }
    uint32_t mul(uint32_t x, uint32_t y) {
        uint32_t cx, cy;
        if (is_const(x, &cx) && is_const(y, &cy)) return constant((uint32_t)(((uint64_t)cx * cy) % vg::P));
        if (is_const(x, &cx)) { if (cx == 0) return x; if (cx == 1) return y; }
        if (is_const(y, &cy)) { if (cy == 0) return y; if (cy == 1) return x; }
        if (x > y) std::swap(x, y);
        // NOTE: degree is tracked on the un-simplified semantics the reference uses (sum of degrees);
        // folding x*1 can only lower it, and max(deg, 3) in log_quotient_degree absorbs that.
        return intern(N_MUL, x, y, nodes[x].degree + nodes[y].degree);
    }
    int max_degree() const { int d = 0; for (uint32_t c : constraints) d = std::max(d, nodes[c].degree); return d; }
};

// AirBuilder over the DAG.
struct SymbolicBuilder {
    Dag* dag;
    struct Expr {
        Dag* d = nullptr;
        uint32_t id = 0;
        Expr operator+(const Expr& o) const { return Expr{d, d->add(id, o.id)}; }
        Expr operator-(const Expr& o) const { return Expr{d, d->sub(id, o.id)}; }
        Expr operator*(const Expr& o) const { return Expr{d, d->mul(id, o.id)}; }
        Expr operator-() const { return Expr{d, d->neg(id)}; }
    };
    explicit SymbolicBuilder(Dag* d) : dag(d) {}
    Expr constant(uint32_t k) const { return Expr{dag, dag->constant(k)}; }
    Expr main(int c, bool next) const { return Expr{dag, dag->var(false, c, next)}; }
    Expr preprocessed(int c, bool next) const { return Expr{dag, dag->var(true, c, next)}; }
    Expr is_first_row() const { return Expr{dag, dag->selector(N_FIRST)}; }
    Expr is_last_row() const { return Expr{dag, dag->selector(N_LAST)}; }
    Expr is_transition() const { return Expr{dag, dag->selector(N_TRANS)}; }
    void assert_zero(const Expr& x) { dag->constraints.push_back(x.id); }
};

// Degree-only builder with the reference's exact (unsimplified) degree rules, used for lqd.
struct DegreeBuilder {
    struct Expr {
        int d = 0;
        Expr operator+(const Expr& o) const { return Expr{std::max(d, o.d)}; }
        Expr operator-(const Expr& o) const { return Expr{std::max(d, o.d)}; }
        Expr operator-() const { return *this; }
        Expr operator*(const Expr& o) const { return Expr{d + o.d}; }
    };
    int max_degree = 0;
    Expr constant(uint32_t) const { return Expr{0}; }
    Expr main(int, bool) const { return Expr{1}; }
    Expr preprocessed(int, bool) const { return Expr{1}; }
    Expr is_first_row() const { return Expr{1}; }
    Expr is_last_row() const { return Expr{1}; }
    Expr is_transition() const { return Expr{0}; }
    void assert_zero(const Expr& x) { max_degree = std::max(max_degree, x.d); }
};
inline unsigned log2_ceil_u(unsigned n) { unsigned k = 0; while ((1u << k) < n) k++; return k; }
inline unsigned log_quotient_degree_from(int max_constraint_degree) { return log2_ceil_u((unsigned)std::max(max_constraint_degree, 3) - 1); }

// ---- device constraint program -------------------------------------------------------------------
enum OpCode : uint8_t { OP_CONST = 0, OP_LOAD_MAIN, OP_LOAD_PREP, OP_SEL_FIRST, OP_SEL_LAST, OP_SEL_TRANS, OP_ADD, OP_SUB, OP_MUL, OP_NEG, OP_ASSERT, OP_NOP };

struct Instr {  // 8 bytes
    uint8_t op, flag;  // flag: is_next for loads
    uint16_t dst;
    uint16_t a, b;  // CONST: a | b<<16 = Montgomery value; LOAD: a = column; ops: source registers; ASSERT: a = register
};
static_assert(sizeof(Instr) == 8, "Instr must be 8 bytes");

struct Program {
    std::vector<Instr> instrs;
    uint32_t num_regs = 0;
    uint32_t num_asserts = 0;
};

// Lower the DAG: dead-node elimination, then a single pass in creation order (a topological order)
// with last-use based register reuse.  A node whose value is asserted is kept alive until its assert,
// which is emitted right after the node itself so its position in the instruction stream equals the
// constraint order.
inline Program compile(const Dag& dag) {
    size_t n = dag.nodes.size();
    // Emission order: process constraints in order; for each, emit its (not yet emitted) dependency
    // cone in node-id order, then the ASSERT.  This localises leaf loads near their first use.
    std::vector<std::vector<uint32_t>> emit_groups;
    std::vector<char> emitted(n, 0);
    std::vector<uint32_t> stack;
    for (uint32_t c : dag.constraints) {
        std::vector<uint32_t> cone;
        stack.push_back(c);
        while (!stack.empty()) {
            uint32_t id = stack.back(); stack.pop_back();
            if (emitted[id]) continue;
            emitted[id] = 1;
            cone.push_back(id);
            const Node& nd = dag.nodes[id];
            if (nd.kind >= N_ADD) { stack.push_back(nd.a); if (nd.kind != N_NEG) stack.push_back(nd.b); }
        }
        std::sort(cone.begin(), cone.end());
        emit_groups.push_back(std::move(cone));
    }
    // order[] = flat emission list with ASSERT markers (id | 0x80000000 => assert of constraint)
    std::vector<uint32_t> order;
    for (size_t g = 0; g < emit_groups.size(); g++) {
        for (uint32_t id : emit_groups[g]) order.push_back(id);
        order.push_back(0x80000000u | dag.constraints[g]);
    }
    // last use position of each node
    std::vector<long> last_use(n, -1);
    for (size_t pos = 0; pos < order.size(); pos++) {
        uint32_t e = order[pos];
        if (e & 0x80000000u) { last_use[e & 0x7fffffffu] = (long)pos; continue; }
        const Node& nd = dag.nodes[e];
        if (nd.kind >= N_ADD) { last_use[nd.a] = (long)pos; if (nd.kind != N_NEG) last_use[nd.b] = (long)pos; }
    }
    Program p;
    std::vector<int> reg_of(n, -1);
    std::vector<uint16_t> free_regs;
    uint32_t next_reg = 0;
    auto alloc = [&]() -> uint16_t {
        if (!free_regs.empty()) { uint16_t r = free_regs.back(); free_regs.pop_back(); return r; }
        return (uint16_t)next_reg++;
    };
    auto release_if_dead = [&](uint32_t id, long pos) {
        if (last_use[id] == pos && reg_of[id] >= 0) { free_regs.push_back((uint16_t)reg_of[id]); reg_of[id] = -1; }
    };
    for (size_t pos = 0; pos < order.size(); pos++) {
        uint32_t e = order[pos];
        if (e & 0x80000000u) {
            uint32_t id = e & 0x7fffffffu;
            p.instrs.push_back({OP_ASSERT, 0, 0, (uint16_t)reg_of[id], 0});
            p.num_asserts++;
            release_if_dead(id, (long)pos);
            continue;
        }
        const Node& nd = dag.nodes[e];
        Instr in{};
        uint16_t ra = 0, rb = 0;
        if (nd.kind >= N_ADD) { ra = (uint16_t)reg_of[nd.a]; rb = nd.kind != N_NEG ? (uint16_t)reg_of[nd.b] : 0; }
        // sources may be released before the destination is allocated (dst may alias a dying source)
        if (nd.kind >= N_ADD) { release_if_dead(nd.a, (long)pos); if (nd.kind != N_NEG && nd.b != nd.a) release_if_dead(nd.b, (long)pos); }
        uint16_t dst = alloc();
        reg_of[e] = dst;
        switch (nd.kind) {
            case N_CONST: { uint32_t m = vg::Fp::from_canonical(nd.a).v; in = {OP_CONST, 0, dst, (uint16_t)(m & 0xffff), (uint16_t)(m >> 16)}; break; }
            case N_MAIN: in = {OP_LOAD_MAIN, (uint8_t)nd.b, dst, (uint16_t)nd.a, 0}; break;
            case N_PREP: in = {OP_LOAD_PREP, (uint8_t)nd.b, dst, (uint16_t)nd.a, 0}; break;
            case N_FIRST: in = {OP_SEL_FIRST, 0, dst, 0, 0}; break;
            case N_LAST: in = {OP_SEL_LAST, 0, dst, 0, 0}; break;
            case N_TRANS: in = {OP_SEL_TRANS, 0, dst, 0, 0}; break;
            case N_ADD: in = {OP_ADD, 0, dst, ra, rb}; break;
            case N_SUB: in = {OP_SUB, 0, dst, ra, rb}; break;
            case N_MUL: in = {OP_MUL, 0, dst, ra, rb}; break;
            case N_NEG: in = {OP_NEG, 0, dst, ra, 0}; break;
        }
        p.instrs.push_back(in);
        if (last_use[e] < 0) { free_regs.push_back(dst); reg_of[e] = -1; }  // unreachable in practice
    }
    p.num_regs = next_reg;
    while (p.instrs.size() % 4) p.instrs.push_back({OP_NOP, 0, 0, 0, 0});  // the device interpreter fetches 4 instructions at a time
    return p;
}

// Host interpreter of a Program over canonical inputs — used by the CPU tests to check that the
// lowering preserves the chip's constraints (compare against a direct template instantiation).
struct HostEval {
    const vg::Fp *main_local, *main_next, *prep_local, *prep_next;
    vg::Fp first, last, trans;
    // returns the asserted values in order
    std::vector<vg::Fp> run(const Program& p) const {
        std::vector<vg::Fp> regs(p.num_regs), out;
        for (const Instr& in : p.instrs) {
            switch (in.op) {
                case OP_CONST: regs[in.dst] = vg::Fp::raw((uint32_t)in.a | ((uint32_t)in.b << 16)); break;
                case OP_LOAD_MAIN: regs[in.dst] = (in.flag ? main_next : main_local)[in.a]; break;
                case OP_LOAD_PREP: regs[in.dst] = (in.flag ? prep_next : prep_local)[in.a]; break;
                case OP_SEL_FIRST: regs[in.dst] = first; break;
                case OP_SEL_LAST: regs[in.dst] = last; break;
                case OP_SEL_TRANS: regs[in.dst] = trans; break;
                case OP_ADD: regs[in.dst] = regs[in.a] + regs[in.b]; break;
                case OP_SUB: regs[in.dst] = regs[in.a] - regs[in.b]; break;
                case OP_MUL: regs[in.dst] = regs[in.a] * regs[in.b]; break;
                case OP_NEG: regs[in.dst] = -regs[in.a]; break;
                case OP_ASSERT: out.push_back(regs[in.a]); break;
                case OP_NOP: break;
            }
        }
        return out;
    }
};

}  // namespace vair
