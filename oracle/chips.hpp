// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).  PARITY UNPINNED vs Plonky3@bdd338d6.
//
// The ORACLE'S OWN transcription of the 14 BasicMachine chips (basic/src/lib.rs:151-166), written from the Rust sources and
// sharing nothing with the product's valida_amd/csrc/chips/basic_machine.hpp (tests/test_chip_transcriptions.py evaluates both
// on random rows and compares constraint vectors and interaction descriptors; tests/golden/reference_shapes.json, extracted
// from the Rust by tools/extract_reference_shapes.py, pins both against the reference).
//
// Style follows the reference rather than the product: every chip's columns are a struct with the reference's field names
// in the reference's declaration order (the `#[derive(AlignedBorrow)] struct XCols<T>` of each columns.rs), a trace row is
// BORROWED as that struct (`main.row_slice(0).borrow()`), and the column map used by the interactions is the struct
// instantiated over indices (`X_COL_MAP`, `make_col_map()`: transmute of [0, 1, .., N-1]).
//
//   chip            columns.rs                      Air::eval                        interactions
//   cpu             cpu/src/columns.rs:8-77         cpu/src/stark.rs:17-314          cpu/src/lib.rs:99-159
//   program         program/src/columns.rs:8-17     program/src/stark.rs:14-41       program/src/lib.rs:50-68  (none)
//   mem             memory/src/columns.rs:8-39      memory/src/stark.rs:16-78 (none) memory/src/lib.rs:205-233
//   add / sub       alu_u32/src/{add,sub}/columns.rs, stark.rs:21-54 / :21-51, mod.rs:53-87
//   mul             alu_u32/src/mul/columns.rs:8-24,  stark.rs:23-82,   mod.rs:68-96
//   div             alu_u32/src/div/columns.rs:8-17,  stark.rs:18-20 (none), mod.rs:55-80
//   shift           alu_u32/src/shift/columns.rs:8-25, stark.rs:21-69,  mod.rs:58-116
//   lt              alu_u32/src/lt/columns.rs:8-36,   stark.rs:21-168,  mod.rs:58-85
//   com             alu_u32/src/com/columns.rs:8-24,  stark.rs:21-49,   mod.rs:56-83
//   bitwise         alu_u32/src/bitwise/columns.rs:8-24, stark.rs:22-74, mod.rs:56-82
//   output          output/src/columns.rs:7-26,     output/src/stark.rs:21-39,       output/src/lib.rs:117-136
//   range           range/src/columns.rs:5-8,       range/src/stark.rs:12-25 (none), range/src/lib.rs:46-55
//   static_data     static_data/src/columns.rs:8-17, static_data/src/stark.rs:25-37, static_data/src/lib.rs:81-96
#pragma once
#include <cstring>
#include <string>
#include <utility>
#include <vector>
#include "field.hpp"

namespace oracle {
namespace chips {

// ---------------------------------------------------------------- opcodes (opcodes/src/lib.rs:3-45)
namespace opcode {
constexpr uint32_t BYTES_PER_INSTR = 24;
constexpr uint32_t LOAD32 = 1, STORE32 = 2, JAL = 3, JALV = 4, BEQ = 5, BNE = 6, IMM32 = 7, STOP = 8, READ_ADVICE = 9, LOADFP = 10, LOADU8 = 11,
                   LOADS8 = 12, STOREU8 = 13;
constexpr uint32_t ADD32 = 100, SUB32 = 101, MUL32 = 102, DIV32 = 103, SDIV32 = 110, LT32 = 104, SHL32 = 105, SHR32 = 106, AND32 = 107, OR32 = 108,
                   XOR32 = 109, NE32 = 111, MULHU32 = 112, SRA32 = 113, MULHS32 = 114, LTE32 = 115, EQ32 = 116, SLT32 = 117, SLE32 = 118;
constexpr uint32_t WRITE = 300;
}  // namespace opcode

// ---------------------------------------------------------------- column structs
template <class T> struct Word { T b[4]; T operator[](int i) const { return b[i]; } };  // machine/src/core.rs:9 (big-endian bytes)
template <class T> struct Operands { T v[5]; T a() const { return v[0]; } T b() const { return v[1]; } T c() const { return v[2]; } T d() const { return v[3]; } T e() const { return v[4]; } };

template <class T> struct InstructionCols { T opcode; Operands<T> operands; };
template <class T> struct OpcodeFlagCols {
    T is_bus_op, is_bus_op_with_mem, is_imm_op, is_left_imm_op, is_load, is_load_u8, is_load_s8, is_store, is_store_u8, is_beq, is_bne, is_jal, is_jalv,
        is_imm32, is_advice, is_stop, is_loadfp;
};
template <class T> struct MemoryChannelCols { T used, is_read, addr; Word<T> value; };
template <class T> struct ChipChannelCols { T clk_or_zero; };
template <class T> struct CpuCols {
    T clk, pc, fp;
    InstructionCols<T> instruction;
    OpcodeFlagCols<T> opcode_flags;
    T diff, diff_inv, not_equal;
    MemoryChannelCols<T> mem_channels[3];  // CPU_MEMORY_CHANNELS
    ChipChannelCols<T> chip_channel;
    T read_addr_1() const { return mem_channels[0].addr; }
    T read_addr_2() const { return mem_channels[1].addr; }
    T write_addr() const { return mem_channels[2].addr; }
    Word<T> read_value_1() const { return mem_channels[0].value; }
    Word<T> read_value_2() const { return mem_channels[1].value; }
    Word<T> write_value() const { return mem_channels[2].value; }
    T read_1_used() const { return mem_channels[0].used; }
    T read_2_used() const { return mem_channels[1].used; }
    T write_used() const { return mem_channels[2].used; }
};
template <class T> struct ProgramCols { T multiplicity; };
template <class T> struct ProgramPreprocessedCols { T pc, opcode; Operands<T> operands; };
template <class T> struct MemoryCols { T addr; Word<T> value; T clk, is_static_initial, is_read, is_write, diff, diff_inv, addr_not_equal, counter, counter_mult; };
template <class T> struct Add32Cols { Word<T> input_1, input_2; T carry[3]; Word<T> output; T is_real; };
template <class T> struct Sub32Cols { Word<T> input_1, input_2; T borrow[3]; Word<T> output; T is_real; };
template <class T> struct Mul32Cols { Word<T> input_1, input_2, output; T r, s, is_mul, is_mulhs, is_mulhu, counter; };
template <class T> struct Div32Cols { Word<T> input_1, input_2, output; T is_div, is_sdiv; };
template <class T> struct Shift32Cols { Word<T> input_1, input_2, output; T bits_2[8]; T temp_1; Word<T> power_of_two; T is_shl, is_shr, is_sra; };
template <class T> struct Lt32Cols {
    Word<T> input_1, input_2;
    T byte_flag[4];
    T bits[9];
    T output, multiplicity, is_lt, is_lte, is_slt, is_sle, diff_inv;
    T top_bits_1[8], top_bits_2[8];
    T different_signs;
};
template <class T> struct Com32Cols { Word<T> input_1, input_2; T diff, diff_inv, not_equal, output, is_ne, is_eq; };
template <class T> struct Bitwise32Cols { Word<T> input_1, input_2; T bits_1[4][8]; T bits_2[4][8]; Word<T> output; T is_and, is_or, is_xor; };
template <class T> struct OutputCols { T clk, value, is_real, diff, counter, counter_mult, opcode; };
template <class T> struct RangeCols { T mult, counter; };
template <class T> struct RangePreprocessedCols { T counter; };  // preprocessed_trace = RowMajorMatrix::new_col(0..MAX), range/src/stark.rs:21-24 (the struct itself is a TODO in columns.rs)
template <class T> struct StaticDataCols { T addr; Word<T> value; T is_real; };

// NUM_X_COLS = size_of::<XCols<u8>>()
template <template <class> class Cols> constexpr size_t num_cols() { return sizeof(Cols<unsigned char>); }
// X_COL_MAP = transmute::<[usize; N], XCols<usize>>(indices_arr())
template <template <class> class Cols> Cols<size_t> col_map() {
    constexpr size_t N = num_cols<Cols>();
    static_assert(sizeof(Cols<size_t>) == N * sizeof(size_t), "column structs must be packed arrays of T");
    size_t idx[N];
    for (size_t i = 0; i < N; i++) idx[i] = i;
    Cols<size_t> m;
    std::memcpy(&m, idx, sizeof m);
    return m;
}
// main.row_slice(k).borrow()
template <template <class> class Cols, class T> const Cols<T>& borrow(const T* row) {
    static_assert(sizeof(Cols<T>) == num_cols<Cols>() * sizeof(T), "column structs must be packed arrays of T");
    return *reinterpret_cast<const Cols<T>*>(row);
}

// ---------------------------------------------------------------- AirBuilder sugar ([P3-RECALL] p3-air default methods)
// B provides: Expr, main_local()/main_next() (const Expr*), is_first_row(), is_last_row(), is_transition(), assert_zero(Expr),
// from_u32(uint32_t).  FilteredAirBuilder multiplies its condition into whatever is asserted; nested filters multiply.
template <class B> struct Filtered {
    using T = typename B::Expr;
    B& inner;
    T condition;
    Filtered when(const T& c) const { return Filtered{inner, condition * c}; }
    Filtered when_ne(const T& x, const T& y) const { return when(x - y); }
    void assert_zero(const T& x) const { inner.assert_zero(condition * x); }
    void assert_eq(const T& x, const T& y) const { assert_zero(x - y); }
    void assert_one(const T& x) const { assert_zero(x - inner.from_u32(1)); }
};
template <class B> struct Air {
    using T = typename B::Expr;
    B& b;
    T one, zero;
    explicit Air(B& bb) : b(bb), one(bb.from_u32(1)), zero(bb.from_u32(0)) {}
    T k(uint32_t v) const { return b.from_u32(v); }
    void assert_zero(const T& x) { b.assert_zero(x); }
    void assert_eq(const T& x, const T& y) { b.assert_zero(x - y); }
    void assert_one(const T& x) { b.assert_zero(x - one); }
    void assert_bool(const T& x) { b.assert_zero(x * (x - one)); }
    Filtered<B> when(const T& c) { return Filtered<B>{b, c}; }
    Filtered<B> when_ne(const T& x, const T& y) { return Filtered<B>{b, x - y}; }
    Filtered<B> when_first_row() { return Filtered<B>{b, b.is_first_row()}; }
    Filtered<B> when_last_row() { return Filtered<B>{b, b.is_last_row()}; }
    Filtered<B> when_transition() { return Filtered<B>{b, b.is_transition()}; }
};

// ---------------------------------------------------------------- cpu (cpu/src/stark.rs)
// fn reduce (stark.rs:308-314): sum_i base[i] * input[i]
template <class T> T reduce(const T (&base)[4], const Word<T>& input) {
    T acc = base[0] * input[0];
    for (int i = 1; i < 4; i++) acc = acc + base[i] * input[i];
    return acc;
}
// zip(x, y).map(|(a, b)| (a - b) * (a - b)).sum()
template <class T> T squared_distance(const Word<T>& x, const Word<T>& y) {
    T acc = (x[0] - y[0]) * (x[0] - y[0]);
    for (int i = 1; i < 4; i++) acc = acc + (x[i] - y[i]) * (x[i] - y[i]);
    return acc;
}

template <class B> void cpu_eval_pc(Air<B>& a, const CpuCols<typename B::Expr>& local, const CpuCols<typename B::Expr>& next, const typename B::Expr (&base)[4]) {
    using T = typename B::Expr;
    const T bytes_per_instr = a.k(opcode::BYTES_PER_INSTR);
    const T should_increment_pc = local.opcode_flags.is_imm32 + local.opcode_flags.is_loadfp + local.opcode_flags.is_bus_op + local.opcode_flags.is_advice;
    const T incremented_pc = local.pc + a.one;
    a.when_transition().when(should_increment_pc).assert_eq(next.pc, incremented_pc);
    // Branch manipulation
    const T equal = a.one - local.not_equal;
    const T next_pc_times_24_if_branching = local.instruction.operands.a();
    const T beq_next_pc_times_24 = equal * next_pc_times_24_if_branching + bytes_per_instr * local.not_equal * incremented_pc;
    const T bne_next_pc_times_24 = bytes_per_instr * equal * incremented_pc + local.not_equal * next_pc_times_24_if_branching;
    a.when_transition().when(local.opcode_flags.is_beq).assert_eq(bytes_per_instr * next.pc, beq_next_pc_times_24);
    a.when_transition().when(local.opcode_flags.is_bne).assert_eq(bytes_per_instr * next.pc, bne_next_pc_times_24);
    // Jump manipulation
    a.when_transition().when(local.opcode_flags.is_jal).assert_eq(bytes_per_instr * next.pc, local.instruction.operands.b());
    a.when_transition().when(local.opcode_flags.is_jalv).assert_eq(bytes_per_instr * next.pc, reduce(base, local.read_value_1()));
}
template <class B> void cpu_eval_fp(Air<B>& a, const CpuCols<typename B::Expr>& local, const CpuCols<typename B::Expr>& next, const typename B::Expr (&base)[4]) {
    a.when_transition().when(local.opcode_flags.is_jal).assert_eq(next.fp, local.fp + local.instruction.operands.c());
    a.when_transition().when(local.opcode_flags.is_jalv).assert_eq(next.fp, local.fp + reduce(base, local.read_value_2()));
    a.when_transition().when(a.one - local.opcode_flags.is_jal - local.opcode_flags.is_jalv).assert_eq(next.fp, local.fp);
}
template <class B> void cpu_eval_equality(Air<B>& a, const CpuCols<typename B::Expr>& local) {
    using T = typename B::Expr;
    a.assert_eq(local.diff, squared_distance(local.read_value_1(), local.read_value_2()));
    a.assert_bool(local.not_equal);
    a.assert_eq(local.not_equal, local.diff * local.diff_inv);
    const T equal = a.one - local.not_equal;
    a.assert_zero(equal * local.diff);
}
template <class B> void cpu_eval_memory_channels(Air<B>& a, const CpuCols<typename B::Expr>& local, const typename B::Expr (&base)[4]) {
    using T = typename B::Expr;
    const T bytes_per_instr = a.k(opcode::BYTES_PER_INSTR);
    const auto& fl = local.opcode_flags;
    const T is_load = fl.is_load, is_store = fl.is_store, is_jal = fl.is_jal, is_jalv = fl.is_jalv, is_beq = fl.is_beq, is_bne = fl.is_bne, is_imm32 = fl.is_imm32,
            is_loadfp = fl.is_loadfp, is_imm_op = fl.is_imm_op, is_left_imm_op = fl.is_left_imm_op, is_bus_op = fl.is_bus_op;
    for (const T* flag : {&is_load, &is_store, &is_jal, &is_jalv, &is_beq, &is_bne, &is_imm32, &is_loadfp, &is_imm_op, &is_left_imm_op, &is_bus_op}) a.assert_bool(*flag);

    const T addr_a = local.fp + local.instruction.operands.a();
    const T addr_b = local.fp + local.instruction.operands.b();
    const T addr_c = local.fp + local.instruction.operands.c();

    a.assert_one(local.mem_channels[0].is_read);
    a.assert_one(local.mem_channels[1].is_read);
    a.assert_zero(local.mem_channels[2].is_read);

    // Read (1)
    a.when(is_jalv + is_beq + is_bne + is_bus_op * (a.one - is_left_imm_op)).assert_eq(local.read_addr_1(), addr_b);
    a.when(is_load + is_store).assert_eq(local.read_addr_1(), addr_c);
    a.when(is_load + is_store + is_jalv + is_beq + is_bne + (a.one - is_left_imm_op) * is_bus_op).assert_one(local.read_1_used());
    a.when(is_jal + is_left_imm_op + is_loadfp + is_imm32).assert_zero(local.read_1_used());
    // Read (2)
    a.when(is_load).assert_eq(local.read_addr_2(), reduce(base, local.read_value_1()));
    a.when(is_store).assert_eq(local.read_addr_2(), addr_b);
    a.when(is_jalv + (a.one - is_imm_op) * is_bus_op).assert_eq(local.read_addr_2(), addr_c);
    a.when(is_load + is_store + is_jalv + (a.one - is_imm_op) * (is_beq + is_bne + is_bus_op)).assert_one(local.read_2_used());
    a.when(is_jal + is_imm_op * (is_beq + is_bne + is_bus_op) + is_loadfp + is_imm32).assert_zero(local.read_2_used());
    // Write
    a.when(is_load + is_jal + is_jalv + is_imm32 + is_bus_op + is_loadfp).assert_eq(local.write_addr(), addr_a);
    a.when(is_store).assert_eq(local.write_addr(), reduce(base, local.read_value_2()));
    a.when(is_store).assert_zero(squared_distance(local.read_value_1(), local.write_value()));
    a.when(is_load).assert_zero(squared_distance(local.read_value_2(), local.write_value()));
    a.when_transition().when(is_jal + is_jalv).assert_eq(bytes_per_instr * (local.pc + a.one), reduce(base, local.write_value()));
    {
        // operands.imm32() = Word([b, c, d, e])  (machine/src/program.rs)
        const Operands<T>& o = local.instruction.operands;
        const Word<T> imm32{{o.b(), o.c(), o.d(), o.e()}};
        a.when(is_imm32).assert_zero(squared_distance(local.write_value(), imm32));
    }
    a.when(is_loadfp).assert_eq(addr_b, reduce(base, local.write_value()));
    a.when(is_store + is_load + is_jal + is_jalv + is_imm32 + is_loadfp + is_bus_op).assert_one(local.write_used());
    a.when(is_beq + is_bne).assert_zero(local.write_used());
}
template <class B> void eval_cpu(B& b) {
    using T = typename B::Expr;
    Air<B> a(b);
    const CpuCols<T>& local = borrow<CpuCols>(b.main_local());
    const CpuCols<T>& next = borrow<CpuCols>(b.main_next());
    const T base[4] = {a.k(1u << 24), a.k(1u << 16), a.k(1u << 8), a.k(1)};
    cpu_eval_pc(a, local, next, base);
    cpu_eval_fp(a, local, next, base);
    cpu_eval_equality(a, local);
    cpu_eval_memory_channels(a, local, base);
    // Clock constraints
    a.when_first_row().assert_zero(local.clk);
    a.when_transition().assert_eq(local.clk + a.one, next.clk);
    a.when(local.opcode_flags.is_bus_op_with_mem).assert_eq(local.clk, local.chip_channel.clk_or_zero);
    a.when(a.one - local.opcode_flags.is_bus_op_with_mem).assert_zero(local.chip_channel.clk_or_zero);
    // Immediate value constraints
    a.assert_bool(local.opcode_flags.is_imm_op + local.opcode_flags.is_left_imm_op);
    a.when(local.opcode_flags.is_imm_op).assert_eq(local.instruction.operands.c(), reduce(base, local.read_value_2()));
    a.when(local.opcode_flags.is_left_imm_op).assert_eq(local.instruction.operands.b(), reduce(base, local.read_value_1()));
    // "Stop" constraints
    a.when_transition().when(local.opcode_flags.is_stop).assert_eq(next.pc, local.pc);
    a.when_last_row().assert_one(local.opcode_flags.is_stop);
}

// ---------------------------------------------------------------- add / sub (alu_u32/src/{add,sub}/stark.rs)
template <class B> void eval_add(B& b) {
    using T = typename B::Expr;
    Air<B> a(b);
    const Add32Cols<T>& local = borrow<Add32Cols>(b.main_local());
    const T one = a.one, base = a.k(1u << 8);
    const T carry_1 = local.carry[0], carry_2 = local.carry[1], carry_3 = local.carry[2];
    const T overflow_0 = local.input_1[3] + local.input_2[3] - local.output[3];
    const T overflow_1 = local.input_1[2] + local.input_2[2] - local.output[2] + carry_1;
    const T overflow_2 = local.input_1[1] + local.input_2[1] - local.output[1] + carry_2;
    const T overflow_3 = local.input_1[0] + local.input_2[0] - local.output[0] + carry_3;
    // Limb constraints
    a.assert_zero(overflow_0 * (overflow_0 - base));
    a.assert_zero(overflow_1 * (overflow_1 - base));
    a.assert_zero(overflow_2 * (overflow_2 - base));
    a.assert_zero(overflow_3 * (overflow_3 - base));
    // Carry constraints
    a.assert_zero(overflow_0 * (carry_1 - one) + (overflow_0 - base) * carry_1);
    a.assert_zero(overflow_1 * (carry_2 - one) + (overflow_1 - base) * carry_2);
    a.assert_zero(overflow_2 * (carry_3 - one) + (overflow_2 - base) * carry_3);
    a.assert_bool(carry_1);
    a.assert_bool(carry_2);
    a.assert_bool(carry_3);
}
template <class B> void eval_sub(B& b) {
    using T = typename B::Expr;
    Air<B> a(b);
    const Sub32Cols<T>& local = borrow<Sub32Cols>(b.main_local());
    const T base = a.k(1u << 8);
    const T borrow_1 = local.borrow[0], borrow_2 = local.borrow[1], borrow_3 = local.borrow[2];
    a.assert_eq(local.output[3], base * borrow_1 + local.input_1[3] - local.input_2[3]);
    a.assert_eq(local.output[2], base * borrow_2 + local.input_1[2] - local.input_2[2] - borrow_1);
    a.assert_eq(local.output[1], base * borrow_3 + local.input_1[1] - local.input_2[1] - borrow_2);
    a.assert_eq(local.output[0], local.input_1[0] - local.input_2[0] - borrow_3);
    a.assert_bool(borrow_1);
    a.assert_bool(borrow_2);
    a.assert_bool(borrow_3);
}

// ---------------------------------------------------------------- mul (alu_u32/src/mul/stark.rs)
// pi_m::<N>: sum over (i, j) in 0..N x 0..N with i + j < N of base[i+j] * input_1[3-i] * input_2[3-j]
template <class T> T pi_m(int N, const T* base, const Word<T>& in1, const Word<T>& in2, const T& zero) {
    T acc = zero;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            if (i + j < N) acc = acc + base[i + j] * in1[3 - i] * in2[3 - j];
    return acc;
}
// sigma_m::<N>: input.rev().take(N).enumerate() -> sum base[i] * x
template <class T> T sigma_m(int N, const T* base, const Word<T>& in, const T& zero) {
    T acc = zero;
    for (int i = 0; i < N; i++) acc = acc + base[i] * in[3 - i];
    return acc;
}
template <class B> void eval_mul(B& b) {
    using T = typename B::Expr;
    Air<B> a(b);
    const Mul32Cols<T>& local = borrow<Mul32Cols>(b.main_local());
    const Mul32Cols<T>& next = borrow<Mul32Cols>(b.main_next());
    const T base_m[4] = {a.k(1), a.k(1u << 8), a.k(1u << 16), a.k(1u << 24)};
    const T pi = pi_m(4, base_m, local.input_1, local.input_2, a.zero);
    const T sigma = sigma_m(4, base_m, local.output, a.zero);
    const T pi_prime = pi_m(2, base_m, local.input_1, local.input_2, a.zero);
    const T sigma_prime = sigma_m(2, base_m, local.output, a.zero);
    // Congruence checks
    a.assert_eq(pi - sigma, local.r * a.k(2));
    a.assert_eq(pi_prime - sigma_prime, local.s * base_m[2]);
    // Range check counter
    a.when_first_row().assert_eq(local.counter, a.one);
    const T counter_diff = next.counter - local.counter;
    a.when_transition().assert_zero(counter_diff * (counter_diff - a.one));
    a.when_last_row().assert_eq(local.counter, a.k(1u << 10));
}

// ---------------------------------------------------------------- shift (alu_u32/src/shift/stark.rs)
template <class B> void eval_shift(B& b) {
    using T = typename B::Expr;
    Air<B> a(b);
    const Shift32Cols<T>& local = borrow<Shift32Cols>(b.main_local());
    const T one = a.one;
    T byte_2 = a.zero;
    for (int i = 0; i < 8; i++) byte_2 = byte_2 + local.bits_2[i] * a.k(1u << i);
    a.assert_eq(local.input_2[3], byte_2);
    for (int i = 0; i < 8; i++) a.assert_bool(local.bits_2[i]);
    const T pow_base[3] = {a.k(1u << 1), a.k(1u << 2), a.k(1u << 4)};
    const T temp_1 = (local.bits_2[0] * pow_base[0]) * (local.bits_2[1] * pow_base[1]) * (local.bits_2[2] * pow_base[2]);
    a.assert_eq(local.temp_1, temp_1);
    a.assert_eq(local.power_of_two[0], local.temp_1 * (one - local.bits_2[3]) * (one - local.bits_2[4]));
    a.assert_eq(local.power_of_two[1], local.temp_1 * local.bits_2[3] * (one - local.bits_2[4]));
    a.assert_eq(local.power_of_two[2], local.temp_1 * (one - local.bits_2[3]) * local.bits_2[4]);
    a.assert_eq(local.power_of_two[3], local.temp_1 * local.bits_2[3] * local.bits_2[4]);
    a.assert_bool(local.is_shl);
    a.assert_bool(local.is_shr);
    a.assert_bool(local.is_sra);
    a.assert_bool(local.is_shl + local.is_shr + local.is_sra);
}

// ---------------------------------------------------------------- lt (alu_u32/src/lt/stark.rs)
template <class B> void eval_lt(B& b) {
    using T = typename B::Expr;
    Air<B> a(b);
    const Lt32Cols<T>& local = borrow<Lt32Cols>(b.main_local());
    const T one = a.one;
    auto weighted_bits = [&](const T* bits, int n) {  // zip(bits, base_2).map(bit * base).sum()
        T acc = a.zero;
        for (int i = 0; i < n; i++) acc = acc + bits[i] * a.k(1u << i);
        return acc;
    };
    const T bit_comp = weighted_bits(local.bits, 9);
    // ensure at most one byte flag is set
    const T flag_sum = local.byte_flag[0] + local.byte_flag[1] + local.byte_flag[2] + local.byte_flag[3];
    a.assert_bool(flag_sum);
    // bytes before the first set byte flag are all equal
    a.when_ne(local.byte_flag[0], one).assert_eq(local.input_1[0], local.input_2[0]);
    a.when_ne(local.byte_flag[0] + local.byte_flag[1], one).assert_eq(local.input_1[1], local.input_2[1]);
    a.when_ne(local.byte_flag[0] + local.byte_flag[1] + local.byte_flag[2], one).assert_eq(local.input_1[2], local.input_2[2]);
    a.when_ne(flag_sum, one).assert_eq(local.input_1[3], local.input_2[3]);
    a.when_ne(flag_sum, one).assert_eq(bit_comp, a.zero);
    // bit decomposition of z = 256 + input_1[n] - input_2[n]
    for (int i = 0; i < 4; i++) {
        a.when(local.byte_flag[i]).assert_eq(a.k(256) + local.input_1[i] - local.input_2[i], bit_comp);
        a.when(local.byte_flag[i]).assert_eq((local.input_1[i] - local.input_2[i]) * local.diff_inv, one);
        a.assert_bool(local.byte_flag[i]);
    }
    // bit decomposition of the top bytes
    const T top_comp_1 = weighted_bits(local.top_bits_1, 8), top_comp_2 = weighted_bits(local.top_bits_2, 8);
    a.assert_eq(top_comp_1, local.input_1[0]);
    a.assert_eq(top_comp_2, local.input_2[0]);

    const T is_signed = local.is_slt + local.is_sle;
    const T is_unsigned = one - is_signed;
    const T same_sign = one - local.different_signs;
    const T are_equal = one - flag_sum;

    a.when(is_unsigned).assert_zero(local.different_signs);
    a.when(is_signed).when_ne(local.top_bits_1[7], local.top_bits_2[7]).assert_eq(local.different_signs, one);
    a.when(local.different_signs).assert_eq(local.byte_flag[0], one);
    a.when(local.different_signs).assert_eq(local.top_bits_1[7] + local.top_bits_2[7], one);

    a.assert_bool(local.is_lt);
    a.assert_bool(local.is_lte);
    a.assert_bool(local.is_slt);
    a.assert_bool(local.is_sle);
    a.assert_bool(local.is_lt + local.is_lte + local.is_slt + local.is_sle);

    // Output constraints
    a.when(local.bits[8]).when(is_unsigned + same_sign).assert_zero(local.output);
    a.when(local.bits[8]).when(local.different_signs).assert_one(local.output);
    a.when_ne(local.bits[8] + are_equal, one).when(is_unsigned + same_sign).assert_one(local.output);
    a.when_ne(local.bits[8] + are_equal, one).when(local.different_signs).assert_zero(local.output);
    a.when(are_equal).when(local.is_lte + local.is_sle).assert_one(local.output);
    a.when(are_equal).when(local.is_lt + local.is_slt).assert_zero(local.output);

    // bits.chain(top_bits_1).chain(top_bits_2): all boolean
    for (int i = 0; i < 9; i++) a.assert_bool(local.bits[i]);
    for (int i = 0; i < 8; i++) a.assert_bool(local.top_bits_1[i]);
    for (int i = 0; i < 8; i++) a.assert_bool(local.top_bits_2[i]);
}

// ---------------------------------------------------------------- com (alu_u32/src/com/stark.rs)
template <class B> void eval_com(B& b) {
    using T = typename B::Expr;
    Air<B> a(b);
    const Com32Cols<T>& local = borrow<Com32Cols>(b.main_local());
    a.assert_eq(local.diff, squared_distance(local.input_1, local.input_2));
    a.assert_bool(local.not_equal);
    a.assert_eq(local.not_equal, local.diff * local.diff_inv);
    const T equal = a.one - local.not_equal;
    a.assert_zero(equal * local.diff);
    a.assert_bool(local.is_ne);
    a.assert_bool(local.is_eq);
    a.assert_bool(local.is_ne + local.is_eq);
    a.assert_eq(local.output, local.is_ne * local.not_equal + local.is_eq * (a.one - local.not_equal));
}

// ---------------------------------------------------------------- bitwise (alu_u32/src/bitwise/stark.rs)
template <class B> void eval_bitwise(B& b) {
    using T = typename B::Expr;
    Air<B> a(b);
    const Bitwise32Cols<T>& local = borrow<Bitwise32Cols>(b.main_local());
    for (int i = 0; i < 4; i++) {  // MEMORY_CELL_BYTES
        T byte_1 = a.zero, byte_2 = a.zero, bitwise_and = a.zero;
        for (int k = 0; k < 8; k++) {
            const T base = a.k(1u << k);
            byte_1 = byte_1 + local.bits_1[i][k] * base;
            byte_2 = byte_2 + local.bits_2[i][k] * base;
            bitwise_and = bitwise_and + local.bits_1[i][k] * local.bits_2[i][k] * base;
        }
        a.assert_eq(local.input_1[i], byte_1);
        a.assert_eq(local.input_2[i], byte_2);
        const T bitwise_or = byte_1 + byte_2 - bitwise_and;
        const T bitwise_xor = byte_1 + byte_2 - a.k(2) * bitwise_and;
        a.when(local.is_and).assert_eq(bitwise_and, local.output[i]);
        a.when(local.is_or).assert_eq(bitwise_or, local.output[i]);
        a.when(local.is_xor).assert_eq(bitwise_xor, local.output[i]);
        for (int k = 0; k < 8; k++) a.assert_bool(local.bits_1[i][k]);
        for (int k = 0; k < 8; k++) a.assert_bool(local.bits_2[i][k]);
    }
    a.assert_bool(local.is_and);
    a.assert_bool(local.is_or);
    a.assert_bool(local.is_xor);
    a.assert_bool(local.is_and + local.is_or + local.is_xor);
}

// ---------------------------------------------------------------- output (output/src/stark.rs), static_data (static_data/src/stark.rs)
template <class B> void eval_output(B& b) {
    using T = typename B::Expr;
    Air<B> a(b);
    const OutputCols<T>& local = borrow<OutputCols>(b.main_local());
    const OutputCols<T>& next = borrow<OutputCols>(b.main_next());
    a.when_transition().assert_eq(local.diff, next.clk - local.clk);
    a.when_transition().assert_eq(next.counter, local.counter + a.one);
    a.when(local.is_real).assert_eq(local.opcode, a.k(opcode::WRITE));
}
template <class B> void eval_static_data(B& b) {
    using T = typename B::Expr;
    Air<B> a(b);
    const StaticDataCols<T>& local = borrow<StaticDataCols>(b.main_local());
    const StaticDataCols<T>& next = borrow<StaticDataCols>(b.main_next());
    a.when_transition().when(local.is_real * next.is_real).assert_eq(next.addr, local.addr + a.one + a.one + a.one + a.one);
}

// ---------------------------------------------------------------- test AIRs of higher degree (NOT in the reference)
// The reference's chips all have constraint degree <= 3 (log_quotient_degree 1).  machine/src/quotient.rs and verify.rs handle any
// log_quotient_degree, so the oracle carries two synthetic single-chip AIRs to exercise that generality:
//   POW5 (chip id 100): columns (x, y);  y - x^5 = 0 (degree 5 -> lqd 2);  when_transition: x' - x - 1 = 0
//   POW9 (chip id 101): columns (x, y);  y - x^9 = 0 (degree 9 -> lqd 3);  when_transition: x' - x - 1 = 0;  when_first_row: x - 3 = 0
template <class B> void eval_pow(B& b, int degree, bool pin_first) {
    using T = typename B::Expr;
    Air<B> a(b);
    const T x = b.main_local()[0], y = b.main_local()[1], xn = b.main_next()[0];
    T p = x;
    for (int k = 1; k < degree; k++) p = p * x;
    a.assert_eq(y, p);
    a.when_transition().assert_eq(xn, x + a.one);
    if (pin_first) a.when_first_row().assert_eq(x, a.k(3));
}
constexpr int TEST_POW5 = 100, TEST_POW9 = 101;

// ---------------------------------------------------------------- the machine (basic/src/lib.rs:151-166)
enum ChipIndex { CPU = 0, PROGRAM, MEM, ADD, SUB, MUL, DIV, SHIFT, LT, COM, BITWISE, OUTPUT, RANGE, STATIC_DATA, NUM_CHIPS };
struct ChipShape { const char* name; size_t width, preprocessed_width; };
inline ChipShape chip_shape(int chip) {
    switch (chip) {
        case CPU: return {"cpu", num_cols<CpuCols>(), 0};
        case PROGRAM: return {"program", num_cols<ProgramCols>(), num_cols<ProgramPreprocessedCols>()};
        case MEM: return {"mem", num_cols<MemoryCols>(), 0};
        case ADD: return {"add", num_cols<Add32Cols>(), 0};
        case SUB: return {"sub", num_cols<Sub32Cols>(), 0};
        case MUL: return {"mul", num_cols<Mul32Cols>(), 0};
        case DIV: return {"div", num_cols<Div32Cols>(), 0};
        case SHIFT: return {"shift", num_cols<Shift32Cols>(), 0};
        case LT: return {"lt", num_cols<Lt32Cols>(), 0};
        case COM: return {"com", num_cols<Com32Cols>(), 0};
        case BITWISE: return {"bitwise", num_cols<Bitwise32Cols>(), 0};
        case OUTPUT: return {"output", num_cols<OutputCols>(), 0};
        case RANGE: return {"range", num_cols<RangeCols>(), num_cols<RangePreprocessedCols>()};
        case STATIC_DATA: return {"static_data", num_cols<StaticDataCols>(), 0};
        case TEST_POW5: return {"pow5", 2, 0};
        case TEST_POW9: return {"pow9", 2, 0};
    }
    return {"?", 0, 0};
}
// Air::eval of chip `chip`; program, mem, div and range have empty evals (program/src/stark.rs:14, memory/src/stark.rs:22-78
// commented out, alu_u32/src/div/stark.rs:18-20, range/src/stark.rs:12-14)
template <class B> void eval(int chip, B& b) {
    switch (chip) {
        case CPU: eval_cpu(b); break;
        case ADD: eval_add(b); break;
        case SUB: eval_sub(b); break;
        case MUL: eval_mul(b); break;
        case SHIFT: eval_shift(b); break;
        case LT: eval_lt(b); break;
        case COM: eval_com(b); break;
        case BITWISE: eval_bitwise(b); break;
        case OUTPUT: eval_output(b); break;
        case STATIC_DATA: eval_static_data(b); break;
        case TEST_POW5: eval_pow(b, 5, false); break;
        case TEST_POW9: eval_pow(b, 9, true); break;
        default: break;
    }
}

// ---------------------------------------------------------------- interactions (machine/src/chip.rs:24-94)
// VirtualPairCol ([P3-RECALL] p3-air): sum of weighted (preprocessed | main) columns + constant
struct PairTerm { bool preprocessed; size_t col; uint32_t weight; };
struct VirtualPairCol {
    std::vector<PairTerm> terms;
    uint32_t constant = 0;
    static VirtualPairCol single_main(size_t c) { VirtualPairCol v; v.terms.push_back({false, c, 1}); return v; }
    static VirtualPairCol constant_of(uint32_t k) { VirtualPairCol v; v.constant = k; return v; }
    static VirtualPairCol sum_main(std::initializer_list<size_t> cols) { VirtualPairCol v; for (size_t c : cols) v.terms.push_back({false, c, 1}); return v; }
    static VirtualPairCol new_main(std::initializer_list<std::pair<size_t, uint32_t>> cw, uint32_t k) {
        VirtualPairCol v;
        for (auto& p : cw) v.terms.push_back({false, p.first, p.second});
        v.constant = k;
        return v;
    }
};
enum class InteractionType { LocalSend, LocalReceive, GlobalSend, GlobalReceive };
struct Interaction {
    std::vector<VirtualPairCol> fields;
    VirtualPairCol count;
    bool global = true;   // BusArgument::Global(bus_index) / Local(bus_index)
    size_t bus_index = 0;
    InteractionType type = InteractionType::GlobalSend;  // from the all_interactions slot the interaction came from
    bool is_local() const { return !global; }
    bool is_send() const { return type == InteractionType::LocalSend || type == InteractionType::GlobalSend; }
};
// BasicMachine's buses (basic/src/lib.rs:1190-1212)
constexpr size_t GENERAL_BUS = 0, PROGRAM_BUS = 1, MEM_BUS = 2, RANGE_BUS = 3;

template <class W> void extend_word(std::vector<VirtualPairCol>& f, const W& w) { for (int i = 0; i < 4; i++) f.push_back(VirtualPairCol::single_main(w[i])); }

// Chip::all_interactions (chip.rs:40-63): local sends, local receives, global sends, global receives.  No BasicMachine chip
// has a live local interaction (all commented out in the reference).
inline std::vector<Interaction> all_interactions(int chip) {
    using V = VirtualPairCol;
    std::vector<Interaction> sends, receives;
    auto send = [&](std::vector<V> fields, V count, size_t bus) { sends.push_back(Interaction{std::move(fields), std::move(count), true, bus, InteractionType::GlobalSend}); };
    auto receive = [&](std::vector<V> fields, V count, size_t bus) { receives.push_back(Interaction{std::move(fields), std::move(count), true, bus, InteractionType::GlobalReceive}); };
    switch (chip) {
        case CPU: {  // global_sends, cpu/src/lib.rs:99-159
            const auto M = col_map<CpuCols>();
            for (int i = 0; i < 3; i++) {  // memory bus channels
                const auto& ch = M.mem_channels[i];
                std::vector<V> f = {V::single_main(ch.is_read), V::single_main(M.clk), V::single_main(ch.addr), V::constant_of(0)};
                extend_word(f, ch.value);
                send(f, V::single_main(ch.used), MEM_BUS);
            }
            std::vector<V> f = {V::single_main(M.instruction.opcode)};  // general bus channel
            for (int i = 0; i < 3; i++) extend_word(f, M.mem_channels[i].value);
            f.push_back(V::single_main(M.chip_channel.clk_or_zero));
            send(f, V::single_main(M.opcode_flags.is_bus_op), GENERAL_BUS);
            break;
        }
        case PROGRAM: break;  // program/src/lib.rs:50-68: vec![]
        case MEM: {           // memory/src/lib.rs:216-233
            const auto M = col_map<MemoryCols>();
            std::vector<V> f = {V::single_main(M.is_read), V::single_main(M.clk), V::single_main(M.addr), V::single_main(M.is_static_initial)};
            extend_word(f, M.value);
            receive(f, V::sum_main({M.is_read, M.is_write}), MEM_BUS);
            break;
        }
        case ADD: {  // alu_u32/src/add/mod.rs:53-87
            const auto M = col_map<Add32Cols>();
            for (int i = 0; i < 4; i++) send({V::single_main(M.output[i])}, V::single_main(M.is_real), RANGE_BUS);
            std::vector<V> f = {V::constant_of(opcode::ADD32)};
            extend_word(f, M.input_1); extend_word(f, M.input_2); extend_word(f, M.output);
            receive(f, V::single_main(M.is_real), GENERAL_BUS);
            break;
        }
        case SUB: {  // alu_u32/src/sub/mod.rs:53-87
            const auto M = col_map<Sub32Cols>();
            for (int i = 0; i < 4; i++) send({V::single_main(M.output[i])}, V::single_main(M.is_real), RANGE_BUS);
            std::vector<V> f = {V::constant_of(opcode::SUB32)};
            extend_word(f, M.input_1); extend_word(f, M.input_2); extend_word(f, M.output);
            receive(f, V::single_main(M.is_real), GENERAL_BUS);
            break;
        }
        case MUL: {  // alu_u32/src/mul/mod.rs:68-96
            const auto M = col_map<Mul32Cols>();
            std::vector<V> f = {V::new_main({{M.is_mul, opcode::MUL32}, {M.is_mulhs, opcode::MULHS32}, {M.is_mulhu, opcode::MULHU32}}, 0)};
            extend_word(f, M.input_1); extend_word(f, M.input_2); extend_word(f, M.output);
            receive(f, V::sum_main({M.is_mul, M.is_mulhs, M.is_mulhu}), GENERAL_BUS);
            break;
        }
        case DIV: {  // alu_u32/src/div/mod.rs:55-80
            const auto M = col_map<Div32Cols>();
            std::vector<V> f = {V::new_main({{M.is_div, opcode::DIV32}, {M.is_sdiv, opcode::SDIV32}}, 0)};
            extend_word(f, M.input_1); extend_word(f, M.input_2); extend_word(f, M.output);
            receive(f, V::sum_main({M.is_div, M.is_sdiv}), GENERAL_BUS);
            break;
        }
        case SHIFT: {  // alu_u32/src/shift/mod.rs:58-116
            const auto M = col_map<Shift32Cols>();
            std::vector<V> fs = {V::new_main({{M.is_shl, opcode::MUL32}, {M.is_shr, opcode::DIV32}, {M.is_sra, opcode::SDIV32}}, 0)};
            extend_word(fs, M.input_1); extend_word(fs, M.power_of_two); extend_word(fs, M.output);
            send(fs, V::sum_main({M.is_shl, M.is_shr, M.is_sra}), GENERAL_BUS);
            std::vector<V> fr = {V::new_main({{M.is_shl, opcode::SHL32}, {M.is_shr, opcode::SHR32}, {M.is_sra, opcode::SRA32}}, 0)};
            extend_word(fr, M.input_1); extend_word(fr, M.input_2); extend_word(fr, M.output);
            receive(fr, V::sum_main({M.is_shl, M.is_shr, M.is_sra}), GENERAL_BUS);
            break;
        }
        case LT: {  // alu_u32/src/lt/mod.rs:58-85
            const auto M = col_map<Lt32Cols>();
            std::vector<V> f = {V::new_main({{M.is_lt, opcode::LT32}, {M.is_lte, opcode::LTE32}, {M.is_slt, opcode::SLT32}, {M.is_sle, opcode::SLE32}}, 0)};
            extend_word(f, M.input_1); extend_word(f, M.input_2);
            for (int i = 0; i < 3; i++) f.push_back(V::constant_of(0));  // (0..MEMORY_CELL_BYTES - 1) zeros, then the output
            f.push_back(V::single_main(M.output));
            receive(f, V::single_main(M.multiplicity), GENERAL_BUS);
            break;
        }
        case COM: {  // alu_u32/src/com/mod.rs:56-83
            const auto M = col_map<Com32Cols>();
            std::vector<V> f = {V::new_main({{M.is_ne, opcode::NE32}, {M.is_eq, opcode::EQ32}}, 0)};
            extend_word(f, M.input_1); extend_word(f, M.input_2);
            for (int i = 0; i < 3; i++) f.push_back(V::constant_of(0));
            f.push_back(V::single_main(M.output));
            receive(f, V::sum_main({M.is_ne, M.is_eq}), GENERAL_BUS);
            break;
        }
        case BITWISE: {  // alu_u32/src/bitwise/mod.rs:56-82
            const auto M = col_map<Bitwise32Cols>();
            std::vector<V> f = {V::new_main({{M.is_and, opcode::AND32}, {M.is_or, opcode::OR32}, {M.is_xor, opcode::XOR32}}, 0)};
            extend_word(f, M.input_1); extend_word(f, M.input_2); extend_word(f, M.output);
            receive(f, V::sum_main({M.is_and, M.is_or, M.is_xor}), GENERAL_BUS);
            break;
        }
        case OUTPUT: {  // output/src/lib.rs:117-136
            const auto M = col_map<OutputCols>();
            std::vector<V> values(3 * 4, V::constant_of(0));  // CPU_MEMORY_CHANNELS * MEMORY_CELL_BYTES zeros
            values[4 - 1] = V::single_main(M.value);          // values[MEMORY_CELL_BYTES - 1]
            std::vector<V> f = {V::single_main(M.opcode)};
            f.insert(f.end(), values.begin(), values.end());
            f.push_back(V::single_main(M.clk));
            receive(f, V::single_main(M.is_real), GENERAL_BUS);
            break;
        }
        case RANGE: {  // range/src/lib.rs:46-55
            const auto M = col_map<RangeCols>();
            receive({V::single_main(M.counter)}, V::single_main(M.mult), RANGE_BUS);
            break;
        }
        case STATIC_DATA: {  // static_data/src/lib.rs:81-96
            const auto M = col_map<StaticDataCols>();
            std::vector<V> f = {V::constant_of(0) /* is_read */, V::constant_of(0) /* clk */, V::single_main(M.addr), V::constant_of(1) /* is_static_initial */};
            extend_word(f, M.value);
            send(f, V::single_main(M.is_real), MEM_BUS);
            break;
        }
    }
    std::vector<Interaction> all = std::move(sends);  // no local interactions: global sends, then global receives
    all.insert(all.end(), receives.begin(), receives.end());
    return all;
}

}  // namespace chips
}  // namespace oracle
