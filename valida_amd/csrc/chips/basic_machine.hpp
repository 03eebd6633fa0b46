// The 14 chips of Valida's BasicMachine (basic/src/lib.rs:65-122, chip order :151-166), their column
// maps, AIR constraints (`Air::eval`) and bus interactions, restated over the AirBuilder surface of
// air/builder.hpp so the same definitions drive (a) the symbolic capture that is compiled to the
// device constraint program and (b) any scalar folder (prover / verifier / debug) a checker wants.
// Bus ids: general = Global(0), program = Global(1), mem = Global(2), range = Global(3)
// (basic/src/lib.rs:1190-1212).
//
// Every eval keeps the reference's assert_* ORDER (it fixes the alpha powers of the fold).
#pragma once
#include <string>
#include "../air/builder.hpp"

namespace vchips {
using namespace vair;

// ---- opcodes used in interactions (opcodes/src/lib.rs:7-45)
enum : uint32_t {
    OP_LOAD32 = 1, OP_STORE32 = 2, OP_JAL = 3, OP_JALV = 4, OP_BEQ = 5, OP_BNE = 6, OP_IMM32 = 7, OP_STOP = 8,
    OP_READ_ADVICE = 9, OP_LOADFP = 10, OP_LOADU8 = 11, OP_LOADS8 = 12, OP_STOREU8 = 13,
    OP_ADD32 = 100, OP_SUB32 = 101, OP_MUL32 = 102, OP_DIV32 = 103, OP_LT32 = 104, OP_SHL32 = 105, OP_SHR32 = 106,
    OP_AND32 = 107, OP_OR32 = 108, OP_XOR32 = 109, OP_SDIV32 = 110, OP_NE32 = 111, OP_MULHU32 = 112, OP_SRA32 = 113,
    OP_MULHS32 = 114, OP_LTE32 = 115, OP_EQ32 = 116, OP_SLT32 = 117, OP_SLE32 = 118, OP_WRITE = 300,
};
constexpr uint32_t BYTES_PER_INSTR = 24;
constexpr int BUS_GENERAL = 0, BUS_PROGRAM = 1, BUS_MEM = 2, BUS_RANGE = 3;

enum ChipId {
    CHIP_CPU = 0, CHIP_PROGRAM, CHIP_MEM, CHIP_ADD, CHIP_SUB, CHIP_MUL, CHIP_DIV, CHIP_SHIFT, CHIP_LT, CHIP_COM,
    CHIP_BITWISE, CHIP_OUTPUT, CHIP_RANGE, CHIP_STATIC_DATA, NUM_CHIPS
};

// ---- column maps -----------------------------------------------------------------------------
namespace cpu {  // cpu/src/columns.rs:8-77
enum {
    CLK = 0, PC = 1, FP = 2, OPCODE = 3, OPERAND_A = 4, OPERAND_B = 5, OPERAND_C = 6, OPERAND_D = 7, OPERAND_E = 8,
    IS_BUS_OP = 9, IS_BUS_OP_WITH_MEM, IS_IMM_OP, IS_LEFT_IMM_OP, IS_LOAD, IS_LOAD_U8, IS_LOAD_S8, IS_STORE, IS_STORE_U8,
    IS_BEQ, IS_BNE, IS_JAL, IS_JALV, IS_IMM32, IS_ADVICE, IS_STOP, IS_LOADFP,  // ..25
    DIFF = 26, DIFF_INV = 27, NOT_EQUAL = 28,
    MEM0 = 29,  // each channel: used, is_read, addr, value[4]
    CH_USED = 0, CH_IS_READ = 1, CH_ADDR = 2, CH_VALUE = 3, CH_STRIDE = 7,
    CLK_OR_ZERO = 50, NUM_COLS = 51
};
constexpr int ch(int i, int f) { return MEM0 + i * CH_STRIDE + f; }
}  // namespace cpu
namespace program { enum { MULTIPLICITY = 0, NUM_COLS = 1, PRE_PC = 0, PRE_OPCODE = 1, PRE_OPERANDS = 2, NUM_PRE_COLS = 7 }; }
namespace mem {  // memory/src/columns.rs:8-39
enum { ADDR = 0, VALUE = 1, CLK = 5, IS_STATIC_INITIAL = 6, IS_READ = 7, IS_WRITE = 8, DIFF = 9, DIFF_INV = 10,
       ADDR_NOT_EQUAL = 11, COUNTER = 12, COUNTER_MULT = 13, NUM_COLS = 14 };
}
namespace add { enum { INPUT_1 = 0, INPUT_2 = 4, CARRY = 8, OUTPUT = 11, IS_REAL = 15, NUM_COLS = 16 }; }  // alu_u32/src/add/columns.rs:8-18
namespace sub { enum { INPUT_1 = 0, INPUT_2 = 4, BORROW = 8, OUTPUT = 11, IS_REAL = 15, NUM_COLS = 16 }; }
namespace mul { enum { INPUT_1 = 0, INPUT_2 = 4, OUTPUT = 8, R = 12, S = 13, IS_MUL = 14, IS_MULHS = 15, IS_MULHU = 16, COUNTER = 17, NUM_COLS = 18 }; }
namespace divc { enum { INPUT_1 = 0, INPUT_2 = 4, OUTPUT = 8, IS_DIV = 12, IS_SDIV = 13, NUM_COLS = 14 }; }
namespace shift { enum { INPUT_1 = 0, INPUT_2 = 4, OUTPUT = 8, BITS_2 = 12, TEMP_1 = 20, POWER_OF_TWO = 21, IS_SHL = 25, IS_SHR = 26, IS_SRA = 27, NUM_COLS = 28 }; }
namespace lt {  // alu_u32/src/lt/columns.rs:8-36
enum { INPUT_1 = 0, INPUT_2 = 4, BYTE_FLAG = 8, BITS = 12, OUTPUT = 21, MULTIPLICITY = 22, IS_LT = 23, IS_LTE = 24, IS_SLT = 25,
       IS_SLE = 26, DIFF_INV = 27, TOP_BITS_1 = 28, TOP_BITS_2 = 36, DIFFERENT_SIGNS = 44, NUM_COLS = 45 };
}
namespace com { enum { INPUT_1 = 0, INPUT_2 = 4, DIFF = 8, DIFF_INV = 9, NOT_EQUAL = 10, OUTPUT = 11, IS_NE = 12, IS_EQ = 13, NUM_COLS = 14 }; }
namespace bitwise { enum { INPUT_1 = 0, INPUT_2 = 4, BITS_1 = 8, BITS_2 = 40, OUTPUT = 72, IS_AND = 76, IS_OR = 77, IS_XOR = 78, NUM_COLS = 79 }; }
namespace output { enum { CLK = 0, VALUE = 1, IS_REAL = 2, DIFF = 3, COUNTER = 4, COUNTER_MULT = 5, OPCODE = 6, NUM_COLS = 7 }; }
namespace range { enum { MULT = 0, COUNTER = 1, NUM_COLS = 2, NUM_PRE_COLS = 1 }; }
namespace static_data { enum { ADDR = 0, VALUE = 1, IS_REAL = 5, NUM_COLS = 6 }; }

struct ChipInfo { const char* name; int width; int preprocessed_width; };
inline const ChipInfo& chip_info(int id) {
    static const ChipInfo infos[NUM_CHIPS] = {
        {"cpu", cpu::NUM_COLS, 0},       {"program", program::NUM_COLS, program::NUM_PRE_COLS},
        {"mem", mem::NUM_COLS, 0},       {"add", add::NUM_COLS, 0},
        {"sub", sub::NUM_COLS, 0},       {"mul", mul::NUM_COLS, 0},
        {"div", divc::NUM_COLS, 0},      {"shift", shift::NUM_COLS, 0},
        {"lt", lt::NUM_COLS, 0},         {"com", com::NUM_COLS, 0},
        {"bitwise", bitwise::NUM_COLS, 0}, {"output", output::NUM_COLS, 0},
        {"range", range::NUM_COLS, range::NUM_PRE_COLS}, {"static_data", static_data::NUM_COLS, 0}};
    return infos[id];
}

// ---- AIRs ------------------------------------------------------------------------------------

// cpu/src/stark.rs:17-306
template <class AB> VAIR_HD void eval_cpu(AB& b) {
    using E = typename AB::Expr;
    auto L = [&](int c) { return b.main(c, false); };
    auto N = [&](int c) { return b.main(c, true); };
    auto K = [&](uint32_t k) { return b.constant(k); };
    const E base[4] = {K(1u << 24), K(1u << 16), K(1u << 8), K(1)};
    auto reduce = [&](int first_col) {  // fn reduce, cpu/src/stark.rs:308-314
        E acc = base[0] * L(first_col);
        for (int i = 1; i < 4; i++) acc = acc + base[i] * L(first_col + i);
        return acc;
    };
    auto sq_diff_sum = [&](int col_a, int col_b) {  // sum_i (a_i - b_i)^2
        E acc = (L(col_a) - L(col_b)) * (L(col_a) - L(col_b));
        for (int i = 1; i < 4; i++) acc = acc + (L(col_a + i) - L(col_b + i)) * (L(col_a + i) - L(col_b + i));
        return acc;
    };
    const int RV1 = cpu::ch(0, cpu::CH_VALUE), RV2 = cpu::ch(1, cpu::CH_VALUE), WV = cpu::ch(2, cpu::CH_VALUE);
    const E one = K(1);
    const E bpi = K(BYTES_PER_INSTR);

    // eval_pc (stark.rs:203-252)
    {
        E should_increment_pc = L(cpu::IS_IMM32) + L(cpu::IS_LOADFP) + L(cpu::IS_BUS_OP) + L(cpu::IS_ADVICE);
        E incremented_pc = L(cpu::PC) + one;
        when_transition(b).when(should_increment_pc).assert_eq(N(cpu::PC), incremented_pc);
        E equal = one - L(cpu::NOT_EQUAL);
        E next_pc_times_24_if_branching = L(cpu::OPERAND_A);
        E beq_next = equal * next_pc_times_24_if_branching + bpi * L(cpu::NOT_EQUAL) * incremented_pc;
        E bne_next = bpi * equal * incremented_pc + L(cpu::NOT_EQUAL) * next_pc_times_24_if_branching;
        when_transition(b).when(L(cpu::IS_BEQ)).assert_eq(bpi * N(cpu::PC), beq_next);
        when_transition(b).when(L(cpu::IS_BNE)).assert_eq(bpi * N(cpu::PC), bne_next);
        when_transition(b).when(L(cpu::IS_JAL)).assert_eq(bpi * N(cpu::PC), L(cpu::OPERAND_B));
        when_transition(b).when(L(cpu::IS_JALV)).assert_eq(bpi * N(cpu::PC), reduce(RV1));
    }
    // eval_fp (stark.rs:254-277)
    {
        when_transition(b).when(L(cpu::IS_JAL)).assert_eq(N(cpu::FP), L(cpu::FP) + L(cpu::OPERAND_C));
        when_transition(b).when(L(cpu::IS_JALV)).assert_eq(N(cpu::FP), L(cpu::FP) + reduce(RV2));
        when_transition(b).when(one - L(cpu::IS_JAL) - L(cpu::IS_JALV)).assert_eq(N(cpu::FP), L(cpu::FP));
    }
    // eval_equality (stark.rs:279-305)
    {
        assert_eq(b, L(cpu::DIFF), sq_diff_sum(RV1, RV2));
        assert_bool(b, L(cpu::NOT_EQUAL));
        assert_eq(b, L(cpu::NOT_EQUAL), L(cpu::DIFF) * L(cpu::DIFF_INV));
        E equal = one - L(cpu::NOT_EQUAL);
        b.assert_zero(equal * L(cpu::DIFF));
    }
    // eval_memory_channels (stark.rs:70-201)
    {
        E is_load = L(cpu::IS_LOAD), is_store = L(cpu::IS_STORE), is_jal = L(cpu::IS_JAL), is_jalv = L(cpu::IS_JALV),
          is_beq = L(cpu::IS_BEQ), is_bne = L(cpu::IS_BNE), is_imm32 = L(cpu::IS_IMM32), is_loadfp = L(cpu::IS_LOADFP),
          is_imm_op = L(cpu::IS_IMM_OP), is_left_imm_op = L(cpu::IS_LEFT_IMM_OP), is_bus_op = L(cpu::IS_BUS_OP);
        assert_bool(b, is_load);
        assert_bool(b, is_store);
        assert_bool(b, is_jal);
        assert_bool(b, is_jalv);
        assert_bool(b, is_beq);
        assert_bool(b, is_bne);
        assert_bool(b, is_imm32);
        assert_bool(b, is_loadfp);
        assert_bool(b, is_imm_op);
        assert_bool(b, is_left_imm_op);
        assert_bool(b, is_bus_op);

        E addr_a = L(cpu::FP) + L(cpu::OPERAND_A);
        E addr_b = L(cpu::FP) + L(cpu::OPERAND_B);
        E addr_c = L(cpu::FP) + L(cpu::OPERAND_C);
        E read_addr_1 = L(cpu::ch(0, cpu::CH_ADDR)), read_addr_2 = L(cpu::ch(1, cpu::CH_ADDR)), write_addr = L(cpu::ch(2, cpu::CH_ADDR));
        E read_1_used = L(cpu::ch(0, cpu::CH_USED)), read_2_used = L(cpu::ch(1, cpu::CH_USED)), write_used = L(cpu::ch(2, cpu::CH_USED));

        assert_one(b, L(cpu::ch(0, cpu::CH_IS_READ)));
        assert_one(b, L(cpu::ch(1, cpu::CH_IS_READ)));
        b.assert_zero(L(cpu::ch(2, cpu::CH_IS_READ)));

        // Read (1)
        when(b, is_jalv + is_beq + is_bne + is_bus_op * (one - is_left_imm_op)).assert_eq(read_addr_1, addr_b);
        when(b, is_load + is_store).assert_eq(read_addr_1, addr_c);
        when(b, is_load + is_store + is_jalv + is_beq + is_bne + (one - is_left_imm_op) * is_bus_op).assert_one(read_1_used);
        when(b, is_jal + is_left_imm_op + is_loadfp + is_imm32).assert_zero(read_1_used);

        // Read (2)
        when(b, is_load).assert_eq(read_addr_2, reduce(RV1));
        when(b, is_store).assert_eq(read_addr_2, addr_b);
        when(b, is_jalv + (one - is_imm_op) * is_bus_op).assert_eq(read_addr_2, addr_c);
        when(b, is_load + is_store + is_jalv + (one - is_imm_op) * (is_beq + is_bne + is_bus_op)).assert_one(read_2_used);
        when(b, is_jal + is_imm_op * (is_beq + is_bne + is_bus_op) + is_loadfp + is_imm32).assert_zero(read_2_used);

        // Write
        when(b, is_load + is_jal + is_jalv + is_imm32 + is_bus_op + is_loadfp).assert_eq(write_addr, addr_a);
        when(b, is_store).assert_eq(write_addr, reduce(RV2));
        when(b, is_store).assert_zero(sq_diff_sum(RV1, WV));
        when(b, is_load).assert_zero(sq_diff_sum(RV2, WV));
        when_transition(b).when(is_jal + is_jalv).assert_eq(bpi * (L(cpu::PC) + one), reduce(WV));
        {
            // write_value vs operands.imm32() = (b, c, d, e)
            E acc = (L(WV) - L(cpu::OPERAND_B)) * (L(WV) - L(cpu::OPERAND_B));
            for (int i = 1; i < 4; i++) acc = acc + (L(WV + i) - L(cpu::OPERAND_B + i)) * (L(WV + i) - L(cpu::OPERAND_B + i));
            when(b, is_imm32).assert_zero(acc);
        }
        when(b, is_loadfp).assert_eq(addr_b, reduce(WV));
        when(b, is_store + is_load + is_jal + is_jalv + is_imm32 + is_loadfp + is_bus_op).assert_one(write_used);
        when(b, is_beq + is_bne).assert_zero(write_used);
    }
    // Clock constraints (stark.rs:33-43)
    when_first_row(b).assert_zero(L(cpu::CLK));
    when_transition(b).assert_eq(L(cpu::CLK) + one, N(cpu::CLK));
    when(b, L(cpu::IS_BUS_OP_WITH_MEM)).assert_eq(L(cpu::CLK), L(cpu::CLK_OR_ZERO));
    when(b, one - L(cpu::IS_BUS_OP_WITH_MEM)).assert_zero(L(cpu::CLK_OR_ZERO));
    // Immediate value constraints (stark.rs:45-56)
    assert_bool(b, L(cpu::IS_IMM_OP) + L(cpu::IS_LEFT_IMM_OP));
    when(b, L(cpu::IS_IMM_OP)).assert_eq(L(cpu::OPERAND_C), reduce(RV2));
    when(b, L(cpu::IS_LEFT_IMM_OP)).assert_eq(L(cpu::OPERAND_B), reduce(RV1));
    // "Stop" constraints (stark.rs:58-66)
    when_transition(b).when(L(cpu::IS_STOP)).assert_eq(N(cpu::PC), L(cpu::PC));
    when_last_row(b).assert_one(L(cpu::IS_STOP));
}

// alu_u32/src/add/stark.rs:21-54
template <class AB> VAIR_HD void eval_add(AB& b) {
    using E = typename AB::Expr;
    auto L = [&](int c) { return b.main(c, false); };
    E one = b.constant(1), base = b.constant(1u << 8);
    E carry_1 = L(add::CARRY), carry_2 = L(add::CARRY + 1), carry_3 = L(add::CARRY + 2);
    E overflow_0 = L(add::INPUT_1 + 3) + L(add::INPUT_2 + 3) - L(add::OUTPUT + 3);
    E overflow_1 = L(add::INPUT_1 + 2) + L(add::INPUT_2 + 2) - L(add::OUTPUT + 2) + carry_1;
    E overflow_2 = L(add::INPUT_1 + 1) + L(add::INPUT_2 + 1) - L(add::OUTPUT + 1) + carry_2;
    E overflow_3 = L(add::INPUT_1 + 0) + L(add::INPUT_2 + 0) - L(add::OUTPUT + 0) + carry_3;
    b.assert_zero(overflow_0 * (overflow_0 - base));
    b.assert_zero(overflow_1 * (overflow_1 - base));
    b.assert_zero(overflow_2 * (overflow_2 - base));
    b.assert_zero(overflow_3 * (overflow_3 - base));
    b.assert_zero(overflow_0 * (carry_1 - one) + (overflow_0 - base) * carry_1);
    b.assert_zero(overflow_1 * (carry_2 - one) + (overflow_1 - base) * carry_2);
    b.assert_zero(overflow_2 * (carry_3 - one) + (overflow_2 - base) * carry_3);
    assert_bool(b, carry_1);
    assert_bool(b, carry_2);
    assert_bool(b, carry_3);
}

// alu_u32/src/sub/stark.rs:21-51
template <class AB> VAIR_HD void eval_sub(AB& b) {
    using E = typename AB::Expr;
    auto L = [&](int c) { return b.main(c, false); };
    E base = b.constant(1u << 8);
    E borrow_1 = L(sub::BORROW), borrow_2 = L(sub::BORROW + 1), borrow_3 = L(sub::BORROW + 2);
    assert_eq(b, L(sub::OUTPUT + 3), base * borrow_1 + L(sub::INPUT_1 + 3) - L(sub::INPUT_2 + 3));
    assert_eq(b, L(sub::OUTPUT + 2), base * borrow_2 + L(sub::INPUT_1 + 2) - L(sub::INPUT_2 + 2) - borrow_1);
    assert_eq(b, L(sub::OUTPUT + 1), base * borrow_3 + L(sub::INPUT_1 + 1) - L(sub::INPUT_2 + 1) - borrow_2);
    assert_eq(b, L(sub::OUTPUT + 0), L(sub::INPUT_1 + 0) - L(sub::INPUT_2 + 0) - borrow_3);
    assert_bool(b, borrow_1);
    assert_bool(b, borrow_2);
    assert_bool(b, borrow_3);
}

// alu_u32/src/mul/stark.rs:23-82
template <class AB> VAIR_HD void eval_mul(AB& b) {
    using E = typename AB::Expr;
    auto L = [&](int c) { return b.main(c, false); };
    auto N = [&](int c) { return b.main(c, true); };
    const E base_m[4] = {b.constant(1), b.constant(1u << 8), b.constant(1u << 16), b.constant(1u << 24)};
    auto pi_m = [&](int n) {
        E acc = b.constant(0);
        bool first = true;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                if (i + j < n) {
                    E t = base_m[i + j] * L(mul::INPUT_1 + 3 - i) * L(mul::INPUT_2 + 3 - j);
                    acc = first ? t : acc + t;
                    first = false;
                }
        return acc;
    };
    auto sigma_m = [&](int n) {
        E acc = base_m[0] * L(mul::OUTPUT + 3);
        for (int i = 1; i < n; i++) acc = acc + base_m[i] * L(mul::OUTPUT + 3 - i);
        return acc;
    };
    E pi = pi_m(4), sigma = sigma_m(4), pi_prime = pi_m(2), sigma_prime = sigma_m(2);
    assert_eq(b, pi - sigma, L(mul::R) * b.constant(2));
    assert_eq(b, pi_prime - sigma_prime, L(mul::S) * base_m[2]);
    when_first_row(b).assert_eq(L(mul::COUNTER), b.constant(1));
    E counter_diff = N(mul::COUNTER) - L(mul::COUNTER);
    when_transition(b).assert_zero(counter_diff * (counter_diff - b.constant(1)));
    when_last_row(b).assert_eq(L(mul::COUNTER), b.constant(1u << 10));
}

// alu_u32/src/shift/stark.rs:21-69
template <class AB> VAIR_HD void eval_shift(AB& b) {
    using E = typename AB::Expr;
    auto L = [&](int c) { return b.main(c, false); };
    E one = b.constant(1);
    E byte_2 = L(shift::BITS_2) * b.constant(1);
    for (int i = 1; i < 8; i++) byte_2 = byte_2 + L(shift::BITS_2 + i) * b.constant(1u << i);
    assert_eq(b, L(shift::INPUT_2 + 3), byte_2);
    for (int i = 0; i < 8; i++) assert_bool(b, L(shift::BITS_2 + i));
    E temp_1 = (L(shift::BITS_2 + 0) * b.constant(1u << 1)) * (L(shift::BITS_2 + 1) * b.constant(1u << 2)) * (L(shift::BITS_2 + 2) * b.constant(1u << 4));
    assert_eq(b, L(shift::TEMP_1), temp_1);
    assert_eq(b, L(shift::POWER_OF_TWO + 0), L(shift::TEMP_1) * (one - L(shift::BITS_2 + 3)) * (one - L(shift::BITS_2 + 4)));
    assert_eq(b, L(shift::POWER_OF_TWO + 1), L(shift::TEMP_1) * L(shift::BITS_2 + 3) * (one - L(shift::BITS_2 + 4)));
    assert_eq(b, L(shift::POWER_OF_TWO + 2), L(shift::TEMP_1) * (one - L(shift::BITS_2 + 3)) * L(shift::BITS_2 + 4));
    assert_eq(b, L(shift::POWER_OF_TWO + 3), L(shift::TEMP_1) * L(shift::BITS_2 + 3) * L(shift::BITS_2 + 4));
    assert_bool(b, L(shift::IS_SHL));
    assert_bool(b, L(shift::IS_SHR));
    assert_bool(b, L(shift::IS_SRA));
    assert_bool(b, L(shift::IS_SHL) + L(shift::IS_SHR) + L(shift::IS_SRA));
}

// alu_u32/src/lt/stark.rs:21-168
template <class AB> VAIR_HD void eval_lt(AB& b) {
    using E = typename AB::Expr;
    auto L = [&](int c) { return b.main(c, false); };
    E one = b.constant(1);
    auto bits_sum = [&](int first, int n) {
        E acc = L(first) * b.constant(1);
        for (int i = 1; i < n; i++) acc = acc + L(first + i) * b.constant(1u << i);
        return acc;
    };
    E bit_comp = bits_sum(lt::BITS, 9);
    E flag_sum = L(lt::BYTE_FLAG) + L(lt::BYTE_FLAG + 1) + L(lt::BYTE_FLAG + 2) + L(lt::BYTE_FLAG + 3);
    assert_bool(b, flag_sum);
    when_ne(b, L(lt::BYTE_FLAG), one).assert_eq(L(lt::INPUT_1 + 0), L(lt::INPUT_2 + 0));
    when_ne(b, L(lt::BYTE_FLAG) + L(lt::BYTE_FLAG + 1), one).assert_eq(L(lt::INPUT_1 + 1), L(lt::INPUT_2 + 1));
    when_ne(b, L(lt::BYTE_FLAG) + L(lt::BYTE_FLAG + 1) + L(lt::BYTE_FLAG + 2), one).assert_eq(L(lt::INPUT_1 + 2), L(lt::INPUT_2 + 2));
    when_ne(b, flag_sum, one).assert_eq(L(lt::INPUT_1 + 3), L(lt::INPUT_2 + 3));
    when_ne(b, flag_sum, one).assert_eq(bit_comp, b.constant(0));
    for (int i = 0; i < 4; i++) {
        when(b, L(lt::BYTE_FLAG + i)).assert_eq(b.constant(256) + L(lt::INPUT_1 + i) - L(lt::INPUT_2 + i), bit_comp);
        when(b, L(lt::BYTE_FLAG + i)).assert_eq((L(lt::INPUT_1 + i) - L(lt::INPUT_2 + i)) * L(lt::DIFF_INV), one);
        assert_bool(b, L(lt::BYTE_FLAG + i));
    }
    E top_comp_1 = bits_sum(lt::TOP_BITS_1, 8), top_comp_2 = bits_sum(lt::TOP_BITS_2, 8);
    assert_eq(b, top_comp_1, L(lt::INPUT_1 + 0));
    assert_eq(b, top_comp_2, L(lt::INPUT_2 + 0));
    E is_signed = L(lt::IS_SLT) + L(lt::IS_SLE);
    E is_unsigned = one - is_signed;
    E same_sign = one - L(lt::DIFFERENT_SIGNS);
    E are_equal = one - flag_sum;
    when(b, is_unsigned).assert_zero(L(lt::DIFFERENT_SIGNS));
    when(b, is_signed).when_ne(L(lt::TOP_BITS_1 + 7), L(lt::TOP_BITS_2 + 7)).assert_eq(L(lt::DIFFERENT_SIGNS), one);
    when(b, L(lt::DIFFERENT_SIGNS)).assert_eq(L(lt::BYTE_FLAG), one);
    when(b, L(lt::DIFFERENT_SIGNS)).assert_eq(L(lt::TOP_BITS_1 + 7) + L(lt::TOP_BITS_2 + 7), one);
    assert_bool(b, L(lt::IS_LT));
    assert_bool(b, L(lt::IS_LTE));
    assert_bool(b, L(lt::IS_SLT));
    assert_bool(b, L(lt::IS_SLE));
    assert_bool(b, L(lt::IS_LT) + L(lt::IS_LTE) + L(lt::IS_SLT) + L(lt::IS_SLE));
    when(b, L(lt::BITS + 8)).when(is_unsigned + same_sign).assert_zero(L(lt::OUTPUT));
    when(b, L(lt::BITS + 8)).when(L(lt::DIFFERENT_SIGNS)).assert_one(L(lt::OUTPUT));
    when_ne(b, L(lt::BITS + 8) + are_equal, one).when(is_unsigned + same_sign).assert_one(L(lt::OUTPUT));
    when_ne(b, L(lt::BITS + 8) + are_equal, one).when(L(lt::DIFFERENT_SIGNS)).assert_zero(L(lt::OUTPUT));
    when(b, are_equal).when(L(lt::IS_LTE) + L(lt::IS_SLE)).assert_one(L(lt::OUTPUT));
    when(b, are_equal).when(L(lt::IS_LT) + L(lt::IS_SLT)).assert_zero(L(lt::OUTPUT));
    for (int i = 0; i < 9; i++) assert_bool(b, L(lt::BITS + i));
    for (int i = 0; i < 8; i++) assert_bool(b, L(lt::TOP_BITS_1 + i));
    for (int i = 0; i < 8; i++) assert_bool(b, L(lt::TOP_BITS_2 + i));
}

// alu_u32/src/com/stark.rs:21-49
template <class AB> VAIR_HD void eval_com(AB& b) {
    using E = typename AB::Expr;
    auto L = [&](int c) { return b.main(c, false); };
    E one = b.constant(1);
    E acc = (L(com::INPUT_1) - L(com::INPUT_2)) * (L(com::INPUT_1) - L(com::INPUT_2));
    for (int i = 1; i < 4; i++) acc = acc + (L(com::INPUT_1 + i) - L(com::INPUT_2 + i)) * (L(com::INPUT_1 + i) - L(com::INPUT_2 + i));
    assert_eq(b, L(com::DIFF), acc);
    assert_bool(b, L(com::NOT_EQUAL));
    assert_eq(b, L(com::NOT_EQUAL), L(com::DIFF) * L(com::DIFF_INV));
    E equal = one - L(com::NOT_EQUAL);
    b.assert_zero(equal * L(com::DIFF));
    assert_bool(b, L(com::IS_NE));
    assert_bool(b, L(com::IS_EQ));
    assert_bool(b, L(com::IS_NE) + L(com::IS_EQ));
    assert_eq(b, L(com::OUTPUT), L(com::IS_NE) * L(com::NOT_EQUAL) + L(com::IS_EQ) * (one - L(com::NOT_EQUAL)));
}

// alu_u32/src/bitwise/stark.rs:22-74
template <class AB> VAIR_HD void eval_bitwise(AB& b) {
    using E = typename AB::Expr;
    auto L = [&](int c) { return b.main(c, false); };
    for (int i = 0; i < 4; i++) {
        int b1 = bitwise::BITS_1 + 8 * i, b2 = bitwise::BITS_2 + 8 * i;
        E byte_1 = L(b1) * b.constant(1), byte_2 = L(b2) * b.constant(1), band = L(b1) * L(b2) * b.constant(1);
        for (int k = 1; k < 8; k++) {
            byte_1 = byte_1 + L(b1 + k) * b.constant(1u << k);
            byte_2 = byte_2 + L(b2 + k) * b.constant(1u << k);
            band = band + L(b1 + k) * L(b2 + k) * b.constant(1u << k);
        }
        assert_eq(b, L(bitwise::INPUT_1 + i), byte_1);
        assert_eq(b, L(bitwise::INPUT_2 + i), byte_2);
        E bor = byte_1 + byte_2 - band;
        E bxor = byte_1 + byte_2 - b.constant(2) * band;
        when(b, L(bitwise::IS_AND)).assert_eq(band, L(bitwise::OUTPUT + i));
        when(b, L(bitwise::IS_OR)).assert_eq(bor, L(bitwise::OUTPUT + i));
        when(b, L(bitwise::IS_XOR)).assert_eq(bxor, L(bitwise::OUTPUT + i));
        for (int k = 0; k < 8; k++) assert_bool(b, L(b1 + k));
        for (int k = 0; k < 8; k++) assert_bool(b, L(b2 + k));
    }
    assert_bool(b, L(bitwise::IS_AND));
    assert_bool(b, L(bitwise::IS_OR));
    assert_bool(b, L(bitwise::IS_XOR));
    assert_bool(b, L(bitwise::IS_AND) + L(bitwise::IS_OR) + L(bitwise::IS_XOR));
}

// output/src/stark.rs:21-39
template <class AB> VAIR_HD void eval_output(AB& b) {
    auto L = [&](int c) { return b.main(c, false); };
    auto N = [&](int c) { return b.main(c, true); };
    when_transition(b).assert_eq(L(output::DIFF), N(output::CLK) - L(output::CLK));
    when_transition(b).assert_eq(N(output::COUNTER), L(output::COUNTER) + b.constant(1));
    when(b, L(output::IS_REAL)).assert_eq(L(output::OPCODE), b.constant(OP_WRITE));
}

// static_data/src/stark.rs:25-37
template <class AB> VAIR_HD void eval_static_data(AB& b) {
    auto L = [&](int c) { return b.main(c, false); };
    auto N = [&](int c) { return b.main(c, true); };
    when_transition(b).when(L(static_data::IS_REAL) * N(static_data::IS_REAL))
        .assert_eq(N(static_data::ADDR), L(static_data::ADDR) + b.constant(1) + b.constant(1) + b.constant(1) + b.constant(1));
}

// Empty AIRs: program (program/src/stark.rs:14), mem (memory/src/stark.rs:22-78, all commented out),
// div (alu_u32/src/div/stark.rs:18-20), range (range/src/stark.rs:12-14).
template <class AB> VAIR_HD void eval_chip(int chip, AB& b) {
    switch (chip) {
        case CHIP_CPU: eval_cpu(b); break;
        case CHIP_ADD: eval_add(b); break;
        case CHIP_SUB: eval_sub(b); break;
        case CHIP_MUL: eval_mul(b); break;
        case CHIP_SHIFT: eval_shift(b); break;
        case CHIP_LT: eval_lt(b); break;
        case CHIP_COM: eval_com(b); break;
        case CHIP_BITWISE: eval_bitwise(b); break;
        case CHIP_OUTPUT: eval_output(b); break;
        case CHIP_STATIC_DATA: eval_static_data(b); break;
        default: break;  // program, mem, div, range
    }
}

// ---- interactions (Chip::all_interactions order: local sends, local receives, global sends,
//      global receives — machine/src/chip.rs:40-63) --------------------------------------------
// Stated ONCE as a visit over (begin, field*, end(count)) calls, so that two visitors read the same definition: the host collector below
// (chip_interactions: the vair::Interaction lists the machine description, the permutation-trace kernels and the FFI use) and the device
// evaluator of the per-point quotient kernel (kernels/quotient.hip), for which every field becomes a column load at a compile-time column.
// Lin = a VirtualPairCol over main columns as Valida's chips use them: k + sum_i w_i * main[col_i], at most four terms.
struct Lin {
    int n;
    int col[4];
    uint32_t w[4];
    uint32_t k;
};
VAIR_HD constexpr Lin lin_col(int c) { return Lin{1, {c, 0, 0, 0}, {1, 0, 0, 0}, 0}; }                                      // VirtualPairCol::single_main
VAIR_HD constexpr Lin lin_const(uint32_t k) { return Lin{0, {0, 0, 0, 0}, {0, 0, 0, 0}, k}; }                               // VirtualPairCol::constant
VAIR_HD constexpr Lin lin_sum2(int a, int b) { return Lin{2, {a, b, 0, 0}, {1, 1, 0, 0}, 0}; }                              // VirtualPairCol::sum_main
VAIR_HD constexpr Lin lin_sum3(int a, int b, int c) { return Lin{3, {a, b, c, 0}, {1, 1, 1, 0}, 0}; }
VAIR_HD constexpr Lin lin_w2(int a, uint32_t wa, int b, uint32_t wb) { return Lin{2, {a, b, 0, 0}, {wa, wb, 0, 0}, 0}; }  // VirtualPairCol::new_main
VAIR_HD constexpr Lin lin_w3(int a, uint32_t wa, int b, uint32_t wb, int c, uint32_t wc) { return Lin{3, {a, b, c, 0}, {wa, wb, wc, 0}, 0}; }
VAIR_HD constexpr Lin lin_w4(int a, uint32_t wa, int b, uint32_t wb, int c, uint32_t wc, int d, uint32_t wd) { return Lin{4, {a, b, c, d}, {wa, wb, wc, wd}, 0}; }

// V: void begin(bool is_send, int bus_index); void field(const Lin&); void end(const Lin& count);
template <class V> VAIR_HD void visit_word(V& v, int first) {
    for (int i = 0; i < 4; i++) v.field(lin_col(first + i));
}
template <class V> VAIR_HD void visit_alu(V& v, bool is_send, const Lin& opcode, int in1, int in2, int outc, const Lin& count) {
    v.begin(is_send, BUS_GENERAL);
    v.field(opcode);
    visit_word(v, in1); visit_word(v, in2); visit_word(v, outc);
    v.end(count);
}
template <class V> VAIR_HD void visit_interactions(int chip, V& v) {
    switch (chip) {
        case CHIP_CPU: {  // cpu/src/lib.rs:99-159
            for (int i = 0; i < 3; i++) {
                v.begin(true, BUS_MEM);
                v.field(lin_col(cpu::ch(i, cpu::CH_IS_READ))); v.field(lin_col(cpu::CLK)); v.field(lin_col(cpu::ch(i, cpu::CH_ADDR))); v.field(lin_const(0));
                visit_word(v, cpu::ch(i, cpu::CH_VALUE));
                v.end(lin_col(cpu::ch(i, cpu::CH_USED)));
            }
            v.begin(true, BUS_GENERAL);
            v.field(lin_col(cpu::OPCODE));
            for (int i = 0; i < 3; i++) visit_word(v, cpu::ch(i, cpu::CH_VALUE));
            v.field(lin_col(cpu::CLK_OR_ZERO));
            v.end(lin_col(cpu::IS_BUS_OP));
            break;
        }
        case CHIP_PROGRAM: break;  // program/src/lib.rs:50-68 (bus commented out)
        case CHIP_MEM:  // memory/src/lib.rs:216-233
            v.begin(false, BUS_MEM);
            v.field(lin_col(mem::IS_READ)); v.field(lin_col(mem::CLK)); v.field(lin_col(mem::ADDR)); v.field(lin_col(mem::IS_STATIC_INITIAL));
            visit_word(v, mem::VALUE);
            v.end(lin_sum2(mem::IS_READ, mem::IS_WRITE));
            break;
        case CHIP_ADD:    // alu_u32/src/add/mod.rs:53-87
        case CHIP_SUB: {  // alu_u32/src/sub/mod.rs:53-87
            const int outc = chip == CHIP_ADD ? (int)add::OUTPUT : (int)sub::OUTPUT, is_real = chip == CHIP_ADD ? (int)add::IS_REAL : (int)sub::IS_REAL;
            for (int i = 0; i < 4; i++) {
                v.begin(true, BUS_RANGE);
                v.field(lin_col(outc + i));
                v.end(lin_col(is_real));
            }
            visit_alu(v, false, lin_const(chip == CHIP_ADD ? OP_ADD32 : OP_SUB32), 0, 4, outc, lin_col(is_real));
            break;
        }
        case CHIP_MUL:  // alu_u32/src/mul/mod.rs:68-96
            visit_alu(v, false, lin_w3(mul::IS_MUL, OP_MUL32, mul::IS_MULHS, OP_MULHS32, mul::IS_MULHU, OP_MULHU32), mul::INPUT_1, mul::INPUT_2, mul::OUTPUT,
                      lin_sum3(mul::IS_MUL, mul::IS_MULHS, mul::IS_MULHU));
            break;
        case CHIP_DIV:  // alu_u32/src/div/mod.rs:55-80
            visit_alu(v, false, lin_w2(divc::IS_DIV, OP_DIV32, divc::IS_SDIV, OP_SDIV32), divc::INPUT_1, divc::INPUT_2, divc::OUTPUT, lin_sum2(divc::IS_DIV, divc::IS_SDIV));
            break;
        case CHIP_SHIFT:  // alu_u32/src/shift/mod.rs:58-116
            visit_alu(v, true, lin_w3(shift::IS_SHL, OP_MUL32, shift::IS_SHR, OP_DIV32, shift::IS_SRA, OP_SDIV32), shift::INPUT_1, shift::POWER_OF_TWO, shift::OUTPUT,
                      lin_sum3(shift::IS_SHL, shift::IS_SHR, shift::IS_SRA));
            visit_alu(v, false, lin_w3(shift::IS_SHL, OP_SHL32, shift::IS_SHR, OP_SHR32, shift::IS_SRA, OP_SRA32), shift::INPUT_1, shift::INPUT_2, shift::OUTPUT,
                      lin_sum3(shift::IS_SHL, shift::IS_SHR, shift::IS_SRA));
            break;
        case CHIP_LT:  // alu_u32/src/lt/mod.rs:58-85
            v.begin(false, BUS_GENERAL);
            v.field(lin_w4(lt::IS_LT, OP_LT32, lt::IS_LTE, OP_LTE32, lt::IS_SLT, OP_SLT32, lt::IS_SLE, OP_SLE32));
            visit_word(v, lt::INPUT_1); visit_word(v, lt::INPUT_2);
            for (int i = 0; i < 3; i++) v.field(lin_const(0));
            v.field(lin_col(lt::OUTPUT));
            v.end(lin_col(lt::MULTIPLICITY));
            break;
        case CHIP_COM:  // alu_u32/src/com/mod.rs:56-83
            v.begin(false, BUS_GENERAL);
            v.field(lin_w2(com::IS_NE, OP_NE32, com::IS_EQ, OP_EQ32));
            visit_word(v, com::INPUT_1); visit_word(v, com::INPUT_2);
            for (int i = 0; i < 3; i++) v.field(lin_const(0));
            v.field(lin_col(com::OUTPUT));
            v.end(lin_sum2(com::IS_NE, com::IS_EQ));
            break;
        case CHIP_BITWISE:  // alu_u32/src/bitwise/mod.rs:56-82
            visit_alu(v, false, lin_w3(bitwise::IS_AND, OP_AND32, bitwise::IS_OR, OP_OR32, bitwise::IS_XOR, OP_XOR32), bitwise::INPUT_1, bitwise::INPUT_2, bitwise::OUTPUT,
                      lin_sum3(bitwise::IS_AND, bitwise::IS_OR, bitwise::IS_XOR));
            break;
        case CHIP_OUTPUT:  // output/src/lib.rs:117-136
            v.begin(false, BUS_GENERAL);
            v.field(lin_col(output::OPCODE));
            for (int i = 0; i < 12; i++) v.field(i == 3 ? lin_col(output::VALUE) : lin_const(0));
            v.field(lin_col(output::CLK));
            v.end(lin_col(output::IS_REAL));
            break;
        case CHIP_RANGE:  // range/src/lib.rs:46-55
            v.begin(false, BUS_RANGE);
            v.field(lin_col(range::COUNTER));
            v.end(lin_col(range::MULT));
            break;
        case CHIP_STATIC_DATA:  // static_data/src/lib.rs:81-96
            v.begin(true, BUS_MEM);
            v.field(lin_const(0)); v.field(lin_const(0)); v.field(lin_col(static_data::ADDR)); v.field(lin_const(1));
            visit_word(v, static_data::VALUE);
            v.end(lin_col(static_data::IS_REAL));
            break;
        default: break;
    }
}

// the host visitor: the Interaction lists (every interaction of the BasicMachine is on a global bus)
struct InteractionCollector {
    std::vector<Interaction> out;
    static VirtualCol to_vcol(const Lin& f) {
        VirtualCol v;
        for (int i = 0; i < f.n; i++) v.terms.push_back({false, f.col[i], f.w[i]});
        v.constant = f.k;
        return v;
    }
    void begin(bool is_send, int bus_index) {
        Interaction it;
        it.bus_kind = BusKind::Global; it.bus_index = bus_index; it.type = is_send ? InteractionType::GlobalSend : InteractionType::GlobalReceive;
        out.push_back(it);
    }
    void field(const Lin& f) { out.back().fields.push_back(to_vcol(f)); }
    void end(const Lin& count) { out.back().count = to_vcol(count); }
};
inline std::vector<Interaction> chip_interactions(int chip) {
    InteractionCollector c;
    visit_interactions(chip, c);
    return c.out;
}

}  // namespace vchips
