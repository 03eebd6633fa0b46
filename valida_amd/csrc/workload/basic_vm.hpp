// Synthetic-workload generator: a minimal interpreter for the BasicMachine opcodes the benchmark
// programs use, plus each chip's `generate_trace`, producing the host-resident RowMajorMatrix inputs
// that `Machine::prove` starts from.  Trace generation is UPSTREAM of the hot path (SURVEY.md §3.5,
// §7.1 step 1b); it lives here so that bench.py and the tests can build inputs without the oracle.
//
// Restated from (reference file:line):
//   Machine::run / step                  basic/src/lib.rs:127-145, :1066-1188
//   core instructions                    cpu/src/lib.rs:430-881
//   add/sub/lt/bitwise instructions      alu_u32/src/{add,sub,lt,bitwise}/mod.rs
//   MemoryChip read/write/generate_trace memory/src/lib.rs:84-194, :236-262
//   CpuChip generate_trace + padding     cpu/src/lib.rs:79-97, :163-373
//   Add32/Sub32/Lt32/Bitwise32 rows      alu_u32/src/add/mod.rs:91-121, sub/mod.rs:91-117, lt/mod.rs:87-166, bitwise/mod.rs:84-129
//   Mul32 rows + min-length counter rows alu_u32/src/mul/mod.rs:38-66, :107-132; instructions :141-263
//   Div32 / Com32 rows (flags only, as the reference fills them)   alu_u32/src/div/mod.rs:84-103, com/mod.rs:87-103; instructions div/mod.rs:112-189, com/mod.rs:110-196
//   Shift32 rows                         alu_u32/src/shift/mod.rs:118-162; instructions :170-333 (SHL also logs a Mul32, SHR / SRA a Div32 / SDiv32
//                                        with the power of two; SRA logs itself as Operation::Shr32, :325-328 — restated as written)
//   Output rows                          output/src/lib.rs:37-100; WRITE :146-173
//   Word arithmetic                      machine/src/core.rs:129-252 (mulhs zero-extends both operands, :146-158 — restated as written)
//   Range / Program traces + preprocessed range/src/lib.rs:32-44, range/src/stark.rs:22-25,
//                                        program/src/lib.rs:38-48, program/src/stark.rs:22-40
// All values are canonical u32 < p.
#pragma once
#include <algorithm>
#include <array>
#include <climits>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <map>
#include <unordered_map>
#include <vector>
#include "../chips/basic_machine.hpp"

namespace vwork {
using namespace vchips;

constexpr uint32_t P = 2013265921u;
inline uint32_t fmul(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % P); }
inline uint32_t fadd(uint32_t a, uint32_t b) { uint32_t s = a + b; return s >= P ? s - P : s; }
inline uint32_t fsub(uint32_t a, uint32_t b) { return a >= b ? a - b : a + P - b; }
inline uint32_t fpow(uint32_t a, uint64_t e) { uint32_t r = 1; while (e) { if (e & 1) r = fmul(r, a); a = fmul(a, a); e >>= 1; } return r; }
inline uint32_t finv(uint32_t a) { return fpow(a, P - 2); }
inline uint32_t from_i32(int32_t x) {  // Operands::from_i32_slice
    uint32_t a = (uint32_t)(x < 0 ? -(int64_t)x : (int64_t)x) % P;
    return (x < 0 && a) ? P - a : a;
}

struct InstructionWord { uint32_t opcode; int32_t ops[5]; };

struct Word { uint8_t b[4]; };  // big-endian (machine/src/core.rs:9)
inline Word word_of(uint32_t v) { return Word{{(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v}}; }
inline uint32_t u32_of(const Word& w) { return ((uint32_t)w.b[0] << 24) | ((uint32_t)w.b[1] << 16) | ((uint32_t)w.b[2] << 8) | w.b[3]; }
inline uint32_t word_reduce(const Word& w) {  // Word::reduce in the field
    return (uint32_t)((((uint64_t)w.b[0] << 24) + ((uint64_t)w.b[1] << 16) + ((uint64_t)w.b[2] << 8) + w.b[3]) % P);
}

struct MemOp { uint32_t clk; uint32_t addr; Word value; bool is_write; };
enum class CpuOp { Store32, Load32, Jal, Jalv, Beq, Bne, Imm32, Bus, BusLeftImm, Stop, LoadFp };
struct CpuRecord { CpuOp op; bool has_imm; Word imm; uint32_t pc, fp; InstructionWord instr; };
struct AluOp { uint32_t opcode; Word a, b, c; };

struct RowMajor {
    size_t height = 0, width = 0;
    std::vector<uint32_t> v;
    RowMajor() {}
    RowMajor(size_t h, size_t w) : height(h), width(w), v(h * w, 0) {}
    uint32_t* row(size_t r) { return &v[r * width]; }
};

inline size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }

struct BasicVm {
    std::vector<InstructionWord> rom;
    uint32_t pc = 0, fp = 0, clock = 0;
    std::unordered_map<uint32_t, uint32_t> cells;  // addr -> u32 value
    std::vector<MemOp> mem_ops;                    // in (clk, issue) order
    std::vector<CpuRecord> cpu_ops;
    std::vector<AluOp> add_ops, sub_ops, lt_ops, bitwise_ops;
    std::vector<AluOp> mul_ops, div_ops, shift_ops, com_ops;       // Mul32Chip / Div32Chip / Shift32Chip / Com32Chip ::operations
    std::vector<std::pair<uint32_t, uint8_t>> output_values;       // OutputChip::values: (clk, byte)
    std::vector<uint32_t> program_counts;
    uint32_t range_counts[256] = {0};
    std::map<uint32_t, Word> static_cells;  // StaticDataChip::cells, copied into memory by initialize_memory (static_data/src/lib.rs:28-32)
    // machine.static_data_mut().write(addr, value) before run (basic/tests/test_static_data.rs:57-58)
    void write_static(uint32_t addr, Word w) { static_cells[addr] = w; cells[addr] = u32_of(w); }

    explicit BasicVm(std::vector<InstructionWord> program, uint32_t initial_fp = 0x1000) : rom(std::move(program)), fp(initial_fp) {
        program_counts.assign(rom.size(), 0);
    }
    Word read(uint32_t addr) {
        auto it = cells.find(addr);
        if (it == cells.end()) throw std::runtime_error("memory chip: read before write: " + std::to_string(addr) + " (pc = " + std::to_string(pc) + ")");
        Word w = word_of(it->second);
        mem_ops.push_back({clock, addr, w, false});
        return w;
    }
    void write(uint32_t addr, Word w) {
        mem_ops.push_back({clock, addr, w, true});
        cells[addr] = u32_of(w);
    }
    void range_check(const Word& w) { for (int i = 0; i < 4; i++) range_counts[w.b[i]]++; }
    void push(CpuOp op, const InstructionWord& iw, uint32_t pc0, uint32_t fp0, bool has_imm = false, Word imm = Word{{0, 0, 0, 0}}) {
        cpu_ops.push_back({op, has_imm, imm, pc0, fp0, iw});
        clock++;
    }
    // One Machine::step (basic/src/lib.rs:1066-1188).  Returns true on STOP.
    bool step() {
        if (pc >= rom.size()) throw std::runtime_error("pc out of range");
        const InstructionWord iw = rom[pc];
        const int32_t* o = iw.ops;
        const uint32_t pc0 = pc, fp0 = fp;
        auto at = [&](int32_t off) { return (uint32_t)((int32_t)fp0 + off); };
        auto second_operand = [&](bool& has_imm, Word& imm) {
            if (o[4] == 1) { imm = word_of((uint32_t)o[2]); has_imm = true; return imm; }
            return read(at(o[2]));
        };
        switch (iw.opcode) {
            case OP_IMM32: {
                write(at(o[0]), Word{{(uint8_t)o[1], (uint8_t)o[2], (uint8_t)o[3], (uint8_t)o[4]}});
                pc += 1; push(CpuOp::Imm32, iw, pc0, fp0); break;
            }
            case OP_LOADFP: {
                write(at(o[0]), word_of(at(o[1])));
                pc += 1; push(CpuOp::LoadFp, iw, pc0, fp0); break;
            }
            case OP_LOAD32: {
                Word a2 = read(at(o[2]));
                Word cell = read(u32_of(a2));
                write(at(o[0]), cell);
                pc += 1; push(CpuOp::Load32, iw, pc0, fp0); break;
            }
            case OP_STORE32: {
                Word waddr = read(at(o[1]));
                Word cell = read(at(o[2]));
                write(u32_of(waddr), cell);
                pc += 1; push(CpuOp::Store32, iw, pc0, fp0); break;
            }
            case OP_JAL: {
                write(at(o[0]), word_of(BYTES_PER_INSTR * (pc0 + 1)));
                pc = (uint32_t)o[1] / BYTES_PER_INSTR;
                fp = at(o[2]);
                push(CpuOp::Jal, iw, pc0, fp0); break;
            }
            case OP_JALV: {
                write(at(o[0]), word_of(BYTES_PER_INSTR * (pc0 + 1)));
                pc = u32_of(read(at(o[1]))) / BYTES_PER_INSTR;
                fp = (uint32_t)((int32_t)fp0 + (int32_t)u32_of(read(at(o[2]))));
                push(CpuOp::Jalv, iw, pc0, fp0); break;
            }
            case OP_BEQ: case OP_BNE: {
                bool has_imm = false; Word imm{{0, 0, 0, 0}};
                Word c1 = read(at(o[1]));
                Word c2 = second_operand(has_imm, imm);
                bool eq = u32_of(c1) == u32_of(c2);
                bool taken = iw.opcode == OP_BEQ ? eq : !eq;
                pc = taken ? (uint32_t)o[0] / BYTES_PER_INSTR : pc0 + 1;
                push(iw.opcode == OP_BEQ ? CpuOp::Beq : CpuOp::Bne, iw, pc0, fp0, has_imm, imm); break;
            }
            case OP_STOP: push(CpuOp::Stop, iw, pc0, fp0); break;
            case OP_ADD32: case OP_SUB32: case OP_AND32: case OP_OR32: case OP_XOR32: {
                bool has_imm = false; Word imm{{0, 0, 0, 0}};
                Word bw = read(at(o[1]));
                Word cw = second_operand(has_imm, imm);
                uint32_t bv = u32_of(bw), cv = u32_of(cw), av;
                switch (iw.opcode) {
                    case OP_ADD32: av = bv + cv; break;
                    case OP_SUB32: av = bv - cv; break;
                    case OP_AND32: av = bv & cv; break;
                    case OP_OR32: av = bv | cv; break;
                    default: av = bv ^ cv; break;
                }
                Word aw = word_of(av);
                write(at(o[0]), aw);
                AluOp rec{iw.opcode, aw, bw, cw};
                if (iw.opcode == OP_ADD32) add_ops.push_back(rec);
                else if (iw.opcode == OP_SUB32) sub_ops.push_back(rec);
                else bitwise_ops.push_back(rec);
                pc += 1; push(CpuOp::Bus, iw, pc0, fp0, has_imm, imm);
                if (iw.opcode == OP_ADD32 || iw.opcode == OP_SUB32) range_check(aw);
                break;
            }
            case OP_LT32: case OP_LTE32: case OP_SLT32: case OP_SLE32: {  // alu_u32/src/lt/mod.rs:168-218
                bool has_imm = false; Word imm{{0, 0, 0, 0}};
                bool left_imm = o[3] == 1;
                Word s1;
                if (left_imm) { s1 = word_of((uint32_t)o[1]); imm = s1; has_imm = true; } else s1 = read(at(o[1]));
                Word s2 = second_operand(has_imm, imm);
                uint32_t u1 = u32_of(s1), u2 = u32_of(s2);
                bool r;
                switch (iw.opcode) {
                    case OP_LT32: r = u1 < u2; break;
                    case OP_LTE32: r = u1 <= u2; break;
                    case OP_SLT32: r = (int32_t)u1 < (int32_t)u2; break;
                    default: r = (int32_t)u1 <= (int32_t)u2; break;
                }
                Word dst = word_of(r ? 1 : 0);
                write(at(o[0]), dst);
                lt_ops.push_back({iw.opcode, dst, s1, s2});
                pc += 1; push(left_imm ? CpuOp::BusLeftImm : CpuOp::Bus, iw, pc0, fp0, has_imm, imm);
                break;
            }
            case OP_MUL32: case OP_MULHS32: case OP_MULHU32: case OP_DIV32: case OP_SDIV32: {  // alu_u32/src/mul/mod.rs:141-263, div/mod.rs:112-189
                bool has_imm = false; Word imm{{0, 0, 0, 0}};
                Word bw = read(at(o[1]));
                Word cw = second_operand(has_imm, imm);
                const uint32_t bv = u32_of(bw), cv = u32_of(cw);
                uint32_t av;
                switch (iw.opcode) {
                    case OP_MUL32: av = bv * cv; break;
                    case OP_MULHS32:  // core.rs:146-158: `bu32 as i64` ZERO-extends, so this is the unsigned high word
                    case OP_MULHU32: av = (uint32_t)(((uint64_t)bv * cv) >> 32); break;
                    case OP_DIV32:
                        if (!cv) throw std::runtime_error("workload VM: division by zero (the reference panics)");
                        av = bv / cv; break;
                    default:
                        if (!cv || ((int32_t)bv == INT32_MIN && (int32_t)cv == -1)) throw std::runtime_error("workload VM: signed division overflow (the reference panics)");
                        av = (uint32_t)((int32_t)bv / (int32_t)cv); break;
                }
                Word aw = word_of(av);
                write(at(o[0]), aw);
                const bool is_mul = iw.opcode == OP_MUL32 || iw.opcode == OP_MULHS32 || iw.opcode == OP_MULHU32;
                (is_mul ? mul_ops : div_ops).push_back({iw.opcode, aw, bw, cw});
                pc += 1; push(CpuOp::Bus, iw, pc0, fp0, has_imm, imm);
                range_check(aw);
                break;
            }
            case OP_SHL32: case OP_SHR32: case OP_SRA32: {  // alu_u32/src/shift/mod.rs:170-333
                bool has_imm = false; Word imm{{0, 0, 0, 0}};
                Word bw = read(at(o[1]));
                Word cw = second_operand(has_imm, imm);
                const uint32_t bv = u32_of(bw), cv = u32_of(cw);
                if (cv >= 32) throw std::runtime_error("workload VM: shift amount >= 32 (overflow in the reference's u32 shift)");
                const uint32_t av = iw.opcode == OP_SHL32 ? bv << cv : iw.opcode == OP_SHR32 ? bv >> cv : (uint32_t)((int32_t)bv >> cv);
                Word aw = word_of(av), dw = word_of(1u << cv);
                write(at(o[0]), aw);
                // the "receive" that matches the shift chip's send on the general bus
                if (iw.opcode == OP_SHL32) mul_ops.push_back({OP_MUL32, aw, bw, dw});
                else div_ops.push_back({iw.opcode == OP_SHR32 ? (uint32_t)OP_DIV32 : (uint32_t)OP_SDIV32, aw, bw, dw});
                shift_ops.push_back({iw.opcode == OP_SHL32 ? (uint32_t)OP_SHL32 : (uint32_t)OP_SHR32, aw, bw, cw});  // SRA is logged as Shr32 (shift/mod.rs:325-328)
                pc += 1; push(CpuOp::Bus, iw, pc0, fp0, has_imm, imm);
                break;
            }
            case OP_NE32: case OP_EQ32: {  // alu_u32/src/com/mod.rs:110-196
                bool has_imm = false; Word imm{{0, 0, 0, 0}};
                Word s1 = read(at(o[1]));
                Word s2 = second_operand(has_imm, imm);
                const bool ne = u32_of(s1) != u32_of(s2);
                Word dst = word_of((iw.opcode == OP_NE32 ? ne : !ne) ? 1 : 0);
                write(at(o[0]), dst);
                com_ops.push_back({iw.opcode, dst, s1, s2});
                pc += 1; push(CpuOp::Bus, iw, pc0, fp0, has_imm, imm);
                break;
            }
            case OP_WRITE: {  // output/src/lib.rs:146-173: one byte (the least significant of the word at fp + b) to the output tape
                if (o[4] != 1 || o[2] != 0) throw std::runtime_error("workload VM: WRITE needs is_imm = 1 and c = 0 (asserted by the reference)");
                Word bw = read(at(o[1]));
                output_values.push_back({clock, bw.b[3]});
                pc += 1; push(CpuOp::Bus, iw, pc0, fp0);
                break;
            }
            default: throw std::runtime_error("workload VM: unsupported opcode " + std::to_string(iw.opcode));
        }
        program_counts[pc0]++;  // read_word(pc) (basic/src/lib.rs:1179)
        return iw.opcode == OP_STOP;
    }
    // Machine::run (basic/src/lib.rs:127-145)
    void run(uint64_t max_cycles = (uint64_t)1 << 32) {
        uint64_t n = 0;
        while (!step()) if (++n > max_cycles) throw std::runtime_error("workload VM: cycle limit exceeded");
        size_t pad = next_pow2(clock) - clock;
        program_counts[pc] += (uint32_t)pad;  // padded STOP reads
    }

    // ---- per-chip generate_trace ---------------------------------------------------------------
    RowMajor cpu_trace() const {
        size_t n = cpu_ops.size(), N = next_pow2(n);
        RowMajor t(N, cpu::NUM_COLS);
        // memory ops grouped by clk, in issue order
        std::vector<size_t> first(n + 1, 0);
        for (auto& m : mem_ops) first[m.clk + 1]++;
        for (size_t i = 0; i < n; i++) first[i + 1] += first[i];
        std::vector<uint32_t> diff(n);
        for (size_t i = 0; i < n; i++) {
            const CpuRecord& rec = cpu_ops[i];
            uint32_t* r = t.row(i);
            r[cpu::PC] = rec.pc; r[cpu::FP] = rec.fp % P; r[cpu::CLK] = (uint32_t)i;
            r[cpu::OPCODE] = rec.instr.opcode;
            for (int k = 0; k < 5; k++) r[cpu::OPERAND_A + k] = from_i32(rec.instr.ops[k]);
            auto set_imm = [&](bool left) {
                if (!rec.has_imm) return;
                r[left ? cpu::IS_LEFT_IMM_OP : cpu::IS_IMM_OP] = 1;
                int chn = left ? 0 : 1;
                for (int k = 0; k < 4; k++) r[cpu::ch(chn, cpu::CH_VALUE) + k] = rec.imm.b[k];
                r[left ? cpu::OPERAND_B : cpu::OPERAND_C] = word_reduce(rec.imm);
            };
            switch (rec.op) {
                case CpuOp::Store32: r[cpu::IS_STORE] = 1; break;
                case CpuOp::Load32: r[cpu::IS_LOAD] = 1; break;
                case CpuOp::Jal: r[cpu::IS_JAL] = 1; break;
                case CpuOp::Jalv: r[cpu::IS_JALV] = 1; break;
                case CpuOp::Beq: r[cpu::IS_BEQ] = 1; set_imm(false); break;
                case CpuOp::Bne: r[cpu::IS_BNE] = 1; set_imm(false); break;
                case CpuOp::Imm32: r[cpu::IS_IMM32] = 1; break;
                case CpuOp::Bus: r[cpu::IS_BUS_OP] = 1; set_imm(false); break;
                case CpuOp::BusLeftImm: r[cpu::IS_BUS_OP] = 1; set_imm(true); break;
                case CpuOp::Stop: r[cpu::IS_STOP] = 1; break;
                case CpuOp::LoadFp: r[cpu::IS_LOADFP] = 1; break;
            }
            // set_memory_channel_values (cpu/src/lib.rs:253-296)
            r[cpu::ch(0, cpu::CH_IS_READ)] = 1; r[cpu::ch(1, cpu::CH_IS_READ)] = 1; r[cpu::ch(2, cpu::CH_IS_READ)] = 0;
            bool is_left_imm = r[cpu::IS_LEFT_IMM_OP] == 1, is_first_read = true;
            for (size_t k = first[i]; k < first[i + 1]; k++) {
                const MemOp& m = mem_ops[k];
                int chn = m.is_write ? 2 : ((is_first_read && !is_left_imm) ? 0 : 1);
                if (!m.is_write && chn == 0) is_first_read = false;
                r[cpu::ch(chn, cpu::CH_USED)] = 1;
                r[cpu::ch(chn, cpu::CH_ADDR)] = m.addr % P;
                for (int b = 0; b < 4; b++) r[cpu::ch(chn, cpu::CH_VALUE) + b] = m.value.b[b];
            }
            // compute_word_diffs (cpu/src/lib.rs:298-330)
            uint32_t d = 0;
            for (int b = 0; b < 4; b++) {
                uint32_t x = fsub(r[cpu::ch(0, cpu::CH_VALUE) + b], r[cpu::ch(1, cpu::CH_VALUE) + b]);
                d = fadd(d, fmul(x, x));
            }
            diff[i] = d;
        }
        for (size_t i = 0; i < n; i++) {
            uint32_t* r = t.row(i);
            r[cpu::DIFF] = diff[i];
            r[cpu::DIFF_INV] = diff[i] ? finv(diff[i]) : 0;
            r[cpu::NOT_EQUAL] = diff[i] ? 1 : 0;
        }
        // pad_to_power_of_two (cpu/src/lib.rs:332-373)
        const uint32_t* last = t.row(n - 1);
        uint32_t lpc = last[cpu::PC], lfp = last[cpu::FP], lclk = last[cpu::CLK];
        for (size_t i = n; i < N; i++) {
            uint32_t* r = t.row(i);
            r[cpu::PC] = lpc; r[cpu::FP] = lfp; r[cpu::CLK] = fadd(lclk, (uint32_t)(i - n + 1) % P);
            r[cpu::IS_STOP] = 1; r[cpu::OPCODE] = OP_STOP;
            r[cpu::ch(0, cpu::CH_IS_READ)] = 1; r[cpu::ch(1, cpu::CH_IS_READ)] = 1;
        }
        return t;
    }
    RowMajor mem_trace() const {  // memory/src/lib.rs:143-194 (no static data in these workloads)
        std::vector<MemOp> ops = mem_ops;
        std::stable_sort(ops.begin(), ops.end(), [](const MemOp& a, const MemOp& b) { return a.addr != b.addr ? a.addr < b.addr : a.clk < b.clk; });
        const size_t n0 = static_cells.size(), n = ops.size(), N = next_pow2(n0 + n);
        RowMajor t(N, mem::NUM_COLS);
        size_t k = 0;
        for (auto& kv : static_cells) {  // static_data_to_row (memory/src/lib.rs:265-284), ascending address
            uint32_t* r = t.row(k);
            r[mem::IS_STATIC_INITIAL] = 1; r[mem::COUNTER] = (uint32_t)k; r[mem::ADDR] = kv.first % P; r[mem::IS_WRITE] = 1;
            for (int b = 0; b < 4; b++) r[mem::VALUE + b] = kv.second.b[b];
            k++;
        }
        for (size_t i = 0; i < n; i++) {
            uint32_t* r = t.row(n0 + i);
            r[mem::CLK] = ops[i].clk; r[mem::COUNTER] = (uint32_t)(n0 + i); r[mem::ADDR] = ops[i].addr % P;
            for (int b = 0; b < 4; b++) r[mem::VALUE + b] = ops[i].value.b[b];
            r[ops[i].is_write ? mem::IS_WRITE : mem::IS_READ] = 1;
        }
        return t;
    }
    static RowMajor alu_rows(const std::vector<AluOp>& ops, int width) {
        return RowMajor(next_pow2(ops.size()), width);  // pad_to_power_of_two: 0 ops -> 1 zero row
    }
    RowMajor add_trace() const {
        RowMajor t = alu_rows(add_ops, add::NUM_COLS);
        for (size_t i = 0; i < add_ops.size(); i++) {
            uint32_t* r = t.row(i); const AluOp& op = add_ops[i];
            for (int k = 0; k < 4; k++) { r[add::INPUT_1 + k] = op.b.b[k]; r[add::INPUT_2 + k] = op.c.b[k]; r[add::OUTPUT + k] = op.a.b[k]; }
            uint32_t c1 = 0, c2 = 0;
            if ((uint32_t)op.b.b[3] + op.c.b[3] > 255) { c1 = 1; r[add::CARRY] = 1; }
            if ((uint32_t)op.b.b[2] + op.c.b[2] + c1 > 255) { c2 = 1; r[add::CARRY + 1] = 1; }
            if ((uint32_t)op.b.b[1] + op.c.b[1] + c2 > 255) r[add::CARRY + 2] = 1;
            r[add::IS_REAL] = 1;
        }
        return t;
    }
    RowMajor sub_trace() const {
        RowMajor t = alu_rows(sub_ops, sub::NUM_COLS);
        for (size_t i = 0; i < sub_ops.size(); i++) {
            uint32_t* r = t.row(i); const AluOp& op = sub_ops[i];
            for (int k = 0; k < 4; k++) { r[sub::INPUT_1 + k] = op.b.b[k]; r[sub::INPUT_2 + k] = op.c.b[k]; r[sub::OUTPUT + k] = op.a.b[k]; }
            if (op.b.b[3] < op.c.b[3]) r[sub::BORROW] = 1;      // reference witness as written (sub/mod.rs:104-112)
            if (op.b.b[2] < op.c.b[2]) r[sub::BORROW + 1] = 1;
            if (op.b.b[1] < op.c.b[1]) r[sub::BORROW + 2] = 1;
            r[sub::IS_REAL] = 1;
        }
        return t;
    }
    RowMajor lt_trace() const {
        RowMajor t = alu_rows(lt_ops, lt::NUM_COLS);
        for (size_t i = 0; i < lt_ops.size(); i++) {
            uint32_t* r = t.row(i); const AluOp& op = lt_ops[i];
            bool is_signed = op.opcode == OP_SLT32 || op.opcode == OP_SLE32;
            r[op.opcode == OP_LT32 ? lt::IS_LT : op.opcode == OP_LTE32 ? lt::IS_LTE : op.opcode == OP_SLT32 ? lt::IS_SLT : lt::IS_SLE] = 1;
            for (int k = 0; k < 4; k++) { r[lt::INPUT_1 + k] = op.b.b[k]; r[lt::INPUT_2 + k] = op.c.b[k]; }
            r[lt::OUTPUT] = op.a.b[3];
            for (int n = 0; n < 4; n++)
                if (op.b.b[n] != op.c.b[n]) {
                    uint32_t z = 256u + op.b.b[n] - op.c.b[n];
                    for (int k = 0; k < 9; k++) r[lt::BITS + k] = (z >> k) & 1;
                    r[lt::BYTE_FLAG + n] = 1;
                    r[lt::DIFF_INV] = finv(fsub(op.b.b[n], op.c.b[n]));
                    break;
                }
            for (int k = 0; k < 8; k++) { r[lt::TOP_BITS_1 + k] = (op.b.b[0] >> k) & 1; r[lt::TOP_BITS_2 + k] = (op.c.b[0] >> k) & 1; }
            r[lt::DIFFERENT_SIGNS] = (is_signed && r[lt::TOP_BITS_1 + 7] != r[lt::TOP_BITS_2 + 7]) ? 1 : 0;
            r[lt::MULTIPLICITY] = 1;
        }
        return t;
    }
    RowMajor bitwise_trace() const {
        RowMajor t = alu_rows(bitwise_ops, bitwise::NUM_COLS);
        for (size_t i = 0; i < bitwise_ops.size(); i++) {
            uint32_t* r = t.row(i); const AluOp& op = bitwise_ops[i];
            r[op.opcode == OP_AND32 ? bitwise::IS_AND : op.opcode == OP_OR32 ? bitwise::IS_OR : bitwise::IS_XOR] = 1;
            for (int k = 0; k < 4; k++) {
                r[bitwise::INPUT_1 + k] = op.b.b[k]; r[bitwise::INPUT_2 + k] = op.c.b[k]; r[bitwise::OUTPUT + k] = op.a.b[k];
                for (int j = 0; j < 8; j++) { r[bitwise::BITS_1 + 8 * k + j] = (op.b.b[k] >> j) & 1; r[bitwise::BITS_2 + 8 * k + j] = (op.c.b[k] >> j) & 1; }
            }
        }
        return t;
    }
    RowMajor mul_trace() const {  // alu_u32/src/mul/mod.rs:38-62: at least 1024 rows (the range-check counter), counter = row + 1 on every row
        RowMajor t(std::max<size_t>(next_pow2(mul_ops.size()), 1024), mul::NUM_COLS);
        for (size_t i = 0; i < t.height; i++) t.row(i)[mul::COUNTER] = (uint32_t)(i + 1);
        for (size_t i = 0; i < mul_ops.size(); i++) {
            uint32_t* r = t.row(i); const AluOp& op = mul_ops[i];
            r[op.opcode == OP_MUL32 ? mul::IS_MUL : op.opcode == OP_MULHS32 ? mul::IS_MULHS : mul::IS_MULHU] = 1;
            for (int k = 0; k < 4; k++) { r[mul::INPUT_1 + k] = op.b.b[k]; r[mul::INPUT_2 + k] = op.c.b[k]; r[mul::OUTPUT + k] = op.a.b[k]; }  // r, s stay 0 (:125-132)
        }
        return t;
    }
    RowMajor div_trace() const {  // alu_u32/src/div/mod.rs:84-103: only the opcode flag ("TODO: Fill in other columns")
        RowMajor t = alu_rows(div_ops, divc::NUM_COLS);
        for (size_t i = 0; i < div_ops.size(); i++) t.row(i)[div_ops[i].opcode == OP_DIV32 ? divc::IS_DIV : divc::IS_SDIV] = 1;
        return t;
    }
    RowMajor shift_trace() const {  // alu_u32/src/shift/mod.rs:118-162
        RowMajor t = alu_rows(shift_ops, shift::NUM_COLS);
        for (size_t i = 0; i < shift_ops.size(); i++) {
            uint32_t* r = t.row(i); const AluOp& op = shift_ops[i];
            r[op.opcode == OP_SHL32 ? shift::IS_SHL : op.opcode == OP_SHR32 ? shift::IS_SHR : shift::IS_SRA] = 1;
            for (int k = 0; k < 4; k++) { r[shift::INPUT_1 + k] = op.b.b[k]; r[shift::INPUT_2 + k] = op.c.b[k]; r[shift::OUTPUT + k] = op.a.b[k]; }
            const uint32_t c3 = op.c.b[3];
            for (int j = 0; j < 8; j++) r[shift::BITS_2 + j] = (c3 >> j) & 1;
            r[shift::TEMP_1] = (c3 & 1) + 2 * ((c3 >> 1) & 1) + 4 * ((c3 >> 2) & 1);  // the exponent, as written (:159-160)
            const Word pw = word_of(1u << (u32_of(op.c) & 31u));
            for (int k = 0; k < 4; k++) r[shift::POWER_OF_TWO + k] = pw.b[k];
        }
        return t;
    }
    RowMajor com_trace() const {  // alu_u32/src/com/mod.rs:87-103: only the opcode flag
        RowMajor t = alu_rows(com_ops, com::NUM_COLS);
        for (size_t i = 0; i < com_ops.size(); i++) t.row(i)[com_ops[i].opcode == OP_NE32 ? com::IS_NE : com::IS_EQ] = 1;
        return t;
    }
    // rows of window i of the output tape (output/src/lib.rs:41-47): (clk_2 - clk_1) / table_len + 1
    static uint64_t output_window_rows(uint32_t clk_1, uint32_t clk_2, uint32_t table_len) { return (uint64_t)((clk_2 - clk_1) / table_len) + 1; }
    RowMajor output_trace() const {  // output/src/lib.rs:37-100
        const size_t n = output_values.size();
        const uint32_t table_len = (uint32_t)n;
        std::vector<std::array<uint32_t, output::NUM_COLS>> rows;
        for (size_t w = 0; w + 1 < n; w++) {
            const uint32_t clk_1 = output_values[w].first, clk_2 = output_values[w + 1].first;
            const uint64_t num = output_window_rows(clk_1, clk_2, table_len);
            const size_t base = rows.size();
            for (uint64_t i = 0; i < num; i++) {
                std::array<uint32_t, output::NUM_COLS> r{};
                if (i == 0) { r[output::IS_REAL] = 1; r[output::CLK] = clk_1 % P; r[output::VALUE] = output_values[w].second; }
                else r[output::CLK] = (uint32_t)(((uint64_t)clk_1 + (uint64_t)table_len * (i + 1)) % P);  // "dummy output to satisfy range check"
                rows.push_back(r);
            }
            for (uint64_t i = 0; i < num; i++) {  // clock diffs inside the window, the last one against clk_2
                const uint32_t next = i + 1 < num ? rows[base + i + 1][output::CLK] : clk_2 % P;
                rows[base + i][output::DIFF] = fsub(next, rows[base + i][output::CLK]);
            }
        }
        if (n) {
            std::array<uint32_t, output::NUM_COLS> r{};
            r[output::IS_REAL] = 1; r[output::CLK] = output_values[n - 1].first % P; r[output::VALUE] = output_values[n - 1].second;
            rows.push_back(r);
        }
        RowMajor t(next_pow2(rows.size()), output::NUM_COLS);  // counter, counter_mult, opcode: never written by the reference
        for (size_t i = 0; i < rows.size(); i++) for (int c = 0; c < output::NUM_COLS; c++) t.row(i)[c] = rows[i][c];
        return t;
    }
    RowMajor range_trace() const {
        RowMajor t(256, range::NUM_COLS);
        for (size_t i = 0; i < 256; i++) { t.row(i)[range::MULT] = range_counts[i] % P; t.row(i)[range::COUNTER] = (uint32_t)i; }
        return t;
    }
    RowMajor static_data_trace() const {  // static_data/src/lib.rs:60-79
        RowMajor t(next_pow2(static_cells.size()), static_data::NUM_COLS);
        size_t k = 0;
        for (auto& kv : static_cells) {
            uint32_t* r = t.row(k++);
            r[static_data::ADDR] = kv.first % P; r[static_data::IS_REAL] = 1;
            for (int b = 0; b < 4; b++) r[static_data::VALUE + b] = kv.second.b[b];
        }
        return t;
    }
    RowMajor program_trace() const {
        RowMajor t(next_pow2(program_counts.size()), program::NUM_COLS);
        for (size_t i = 0; i < program_counts.size(); i++) t.row(i)[0] = program_counts[i] % P;
        return t;
    }
    RowMajor program_preprocessed() const {
        RowMajor t(next_pow2(rom.size()), program::NUM_PRE_COLS);
        for (size_t i = 0; i < t.height; i++) {
            uint32_t* r = t.row(i);
            r[0] = (uint32_t)i;
            if (i < rom.size()) { r[1] = rom[i].opcode; for (int k = 0; k < 5; k++) r[2 + k] = from_i32(rom[i].ops[k]); }
        }
        return t;
    }
    static RowMajor range_preprocessed() {
        RowMajor t(256, 1);
        for (size_t i = 0; i < 256; i++) t.row(i)[0] = (uint32_t)i;
        return t;
    }
    // All 14 main traces in chip order (basic/src/lib.rs:151-166).
    std::vector<RowMajor> main_traces() const {
        std::vector<RowMajor> out(NUM_CHIPS);
        out[CHIP_CPU] = cpu_trace();
        out[CHIP_PROGRAM] = program_trace();
        out[CHIP_MEM] = mem_trace();
        out[CHIP_ADD] = add_trace();
        out[CHIP_SUB] = sub_trace();
        out[CHIP_MUL] = mul_trace();
        out[CHIP_DIV] = div_trace();
        out[CHIP_SHIFT] = shift_trace();
        out[CHIP_LT] = lt_trace();
        out[CHIP_COM] = com_trace();
        out[CHIP_BITWISE] = bitwise_trace();
        out[CHIP_OUTPUT] = output_trace();
        out[CHIP_RANGE] = range_trace();
        out[CHIP_STATIC_DATA] = static_data_trace();
        return out;
    }
};

// fib_program (basic/tests/test_prover.rs:35-188) with the loop bound `n` as the immediate of
// instruction 1 (big-endian bytes in operands b..e).
inline std::vector<InstructionWord> fib_program(uint32_t n) {
    const int32_t B = BYTES_PER_INSTR;
    const int32_t fib_bb0 = 8 * B, bb0_1 = 13 * B, bb0_2 = 15 * B, bb0_3 = 19 * B, bb0_4 = 21 * B;
    return {
        {OP_IMM32, {-4, 0, 0, 0, 0}},
        {OP_IMM32, {-8, (int32_t)(n >> 24), (int32_t)((n >> 16) & 255), (int32_t)((n >> 8) & 255), (int32_t)(n & 255)}},
        {OP_ADD32, {-16, -8, 0, 0, 1}},
        {OP_IMM32, {-20, 0, 0, 0, 28}},
        {OP_JAL, {-28, fib_bb0, -28, 0, 0}},
        {OP_ADD32, {-12, -24, 0, 0, 1}},
        {OP_ADD32, {4, -12, 0, 0, 1}},
        {OP_STOP, {0, 0, 0, 0, 0}},
        // fib:
        {OP_ADD32, {-4, 12, 0, 0, 1}},
        {OP_IMM32, {-8, 0, 0, 0, 0}},
        {OP_IMM32, {-12, 0, 0, 0, 1}},
        {OP_IMM32, {-16, 0, 0, 0, 0}},
        {OP_BEQ, {bb0_1, 0, 0, 0, 0}},
        // .LBB0_1:
        {OP_BNE, {bb0_2, -16, -4, 0, 0}},
        {OP_BEQ, {bb0_4, 0, 0, 0, 0}},
        // .LBB0_2:
        {OP_ADD32, {-20, -8, -12, 0, 0}},
        {OP_ADD32, {-8, -12, 0, 0, 1}},
        {OP_ADD32, {-12, -20, 0, 0, 1}},
        {OP_BEQ, {bb0_3, 0, 0, 0, 0}},
        // .LBB0_3:
        {OP_ADD32, {-16, -16, 1, 0, 1}},
        {OP_BEQ, {bb0_1, 0, 0, 0, 0}},
        // .LBB0_4:
        {OP_ADD32, {4, -8, 0, 0, 1}},
        {OP_JALV, {-4, 0, 8, 0, 0}},
    };
}

// Synthetic ALU-heavy loop (workload C4 of SURVEY.md §8): 9 instructions per iteration
//   add, sub, xor, and, or, lt, addi, addi, bne
// exercising the add / sub / bitwise / lt chips and the range bus.  The subtraction is 0xFFFFFFFF - t, which
// never borrows: the reference's Sub32 witness (alu_u32/src/sub/mod.rs:104-112) ignores incoming borrows,
// so only borrow-free subtractions satisfy its own AIR.
inline std::vector<InstructionWord> alu_program(uint32_t iters) {
    const int32_t B = BYTES_PER_INSTR, loop = 4 * B;
    auto b = [](uint32_t v, int i) { return (int32_t)((v >> (24 - 8 * i)) & 255); };
    const uint32_t x0 = 0x01234567u, y0 = 0x9E3779B9u;
    return {
        {OP_IMM32, {-4, b(x0, 0), b(x0, 1), b(x0, 2), b(x0, 3)}},    // x
        {OP_IMM32, {-8, b(y0, 0), b(y0, 1), b(y0, 2), b(y0, 3)}},    // y
        {OP_IMM32, {-12, 0, 0, 0, 0}},                               // i
        {OP_IMM32, {-16, 255, 255, 255, 255}},                       // ones
        // loop:
        {OP_ADD32, {-20, -4, -8, 0, 0}},    // t1 = x + y
        {OP_SUB32, {-24, -16, -20, 0, 0}},  // t2 = ~t1
        {OP_XOR32, {-28, -20, -24, 0, 0}},  // t3 = t1 ^ t2
        {OP_AND32, {-32, -28, -8, 0, 0}},   // t4 = t3 & y
        {OP_OR32, {-36, -32, -4, 0, 0}},    // t5 = t4 | x
        {OP_LT32, {-40, -36, -20, 0, 0}},   // t6 = t5 < t1
        {OP_ADD32, {-4, -20, 12345, 0, 1}}, // x = t1 + 12345
        {OP_ADD32, {-12, -12, 1, 0, 1}},    // i += 1
        {OP_BNE, {loop, -12, (int32_t)iters, 0, 1}},
        {OP_STOP, {0, 0, 0, 0, 0}},
    };
}

// The reference's other pinned prover programs (basic/tests/test_prover.rs:190-402): each is proved and verified
// there, and the listed memory cells are asserted (:489-640).
inline std::vector<InstructionWord> left_imm_ops_program() {  // test_prover.rs:190-262
    return {
        {OP_IMM32, {-4, 0, 0, 0, 3}},    {OP_IMM32, {-8, 0, 0, 1, 0}},
        {OP_LT32, {4, 3, -4, 1, 0}},     {OP_LTE32, {8, 3, -4, 1, 0}},    {OP_LT32, {12, 4, -4, 1, 0}},  {OP_LTE32, {16, 4, -4, 1, 0}},
        {OP_LT32, {20, 2, -4, 1, 0}},    {OP_LTE32, {24, 2, -4, 1, 0}},   {OP_LT32, {28, 256, -4, 1, 0}}, {OP_LTE32, {32, 256, -4, 1, 0}},
        {OP_LT32, {36, 3, -8, 1, 0}},    {OP_LTE32, {40, 3, -8, 1, 0}},   {OP_STOP, {0, 0, 0, 0, 0}},
    };
}
inline std::vector<InstructionWord> signed_inequality_program() {  // test_prover.rs:264-379
    std::vector<InstructionWord> p = {
        {OP_IMM32, {-4, 0, 0, 0, 1}}, {OP_IMM32, {-8, 255, 255, 255, 255}}, {OP_IMM32, {-12, 255, 255, 255, 254}},
    };
    const int32_t ops[8][5] = {{4, -12, -8, 0, 0}, {8, -12, -4, 0, 0}, {12, -4, -1, 0, 1}, {16, -1, -8, 1, 0},
                               {20, -1, -8, 1, 0}, {24, -1, -12, 1, 0}, {28, -8, -12, 0, 0}, {32, -8, -4, 0, 0}};
    for (int i = 0; i < 8; i++) p.push_back({i == 4 ? OP_SLE32 : OP_SLT32, {ops[i][0], ops[i][1], ops[i][2], ops[i][3], ops[i][4]}});
    for (int i = 0; i < 8; i++) p.push_back({i == 4 ? OP_LTE32 : OP_LT32, {ops[i][0] + 32, ops[i][1], ops[i][2], ops[i][3], ops[i][4]}});
    p.push_back({OP_STOP, {0, 0, 0, 0, 0}});
    return p;
}
// basic/tests/test_static_data.rs:31-59: loops until the statically initialised cell 0x10 holds 0x25 (cells 0x10 -> 0x25
// and 0x14 -> 0x32 are written through the static-data chip before the run)
inline std::vector<InstructionWord> static_data_program() {
    return {{OP_IMM32, {0, 0, 0, 0, 0x10}}, {OP_LOAD32, {-4, 0, 0, 0, 0}}, {OP_BNE, {0, -4, 0x25, 0, 1}}, {OP_STOP, {0, 0, 0, 0, 0}}};
}
inline std::vector<InstructionWord> loadfp_program() {  // test_prover.rs:381-402
    return {{OP_LOADFP, {4, 0, 0, 0, 0}}, {OP_LOADFP, {8, 3, 0, 0, 0}}, {OP_STOP, {0, 0, 0, 0, 0}}};
}

// Every chip of the BasicMachine busy at once (no counterpart among the reference's tests, whose programs leave mul / div / shift / com / output
// idle; SURVEY.md §0.6: several of these chips are stubs in the reference, so a proof of this program is well-formed but NOT accepted by a
// verifier — it exists to exercise the five remaining trace generators).  Per iteration: mul, mulhu, mulhs, muli, div, sdiv, shl, shr,
// sra, shli, ne, eq, add, sub, xor, lt, write, addi, addi, bne.
inline std::vector<InstructionWord> mixed_ops_program(uint32_t iters) {
    const int32_t B = BYTES_PER_INSTR, loop = 5 * B;
    auto b = [](uint32_t v, int i) { return (int32_t)((v >> (24 - 8 * i)) & 255); };
    const uint32_t x0 = 0x12345678u, y0 = 0x9ABCDEF1u;
    return {
        {OP_IMM32, {-4, b(x0, 0), b(x0, 1), b(x0, 2), b(x0, 3)}},   // x
        {OP_IMM32, {-8, b(y0, 0), b(y0, 1), b(y0, 2), b(y0, 3)}},   // y
        {OP_IMM32, {-12, 0, 0, 0, 0}},                              // i
        {OP_IMM32, {-16, 0, 0, 0, 5}},                              // s = 5 (shift amount / small divisor)
        {OP_IMM32, {-20, 255, 255, 255, 255}},                      // ones
        // loop:
        {OP_MUL32, {-24, -4, -8, 0, 0}},     // t1 = x * y
        {OP_MULHU32, {-28, -4, -8, 0, 0}},   // t2 = hi(x * y)
        {OP_MULHS32, {-32, -8, -4, 0, 0}},
        {OP_MUL32, {-36, -4, 7, 0, 1}},      // t4 = x * 7
        {OP_DIV32, {-40, -8, -16, 0, 0}},    // t5 = y / 5
        {OP_SDIV32, {-44, -8, 3, 0, 1}},     // t6 = (i32)y / 3
        {OP_SHL32, {-48, -4, -16, 0, 0}},    // t7 = x << 5
        {OP_SHR32, {-52, -8, -16, 0, 0}},    // t8 = y >> 5
        {OP_SRA32, {-56, -8, -16, 0, 0}},    // t9 = (i32)y >> 5
        {OP_SHL32, {-60, -4, 3, 0, 1}},      // t10 = x << 3
        {OP_NE32, {-64, -24, -28, 0, 0}},
        {OP_EQ32, {-68, -4, -4, 0, 0}},
        {OP_ADD32, {-72, -24, -28, 0, 0}},   // t13 = t1 + t2
        {OP_SUB32, {-76, -20, -72, 0, 0}},   // ~t13 (borrow-free)
        {OP_XOR32, {-4, -72, -4, 0, 0}},     // x ^= t13
        {OP_LT32, {-80, -4, -8, 0, 0}},
        {OP_WRITE, {0, -4, 0, 0, 1}},        // output the low byte of x
        {OP_ADD32, {-8, -8, 0x01010101, 0, 1}},  // y += ..
        {OP_ADD32, {-12, -12, 1, 0, 1}},     // i += 1
        {OP_BNE, {loop, -12, (int32_t)iters, 0, 1}},
        {OP_WRITE, {0, -8, 0, 0, 1}},
        {OP_STOP, {0, 0, 0, 0, 0}},
    };
}

}  // namespace vwork
