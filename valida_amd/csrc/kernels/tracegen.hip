// Trace generation on the device: `Chip::generate_trace` (machine/src/chip.rs:22) of the big
// BasicMachine chips, from the VM's operation logs instead of from host-built RowMajorMatrix traces:
//   cpu      cpu/src/lib.rs:79-97,163-236   op_to_row, set_memory_channel_values :253-296, compute_word_diffs
//                                            :298-330, pad_to_power_of_two :332-373
//   mem      memory/src/lib.rs:143-194       sort by (addr, clk), op_to_row, zero padding
//   add      alu_u32/src/add/mod.rs:38-51,91-121      sub  alu_u32/src/sub/mod.rs:91-121
//   lt       alu_u32/src/lt/mod.rs:87-166             bitwise  alu_u32/src/bitwise/mod.rs:84-129
// One thread per trace row; every column is written straight into the column-major Montgomery matrix the
// commitment phase consumes (natural row order), so neither the 516 MB row-major upload nor k_ingest is needed.
// The memory chip's (addr, clk) order is a stable LSD radix sort (4 passes of 8 bits) of the clk-ordered log by address.
#include <algorithm>
#include <cstring>
#include "launch.hpp"
#include "../chips/basic_machine.hpp"

namespace vk {

using namespace vchips;

__device__ __forceinline__ void put(DMatView m, int col, uint64_t row, uint32_t canonical) {
    m.data[(uint64_t)col * m.stride + row] = Fp::from_canonical(canonical).v;
}
__device__ __forceinline__ void put_raw(DMatView m, int col, uint64_t row, Fp v) { m.data[(uint64_t)col * m.stride + row] = v.v; }
__device__ __forceinline__ uint32_t byte_of(uint32_t word, int k) { return (word >> (24 - 8 * k)) & 0xffu; }  // big-endian byte k (machine/src/core.rs:9)
__device__ __forceinline__ uint32_t from_i32(int32_t x) {  // Val::from_canonical / negative operands as p - |x|
    uint32_t a = (uint32_t)(x < 0 ? -(int64_t)x : (int64_t)x) % vg::P;
    return (x < 0 && a) ? vg::P - a : a;
}

// ---- cpu -----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tracegen_cpu(const TgCpuOp* __restrict__ ops, uint64_t n, const TgMemOp* __restrict__ mem, uint64_t n_mem, DMatView t) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.height) return;
    uint32_t r[cpu::NUM_COLS];
#pragma unroll
    for (int c = 0; c < cpu::NUM_COLS; c++) r[c] = 0;
    r[cpu::ch(0, cpu::CH_IS_READ)] = 1; r[cpu::ch(1, cpu::CH_IS_READ)] = 1;
    Fp diff = Fp::zero();
    if (i < n) {
        const TgCpuOp o = ops[i];
        r[cpu::PC] = o.pc; r[cpu::FP] = o.fp % vg::P; r[cpu::CLK] = (uint32_t)i; r[cpu::OPCODE] = o.opcode;
#pragma unroll
        for (int k = 0; k < 5; k++) r[cpu::OPERAND_A + k] = from_i32(o.operands[k]);
        const bool left = o.kind == TG_CPU_BUS_LEFT_IMM;
        // flag column of the op kind (Operation -> is_* flags, cpu/src/lib.rs:163-236)
        r[cpu::IS_STORE] = o.kind == TG_CPU_STORE32; r[cpu::IS_LOAD] = o.kind == TG_CPU_LOAD32; r[cpu::IS_JAL] = o.kind == TG_CPU_JAL;
        r[cpu::IS_JALV] = o.kind == TG_CPU_JALV; r[cpu::IS_BEQ] = o.kind == TG_CPU_BEQ; r[cpu::IS_BNE] = o.kind == TG_CPU_BNE;
        r[cpu::IS_IMM32] = o.kind == TG_CPU_IMM32; r[cpu::IS_BUS_OP] = (o.kind == TG_CPU_BUS || left); r[cpu::IS_STOP] = o.kind == TG_CPU_STOP;
        r[cpu::IS_LOADFP] = o.kind == TG_CPU_LOADFP;
        uint32_t val[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, used[3] = {0, 0, 0}, addr[3] = {0, 0, 0};
        const bool imm_kind = o.kind == TG_CPU_BEQ || o.kind == TG_CPU_BNE || o.kind == TG_CPU_BUS || left;
        if (o.has_imm && imm_kind) {  // the immediate rides in the read channel it replaces; operand = Word::reduce
            if (left) { r[cpu::IS_LEFT_IMM_OP] = 1; r[cpu::OPERAND_B] = o.imm % vg::P; }
            else { r[cpu::IS_IMM_OP] = 1; r[cpu::OPERAND_C] = o.imm % vg::P; }
#pragma unroll
            for (int k = 0; k < 4; k++) { if (left) val[0][k] = byte_of(o.imm, k); else val[1][k] = byte_of(o.imm, k); }
        }
        // set_memory_channel_values: this cycle's memory operations in issue order
        const uint64_t m0 = o.mem_first, m1 = i + 1 < n ? ops[i + 1].mem_first : n_mem;
        bool first_read = true;
        for (uint64_t k = m0; k < m1; k++) {
            const TgMemOp m = mem[k];
            const int chn = m.is_write ? 2 : ((first_read && !left) ? 0 : 1);
            if (!m.is_write && chn == 0) first_read = false;
#pragma unroll
            for (int c = 0; c < 3; c++)
                if (c == chn) {
                    used[c] = 1; addr[c] = m.addr % vg::P;
#pragma unroll
                    for (int b = 0; b < 4; b++) val[c][b] = byte_of(m.value, b);
                }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            r[cpu::ch(c, cpu::CH_USED)] = used[c]; r[cpu::ch(c, cpu::CH_ADDR)] = addr[c];
#pragma unroll
            for (int b = 0; b < 4; b++) r[cpu::ch(c, cpu::CH_VALUE) + b] = val[c][b];
        }
        // compute_word_diffs: sum of squared byte differences of the two read channels, its inverse, the flag
#pragma unroll
        for (int b = 0; b < 4; b++) { Fp x = Fp::from_canonical(val[0][b]) - Fp::from_canonical(val[1][b]); diff += x * x; }
    } else {
        // pad_to_power_of_two: STOP rows continuing the clock
        const TgCpuOp o = ops[n - 1];
        r[cpu::PC] = o.pc; r[cpu::FP] = o.fp % vg::P; r[cpu::CLK] = (uint32_t)(((n - 1) % vg::P + (i - n + 1) % vg::P) % vg::P);
        r[cpu::IS_STOP] = 1; r[cpu::OPCODE] = OP_STOP;
    }
#pragma unroll
    for (int c = 0; c < cpu::NUM_COLS; c++)
        if (c != cpu::DIFF && c != cpu::DIFF_INV && c != cpu::NOT_EQUAL) put(t, c, i, r[c]);
    put_raw(t, cpu::DIFF, i, diff);
    put_raw(t, cpu::DIFF_INV, i, diff.inv());  // 0 -> 0
    put(t, cpu::NOT_EQUAL, i, diff.is_zero() ? 0u : 1u);
}

// ---- memory ----------------------------------------------------------------------------------------------
__global__ void k_tg_mem_keys(const TgMemOp* __restrict__ mem, uint64_t n, uint32_t* keys, uint32_t* idx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = mem[i].addr; idx[i] = (uint32_t)i; }
}
__global__ void __launch_bounds__(256) k_tracegen_mem(const TgMemOp* __restrict__ mem, const uint32_t* __restrict__ order, uint64_t n,
                                                      const uint32_t* __restrict__ static_cells, uint64_t n_static, DMatView t) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.height) return;
    uint32_t r[mem::NUM_COLS];
#pragma unroll
    for (int c = 0; c < mem::NUM_COLS; c++) r[c] = 0;
    if (i < n_static) {  // static_data_to_row (memory/src/lib.rs:265-284): the statically initialised cells come first
        const uint32_t addr = static_cells[2 * i], value = static_cells[2 * i + 1];
        r[mem::IS_STATIC_INITIAL] = 1; r[mem::COUNTER] = (uint32_t)i; r[mem::ADDR] = addr % vg::P; r[mem::IS_WRITE] = 1;
#pragma unroll
        for (int b = 0; b < 4; b++) r[mem::VALUE + b] = byte_of(value, b);
    } else if (i < n_static + n) {
        const TgMemOp m = mem[order[i - n_static]];
        r[mem::CLK] = m.clk; r[mem::COUNTER] = (uint32_t)i; r[mem::ADDR] = m.addr % vg::P;
#pragma unroll
        for (int b = 0; b < 4; b++) r[mem::VALUE + b] = byte_of(m.value, b);
        r[mem::IS_WRITE] = m.is_write ? 1 : 0; r[mem::IS_READ] = m.is_write ? 0 : 1;
    }
#pragma unroll
    for (int c = 0; c < mem::NUM_COLS; c++) put(t, c, i, r[c]);
}

// ---- 32-bit ALU chips -----------------------------------------------------------------------------------
template <int CHIP> __global__ void __launch_bounds__(256) k_tracegen_alu(const TgAluOp* __restrict__ ops, uint64_t n, DMatView t) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.height) return;
    constexpr int W = CHIP == CHIP_ADD ? (int)add::NUM_COLS : CHIP == CHIP_SUB ? (int)sub::NUM_COLS : CHIP == CHIP_LT ? (int)lt::NUM_COLS : (int)bitwise::NUM_COLS;
    uint32_t r[W];
#pragma unroll
    for (int c = 0; c < W; c++) r[c] = 0;
    Fp lt_diff_inv = Fp::zero();
    if (i < n) {
        const TgAluOp op = ops[i];
        uint32_t a[4], b[4], c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { a[k] = byte_of(op.a, k); b[k] = byte_of(op.b, k); c[k] = byte_of(op.c, k); }
        if (CHIP == CHIP_ADD) {
#pragma unroll
            for (int k = 0; k < 4; k++) { r[add::INPUT_1 + k] = b[k]; r[add::INPUT_2 + k] = c[k]; r[add::OUTPUT + k] = a[k]; }
            const uint32_t c1 = b[3] + c[3] > 255 ? 1 : 0, c2 = b[2] + c[2] + c1 > 255 ? 1 : 0, c3 = b[1] + c[1] + c2 > 255 ? 1 : 0;
            r[add::CARRY] = c1; r[add::CARRY + 1] = c2; r[add::CARRY + 2] = c3;
            r[add::IS_REAL] = 1;
        } else if (CHIP == CHIP_SUB) {
#pragma unroll
            for (int k = 0; k < 4; k++) { r[sub::INPUT_1 + k] = b[k]; r[sub::INPUT_2 + k] = c[k]; r[sub::OUTPUT + k] = a[k]; }
            r[sub::BORROW] = b[3] < c[3]; r[sub::BORROW + 1] = b[2] < c[2]; r[sub::BORROW + 2] = b[1] < c[1];  // witness as written (sub/mod.rs:104-112)
            r[sub::IS_REAL] = 1;
        } else if (CHIP == CHIP_LT) {
            const bool is_signed = op.opcode == OP_SLT32 || op.opcode == OP_SLE32;
            r[lt::IS_LT] = op.opcode == OP_LT32; r[lt::IS_LTE] = op.opcode == OP_LTE32; r[lt::IS_SLT] = op.opcode == OP_SLT32; r[lt::IS_SLE] = op.opcode == OP_SLE32;
#pragma unroll
            for (int k = 0; k < 4; k++) { r[lt::INPUT_1 + k] = b[k]; r[lt::INPUT_2 + k] = c[k]; }
            r[lt::OUTPUT] = a[3];
            bool found = false;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (!found && b[k] != c[k]) {
                    found = true;
                    const uint32_t z = 256u + b[k] - c[k];
#pragma unroll
                    for (int j = 0; j < 9; j++) r[lt::BITS + j] = (z >> j) & 1;
                    r[lt::BYTE_FLAG + k] = 1;
                    lt_diff_inv = (Fp::from_canonical(b[k]) - Fp::from_canonical(c[k])).inv();
                }
#pragma unroll
            for (int j = 0; j < 8; j++) { r[lt::TOP_BITS_1 + j] = (b[0] >> j) & 1; r[lt::TOP_BITS_2 + j] = (c[0] >> j) & 1; }
            r[lt::DIFFERENT_SIGNS] = (is_signed && ((b[0] >> 7) != (c[0] >> 7))) ? 1 : 0;
            r[lt::MULTIPLICITY] = 1;
        } else {
            r[bitwise::IS_AND] = op.opcode == OP_AND32; r[bitwise::IS_OR] = op.opcode == OP_OR32; r[bitwise::IS_XOR] = op.opcode == OP_XOR32;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                r[bitwise::INPUT_1 + k] = b[k]; r[bitwise::INPUT_2 + k] = c[k]; r[bitwise::OUTPUT + k] = a[k];
#pragma unroll
                for (int j = 0; j < 8; j++) { r[bitwise::BITS_1 + 8 * k + j] = (b[k] >> j) & 1; r[bitwise::BITS_2 + 8 * k + j] = (c[k] >> j) & 1; }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < W; c++)
        if (!(CHIP == CHIP_LT && c == lt::DIFF_INV)) put(t, c, i, r[c]);
    if (CHIP == CHIP_LT) put_raw(t, lt::DIFF_INV, i, lt_diff_inv);
}

// ---- range and program: multiplicity tables ---------------------------------------------------------------------
// range/src/lib.rs:32-55: row n = (mult = number of range checks of byte n, counter = n).  The checks are those the ALU
// instructions issue on their result words (alu_u32/src/add/mod.rs, sub/mod.rs: `range_check(a)` on execute), so the
// table is the byte histogram of the add and sub logs.  program/src/lib.rs:50-68: multiplicity of pc = number of
// instruction fetches at pc — the histogram of the cpu log's pc column plus the padded STOP rows at the final pc.
// Which words are range-checked is decided by the INSTRUCTION (its execute calls state.range_check(a)), not by the chip that logs the
// operation — a shift leaves a Mul32 / Div32 in the mul / div log without a range check — so the histogram walks the cpu log: a bus
// operation whose opcode is add, sub, mul, mulhs, mulhu, div or sdiv contributes the four bytes of the word its cycle wrote (the last
// memory operation of the cycle).
__global__ void __launch_bounds__(256) k_tg_range_histogram(const TgCpuOp* __restrict__ ops, uint64_t n, const TgMemOp* __restrict__ mem, uint64_t n_mem,
                                                            uint32_t* __restrict__ counts /* 256, zeroed */) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const TgCpuOp op = ops[i];
        if (op.kind != TG_CPU_BUS && op.kind != TG_CPU_BUS_LEFT_IMM) continue;
        const uint32_t oc = op.opcode;
        if (!(oc == OP_ADD32 || oc == OP_SUB32 || oc == OP_MUL32 || oc == OP_MULHS32 || oc == OP_MULHU32 || oc == OP_DIV32 || oc == OP_SDIV32)) continue;
        const uint64_t end = i + 1 < n ? ops[i + 1].mem_first : n_mem;
        if (end == 0 || end <= op.mem_first) continue;
        const TgMemOp w = mem[end - 1];
        if (!w.is_write) continue;
#pragma unroll
        for (int k = 0; k < 4; k++) atomicAdd(&h[byte_of(w.value, k)], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], h[threadIdx.x]);
}
// Loops make a few program counters very hot: fetches are first counted in an LDS table per block (pcs below PC_LDS_BINS),
// and only the non-zero bins reach the global table.
constexpr int PC_LDS_BINS = 4096, PC_ITEMS = 16;
__global__ void __launch_bounds__(256) k_tg_pc_histogram(const TgCpuOp* __restrict__ ops, uint64_t n, uint64_t padded_n, uint32_t rom_len, uint32_t* __restrict__ counts /* zeroed */) {
    __shared__ uint32_t h[PC_LDS_BINS];
    for (int b = threadIdx.x; b < PC_LDS_BINS; b += blockDim.x) h[b] = 0;
    __syncthreads();
    const uint64_t first = (uint64_t)blockIdx.x * (256 * PC_ITEMS);
#pragma unroll 4
    for (int u = 0; u < PC_ITEMS; u++) {
        const uint64_t i = first + u * 256 + threadIdx.x;
        if (i >= n) continue;
        const uint32_t pc = ops[i].pc;
        const uint32_t add = i + 1 == n ? (uint32_t)(1 + padded_n - n) : 1u;  // the last instruction (STOP) is re-fetched by every padding row
        if (pc < (uint32_t)PC_LDS_BINS) atomicAdd(&h[pc], add);
        else if (pc < rom_len) atomicAdd(&counts[pc], add);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < PC_LDS_BINS; b += blockDim.x)
        if (h[b] && (uint32_t)b < rom_len) atomicAdd(&counts[b], h[b]);
}
__global__ void __launch_bounds__(256) k_tracegen_counts(const uint32_t* __restrict__ counts, uint64_t n_counts, int with_counter, DMatView t) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.height) return;
    put(t, 0, i, i < n_counts ? counts[i] : 0u);
    if (with_counter) put(t, 1, i, (uint32_t)i);
}

// ---- mul / div / shift / com: rows exactly as the reference fills them (see workload/basic_vm.hpp for the file:line of every rule) ----
template <int CHIP> __global__ void __launch_bounds__(256) k_tracegen_alu2(const TgAluOp* __restrict__ ops, uint64_t n, DMatView t) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.height) return;
    constexpr int W = CHIP == CHIP_MUL ? (int)mul::NUM_COLS : CHIP == CHIP_DIV ? (int)divc::NUM_COLS : CHIP == CHIP_SHIFT ? (int)shift::NUM_COLS : (int)com::NUM_COLS;
    uint32_t r[W];
#pragma unroll
    for (int c = 0; c < W; c++) r[c] = 0;
    if (CHIP == CHIP_MUL) r[mul::COUNTER] = (uint32_t)(i + 1);  // every row, padding included (alu_u32/src/mul/mod.rs:47-60)
    if (i < n) {
        const TgAluOp op = ops[i];
        uint32_t a[4], b[4], c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { a[k] = byte_of(op.a, k); b[k] = byte_of(op.b, k); c[k] = byte_of(op.c, k); }
        if (CHIP == CHIP_MUL) {
            r[mul::IS_MUL] = op.opcode == OP_MUL32; r[mul::IS_MULHS] = op.opcode == OP_MULHS32; r[mul::IS_MULHU] = op.opcode == OP_MULHU32;
#pragma unroll
            for (int k = 0; k < 4; k++) { r[mul::INPUT_1 + k] = b[k]; r[mul::INPUT_2 + k] = c[k]; r[mul::OUTPUT + k] = a[k]; }
        } else if (CHIP == CHIP_DIV) {
            r[divc::IS_DIV] = op.opcode == OP_DIV32; r[divc::IS_SDIV] = op.opcode == OP_SDIV32;
        } else if (CHIP == CHIP_SHIFT) {
            r[shift::IS_SHL] = op.opcode == OP_SHL32; r[shift::IS_SHR] = op.opcode == OP_SHR32; r[shift::IS_SRA] = op.opcode == OP_SRA32;
#pragma unroll
            for (int k = 0; k < 4; k++) { r[shift::INPUT_1 + k] = b[k]; r[shift::INPUT_2 + k] = c[k]; r[shift::OUTPUT + k] = a[k]; }
#pragma unroll
            for (int j = 0; j < 8; j++) r[shift::BITS_2 + j] = (c[3] >> j) & 1;
            r[shift::TEMP_1] = (c[3] & 1) + 2 * ((c[3] >> 1) & 1) + 4 * ((c[3] >> 2) & 1);
            const uint32_t pw = 1u << (op.c & 31u);
#pragma unroll
            for (int k = 0; k < 4; k++) r[shift::POWER_OF_TWO + k] = byte_of(pw, k);
        } else {
            r[com::IS_NE] = op.opcode == OP_NE32; r[com::IS_EQ] = op.opcode == OP_EQ32;
        }
    }
#pragma unroll
    for (int c = 0; c < W; c++) put(t, c, i, r[c]);
}

// ---- output (output/src/lib.rs:37-100): window w of the tape (entries w, w + 1) occupies rows [row0[w], row0[w + 1]): the real write, then
// dummy rows clk_1 + table_len (i + 1); diff = next row's clk - this row's (the window's last row against clk_2); one final real row.
// row0 (n entries: first row of every window, the last one = the final row) is computed by the host when the log is uploaded.
__global__ void __launch_bounds__(256) k_tracegen_output(const TgOutOp* __restrict__ vals, const uint32_t* __restrict__ row0, uint64_t n, uint64_t n_rows, DMatView t) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.height) return;
    uint32_t r[output::NUM_COLS];
#pragma unroll
    for (int c = 0; c < (int)output::NUM_COLS; c++) r[c] = 0;
    if (i < n_rows) {
        // the window of row i: the last w with row0[w] <= i
        uint64_t lo = 0, hi = n - 1;
        while (lo < hi) { const uint64_t mid = (lo + hi + 1) >> 1; if (row0[mid] <= i) lo = mid; else hi = mid - 1; }
        const uint64_t w = lo, k = i - row0[w];
        const uint32_t table_len = (uint32_t)n, clk_1 = vals[w].clk;
        auto clk_of = [&](uint64_t kk) { return kk == 0 ? clk_1 % vg::P : (uint32_t)(((uint64_t)clk_1 + (uint64_t)table_len * (kk + 1)) % vg::P); };
        r[output::CLK] = clk_of(k);
        if (k == 0) { r[output::IS_REAL] = 1; r[output::VALUE] = vals[w].byte & 255u; }
        if (w + 1 < n) {
            const uint64_t num = row0[w + 1] - row0[w];
            const uint32_t next = k + 1 < num ? clk_of(k + 1) : vals[w + 1].clk % vg::P;
            r[output::DIFF] = next >= r[output::CLK] ? next - r[output::CLK] : next + vg::P - r[output::CLK];
        }
    }
#pragma unroll
    for (int c = 0; c < (int)output::NUM_COLS; c++) put(t, c, i, r[c]);
}

// ---- chips that received no operations: padding-only traces ------------------------------------------------------------
// mul (alu_u32/src/mul/mod.rs:38-62): max(1024, ..) rows whose only non-zero column is counter = row + 1; div / shift / com /
// output (pad_to_power_of_two of an empty table): one zero row; static_data (static_data/src/lib.rs:60-79): one row
// (addr, value bytes, is_real = 1) per initialised cell.  mode: 0 = zeros, 1 = mul counter, 2 = static-data rows.
__global__ void __launch_bounds__(256) k_tracegen_idle(int mode, const uint32_t* __restrict__ static_cells, uint64_t n_static, DMatView t) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.height) return;
    for (int c = 0; c < (int)t.width; c++) {
        uint32_t v = 0;
        if (mode == 1 && c == mul::COUNTER) v = (uint32_t)(i + 1);
        if (mode == 2 && i < n_static) {
            const uint32_t addr = static_cells[2 * i], value = static_cells[2 * i + 1];
            if (c == static_data::ADDR) v = addr % vg::P;
            else if (c >= static_data::VALUE && c < static_data::VALUE + 4) v = byte_of(value, c - static_data::VALUE);
            else if (c == static_data::IS_REAL) v = 1;
        }
        put(t, c, i, v);
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------
static unsigned blocks_for(uint64_t rows) { return (unsigned)((rows + 255) / 256); }

void launch_tracegen_cpu(hipStream_t st, const TgCpuOp* ops, uint64_t n, const TgMemOp* mem, uint64_t n_mem, DMatView t) {
    ProfScope ps("k_tracegen_cpu", st, 48.0 * n + 16.0 * n_mem + 4.0 * t.height * t.width);
    VK_LAUNCH(k_tracegen_cpu, dim3(blocks_for(t.height)), dim3(256), 0, st, ops, n, mem, n_mem, t);
}

// ---- stable LSD radix sort of (address, index) pairs: 4 passes x 8 bits ----------------------------------------------
// A block owns RS_ITEMS * 256 consecutive pairs.  Pass = (1) per-block digit histograms (LDS atomics), (2) exclusive scan
// of the digit-major table counts[digit][block], (3) scatter: the block walks its pairs in order, 256 at a time; a pair's
// slot is table base + pairs of that digit seen in earlier rounds + earlier waves of the round + lower lanes of its wave
// (eight ballots isolate the lanes holding the same digit) — which keeps equal keys in input order.
constexpr int RS_ITEMS = 32, RS_BLOCK = RS_ITEMS * 256;

__global__ void __launch_bounds__(256) k_rs_count(const uint32_t* __restrict__ keys, uint64_t n, int shift, uint32_t* __restrict__ counts, uint32_t n_blocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * RS_BLOCK;
#pragma unroll
    for (int u = 0; u < RS_ITEMS; u++) {
        const uint64_t i = base + u * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[(uint64_t)threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

// In-place exclusive scan of `total` counters by one 1024-thread block (contiguous chunk per thread).
__global__ void __launch_bounds__(1024) k_rs_scan(uint32_t* __restrict__ counts, uint64_t total) {
    __shared__ uint32_t sums[1024];
    const uint64_t chunk = (total + 1023) / 1024, lo = (uint64_t)threadIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
    uint32_t s = 0;
    for (uint64_t i = lo; i < hi; i++) s += counts[i];
    sums[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t t = sums[threadIdx.x];
        if ((int)threadIdx.x >= off) t += sums[threadIdx.x - off];
        __syncthreads();
        sums[threadIdx.x] = t;
        __syncthreads();
    }
    uint32_t run = sums[threadIdx.x] - s;  // exclusive prefix of this thread's chunk
    for (uint64_t i = lo; i < hi; i++) { const uint32_t c = counts[i]; counts[i] = run; run += c; }
}

__global__ void __launch_bounds__(256) k_rs_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t n, int shift,
                                                    const uint32_t* __restrict__ counts, uint32_t n_blocks, uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t base[256], wcount[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    base[threadIdx.x] = counts[(uint64_t)threadIdx.x * n_blocks + blockIdx.x];
    const uint64_t first = (uint64_t)blockIdx.x * RS_BLOCK;
    for (int u = 0; u < RS_ITEMS; u++) {
#pragma unroll
        for (int w = 0; w < 4; w++) wcount[w][threadIdx.x] = 0;
        __syncthreads();
        const uint64_t i = first + u * 256 + threadIdx.x;
        const bool live = i < n;
        const uint32_t key = live ? keys[i] : 0u, val = live ? vals[i] : 0u, d = (key >> shift) & 255u;
        unsigned long long peers = __ballot(live);  // lanes of this wave holding a live pair with the same digit
#pragma unroll
        for (int b = 0; b < 8; b++) { const unsigned long long bal = __ballot((d >> b) & 1u); peers &= ((d >> b) & 1u) ? bal : ~bal; }
        const uint32_t rank_in_wave = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (live && rank_in_wave == 0) wcount[wave][d] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (live) {
            uint32_t before = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) if (w < wave) before += wcount[w][d];
            const uint32_t pos = base[d] + before + rank_in_wave;
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        base[threadIdx.x] += wcount[0][threadIdx.x] + wcount[1][threadIdx.x] + wcount[2][threadIdx.x] + wcount[3][threadIdx.x];
        __syncthreads();
    }
}

// scratch words for the digit tables of an n-pair sort
size_t tracegen_mem_sort_scratch_bytes(uint64_t n) { return (size_t)256 * ((n + RS_BLOCK - 1) / RS_BLOCK) * 4; }

// keys2 / vals2: [2n] each (ping-pong halves, input in the first half); after four passes the result is back in the FIRST half.
static void radix_sort_pairs(hipStream_t st, uint32_t* keys2, uint32_t* vals2, uint64_t n, uint32_t* counts) {
    const uint32_t n_blocks = (uint32_t)((n + RS_BLOCK - 1) / RS_BLOCK);
    for (int pass = 0; pass < 4; pass++) {
        uint32_t* kin = keys2 + (pass & 1) * n;
        uint32_t* vin = vals2 + (pass & 1) * n;
        uint32_t* kout = keys2 + ((pass + 1) & 1) * n;
        uint32_t* vout = vals2 + ((pass + 1) & 1) * n;
        VK_LAUNCH(k_rs_count, dim3(n_blocks), dim3(256), 0, st, (const uint32_t*)kin, n, 8 * pass, counts, n_blocks);
        VK_LAUNCH(k_rs_scan, dim3(1), dim3(1024), 0, st, counts, (uint64_t)256 * n_blocks);
        VK_LAUNCH(k_rs_scatter, dim3(n_blocks), dim3(256), 0, st, (const uint32_t*)kin, (const uint32_t*)vin, n, 8 * pass, (const uint32_t*)counts, n_blocks, kout, vout);
    }
}

// keys/idx: 2 x n words each (in, out); sort_tmp: tracegen_mem_sort_scratch_bytes(n)
hipError_t launch_tracegen_mem(hipStream_t st, const TgMemOp* mem, uint64_t n, const uint32_t* static_cells, uint64_t n_static, uint32_t* keys2, uint32_t* idx2,
                               void* sort_tmp, size_t sort_tmp_bytes, DMatView t) {
    if (n) {
        ProfScope ps("k_tracegen_mem_sort", st, 16.0 * n + 4.0 * 8.0 * 2.0 * n);
        if (sort_tmp_bytes < tracegen_mem_sort_scratch_bytes(n)) return hipErrorInvalidValue;
        VK_LAUNCH(k_tg_mem_keys, dim3(blocks_for(n)), dim3(256), 0, st, mem, n, keys2, idx2);
        radix_sort_pairs(st, keys2, idx2, n, (uint32_t*)sort_tmp);
    }
    ProfScope ps("k_tracegen_mem", st, 20.0 * n + 4.0 * t.height * t.width);
    VK_LAUNCH(k_tracegen_mem, dim3(blocks_for(t.height)), dim3(256), 0, st, mem, (const uint32_t*)idx2, n, static_cells, n_static, t);
    return hipSuccess;
}

// counts: scratch of max(256, rom_len) words
hipError_t launch_tracegen_range(hipStream_t st, const TgCpuOp* ops, uint64_t n, const TgMemOp* mem, uint64_t n_mem, uint32_t* counts, DMatView t) {
    hipError_t e = hipMemsetAsync(counts, 0, 256 * 4, st);
    if (e != hipSuccess) return e;
    ProfScope ps("k_tracegen_tables", st, 48.0 * n + 8.0 * t.height);
    if (n) VK_LAUNCH(k_tg_range_histogram, dim3((unsigned)std::min<uint64_t>(1024, (n + 255) / 256)), dim3(256), 0, st, ops, n, mem, n_mem, counts);
    VK_LAUNCH(k_tracegen_counts, dim3(blocks_for(t.height)), dim3(256), 0, st, (const uint32_t*)counts, (uint64_t)256, 1, t);
    return hipSuccess;
}
hipError_t launch_tracegen_program(hipStream_t st, const TgCpuOp* ops, uint64_t n, uint64_t padded_n, uint32_t rom_len, uint32_t* counts, DMatView t) {
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)rom_len * 4, st);
    if (e != hipSuccess) return e;
    ProfScope ps("k_tracegen_tables", st, 48.0 * n + 4.0 * t.height);
    VK_LAUNCH(k_tg_pc_histogram, dim3((unsigned)((n + 256 * PC_ITEMS - 1) / (256 * PC_ITEMS))), dim3(256), 0, st, ops, n, padded_n, rom_len, counts);
    VK_LAUNCH(k_tracegen_counts, dim3(blocks_for(t.height)), dim3(256), 0, st, (const uint32_t*)counts, (uint64_t)rom_len, 0, t);
    return hipSuccess;
}

void launch_tracegen_idle(hipStream_t st, int mode, const uint32_t* static_cells, uint64_t n_static, DMatView t) {
    ProfScope ps("k_tracegen_tables", st, 4.0 * t.height * t.width);
    VK_LAUNCH(k_tracegen_idle, dim3(blocks_for(t.height)), dim3(256), 0, st, mode, static_cells, n_static, t);
}

void launch_tracegen_alu(hipStream_t st, int chip, const TgAluOp* ops, uint64_t n, DMatView t) {
    ProfScope ps("k_tracegen_alu", st, 16.0 * n + 4.0 * t.height * t.width);
    const dim3 g(blocks_for(t.height)), b(256);
    switch (chip) {
        case CHIP_ADD: VK_LAUNCH(k_tracegen_alu<CHIP_ADD>, g, b, 0, st, ops, n, t); break;
        case CHIP_SUB: VK_LAUNCH(k_tracegen_alu<CHIP_SUB>, g, b, 0, st, ops, n, t); break;
        case CHIP_LT: VK_LAUNCH(k_tracegen_alu<CHIP_LT>, g, b, 0, st, ops, n, t); break;
        default: VK_LAUNCH(k_tracegen_alu<CHIP_BITWISE>, g, b, 0, st, ops, n, t); break;
    }
}

void launch_tracegen_alu2(hipStream_t st, int chip, const TgAluOp* ops, uint64_t n, DMatView t) {
    ProfScope ps("k_tracegen_alu", st, 16.0 * n + 4.0 * t.height * t.width);
    const dim3 g(blocks_for(t.height)), b(256);
    switch (chip) {
        case CHIP_MUL: VK_LAUNCH(k_tracegen_alu2<CHIP_MUL>, g, b, 0, st, ops, n, t); break;
        case CHIP_DIV: VK_LAUNCH(k_tracegen_alu2<CHIP_DIV>, g, b, 0, st, ops, n, t); break;
        case CHIP_SHIFT: VK_LAUNCH(k_tracegen_alu2<CHIP_SHIFT>, g, b, 0, st, ops, n, t); break;
        default: VK_LAUNCH(k_tracegen_alu2<CHIP_COM>, g, b, 0, st, ops, n, t); break;
    }
}
void launch_tracegen_output(hipStream_t st, const TgOutOp* vals, const uint32_t* row0, uint64_t n, uint64_t n_rows, DMatView t) {
    ProfScope ps("k_tracegen_tables", st, 12.0 * n + 4.0 * t.height * t.width);
    VK_LAUNCH(k_tracegen_output, dim3(blocks_for(t.height)), dim3(256), 0, st, vals, row0, n, n_rows, t);
}

}  // namespace vk
