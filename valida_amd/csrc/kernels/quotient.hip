// Quotient-polynomial evaluation + decomposition on the device, realising
//   quotient_values          machine/src/quotient.rs:70-238
//   ProverConstraintFolder   machine/src/folding_builder.rs:32-125  (acc = acc*alpha + c per assert)
//   eval_permutation_constraints  machine/src/chip.rs:210-289
//   decompose_and_flatten    machine/src/quotient.rs:63-67 (Plonky3 uni-stark)
// for log_quotient_degree = 1 (every BasicMachine chip: max constraint degree 3).
//
// Two realisations of the chip's `Air::eval`, bit-identical in their values:
//   * the in-tree BasicMachine chips: their unchanged `eval` templates (chips/basic_machine.hpp) are instantiated over a
//     device-side folder and compiled ahead of time, one kernel per chip (k_quotient<3, CHIP>);
//   * AIRs captured at run time through the vgpu_air_* FFI: a linear register program (air/symbolic.hpp) interpreted per
//     row, register file in VGPRs (s_set_gpr_idx, <= 64 registers) or LDS (slot-major), program words by scalar loads.
// The Horner fold acc = acc*alpha + c is evaluated as sum_k alpha^(K-1-k) c_k with the alpha powers precomputed (same
// field element, 5 instead of 25 multiplications per constraint), accumulated lazily in the native kernels.
//
// Domain bookkeeping: the quotient domain s*H_{2n} is the first 2n storage rows of the bit-reversed
// LDE; storage row j <-> natural index bitrev(j).  Thread m owns storage rows 2m, 2m+1 = natural
// i = bitrev_k(m) and i + n, i.e. x and -x: exactly the pair decompose() butterflies, so the n x 10
// quotient-chunk row i is produced in place (written at position m = bit-reversed order, which is what
// the inverse NTT of the following commit consumes).
#include <cstdlib>
#include <stdexcept>
#include "launch.hpp"
#include "interactions.hpp"
#include "../chips/basic_machine.hpp"

namespace vk {

struct PointCtx {
    uint64_t row, next_row;  // storage rows of local / next
    Fp is_first, is_last, is_trans;
};

// Register file of the interpreter.  Register indices come from the program (wave-uniform), so a
// VGPR-resident 32-lane vector indexed through s_set_gpr_idx (no LDS round trip, no scratch) holds the
// registers: one bank for programs with <= 32 registers, two for <= 64; larger programs fall back to an
// LDS file (slot-major reg[slot][thread], conflict-free).
typedef uint32_t v32u __attribute__((ext_vector_type(32)));

#define VG_INTERP_BODY(GET, SET)                                                                                                        \
    switch (in.op) {                                                                                                                    \
        case vair::OP_CONST: SET(in.dst, (uint32_t)in.a | ((uint32_t)in.b << 16)); break;                                               \
        case vair::OP_LOAD_MAIN: SET(in.dst, in.flag ? a.main_nx[(uint64_t)in.a * a.main_lde.stride + p.next_row] : a.main_lde.data[(uint64_t)in.a * a.main_lde.stride + p.row]); break; \
        case vair::OP_LOAD_PREP: SET(in.dst, in.flag ? a.prep_nx[(uint64_t)in.a * a.prep_lde.stride + p.next_row] : a.prep_lde.data[(uint64_t)in.a * a.prep_lde.stride + p.row]); break; \
        case vair::OP_SEL_FIRST: SET(in.dst, p.is_first.v); break;                                                                      \
        case vair::OP_SEL_LAST: SET(in.dst, p.is_last.v); break;                                                                        \
        case vair::OP_SEL_TRANS: SET(in.dst, p.is_trans.v); break;                                                                      \
        case vair::OP_ADD: { uint32_t x = GET(in.a), y = GET(in.b); SET(in.dst, (Fp::raw(x) + Fp::raw(y)).v); } break;                  \
        case vair::OP_SUB: { uint32_t x = GET(in.a), y = GET(in.b); SET(in.dst, (Fp::raw(x) - Fp::raw(y)).v); } break;                  \
        case vair::OP_MUL: { uint32_t x = GET(in.a), y = GET(in.b); SET(in.dst, (Fp::raw(x) * Fp::raw(y)).v); } break;                  \
        case vair::OP_NEG: { uint32_t x = GET(in.a); SET(in.dst, (-Fp::raw(x)).v); } break;                                             \
        case vair::OP_ASSERT: { uint32_t x = GET(in.a); acc += ext_from_words(a.consts + 5 * k) * Fp::raw(x); k++; } break;             \
        default: break; /* OP_NOP padding */                                                                                            \
    }

// Interpret the chip program at one point; returns sum_k alpha_pow[k] * c_k over the chip's constraints.
// Instructions are fetched four at a time (programs are padded with NOPs to a multiple of 4).
template <int NB>
__device__ __forceinline__ Ext5 run_program_vgpr(const QuotientArgs& a, const PointCtx& p) {
    Ext5 acc = Ext5::zero();
    uint32_t k = 0;
    v32u r0 = {}, r1 = {};
#define VG_GET(i) (NB == 1 ? r0[(i) & 31] : ((i) < 32 ? r0[(i) & 31] : r1[(i) & 31]))
#define VG_SET(i, val) do { uint32_t _v = (val); if (NB == 1 || (i) < 32) r0[(i) & 31] = _v; else r1[(i) & 31] = _v; } while (0)
    for (uint32_t pc = 0; pc < a.n_instrs; pc += 4) {
        vair::Instr quad[4];
        __builtin_memcpy(quad, a.prog + pc, 32);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const vair::Instr in = quad[u];
            VG_INTERP_BODY(VG_GET, VG_SET)
        }
    }
#undef VG_GET
#undef VG_SET
    return acc;
}

__device__ __forceinline__ Ext5 run_program_lds(const QuotientArgs& a, const PointCtx& p, uint32_t* regs /* LDS, slot stride = blockDim.x */) {
    Ext5 acc = Ext5::zero();
    uint32_t k = 0;
    const uint32_t S = blockDim.x;
    uint32_t* r = regs + threadIdx.x;
#define VG_GET(i) (r[(uint32_t)(i) * S])
#define VG_SET(i, val) (r[(uint32_t)(i) * S] = (val))
    for (uint32_t pc = 0; pc < a.n_instrs; pc++) {
        const vair::Instr in = a.prog[pc];
        VG_INTERP_BODY(VG_GET, VG_SET)
    }
#undef VG_GET
#undef VG_SET
    return acc;
}

// ---- the chips' own `eval` templates, compiled for gfx950 ---------------------------------------------------
// ProverConstraintFolder (machine/src/folding_builder.rs:32-125) as a device-side AirBuilder: the BasicMachine
// chips' unchanged `eval` templates (chips/basic_machine.hpp) are instantiated over it, so every constraint
// becomes straight-line VALU code with column loads the compiler schedules and CSEs — no interpretation.
// assert_zero(c_k) accumulates alpha^(K-1-k) * c_k LAZILY: five v_mad_u64_u32 per constraint into 64-bit limb
// accumulators, one Montgomery reduction per limb for every four constraints (4 p^2 < 2^64).
#ifndef VGPU_QUOT_SADDR
#define VGPU_QUOT_SADDR 1  // 1: column loads as uniform base (SGPR pair: data + col * stride) + the point's 32-bit byte offset (global_load ... v_off, s[base]); 0: 64-bit per-lane addresses (A/B)
#endif
struct DeviceFolder {
    using Expr = Fp;
#if VGPU_QUOT_SADDR
    const uint32_t* __restrict__ main_p;  // main LDE column 0 (wave-uniform) for the local row / the next row; the point's rows as byte offsets
    const uint32_t* __restrict__ main_n;
    uint64_t mstride;
    const uint32_t* __restrict__ prep_p;
    const uint32_t* __restrict__ prep_n;
    uint64_t pstride;
    uint32_t roff, noff;
#else
    const uint32_t* __restrict__ main_p;  // main LDE column 0 at this point's local row / next row
    const uint32_t* __restrict__ main_n;
    uint64_t mstride;
    const uint32_t* __restrict__ prep_p;
    const uint32_t* __restrict__ prep_n;
    uint64_t pstride;
#endif
    Fp first, last, trans;
    const uint32_t* __restrict__ apow;  // alpha powers, 5 words per constraint (wave-uniform: scalar loads)
    uint64_t t[5];
    Ext5 total;
    int k, pending;
    __device__ __forceinline__ Fp constant(uint32_t c) const { return Fp::from_canonical(c); }
#if VGPU_QUOT_SADDR
    __device__ __forceinline__ Fp main(int col, bool next) const { return Fp::raw(load_at((next ? main_n : main_p) + (uint64_t)col * mstride, next ? noff : roff)); }
    __device__ __forceinline__ Fp preprocessed(int col, bool next) const { return Fp::raw(load_at((next ? prep_n : prep_p) + (uint64_t)col * pstride, next ? noff : roff)); }
#else
    __device__ __forceinline__ Fp main(int col, bool next) const { return Fp::raw((next ? main_n : main_p)[(uint64_t)col * mstride]); }
    __device__ __forceinline__ Fp preprocessed(int col, bool next) const { return Fp::raw((next ? prep_n : prep_p)[(uint64_t)col * pstride]); }
#endif
    __device__ __forceinline__ Fp is_first_row() const { return first; }
    __device__ __forceinline__ Fp is_last_row() const { return last; }
    __device__ __forceinline__ Fp is_transition() const { return trans; }
    __device__ __forceinline__ void flush() {
#pragma unroll
        for (int c = 0; c < 5; c++) { total.c[c] += Fp::raw(vg::monty_reduce_wide(t[c])); t[c] = 0; }
        pending = 0;
    }
    // (two alternating accumulator sets were measured in round 6: the cpu chip's kernel drops to 3 waves per SIMD and runs 7x slower — the
    // accumulators are not a dependency chain, the powers of alpha are precomputed; profiles/r06_ab_quotient_acc2_poseidon_waves.txt)
    __device__ __forceinline__ void assert_zero(const Fp& e) {
#pragma unroll
        for (int c = 0; c < 5; c++) t[c] += (uint64_t)apow[5 * k + c] * e.v;
        k++;
        if (++pending == 4) flush();
    }
};

// CHIP < 0: a chip without AIR constraints (program, mem, div, range) — only the permutation constraints.
template <int CHIP>
__device__ __forceinline__ Ext5 run_native(const QuotientArgs& a, const PointCtx& p) {
    if (CHIP < 0) return Ext5::zero();
    DeviceFolder f;
#if VGPU_QUOT_SADDR
    f.main_p = a.main_lde.data; f.main_n = a.main_nx; f.mstride = a.main_lde.stride;
    f.prep_p = a.prep_lde.data; f.prep_n = a.prep_nx; f.pstride = a.prep_lde.stride;
    f.roff = (uint32_t)p.row * 4u; f.noff = (uint32_t)p.next_row * 4u;  // LDE heights <= 2^27 rows (the field's two-adicity): the byte offset fits 32 bits
#else
    f.main_p = a.main_lde.data + p.row; f.main_n = a.main_nx + p.next_row; f.mstride = a.main_lde.stride;
    f.prep_p = a.prep_lde.data + p.row; f.prep_n = a.prep_nx + p.next_row; f.pstride = a.prep_lde.stride;
#endif
    f.first = p.is_first; f.last = p.is_last; f.trans = p.is_trans;
    f.apow = a.consts;
    f.total = Ext5::zero();
    f.k = 0; f.pending = 0;
#pragma unroll
    for (int c = 0; c < 5; c++) f.t[c] = 0;
    vchips::eval_chip(CHIP, f);  // CHIP is a compile-time constant: the switch folds to the one chip
    if (f.pending) f.flush();
    return f.total;
}

// eval_permutation_constraints at one point: M reciprocal constraints + transition/first/last.
__device__ __forceinline__ Ext5 load_ext_at(const uint32_t* ubase, uint64_t stride, uint32_t byte_off) {
    Ext5 e;
#pragma unroll
    for (int k = 0; k < 5; k++) e.c[k] = Fp::raw(load_at(ubase + (uint64_t)k * stride, byte_off));
    return e;
}
__device__ __forceinline__ Ext5 perm_constraints(const QuotientArgs& a, const PointCtx& p) {
    const uint32_t* iw = a.iw;
    const uint32_t M = iw[0], maxf = iw[1];
    const uint32_t* apow = a.consts + 5 * a.n_air_asserts;  // alpha powers for the perm constraints
    const uint32_t* bus = a.consts + 5 * a.K;
    const uint32_t* betas = bus + 5 * M;
    const Ext5 cumulative_sum = ext_from_words(betas + 5 * maxf);
    Ext5 acc = Ext5::zero(), rhs = Ext5::zero(), phi_0 = Ext5::zero();
    const uint32_t roff = (uint32_t)p.row * 4u, noff = (uint32_t)p.next_row * 4u;  // every column base below is wave-uniform: loads are base (SGPRs) + this offset
    for (uint32_t m = 0; m < M; m++) {
        uint32_t pos = iw[2 + m];
        const bool is_send = iw[pos] != 0;
        const uint32_t nf = iw[pos + 1];
        pos += 2;
        uint32_t pos_n = pos;
        Fp mult_local = eval_vcol_at(iw, pos, a.main_lde.data, a.main_lde.stride, a.prep_lde.data, a.prep_lde.stride, roff);
        Fp mult_next = eval_vcol_at(iw, pos_n, a.main_nx, a.main_lde.stride, a.prep_nx, a.prep_lde.stride, noff);
        // rlc = alpha_bus + sum_j beta^j f_j: Ext5 x base products accumulated four at a time per limb
        Ext5 rlc = ext_from_words(bus + 5 * m);
        for (uint32_t j0 = 0; j0 < nf; j0 += 4) {
            uint64_t t[5] = {0, 0, 0, 0, 0};
            const uint32_t je = nf - j0 < 4 ? nf - j0 : 4;
            for (uint32_t j = 0; j < je; j++) {
                Fp f = eval_vcol_at(iw, pos, a.main_lde.data, a.main_lde.stride, a.prep_lde.data, a.prep_lde.stride, roff);
#pragma unroll
                for (int c = 0; c < 5; c++) t[c] += (uint64_t)betas[5 * (j0 + j) + c] * f.v;
            }
#pragma unroll
            for (int c = 0; c < 5; c++) rlc.c[c] += Fp::raw(vg::monty_reduce_wide(t[c]));
        }
        const uint32_t* pcol = a.perm_lde.data + (uint64_t)(5 * m) * a.perm_lde.stride;
        Ext5 pl = load_ext_at(pcol, a.perm_lde.stride, roff), pn = load_ext_at(a.perm_nx + (uint64_t)(5 * m) * a.perm_lde.stride, a.perm_lde.stride, noff);
        acc += ext_from_words(apow + 5 * m) * (rlc * pl - Fp::one());  // assert_one_ext(rlc * perm_local[m])
        Ext5 tl = pl * mult_local, tn = pn * mult_next;
        if (is_send) { phi_0 += tl; rhs += tn; } else { phi_0 -= tl; rhs -= tn; }
    }
    const uint32_t* phicol = a.perm_lde.data + (uint64_t)(5 * M) * a.perm_lde.stride;
    Ext5 phi_local = load_ext_at(phicol, a.perm_lde.stride, roff), phi_next = load_ext_at(a.perm_nx + (uint64_t)(5 * M) * a.perm_lde.stride, a.perm_lde.stride, noff);
    acc += ext_from_words(apow + 5 * M) * (((phi_next - phi_local) - rhs) * p.is_trans);
    acc += ext_from_words(apow + 5 * (M + 1)) * ((phi_local - phi_0) * p.is_first);
    acc += ext_from_words(apow + 5 * (M + 2)) * ((phi_local - cumulative_sum) * p.is_last);
    return acc;
}

// ---- the same constraints for the in-tree chips with their interactions COMPILED IN (round 5) ---------------------------------------------
// perm_constraints above walks the encoded interactions: every column index comes from a scalar load that the column load has to wait for, one
// at a time — the mem chip (no AIR constraints at all, 48 loads per point) ran at 2.0 TB/s.  The BasicMachine's interactions are static
// (chips/basic_machine.hpp: visit_interactions, the one definition the host's Interaction lists are collected from as well), so for a native chip
// the visit is instantiated over this evaluator: every field is a load at a compile-time column, the whole point's loads are the compiler's to
// hoist and batch, the betas / bus alphas / alpha powers are scalar loads at compile-time offsets.  Same values (exact field arithmetic).
struct DevicePerm {
    const QuotientArgs& a;
    uint32_t roff, noff;
    const uint32_t* apow;
    const uint32_t* bus;
    const uint32_t* betas;
    Ext5 acc, rhs, phi_0, rlc;
    uint64_t t[5];
    int pend, m, j;
    bool send;
    __device__ __forceinline__ Fp value(const vchips::Lin& f, bool next) const {
        Fp v = Fp::from_canonical(f.k);
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (i < f.n) {
                const Fp x = Fp::raw(load_at((next ? a.main_nx : a.main_lde.data) + (uint64_t)f.col[i] * a.main_lde.stride, next ? noff : roff));
                v += f.w[i] == 1u ? x : x * Fp::from_canonical(f.w[i]);
            }
        return v;
    }
    __device__ __forceinline__ void flush() {
#pragma unroll
        for (int c = 0; c < 5; c++) { rlc.c[c] += Fp::raw(vg::monty_reduce_wide(t[c])); t[c] = 0; }
        pend = 0;
    }
    __device__ __forceinline__ void begin(bool is_send, int) {
        send = is_send;
        rlc = ext_from_words(bus + 5 * m);
#pragma unroll
        for (int c = 0; c < 5; c++) t[c] = 0;
        pend = 0; j = 0;
    }
    __device__ __forceinline__ void field(const vchips::Lin& f) {
        if (!(f.n == 0 && f.k == 0)) {  // a zero field adds nothing to the combination
            const Fp x = value(f, false);
#pragma unroll
            for (int c = 0; c < 5; c++) t[c] += (uint64_t)betas[5 * j + c] * x.v;
            if (++pend == 4) flush();
        }
        j++;
    }
    __device__ __forceinline__ void end(const vchips::Lin& count) {
        if (pend) flush();
        const Fp mult_local = value(count, false), mult_next = value(count, true);
        const Ext5 pl = load_ext_at(a.perm_lde.data + (uint64_t)(5 * m) * a.perm_lde.stride, a.perm_lde.stride, roff);
        const Ext5 pn = load_ext_at(a.perm_nx + (uint64_t)(5 * m) * a.perm_lde.stride, a.perm_lde.stride, noff);
        acc += ext_from_words(apow + 5 * m) * (rlc * pl - Fp::one());  // assert_one_ext(rlc * perm_local[m])
        const Ext5 tl = pl * mult_local, tn = pn * mult_next;
        if (send) { phi_0 += tl; rhs += tn; } else { phi_0 -= tl; rhs -= tn; }
        m++;
    }
};
template <int CHIP>
__device__ __forceinline__ Ext5 perm_constraints_native(const QuotientArgs& a, const PointCtx& p) {
    const uint32_t M = a.iw[0], maxf = a.iw[1];
    DevicePerm v{a, (uint32_t)p.row * 4u, (uint32_t)p.next_row * 4u, a.consts + 5 * a.n_air_asserts, a.consts + 5 * a.K, a.consts + 5 * a.K + 5 * M,
                 Ext5::zero(), Ext5::zero(), Ext5::zero(), Ext5::zero(), {0, 0, 0, 0, 0}, 0, 0, 0, false};
    vchips::visit_interactions(CHIP, v);  // CHIP is a compile-time constant: the visit unrolls into straight-line code; v.m ends at M
    const Ext5 cumulative_sum = ext_from_words(v.betas + 5 * maxf);
    const Ext5 phi_local = load_ext_at(a.perm_lde.data + (uint64_t)(5 * v.m) * a.perm_lde.stride, a.perm_lde.stride, v.roff);
    const Ext5 phi_next = load_ext_at(a.perm_nx + (uint64_t)(5 * v.m) * a.perm_lde.stride, a.perm_lde.stride, v.noff);
    Ext5 acc = v.acc;
    acc += ext_from_words(v.apow + 5 * v.m) * (((phi_next - phi_local) - v.rhs) * p.is_trans);
    acc += ext_from_words(v.apow + 5 * (v.m + 1)) * ((phi_local - v.phi_0) * p.is_first);
    acc += ext_from_words(v.apow + 5 * (v.m + 2)) * ((phi_local - cumulative_sum) * p.is_last);
    return acc;
}

// RFKIND 0: interpreter with an LDS register file, 1: one VGPR bank (<= 32 registers), 2: two banks (<= 64),
// 3: the chip's eval template compiled natively (CHIP = vchips::ChipId, or -1 for a chip without AIR constraints)
template <int RFKIND, int CHIP = -1>
__global__ void __launch_bounds__(256) k_quotient(QuotientArgs a, DeviceTables tb) {
    extern __shared__ uint32_t regs[];
    const uint64_t n = 1ull << a.log_n;
    // Which pair a thread owns.  The successor x g of a point is natural index + 2, i.e. in terms of the pair index m (bit b of m = natural
    // bit log_n - 1 - b) an increment at bit log_n - 2 with the carry running DOWN: the set {all 8 values of bits log_n - 2 .. log_n - 4}
    // x {32 consecutive low values} is closed under it except when those three bits are all ones.  A workgroup takes such a set (8 groups of
    // 32 pairs = 64 consecutive storage rows each): 7 of 8 `next` rows it reads are `local` rows of its own threads, read at about the same
    // time (the chip code touches a column's local and next value together) — they come from L1 / the XCD's L2 instead of HBM a second
    // time.  Work is only permuted: results are position for position the same.  Heights below 2^9 keep the linear map.
    uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
#ifndef VGPU_QUOT_TILE
#define VGPU_QUOT_TILE 1  // 0: the linear map (A/B builds, tools/build_variant.py)
#endif
    const bool tiled = VGPU_QUOT_TILE && a.log_n >= 9 && blockDim.x == 256;
    // Natural-order output (out_natural): the tile is 16 groups (pair-index bits log_n-1 .. log_n-4: the same carry chain, closed 7 times of 8)
    // x 16 consecutive low values, so that for a fixed low value the 16 groups are 16 CONSECUTIVE natural chunk rows
    // (natural index = bitrev(m): its low four bits are the group bits): after an LDS transpose the rows leave in 64-byte runs.
    const bool nat_tile = tiled && a.out_natural;
    if (nat_tile) {
        m = ((uint64_t)(threadIdx.x >> 4) << (a.log_n - 4)) | ((uint64_t)blockIdx.x << 4) | (threadIdx.x & 15u);
    } else if (tiled) {
        const uint32_t B = blockIdx.x, top = B >> (a.log_n - 9), mid = B & ((1u << (a.log_n - 9)) - 1u);
        m = ((uint64_t)top << (a.log_n - 1)) | ((uint64_t)(threadIdx.x >> 5) << (a.log_n - 4)) | ((uint64_t)mid << 5) | (threadIdx.x & 31u);
    }
    if (m >= n) return;  // never in a tiled launch (log_n >= 9: whole workgroups); the register-file LDS slots are thread-private
    const int kq = a.log_n + 1;
    const uint32_t Qmask = (uint32_t)(2 * n - 1);
    const uint32_t j0 = (uint32_t)(2 * m), j1 = j0 + 1;
    const uint32_t i0 = vg::reverse_bits_len((uint32_t)m, (unsigned)a.log_n);  // natural index of storage row 2m
    const uint32_t i1 = i0 + (uint32_t)n;                                      // natural index of storage row 2m+1
    const Fp s = Fp::raw(a.coset_shift), g_inv = Fp::raw(a.g_inv);
    const Fp x0 = s * domain_point(tb, j0), x1 = -x0;  // w_Q^{i0 + n} = -w_Q^{i0}
    // selectors: Z_H(x)/(x-1), Z_H(x)/(x-g^-1), x-g^-1  (quotient.rs:106-108,129-131); one shared inversion
    Fp d00 = x0 - Fp::one(), d01 = x0 - g_inv, d10 = x1 - Fp::one(), d11 = x1 - g_inv;
    Fp p01 = d00 * d01, p23 = d10 * d11;
    Fp inv_all = (p01 * p23).inv();
    Fp ip01 = inv_all * p23, ip23 = inv_all * p01;
    Fp id00 = ip01 * d01, id01 = ip01 * d00, id10 = ip23 * d11, id11 = ip23 * d10;
    const uint32_t par0 = i0 & 1u, par1 = i1 & 1u;
    PointCtx p0, p1;
    p0.row = j0; p0.next_row = vg::reverse_bits_len((i0 + a.next_step_p1 - 1u) & Qmask, (unsigned)kq);
    p1.row = j1; p1.next_row = vg::reverse_bits_len((i1 + a.next_step_p1 - 1u) & Qmask, (unsigned)kq);
    p0.is_trans = d01; p0.is_first = Fp::raw(a.zh[par0]) * id00; p0.is_last = Fp::raw(a.zh[par0]) * id01;
    p1.is_trans = d11; p1.is_first = Fp::raw(a.zh[par1]) * id10; p1.is_last = Fp::raw(a.zh[par1]) * id11;
    // The two points are evaluated one after the other (not unrolled): interleaving them only doubles the live
    // registers of the compiled chip code.
    Ext5 q0 = Ext5::zero(), q1 = Ext5::zero();
#pragma unroll 1
    for (int pt = 0; pt < 2; pt++) {
        PointCtx p;  // field-wise selects (a reference to p0 / p1 would put both in scratch)
        p.row = pt ? p1.row : p0.row; p.next_row = pt ? p1.next_row : p0.next_row;
        p.is_first = pt ? p1.is_first : p0.is_first; p.is_last = pt ? p1.is_last : p0.is_last; p.is_trans = pt ? p1.is_trans : p0.is_trans;
        Ext5 q;
        if (RFKIND == 3) q = run_native<CHIP>(a, p);
        else if (RFKIND == 0) q = run_program_lds(a, p, regs);
        else q = run_program_vgpr<RFKIND>(a, p);
        q = (q + perm_constraints(a, p)) * Fp::raw(a.zh_inv[pt ? par1 : par0]);
        if (pt) q1 = q; else q0 = q;
    }
    // decompose (App. B11): even = (a+b)/2, odd = (a-b)/(2 x0)
    Fp x0_inv = Fp::raw(a.coset_shift_inv) * inv_domain_point(tb, j0);
    Ext5 sum = q0 + q1, diff = (q0 - q1) * x0_inv;
    if (nat_tile) {
        // thread (group A, low r) holds chunk row i0 = bitrev4(r) << (log_n - 4) | bitrev(block) << 4 | bitrev4(A); it stages it at slot
        // (r << 4) | bitrev4(A), and thread t then stores slot t = the row bitrev4(t >> 4) << (log_n - 4) | bitrev(block) << 4 | (t & 15):
        // sixteen neighbouring lanes, sixteen consecutive rows
        __shared__ uint32_t tr[10 * 256];
        const uint32_t A = threadIdx.x >> 4, r = threadIdx.x & 15u, slot = (r << 4) | (__brev(A) >> 28);
#pragma unroll
        for (int c = 0; c < 5; c++) { tr[c * 256 + slot] = sum.c[c].halve().v; tr[(5 + c) * 256 + slot] = diff.c[c].halve().v; }
        __syncthreads();
        const uint64_t row = ((uint64_t)(__brev(A) >> 28) << (a.log_n - 4)) | ((uint64_t)(a.log_n > 8 ? __brev((uint32_t)blockIdx.x) >> (32 - (a.log_n - 8)) : 0u) << 4) | r;
#pragma unroll
        for (int c = 0; c < 10; c++) a.out.data[(uint64_t)c * a.out.stride + row] = tr[c * 256 + threadIdx.x];
        return;
    }
    const uint64_t pos = a.out_natural ? (uint64_t)i0 : m;  // small heights: natural position directly
#pragma unroll
    for (int c = 0; c < 5; c++) {
        a.out.data[(uint64_t)c * a.out.stride + pos] = sum.c[c].halve().v;
        a.out.data[(uint64_t)(5 + c) * a.out.stride + pos] = diff.c[c].halve().v;
    }
}

// ---- thread per POINT (round 4; the compiled chips) ------------------------------------------------------------------------------------
// k_quotient above gives a thread the PAIR (x, -x) = storage rows 2m, 2m + 1 and evaluates the two points one after the other: a wave's
// loads of a column then touch every second row of a 512-byte stretch, twice — once per point, a whole chip's worth of other columns apart:
// the second visit misses the caches, and the PMC passes see 2.0-2.2x the algorithmic bytes.  Here adjacent LANES take the two points of a
// pair: a wave reads 64 consecutive storage rows of a column in ONE fully used 256-byte request, every LDE word is fetched once, a thread
// holds one point's live values instead of being written for two.  The pair's decomposition (quotient.rs:63-67, App. B11: even = (a + b) / 2,
// odd = (a - b) / (2 x)) needs both values: one DPP quad_perm exchange of the five limbs, then the even lane forms the sum (chunk columns
// 0..4) and the odd lane the difference (columns 5..9).  Same successor-closed workgroup tiles as k_quotient — 8 groups x 32 pairs (16 x 16
// with the natural-order store) — now 512 threads.  One inversion per point instead of one per pair (+37 of some hundreds of products).
// Values identical to k_quotient's: every parity test runs through it.  VGPU_QUOT_PER_POINT=0 (environment) selects the pair kernel (A/B).
#ifndef VGPU_QUOT_PT_WAVES
#define VGPU_QUOT_PT_WAVES 6  // waves per SIMD the register allocation must allow (0: the compiler's choice — 4 for the cpu chip's 105 VGPRs); A/B builds
#endif
template <int CHIP>
__global__ void __launch_bounds__(512)
#if VGPU_QUOT_PT_WAVES > 0
__attribute__((amdgpu_waves_per_eu(VGPU_QUOT_PT_WAVES)))
#endif
k_quotient_pt(QuotientArgs a, DeviceTables tb) {
    const uint64_t n = 1ull << a.log_n;
    const uint32_t T = threadIdx.x >> 1, pt = threadIdx.x & 1u;  // the pair's slot in the workgroup, and which of its two points
    uint64_t m = (uint64_t)blockIdx.x * 256 + T;
    const bool tiled = a.log_n >= 9;
    const bool nat_tile = tiled && a.out_natural;
    if (nat_tile) {
        m = ((uint64_t)(T >> 4) << (a.log_n - 4)) | ((uint64_t)blockIdx.x << 4) | (T & 15u);
    } else if (tiled) {
        const uint32_t B = blockIdx.x, top = B >> (a.log_n - 9), mid = B & ((1u << (a.log_n - 9)) - 1u);
        m = ((uint64_t)top << (a.log_n - 1)) | ((uint64_t)(T >> 5) << (a.log_n - 4)) | ((uint64_t)mid << 5) | (T & 31u);
    }
    if (m >= n) return;  // both lanes of a pair leave together; never in a tiled launch
    const int kq = a.log_n + 1;
    const uint32_t Qmask = (uint32_t)(2 * n - 1);
    const uint32_t j = (uint32_t)(2 * m) + pt;                                            // storage row of this point
    const uint32_t i0 = vg::reverse_bits_len((uint32_t)m, (unsigned)a.log_n);           // natural index of the pair's even row
    const uint32_t i = i0 + pt * (uint32_t)n;                                             // ... of this point
    const Fp s = Fp::raw(a.coset_shift), g_inv = Fp::raw(a.g_inv);
    const Fp x0 = s * domain_point(tb, (uint32_t)(2 * m));
    const Fp x = pt ? -x0 : x0;                                                           // w_Q^(i0 + n) = -w_Q^i0
    const Fp d0 = x - Fp::one(), d1 = x - g_inv;
    const Fp inv = (d0 * d1).inv();
    const uint32_t par = i & 1u;
    PointCtx p;
    p.row = j;
    p.next_row = vg::reverse_bits_len((i + a.next_step_p1 - 1u) & Qmask, (unsigned)kq);
    p.is_trans = d1;
    p.is_first = Fp::raw(a.zh[par]) * (inv * d1);
    p.is_last = Fp::raw(a.zh[par]) * (inv * d0);
#ifndef VGPU_QUOT_PERM_NATIVE
#define VGPU_QUOT_PERM_NATIVE 1  // 0: the descriptor-driven permutation constraints (A/B builds)
#endif
    Ext5 q = run_native<CHIP>(a, p);
    q = (q + (VGPU_QUOT_PERM_NATIVE ? perm_constraints_native<CHIP>(a, p) : perm_constraints(a, p))) * Fp::raw(a.zh_inv[par]);
    Ext5 o;  // the partner lane's value
#pragma unroll
    for (int c = 0; c < 5; c++) o.c[c] = Fp::raw((uint32_t)__builtin_amdgcn_update_dpp(0, (int)q.c[c].v, 0xB1, 0xF, 0xF, true));
    const Fp x0_inv = Fp::raw(a.coset_shift_inv) * inv_domain_point(tb, (uint32_t)(2 * m));
    Fp outv[5];
#pragma unroll
    for (int c = 0; c < 5; c++) outv[c] = (pt ? (o.c[c] - q.c[c]) * x0_inv : q.c[c] + o.c[c]).halve();  // odd lane: (q0 - q1) / (2 x0); even lane: (q0 + q1) / 2
    const uint32_t col0 = 5u * pt;
    if (nat_tile) {
        __shared__ uint32_t tr[10 * 256];
        const uint32_t A = T >> 4, r = T & 15u, slot = (r << 4) | (__brev(A) >> 28);
#pragma unroll
        for (int c = 0; c < 5; c++) tr[(col0 + c) * 256 + slot] = outv[c].v;
        __syncthreads();
        // 512 threads store the 10 x 256 tile: the first 256 the sum columns, the others the difference columns; slot t <-> row as in k_quotient
        const uint32_t t = threadIdx.x & 255u, half = threadIdx.x >> 8, A2 = t >> 4, r2 = t & 15u;
        const uint64_t row = ((uint64_t)(__brev(A2) >> 28) << (a.log_n - 4)) | ((uint64_t)(a.log_n > 8 ? __brev((uint32_t)blockIdx.x) >> (32 - (a.log_n - 8)) : 0u) << 4) | r2;
#pragma unroll
        for (int c = 0; c < 5; c++) a.out.data[(uint64_t)(5 * half + c) * a.out.stride + row] = tr[(5 * half + c) * 256 + t];
        return;
    }
    const uint64_t pos = a.out_natural ? (uint64_t)i0 : m;
#pragma unroll
    for (int c = 0; c < 5; c++) a.out.data[(uint64_t)(col0 + c) * a.out.stride + pos] = outv[c].v;
}

// ---- log_quotient_degree >= 2 (AIRs of degree 4..9 captured through vgpu_air_*; no chip of the reference needs it) -------------
// The quotient domain s*H_{Qn}, Q = 2^lqd, is the first Qn storage rows; thread m owns storage rows Qm .. Qm+Q-1 = natural indices
// i0 + n bitrev_lqd(r), i.e. the Q points x0 w_Q^{bitrev(r)}: every pair decompose() butterflies at any level of its recursion
// (machine/src/quotient.rs:63-67 -> p3_uni_stark::decompose, App. B11) lives in this thread.  At each level adjacent storage
// entries are (f(y), f(-y)): even = (a + b) / 2, odd = (a - b) / (2 y); the evens (then the odds) of a group are again in
// bit-reversed order of the squared domain, so the same step recurses with y^2.  Chunk order = decompose(even) ++ decompose(odd).
template <int RFKIND>
__global__ void __launch_bounds__(256) k_quotient_general(QuotientArgs a, DeviceTables tb) {
    extern __shared__ uint32_t regs[];
    constexpr int QMAX = 8;
    const int lqd = a.lqd, Q = 1 << lqd;
    const uint64_t n = 1ull << a.log_n;
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    const int kq = a.log_n + lqd;
    const uint32_t Qmask = (uint32_t)(((uint64_t)Q << a.log_n) - 1);
    const uint32_t i0 = vg::reverse_bits_len((uint32_t)m, (unsigned)a.log_n);
    const Fp s = Fp::raw(a.coset_shift), g_inv = Fp::raw(a.g_inv);
    Ext5 v[QMAX];
#pragma unroll 1
    for (int r = 0; r < Q; r++) {
        const uint32_t j = (uint32_t)(Q * m) + (uint32_t)r;                                       // storage row
        const uint32_t i = i0 + (uint32_t)(n * vg::reverse_bits_len((uint32_t)r, (unsigned)lqd));  // its natural index
        const Fp x = s * domain_point(tb, j);
        const Fp zh = Fp::raw(a.zh[i & (uint32_t)(Q - 1)]);
        PointCtx p;
        p.row = j;
        p.next_row = vg::reverse_bits_len((i + (uint32_t)Q) & Qmask, (unsigned)kq);
        p.is_trans = x - g_inv;
        p.is_first = zh * (x - Fp::one()).inv();
        p.is_last = zh * p.is_trans.inv();
        Ext5 q = RFKIND == 0 ? run_program_lds(a, p, regs) : run_program_vgpr<(RFKIND == 0 ? 1 : RFKIND)>(a, p);
        v[r] = (q + perm_constraints(a, p)) * Fp::raw(a.zh_inv[i & (uint32_t)(Q - 1)]);
    }
    // recursive even / odd decomposition, in registers
    Fp ybase = s * domain_point(tb, (uint32_t)(Q * m));  // x0 of this thread; squared at every level
    for (int level = 0; level < lqd; level++) {
        const int G = Q >> level;                         // group size at this level; groups are contiguous
        const int lg = lqd - level;
        Ext5 t[QMAX];
        for (int base = 0; base < Q; base += G)
            for (int u = 0; u < G / 2; u++) {
                // y = ybase * w_G^{bitrev_lg(2u)}; only its inverse is needed
                const Fp w = vg::two_adic_generator((unsigned)lg).pow(vg::reverse_bits_len((uint32_t)(2 * u), (unsigned)lg));
                const Fp yinv = (ybase * w).inv();
                const Ext5 av = v[base + 2 * u], bv = v[base + 2 * u + 1];
                Ext5 e = av + bv, o = (av - bv) * yinv;
                for (int c = 0; c < 5; c++) { e.c[c] = e.c[c].halve(); o.c[c] = o.c[c].halve(); }
                t[base + u] = e;
                t[base + G / 2 + u] = o;
            }
        for (int r = 0; r < Q; r++) v[r] = t[r];
        ybase = ybase * ybase;
    }
    for (int ch = 0; ch < Q; ch++)
        for (int c = 0; c < 5; c++) a.out.data[(uint64_t)(5 * ch + c) * a.out.stride + m] = v[ch].c[c].v;
}

// ---- debug check of the witness (machine/src/check_constraints.rs via basic/src/lib.rs:270-372, debug builds) ---------
// Every constraint evaluated on the TRACE domain itself: row r, next = r + 1 mod n, is_first / is_last / is_transition as
// 0/1 selectors (DebugConstraintBuilder, machine/src/debug_builder.rs:7-114).  The first failing (row, constraint) is
// reported through an atomicMin on (row << 16 | code): code = index of the chip's AIR constraint (native chips),
// 0xFFFD = some AIR constraint of an interpreted program, 0xFFFE = a permutation / running-sum constraint.
struct DebugFolder {
    using Expr = Fp;
    const uint32_t* __restrict__ main_p;
    const uint32_t* __restrict__ main_n;
    uint64_t mstride;
    const uint32_t* __restrict__ prep_p;
    const uint32_t* __restrict__ prep_n;
    uint64_t pstride;
    Fp first, last, trans;
    uint32_t k, bad;
    __device__ __forceinline__ Fp constant(uint32_t c) const { return Fp::from_canonical(c); }
    __device__ __forceinline__ Fp main(int col, bool next) const { return Fp::raw((next ? main_n : main_p)[(uint64_t)col * mstride]); }
    __device__ __forceinline__ Fp preprocessed(int col, bool next) const { return Fp::raw((next ? prep_n : prep_p)[(uint64_t)col * pstride]); }
    __device__ __forceinline__ Fp is_first_row() const { return first; }
    __device__ __forceinline__ Fp is_last_row() const { return last; }
    __device__ __forceinline__ Fp is_transition() const { return trans; }
    __device__ __forceinline__ void assert_zero(const Fp& e) { if (!e.is_zero() && bad == 0xffffffffu) bad = k; k++; }
};

template <int CHIP>  // vchips::ChipId, -1 = no AIR constraints, QuotientArgs::INTERPRET = interpreted program (LDS register file)
__global__ void __launch_bounds__(256) k_check_constraints(QuotientArgs a, unsigned long long* __restrict__ first_bad) {
    extern __shared__ uint32_t regs[];
    const uint64_t n = 1ull << a.log_n;
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    PointCtx p;
    p.row = r; p.next_row = (r + 1) & (n - 1);
    p.is_first = r == 0 ? Fp::one() : Fp::zero();
    p.is_last = r == n - 1 ? Fp::one() : Fp::zero();
    p.is_trans = r == n - 1 ? Fp::zero() : Fp::one();
    uint32_t code = 0xffffffffu;
    if (CHIP == QuotientArgs::INTERPRET) {
        if (!run_program_lds(a, p, regs).is_zero()) code = 0xFFFDu;  // random fold of all AIR constraints
    } else if (CHIP >= 0) {
        DebugFolder f;
        f.main_p = a.main_lde.data + p.row; f.main_n = a.main_nx + p.next_row; f.mstride = a.main_lde.stride;
        f.prep_p = a.prep_lde.data + p.row; f.prep_n = a.prep_nx + p.next_row; f.pstride = a.prep_lde.stride;
        f.first = p.is_first; f.last = p.is_last; f.trans = p.is_trans;
        f.k = 0; f.bad = 0xffffffffu;
        vchips::eval_chip(CHIP, f);
        code = f.bad;
    }
    if (code == 0xffffffffu && !perm_constraints(a, p).is_zero()) code = 0xFFFEu;
    if (code != 0xffffffffu) atomicMin(first_bad, (unsigned long long)((r << 16) | code));
}

void launch_check_constraints(hipStream_t st, const QuotientArgs& a_in, unsigned long long* first_bad_dev) {
    const QuotientArgs a = a_in.normalised();  // next rows come from the same traces
    const uint64_t n = 1ull << a.log_n;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    ProfScope ps("k_check_constraints", st, 4.0 * n * (a.main_lde.width + a.perm_lde.width + a.prep_lde.width));
    switch (a.native_chip) {
#define VG_CHECK(C) case vchips::C: VK_LAUNCH((k_check_constraints<vchips::C>), grid, block, 0, st, a, first_bad_dev); break;
        VG_CHECK(CHIP_CPU) VG_CHECK(CHIP_ADD) VG_CHECK(CHIP_SUB) VG_CHECK(CHIP_MUL) VG_CHECK(CHIP_SHIFT) VG_CHECK(CHIP_LT)
        VG_CHECK(CHIP_COM) VG_CHECK(CHIP_BITWISE) VG_CHECK(CHIP_OUTPUT) VG_CHECK(CHIP_STATIC_DATA)
#undef VG_CHECK
        case QuotientArgs::INTERPRET: {
            unsigned threads = 256;
            while (threads > 64 && (size_t)a.n_regs * threads * 4 > 64 * 1024) threads >>= 1;
            VK_LAUNCH((k_check_constraints<QuotientArgs::INTERPRET>), dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), (size_t)a.n_regs * threads * 4, st, a,
                               first_bad_dev);
            break;
        }
        default: VK_LAUNCH((k_check_constraints<-1>), grid, block, 0, st, a, first_bad_dev); break;
    }
}

void launch_quotient(hipStream_t st, const QuotientArgs& a_in, const DeviceTables& tb) {
    const QuotientArgs a = a_in.normalised();  // one GPU: next rows from the same LDEs, 2^lqd further in natural index
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_quotient<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 148 * 1024);  // 160 KiB minus the 10 KiB static transpose tile of the natural-order store
        (void)hipFuncSetAttribute((const void*)k_quotient_general<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    uint64_t n = 1ull << a.log_n;
    if (a.lqd != 1) {  // higher-degree AIRs: the interpreted program over 2^lqd points per thread
        if (a.lqd < 1 || a.lqd > 3 || a.native_chip != QuotientArgs::INTERPRET) throw std::runtime_error("quotient: log_quotient_degree must be 1..3 (and above 1 only for interpreted AIRs)");
        ProfScope ps("k_quotient", st, 4.0 * n * ((double)(1 << a.lqd) * (a.main_lde.width + a.perm_lde.width + a.prep_lde.width) + 5.0 * (1 << a.lqd)));
        unsigned threads = 256;
        if (a.n_regs <= 64) {
            dim3 grid((unsigned)((n + threads - 1) / threads));
            if (a.n_regs <= 32) VK_LAUNCH(k_quotient_general<1>, grid, dim3(threads), 0, st, a, tb);
            else VK_LAUNCH(k_quotient_general<2>, grid, dim3(threads), 0, st, a, tb);
            return;
        }
        while (threads > 64 && (size_t)a.n_regs * threads * 4 > 64 * 1024) threads >>= 1;
        VK_LAUNCH(k_quotient_general<0>, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), (size_t)a.n_regs * threads * 4, st, a, tb);
        return;
    }
    static const char* names[16] = {"k_quotient.cpu", "k_quotient.program", "k_quotient.mem", "k_quotient.add", "k_quotient.sub", "k_quotient.mul", "k_quotient.div",
                                    "k_quotient.shift", "k_quotient.lt", "k_quotient.com", "k_quotient.bitwise", "k_quotient.output", "k_quotient.range",
                                    "k_quotient.static_data", "k_quotient", "k_quotient"};
    static const bool per_point = [] { const char* e = getenv("VGPU_QUOT_PER_POINT"); return !(e && e[0] == '0'); }();
    const char* pname = getenv("VGPU_PROF_QUOTIENT_BY_CHIP") && a.native_chip >= 0 && a.native_chip < 14 ? names[a.native_chip]
                        : (per_point && a.native_chip != QuotientArgs::INTERPRET ? "k_quotient_pt" : "k_quotient");  // rocprof's kernel names
    ProfScope ps(pname, st, 4.0 * n * (2.0 * (a.main_lde.width + a.perm_lde.width + a.prep_lde.width) + 10.0));
    if (a.native_chip != QuotientArgs::INTERPRET && per_point) {
        // thread per point, 512 threads = 256 pairs per workgroup (k_quotient_pt)
        const dim3 grid((unsigned)((n + 255) / 256)), block(512);
        switch (a.native_chip) {
#define VG_NATIVE_PT(C) case vchips::C: VK_LAUNCH((k_quotient_pt<vchips::C>), grid, block, 0, st, a, tb); break;
            VG_NATIVE_PT(CHIP_CPU) VG_NATIVE_PT(CHIP_ADD) VG_NATIVE_PT(CHIP_SUB) VG_NATIVE_PT(CHIP_MUL) VG_NATIVE_PT(CHIP_SHIFT) VG_NATIVE_PT(CHIP_LT)
            VG_NATIVE_PT(CHIP_COM) VG_NATIVE_PT(CHIP_BITWISE) VG_NATIVE_PT(CHIP_OUTPUT) VG_NATIVE_PT(CHIP_STATIC_DATA)
            VG_NATIVE_PT(CHIP_PROGRAM) VG_NATIVE_PT(CHIP_MEM) VG_NATIVE_PT(CHIP_DIV) VG_NATIVE_PT(CHIP_RANGE)  // no AIR constraints: their interactions alone
#undef VG_NATIVE_PT
            default: throw std::logic_error("quotient: a native chip id outside the BasicMachine");
        }
        return;
    }
    if (a.native_chip != QuotientArgs::INTERPRET) {
        // the BasicMachine chips: eval compiled ahead of time, one kernel per chip with constraints
        const dim3 grid((unsigned)((n + 255) / 256)), block(256);
        switch (a.native_chip) {
#define VG_NATIVE(C) case vchips::C: VK_LAUNCH((k_quotient<3, vchips::C>), grid, block, 0, st, a, tb); break;
            VG_NATIVE(CHIP_CPU) VG_NATIVE(CHIP_ADD) VG_NATIVE(CHIP_SUB) VG_NATIVE(CHIP_MUL) VG_NATIVE(CHIP_SHIFT) VG_NATIVE(CHIP_LT)
            VG_NATIVE(CHIP_COM) VG_NATIVE(CHIP_BITWISE) VG_NATIVE(CHIP_OUTPUT) VG_NATIVE(CHIP_STATIC_DATA)
#undef VG_NATIVE
            default: VK_LAUNCH((k_quotient<3, -1>), grid, block, 0, st, a, tb); break;
        }
        return;
    }
    if (a.n_regs <= 64) {
        const unsigned threads = 256;
        dim3 grid((unsigned)((n + threads - 1) / threads));
        if (a.n_regs <= 32) VK_LAUNCH(k_quotient<1>, grid, dim3(threads), 0, st, a, tb);
        else VK_LAUNCH(k_quotient<2>, grid, dim3(threads), 0, st, a, tb);
        return;
    }
    unsigned threads = 256;
    while (threads > 64 && (size_t)a.n_regs * threads * 4 > 64 * 1024) threads >>= 1;
    size_t lds = (size_t)a.n_regs * threads * 4;
    VK_LAUNCH(k_quotient<0>, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), lds, st, a, tb);
}

}  // namespace vk
