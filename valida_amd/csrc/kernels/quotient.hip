// Quotient-polynomial evaluation + decomposition on the device, realising
//   quotient_values          machine/src/quotient.rs:70-238
//   ProverConstraintFolder   machine/src/folding_builder.rs:32-125  (acc = acc*alpha + c per assert)
//   eval_permutation_constraints  machine/src/chip.rs:210-289
//   decompose_and_flatten    machine/src/quotient.rs:63-67 (Plonky3 uni-stark)
// for log_quotient_degree = 1 (every BasicMachine chip: max constraint degree 3).
//
// The chip's `Air::eval` arrives as a linear register program (air/symbolic.hpp) compiled once from
// the chip's unchanged constraint definitions; every thread interprets it on its row with the
// register file in LDS (slot-major: reg[slot][thread], bank-conflict free) and program words fetched
// by scalar loads.  The Horner fold acc = acc*alpha + c is evaluated as sum_k alpha^(K-1-k) c_k with
// the alpha powers precomputed (same field element, 5 instead of 25 multiplications per constraint).
//
// Domain bookkeeping: the quotient domain s*H_{2n} is the first 2n storage rows of the bit-reversed
// LDE; storage row j <-> natural index bitrev(j).  Thread m owns storage rows 2m, 2m+1 = natural
// i = bitrev_k(m) and i + n, i.e. x and -x: exactly the pair decompose() butterflies, so the n x 10
// quotient-chunk row i is produced in place (written at position m = bit-reversed order, which is what
// the inverse NTT of the following commit consumes).
#include "launch.hpp"
#include "interactions.hpp"

namespace vk {

struct PointCtx {
    uint64_t row, next_row;  // storage rows of local / next
    Fp is_first, is_last, is_trans;
};

// Interpret the chip program at one point; returns sum_k alpha_pow[k] * c_k over the chip's constraints.
__device__ __forceinline__ Ext5 run_program(const QuotientArgs& a, const PointCtx& p, uint32_t* regs /* LDS, slot stride = blockDim.x */) {
    Ext5 acc = Ext5::zero();
    const uint32_t S = blockDim.x;
    uint32_t* r = regs + threadIdx.x;
    uint32_t k = 0;
    for (uint32_t pc = 0; pc < a.n_instrs; pc++) {
        const vair::Instr in = a.prog[pc];
        switch (in.op) {
            case vair::OP_CONST: r[in.dst * S] = (uint32_t)in.a | ((uint32_t)in.b << 16); break;
            case vair::OP_LOAD_MAIN: r[in.dst * S] = a.main_lde.data[(uint64_t)in.a * a.main_lde.stride + (in.flag ? p.next_row : p.row)]; break;
            case vair::OP_LOAD_PREP: r[in.dst * S] = a.prep_lde.data[(uint64_t)in.a * a.prep_lde.stride + (in.flag ? p.next_row : p.row)]; break;
            case vair::OP_SEL_FIRST: r[in.dst * S] = p.is_first.v; break;
            case vair::OP_SEL_LAST: r[in.dst * S] = p.is_last.v; break;
            case vair::OP_SEL_TRANS: r[in.dst * S] = p.is_trans.v; break;
            case vair::OP_ADD: r[in.dst * S] = (Fp::raw(r[in.a * S]) + Fp::raw(r[in.b * S])).v; break;
            case vair::OP_SUB: r[in.dst * S] = (Fp::raw(r[in.a * S]) - Fp::raw(r[in.b * S])).v; break;
            case vair::OP_MUL: r[in.dst * S] = (Fp::raw(r[in.a * S]) * Fp::raw(r[in.b * S])).v; break;
            case vair::OP_NEG: r[in.dst * S] = (-Fp::raw(r[in.a * S])).v; break;
            case vair::OP_ASSERT: acc += ext_from_words(a.consts + 5 * k) * Fp::raw(r[in.a * S]); k++; break;
        }
    }
    return acc;
}

// eval_permutation_constraints at one point: M reciprocal constraints + transition/first/last.
__device__ __forceinline__ Ext5 perm_constraints(const QuotientArgs& a, const PointCtx& p) {
    const uint32_t* iw = a.iw;
    const uint32_t M = iw[0], maxf = iw[1];
    const uint32_t* apow = a.consts + 5 * a.n_air_asserts;  // alpha powers for the perm constraints
    const uint32_t* bus = a.consts + 5 * a.K;
    const uint32_t* betas = bus + 5 * M;
    const Ext5 cumulative_sum = ext_from_words(betas + 5 * maxf);
    Ext5 acc = Ext5::zero(), rhs = Ext5::zero(), phi_0 = Ext5::zero();
    for (uint32_t m = 0; m < M; m++) {
        uint32_t pos = iw[2 + m];
        const bool is_send = iw[pos] != 0;
        const uint32_t nf = iw[pos + 1];
        pos += 2;
        uint32_t pos_n = pos;
        Fp mult_local = eval_vcol(iw, pos, a.main_lde.data, a.main_lde.stride, a.prep_lde.data, a.prep_lde.stride, p.row);
        Fp mult_next = eval_vcol(iw, pos_n, a.main_lde.data, a.main_lde.stride, a.prep_lde.data, a.prep_lde.stride, p.next_row);
        Ext5 rlc = ext_from_words(bus + 5 * m);
        for (uint32_t j = 0; j < nf; j++) {
            Fp f = eval_vcol(iw, pos, a.main_lde.data, a.main_lde.stride, a.prep_lde.data, a.prep_lde.stride, p.row);
            rlc += ext_from_words(betas + 5 * j) * f;
        }
        const uint32_t* pcol = a.perm_lde.data + (uint64_t)(5 * m) * a.perm_lde.stride;
        Ext5 pl = load_ext(pcol, a.perm_lde.stride, p.row), pn = load_ext(pcol, a.perm_lde.stride, p.next_row);
        acc += ext_from_words(apow + 5 * m) * (rlc * pl - Fp::one());  // assert_one_ext(rlc * perm_local[m])
        Ext5 tl = pl * mult_local, tn = pn * mult_next;
        if (is_send) { phi_0 += tl; rhs += tn; } else { phi_0 -= tl; rhs -= tn; }
    }
    const uint32_t* phicol = a.perm_lde.data + (uint64_t)(5 * M) * a.perm_lde.stride;
    Ext5 phi_local = load_ext(phicol, a.perm_lde.stride, p.row), phi_next = load_ext(phicol, a.perm_lde.stride, p.next_row);
    acc += ext_from_words(apow + 5 * M) * (((phi_next - phi_local) - rhs) * p.is_trans);
    acc += ext_from_words(apow + 5 * (M + 1)) * ((phi_local - phi_0) * p.is_first);
    acc += ext_from_words(apow + 5 * (M + 2)) * ((phi_local - cumulative_sum) * p.is_last);
    return acc;
}

__global__ void k_quotient(QuotientArgs a, DeviceTables tb) {
    extern __shared__ uint32_t regs[];
    const uint64_t n = 1ull << a.log_n;
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;  // no block-level synchronisation below: LDS slots are thread-private
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
    p0.row = j0; p0.next_row = vg::reverse_bits_len((i0 + 2u) & Qmask, (unsigned)kq);
    p1.row = j1; p1.next_row = vg::reverse_bits_len((i1 + 2u) & Qmask, (unsigned)kq);
    p0.is_trans = d01; p0.is_first = Fp::raw(a.zh[par0]) * id00; p0.is_last = Fp::raw(a.zh[par0]) * id01;
    p1.is_trans = d11; p1.is_first = Fp::raw(a.zh[par1]) * id10; p1.is_last = Fp::raw(a.zh[par1]) * id11;
    Ext5 q0 = (run_program(a, p0, regs) + perm_constraints(a, p0)) * Fp::raw(a.zh_inv[par0]);
    Ext5 q1 = (run_program(a, p1, regs) + perm_constraints(a, p1)) * Fp::raw(a.zh_inv[par1]);
    // decompose (App. B11): even = (a+b)/2, odd = (a-b)/(2 x0)
    Fp x0_inv = Fp::raw(a.coset_shift_inv) * inv_domain_point(tb, j0);
    Ext5 sum = q0 + q1, diff = (q0 - q1) * x0_inv;
#pragma unroll
    for (int c = 0; c < 5; c++) {
        a.out.data[(uint64_t)c * a.out.stride + m] = sum.c[c].halve().v;
        a.out.data[(uint64_t)(5 + c) * a.out.stride + m] = diff.c[c].halve().v;
    }
}

void launch_quotient(hipStream_t st, const QuotientArgs& a, const DeviceTables& tb) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_quotient, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    unsigned threads = 256;
    while (threads > 64 && (size_t)a.n_regs * threads * 4 > 64 * 1024) threads >>= 1;
    size_t lds = (size_t)(a.n_regs ? a.n_regs : 1) * threads * 4;
    uint64_t n = 1ull << a.log_n;
    ProfScope ps("k_quotient", st, 4.0 * n * (2.0 * (a.main_lde.width + a.perm_lde.width + a.prep_lde.width) + 10.0));
    hipLaunchKernelGGL(k_quotient, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), lds, st, a, tb);
}

}  // namespace vk
