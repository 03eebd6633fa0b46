// The in-register radix-2 butterfly network shared by the NTT kernels (ntt.hip) and the Poseidon-16 MDS layer (poseidon_mmcs.hip: a
// 16-point cyclic convolution as DIF transform, pointwise product, DIT transform).  Reference: the DFTs behind
// `pcs.commit_batches` (basic/src/lib.rs:199,223,258,599 -> Plonky3 Radix2Dit / coset_lde_batch) and CosetMds<16>
// (basic/tests/test_prover.rs:418-422).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "../field.hpp"

namespace vk {
using vg::Fp;

// The R stages of one round on the 2^R points a work item holds in registers.  `low` = the item's low index bits (below the round's
// stages), lowbits their count: stage s = lowbits + 1 + st pairs the points differing in bit st with the twiddle w_{2^s}^(low + (k << lowbits)).
// LB0 (lowbits == 0, i.e. stages 1 .. R: the first DIT / last DIF round): the twiddle of butterfly p is w^(p & (half - 1)) with a compile-time
// exponent, and exponent 0 is the factor 1 — 15 of a radix-16 round's 32 butterflies need no multiplication at all.
#ifndef VGPU_NTT_LAZY
#define VGPU_NTT_LAZY 1  // 0: every butterfly output fully reduced (A/B builds)
#endif
// `held`: the round's 2^R - 1 twiddles already in registers, in load order (stage st at offset 2^st - 1) — a persistent block whose
// work items sit at the same tile position for every tile loads them once (k_lde_mid12's outer rounds)
template <int R, bool DIT, bool LB0>
__device__ __forceinline__ void butterflies(Fp (&x)[1 << R], const uint32_t* tw, int low, int lowbits, const Fp* held = nullptr) {
    constexpr int G = 1 << R;
#pragma unroll
    for (int step = 0; step < R; step++) {
        const int st = DIT ? step : R - 1 - step;  // stage s = s_lo + st pairs g differing in bit st
        const int half = 1 << st;
        const uint32_t* t = tw + ((1 << (lowbits + st)) - 1) + low;
        Fp wv[G / 2];
#pragma unroll
        for (int k = 0; k < G / 2; k++)
            if (k < half && !(LB0 && k == 0)) wv[k] = held ? held[half - 1 + k] : Fp::raw(t[k << lowbits]);
#pragma unroll
        for (int p = 0; p < G / 2; p++) {
            const int g0 = ((p >> st) << (st + 1)) | (p & (half - 1)), g1 = g0 | half;
            if (LB0 && (p & (half - 1)) == 0) { Fp u = x[g0], v = x[g1]; x[g0] = u + v; x[g1] = u - v; continue; }
            const Fp wk = wv[p & (half - 1)];
            if (DIT) {
                // LAZY outputs: an output that the NEXT stage of this round multiplies by a twiddle (its index has bit st + 1 set, and in
                // the stage-1 round its twiddle is not the trivial one) may stay unreduced in [0, 2p) — the Montgomery product only needs
                // a * b < p * 2^32 — which drops the correction of u + v (2 instructions) and of u - v (1).  The operand `u` of a
                // butterfly is never lazy: u + v must stay below 2^32.  Everything is decided at compile time (the round is unrolled).
                Fp u = x[g0], v = x[g1] * wk;
                const bool next_mul = VGPU_NTT_LAZY && step + 1 < R && (g0 & (half << 1)) != 0;
                const bool lazy1 = next_mul, lazy0 = next_mul && !(LB0 && (g0 & (half - 1)) == 0);
                x[g0] = lazy0 ? Fp::raw(u.v + v.v) : u + v;
                x[g1] = lazy1 ? Fp::raw(u.v + (vg::P - v.v)) : u - v;
            } else {
                // (u - v) * w with the difference left unreduced in (0, 2p): the Montgomery product only needs a * b < p * 2^32
                Fp u = x[g0], v = x[g1];
                x[g0] = u + v;
                x[g1] = Fp::raw(vg::monty_reduce((uint64_t)(u.v + (vg::P - v.v)) * wk.v));
            }
        }
    }
#ifdef HIPEMU_CHECKS  // host emulation only (tests/emu): no lazy value may leave a round
#pragma unroll
    for (int g = 0; g < G; g++) if (x[g].v >= vg::P) { fprintf(stderr, "butterflies: unreduced value leaves the round\n"); abort(); }
#endif
}

}  // namespace vk
