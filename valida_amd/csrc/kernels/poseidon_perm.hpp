// Poseidon<BabyBear, CosetMds<16>, 16, 5> (4 + 22 + 4 rounds; basic/tests/test_prover.rs:418-422) as a device function, one permutation
// per thread with the 16-element state in VGPRs: shared by the Poseidon-16 MMCS kernels (poseidon_mmcs.hip) and the proof-of-work search
// (open.hip, k_pow_grind).  Tables: host/poseidon_opt.hpp (poseidon_device_image), wave-uniform, read through the scalar cache.
#pragma once
#include "device_common.hpp"
#include "butterfly.hpp"

namespace vk {

struct PoseidonTab {
    const uint32_t* __restrict__ rc;   // [30][16] Montgomery
    const uint32_t* __restrict__ mds;  // [16] circulant coefficients: M[j][i] = mds[(j - i) & 15]
    const uint32_t* __restrict__ opt;  // sparse-partial-round tables (host/poseidon_opt.hpp layout), or null: plain rounds
};
// offsets into `opt` (words) — must match vhost::PoseidonOptTables
constexpr int POPT_RC_FULL = 0, POPT_T = 128, POPT_SPARSE = 152, POPT_F = 152 + 21 * 32, POPT_FFT_FWD = POPT_F + 256, POPT_FFT_INV = POPT_FFT_FWD + 16,
              POPT_FFT_LAM = POPT_FFT_INV + 16, POPT_BLK_N8 = POPT_FFT_LAM + 16, POPT_BLK_C4 = POPT_BLK_N8 + 64, POPT_BLK_N4 = POPT_BLK_C4 + 16;

// y = M x with M circulant; four products share one Montgomery reduction (4 p^2 < 2^64)
__device__ __forceinline__ void poseidon_mds(Fp (&st)[16], const uint32_t (&m)[16]) {
    Fp out[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        Fp acc = Fp::zero();
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += 4) {
            uint64_t t = 0;
#pragma unroll
            for (int i = i0; i < i0 + 4; i++) t += (uint64_t)m[(j - i) & 15] * st[i].v;
            acc += Fp::raw(vg::monty_reduce_wide(t));
        }
        out[j] = acc;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) st[i] = out[i];
}
// The same product as a 16-point cyclic convolution (the matrix is circulant): DIF transform (natural in, bit-reversed out), pointwise
// product with lambda = DFT(coefficients) / 16 in that order, DIT transform back — 17 + 16 + 17 products and 128 additions / subtractions,
// ~650 instructions instead of ~900, and no second copy of the state.  Tables: host/poseidon_opt.hpp (FFT_*), wave-uniform.
#ifndef VGPU_POSEIDON_MDS
#define VGPU_POSEIDON_MDS 2  // the MDS layer of the full rounds: 0 dense product, 1 transforms, 2 CRT blocks (A/B builds)
#endif
__device__ __forceinline__ void poseidon_mds_convolution(Fp (&st)[16], const uint32_t* __restrict__ o) {
    butterflies<4, false, true>(st, o + POPT_FFT_FWD, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; i++) st[i] *= Fp::raw(o[POPT_FFT_LAM + i]);
    butterflies<4, true, true>(st, o + POPT_FFT_INV, 0, 0);
}
// ... and as the CRT split x^16 - 1 = (x^8 + 1)(x^4 + 1)(x^4 - 1) of that convolution (host/poseidon_opt.hpp, BLK_*): 24 additions /
// subtractions, a negacyclic 8 x 8, a negacyclic 4 x 4 and a cyclic 4 x 4 product with lazily accumulated terms (96 multiply-adds, 24
// reductions), 24 additions / subtractions back.  No twiddle products; ~430 instructions.
template <int K> __device__ __forceinline__ Fp poseidon_dot(const uint32_t* __restrict__ row, const Fp (&v)[K]) {
    Fp acc = Fp::zero();
#pragma unroll
    for (int i0 = 0; i0 < K; i0 += 4) {
        uint64_t t = 0;
#pragma unroll
        for (int i = i0; i < i0 + 4; i++) t += (uint64_t)row[i] * v[i].v;
        const Fp r = Fp::raw(vg::monty_reduce_wide(t));
        acc = i0 == 0 ? r : acc + r;
    }
    return acc;
}
__device__ __forceinline__ void poseidon_mds_blocks(Fp (&st)[16], const uint32_t* __restrict__ o) {
    Fp am[8], ap[8], app[4], apm[4];
#pragma unroll
    for (int i = 0; i < 8; i++) { am[i] = st[i] - st[i + 8]; ap[i] = st[i] + st[i + 8]; }
#pragma unroll
    for (int i = 0; i < 4; i++) { apm[i] = ap[i] - ap[i + 4]; app[i] = ap[i] + ap[i + 4]; }
    Fp ym[8], ypp[4], ypm[4];
#pragma unroll
    for (int k = 0; k < 8; k++) ym[k] = poseidon_dot<8>(o + POPT_BLK_N8 + 8 * k, am);
#pragma unroll
    for (int k = 0; k < 4; k++) { ypp[k] = poseidon_dot<4>(o + POPT_BLK_C4 + 4 * k, app); ypm[k] = poseidon_dot<4>(o + POPT_BLK_N4 + 4 * k, apm); }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const Fp lo = ypp[i] + ypm[i], hi = ypp[i] - ypm[i];
        st[i] = lo + ym[i]; st[i + 8] = lo - ym[i];
        st[i + 4] = hi + ym[i + 4]; st[i + 12] = hi - ym[i + 4];
    }
}
__device__ __forceinline__ Fp poseidon_sbox(Fp x) { const Fp x2 = x * x; return x2 * x2 * x; }

// row . state for one row of 16 wave-uniform coefficients
__device__ __forceinline__ Fp poseidon_dot16(const uint32_t* __restrict__ row, const Fp (&st)[16]) {
    Fp acc = Fp::zero();
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 4) {
        uint64_t t = 0;
#pragma unroll
        for (int i = i0; i < i0 + 4; i++) t += (uint64_t)row[i] * st[i].v;
        acc += Fp::raw(vg::monty_reduce_wide(t));
    }
    return acc;
}

__device__ __forceinline__ void poseidon16_permute(Fp (&st)[16], const PoseidonTab& tab) {
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = tab.mds[i];
    if (tab.opt == nullptr) {  // plain form: 30 rounds, dense MDS in every one
#pragma unroll 1
        for (int r = 0; r < 30; r++) {
            const uint32_t* rc = tab.rc + 16 * r;
#pragma unroll
            for (int i = 0; i < 16; i++) st[i] += Fp::raw(rc[i]);
            if (r < 4 || r >= 26) {
#pragma unroll
                for (int i = 0; i < 16; i++) st[i] = poseidon_sbox(st[i]);
            } else st[0] = poseidon_sbox(st[0]);
            poseidon_mds(st, m);
        }
        return;
    }
    // 4 full rounds, 21 SPARSE partial rounds (31 products each instead of 256), one dense partial round, 4 full rounds — the same
    // permutation (host/poseidon_opt.hpp derives the tables and checks them against the plain form)
    const uint32_t* __restrict__ o = tab.opt;
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        const uint32_t* rc = o + POPT_RC_FULL + 16 * r;
#pragma unroll
        for (int i = 0; i < 16; i++) st[i] = poseidon_sbox(st[i] + Fp::raw(rc[i]));
        if (VGPU_POSEIDON_MDS == 2) poseidon_mds_blocks(st, o); else if (VGPU_POSEIDON_MDS == 1) poseidon_mds_convolution(st, o); else poseidon_mds(st, m);
    }
    st[0] += Fp::raw(o[POPT_T]);
#pragma unroll 1
    for (int i = 0; i < 21; i++) {
        const uint32_t* __restrict__ s = o + POPT_SPARSE + 32 * i;
        const Fp x0 = poseidon_sbox(st[0]);
        Fp tmp[16];
        tmp[0] = x0;
#pragma unroll
        for (int b = 1; b < 16; b++) tmp[b] = st[b];
        const Fp n0 = poseidon_dot16(s, tmp);             // a x0 + u . x^
#pragma unroll
        for (int a = 1; a < 16; a++) st[a] += Fp::raw(s[15 + a]) * x0;  // x^ + w x0
        st[0] = n0 + Fp::raw(o[POPT_T + 1 + i]);
    }
    {
        st[0] = poseidon_sbox(st[0]);
        Fp out[16];
#pragma unroll
        for (int a = 0; a < 16; a++) out[a] = poseidon_dot16(o + POPT_F + 16 * a, st);
#pragma unroll
        for (int a = 0; a < 16; a++) st[a] = out[a];
    }
#pragma unroll 1
    for (int r = 4; r < 8; r++) {
        const uint32_t* rc = o + POPT_RC_FULL + 16 * r;
#pragma unroll
        for (int i = 0; i < 16; i++) st[i] = poseidon_sbox(st[i] + Fp::raw(rc[i]));
        if (VGPU_POSEIDON_MDS == 2) poseidon_mds_blocks(st, o); else if (VGPU_POSEIDON_MDS == 1) poseidon_mds_convolution(st, o); else poseidon_mds(st, m);
    }
}

// pos_dev: [480 rc][16 mds][16 state][8 ..] as the device challenger uses, followed at word 1024 by the sparse-round / convolution tables
// when they are valid
__host__ __device__ __forceinline__ PoseidonTab tab_of(const uint32_t* pos_dev, bool sparse) { return PoseidonTab{pos_dev, pos_dev + 480, sparse ? pos_dev + 1024 : nullptr}; }

}  // namespace vk
