// Poseidon<BabyBear, CosetMds<16>, 16, 5> (4 + 22 + 4 rounds; basic/tests/test_prover.rs:418-422) as a device function, one permutation
// per thread with the 16-element state in VGPRs: shared by the Poseidon-16 MMCS kernels (poseidon_mmcs.hip) and the proof-of-work search
// (open.hip, k_pow_grind).  Tables: host/poseidon_opt.hpp (poseidon_device_image), wave-uniform, read through the scalar cache.
#pragma once
#include "device_common.hpp"
#include "butterfly.hpp"

namespace vk {

// The permutation's tables are wave-uniform and read with scalar loads.  Left alone, hipcc hoists every load of a round body to its top — a
// full round then wants 112 values in ~100 SGPRs and the rest is spilled to VGPR lanes (v_writelane / v_readlane: VALU instructions, 8 % of the
// compress kernel).  VGPU_POSEIDON_FENCE=1 puts a compiler-level memory fence between the sections of a round, so that a section's table
// values are loaded when it starts and are dead when it ends.  =0: A/B builds.
#ifndef VGPU_POSEIDON_FENCE
#define VGPU_POSEIDON_FENCE 1
#endif
#if VGPU_POSEIDON_FENCE && defined(__HIP_DEVICE_COMPILE__)
#define POSEIDON_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define POSEIDON_FENCE() do { } while (0)
#endif
// the tables as CONSTANT-address-space memory: a uniform load from it is a scalar load whatever else the kernel does to memory
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) uint32_t* popt_ptr;
__device__ __forceinline__ popt_ptr popt_of(const uint32_t* p) { return reinterpret_cast<popt_ptr>(reinterpret_cast<uintptr_t>(p)); }
// K table words (K a multiple of 4, 16-byte aligned) as VOLATILE 4-word scalar loads: the loads stay where the code puts them (not hoisted to
// the top of the round, not merged across sections), four words per scalar-memory instruction
template <int K> __device__ __forceinline__ void popt_load(popt_ptr p, uint32_t (&out)[K]) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef const volatile __attribute__((address_space(4))) u4* vp;
#pragma unroll
    for (int i = 0; i < K; i += 4) {
        const u4 v = *reinterpret_cast<vp>(p + i);
        out[i] = v.x; out[i + 1] = v.y; out[i + 2] = v.z; out[i + 3] = v.w;
    }
}
#else
typedef const uint32_t* popt_ptr;
__host__ __device__ inline popt_ptr popt_of(const uint32_t* p) { return p; }
template <int K> __host__ __device__ inline void popt_load(popt_ptr p, uint32_t (&out)[K]) { for (int i = 0; i < K; i++) out[i] = p[i]; }
#endif

struct PoseidonTab {
    const uint32_t* __restrict__ rc;   // [30][16] Montgomery
    const uint32_t* __restrict__ mds;  // [16] circulant coefficients: M[j][i] = mds[(j - i) & 15]
    const uint32_t* __restrict__ opt;  // sparse-partial-round tables (host/poseidon_opt.hpp layout), or null: plain rounds
};
// offsets into `opt` (words) — must match vhost::PoseidonOptTables
constexpr int POPT_RC_FULL = 0, POPT_T = 128, POPT_SPARSE = 152, POPT_F = 152 + 21 * 32, POPT_FFT_FWD = POPT_F + 256, POPT_FFT_INV = POPT_FFT_FWD + 16,
              POPT_FFT_LAM = POPT_FFT_INV + 16, POPT_BLK_N8 = POPT_FFT_LAM + 16, POPT_BLK_C4 = POPT_BLK_N8 + 64, POPT_BLK_N4 = POPT_BLK_C4 + 16,
              POPT_CROSS = POPT_BLK_N4 + 16, POPT_WORDS = POPT_CROSS + 24 * 4;

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
template <int K> __device__ __forceinline__ Fp poseidon_dot(popt_ptr rowp, const Fp (&v)[K]) {
    uint32_t row[K];
    popt_load<K>(rowp, row);
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
__device__ __forceinline__ void poseidon_mds_blocks(Fp (&st)[16], popt_ptr o) {
    Fp am[8], ap[8], app[4], apm[4];
#pragma unroll
    for (int i = 0; i < 8; i++) { am[i] = st[i] - st[i + 8]; ap[i] = st[i] + st[i + 8]; }
#pragma unroll
    for (int i = 0; i < 4; i++) { apm[i] = ap[i] - ap[i + 4]; app[i] = ap[i] + ap[i + 4]; }
    Fp ym[8], ypp[4], ypm[4];
    POSEIDON_FENCE();
#pragma unroll
    for (int k = 0; k < 4; k++) ym[k] = poseidon_dot<8>(o + POPT_BLK_N8 + 8 * k, am);
    POSEIDON_FENCE();
#pragma unroll
    for (int k = 4; k < 8; k++) ym[k] = poseidon_dot<8>(o + POPT_BLK_N8 + 8 * k, am);
    POSEIDON_FENCE();
#pragma unroll
    for (int k = 0; k < 4; k++) { ypp[k] = poseidon_dot<4>(o + POPT_BLK_C4 + 4 * k, app); ypm[k] = poseidon_dot<4>(o + POPT_BLK_N4 + 4 * k, apm); }
    POSEIDON_FENCE();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const Fp lo = ypp[i] + ypm[i], hi = ypp[i] - ypm[i];
        st[i] = lo + ym[i]; st[i + 8] = lo - ym[i];
        st[i + 4] = hi + ym[i + 4]; st[i + 12] = hi - ym[i + 4];
    }
}
// x^5.  VGPU_POSEIDON_SBOX=1 (default): SIGNED Montgomery products — (t - (t_lo p^-1 mod+- 2^32) p) / 2^32 of a signed 64-bit product with
// |t| < p 2^31 lies in (-p, p) with NO correction step, and a product of two such values is again below p 2^31 — so x^2, x^4 and x^5 chain
// without the three-instruction corrections and ONE correction brings the result back to [0, p): 15 instead of 18 VALU instructions per
// S-box (150 S-boxes per permutation), the same residue, hence the same canonical value.  =0: three plain Montgomery products (A/B builds).
#ifndef VGPU_POSEIDON_SBOX
#define VGPU_POSEIDON_SBOX 1
#endif
__device__ __forceinline__ int32_t monty_signed(int64_t t) {
    const int32_t m = (int32_t)((uint32_t)t * 0x88000001u);
#if defined(__HIP_DEVICE_COMPILE__)
    const int32_t u = __mulhi(m, (int32_t)vg::P);
#else
    const int32_t u = (int32_t)(((int64_t)m * (int64_t)vg::P) >> 32);
#endif
    return (int32_t)(t >> 32) - u;
}
// x1 in [-p, p] (any representative of the residue in that range)
__device__ __forceinline__ Fp poseidon_sbox_i(int32_t x1) {
    const int32_t x2 = monty_signed((int64_t)x1 * x1);
    const int32_t x4 = monty_signed((int64_t)x2 * x2);
    const int32_t x5 = monty_signed((int64_t)x4 * x1);
    return Fp::raw((uint32_t)(x5 + ((x5 >> 31) & (int32_t)vg::P)));
}
__device__ __forceinline__ Fp poseidon_sbox(Fp x) {
#if VGPU_POSEIDON_SBOX
    return poseidon_sbox_i((int32_t)x.v);  // < p < 2^31
#else
    const Fp x2 = x * x;
    return x2 * x2 * x;
#endif
}
// S-box of x + c for a wave-uniform constant c: the signed product takes x + (c - p) in [-p, p) as it is — ONE addition (the constant's
// c - p is scalar-unit work) where the modular addition costs three instructions
__device__ __forceinline__ Fp poseidon_sbox_plus(Fp x, uint32_t c) {
#if VGPU_POSEIDON_SBOX
    return poseidon_sbox_i((int32_t)x.v + (int32_t)(c - vg::P));
#else
    return poseidon_sbox(x + Fp::raw(c));
#endif
}

// row . state for one row of 16 wave-uniform coefficients
__device__ __forceinline__ Fp poseidon_dot16(popt_ptr rowp, const Fp (&st)[16]) {
    uint32_t row[16];
    popt_load<16>(rowp, row);
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

#ifndef VGPU_POSEIDON_DEFER
#define VGPU_POSEIDON_DEFER 1  // the sparse partial rounds in groups of four with deferred updates (below); 0: round by round (A/B builds)
#endif
// N consecutive sparse rounds starting at round g0 (a multiple of 4), st[1..15] updated at the end
template <int N> __device__ __forceinline__ void poseidon_sparse_group(Fp (&st)[16], popt_ptr o, int g0) {
    uint64_t upd[16];
    Fp x0s[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        POSEIDON_FENCE();
        uint32_t s[32], cr[4];
        popt_load<32>(o + POPT_SPARSE + 32 * (g0 + k), s);
        popt_load<4>(o + POPT_CROSS + 4 * (g0 + k), cr);
        const Fp x0 = poseidon_sbox_plus(st[0], o[POPT_T + g0 + k]);  // the round's scalar rides in the S-box's input
        x0s[k] = x0;
        // a x0 + u . x^(group start) + sum_j cross[j] x0_j: 16 + k terms, four per reduction
        uint64_t t = (uint64_t)s[0] * x0.v;
#pragma unroll
        for (int b = 1; b < 4; b++) t += (uint64_t)s[b] * st[b].v;
        Fp n0 = Fp::raw(vg::monty_reduce_wide(t));
#pragma unroll
        for (int b0 = 4; b0 < 16; b0 += 4) {
            t = 0;
#pragma unroll
            for (int b = b0; b < b0 + 4; b++) t += (uint64_t)s[b] * st[b].v;
            n0 += Fp::raw(vg::monty_reduce_wide(t));
        }
        if (k > 0) {
            t = 0;
#pragma unroll
            for (int j = 0; j < k; j++) t += (uint64_t)cr[j] * x0s[j].v;
            n0 += Fp::raw(vg::monty_reduce_wide(t));
        }
#pragma unroll
        for (int a = 1; a < 16; a++) upd[a] = k == 0 ? (uint64_t)s[15 + a] * x0.v : upd[a] + (uint64_t)s[15 + a] * x0.v;
        st[0] = n0;
    }
#pragma unroll
    for (int a = 1; a < 16; a++) st[a] += Fp::raw(vg::monty_reduce_wide(upd[a]));
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
    popt_ptr o = popt_of(tab.opt);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        POSEIDON_FENCE();
        uint32_t rc[16];
        popt_load<16>(o + POPT_RC_FULL + 16 * r, rc);
#pragma unroll
        for (int i = 0; i < 16; i++) st[i] = poseidon_sbox_plus(st[i], rc[i]);
        if (VGPU_POSEIDON_MDS == 2) poseidon_mds_blocks(st, o); else if (VGPU_POSEIDON_MDS == 1) poseidon_mds_convolution(st, tab.opt); else poseidon_mds(st, m);
    }
#if !VGPU_POSEIDON_DEFER
    st[0] += Fp::raw(o[POPT_T]);
#endif
#if VGPU_POSEIDON_DEFER
    // The 21 sparse rounds in groups of four with the updates of coordinates 1..15 DEFERRED to the group's end: a round's dot product reads
    // the coordinates as they stood at the group's start and adds cross[r][j] x0_j for the group's earlier rounds (u_r . w_j, host table),
    // all in the same lazily reduced sums; the updates w_r[b] x0_r accumulate as one 64-bit multiply-add each and are reduced once per group
    // (four products < 4 p^2 < 2^64) — 118 instead of 191 instructions per round.
#pragma unroll 1
    for (int g0 = 0; g0 < 20; g0 += 4) poseidon_sparse_group<4>(st, o, g0);
    poseidon_sparse_group<1>(st, o, 20);
#else
#pragma unroll 1
    for (int i = 0; i < 21; i++) {
        popt_ptr s = o + POPT_SPARSE + 32 * i;
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
#endif
    {
#if VGPU_POSEIDON_DEFER
        st[0] = poseidon_sbox_plus(st[0], o[POPT_T + 21]);  // the last partial round's scalar, still pending
#else
        st[0] = poseidon_sbox(st[0]);
#endif
        Fp out[16];
#pragma unroll
        for (int a = 0; a < 16; a++) { if ((a & 3) == 0) POSEIDON_FENCE(); out[a] = poseidon_dot16(o + POPT_F + 16 * a, st); }
        POSEIDON_FENCE();
#pragma unroll
        for (int a = 0; a < 16; a++) st[a] = out[a];
    }
#pragma unroll 1
    for (int r = 4; r < 8; r++) {
        POSEIDON_FENCE();
        uint32_t rc[16];
        popt_load<16>(o + POPT_RC_FULL + 16 * r, rc);
#pragma unroll
        for (int i = 0; i < 16; i++) st[i] = poseidon_sbox_plus(st[i], rc[i]);
        if (VGPU_POSEIDON_MDS == 2) poseidon_mds_blocks(st, o); else if (VGPU_POSEIDON_MDS == 1) poseidon_mds_convolution(st, tab.opt); else poseidon_mds(st, m);
    }
}

// pos_dev: [480 rc][16 mds][16 state][8 ..] as the device challenger uses, followed at word 1024 by the sparse-round / convolution tables
// when they are valid
__host__ __device__ __forceinline__ PoseidonTab tab_of(const uint32_t* pos_dev, bool sparse) { return PoseidonTab{pos_dev, pos_dev + 480, sparse ? pos_dev + 1024 : nullptr}; }

}  // namespace vk
