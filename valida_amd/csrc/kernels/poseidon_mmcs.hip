// Poseidon-16 Merkle commitment kernels: the MMCS of BASELINE.json's north-star configuration ("batched Poseidon permutations
// for Merkle-tree commitment"), selected with vgpu_config.hash_kind = VGPU_HASH_POSEIDON16.  The reference itself instantiates
// Poseidon only inside the challenger (basic/tests/test_prover.rs:418-431) and commits with Keccak (merkle.hip); this variant
// is FieldMerkleTreeMmcs<BabyBear, PaddingFreeSponge<Perm16, 16, 8, 8>, TruncatedPermutation<Perm16, 2, 8, 16>, 8> over the same
// Poseidon<BabyBear, CosetMds<16>, 16, 5> permutation (conventions of p3-symmetric, recalled — unpinned
// like every Plonky3 convention):
//   leaf digest(row)  = sponge: for each chunk of 8 row elements OVERWRITE state[0..len) with it, permute; output state[0..8)
//   parent            = first 8 elements of Perm16(left || right)
//   injection         = C(parent, H(rows)) exactly as in the Keccak tree (FieldMerkleTree, App. B5)
// One thread per leaf / parent, the 16-element state in VGPRs in Montgomery form (LDE elements are absorbed as they lie in HBM,
// no conversion); digests are 8 canonical words like the Keccak ones.  The 480 round constants and the 16 circulant MDS
// coefficients are wave-uniform: they are read through the scalar cache (s_load_dwordx16 per round) rather than staged in LDS —
// a uniform LDS read still costs a ds_read per value per wave, a scalar load costs no VALU/LDS issue slot at all.
// Integer-VALU-bound: ~0.8 k instructions per dense round (16 x 16 lazily accumulated products); the 22 partial rounds run in their
// sparse-matrix form (31 products each, one dense round at the end) with deferred updates in groups of four rounds, the S-boxes on
// signed Montgomery products: ~9.5 k instructions per permutation (26 k in the plain form, 11.4 k in round 3).
#include <cstdlib>
#include "launch.hpp"
#include "poseidon_perm.hpp"
#include "challenger_dev.hpp"

namespace vk {

struct PPtrCols {
    const uint32_t* const* p;
    __device__ __forceinline__ const uint32_t* operator[](int k) const { return p[k]; }
};
struct PStridedCols {
    const uint32_t* base;
    uint64_t stride;
    __device__ __forceinline__ const uint32_t* operator[](int k) const { return base + (uint64_t)k * stride; }
};

// PaddingFreeSponge<Perm16, 16, 8, 8>::hash_iter over one row
template <class Cols>
__device__ __forceinline__ void poseidon_hash_row(const Cols cols, int n_elems, uint64_t r, const PoseidonTab& tab, uint32_t (&out)[8]) {
    Fp st[16];
#pragma unroll
    for (int i = 0; i < 16; i++) st[i] = Fp::zero();
    for (int base = 0; base < n_elems; base += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (base + k < n_elems) st[k] = Fp::raw(cols[base + k][r]);
        poseidon16_permute(st, tab);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = st[i].canonical();
}
// TruncatedPermutation<Perm16, 2, 8, 16>::compress
__device__ __forceinline__ void poseidon_compress2(const uint32_t (&l)[8], const uint32_t (&r)[8], const PoseidonTab& tab, uint32_t (&out)[8]) {
    Fp st[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { st[i] = Fp::from_canonical(l[i]); st[8 + i] = Fp::from_canonical(r[i]); }
    poseidon16_permute(st, tab);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = st[i].canonical();
}

__device__ __forceinline__ void p_load_digest(const uint32_t* p, uint32_t (&d)[8]) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    d[0] = lo.x; d[1] = lo.y; d[2] = lo.z; d[3] = lo.w; d[4] = hi.x; d[5] = hi.y; d[6] = hi.z; d[7] = hi.w;
}
__device__ __forceinline__ void p_store_digest(uint32_t* p, const uint32_t (&d)[8]) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(d[0], d[1], d[2], d[3]);
    q[1] = make_uint4(d[4], d[5], d[6], d[7]);
}

// VGPU_POSEIDON_LDS=1: the kernels read the permutation's tables (1.3 k words of matrix entries and constants) from a copy in LDS — one
// broadcast ds_read per value — instead of through the scalar cache.  Scalar loads cost no VALU slot by themselves, but a full round needs
// 112 table values at once and the wave has ~100 SGPRs: hipcc spilled the rest to VGPR lanes, and the v_writelane / v_readlane pairs (8 % of
// the compress kernel's VALU instructions, plus their hazard no-ops) ARE VALU work.  =0: scalar loads (A/B builds).
#ifndef VGPU_POSEIDON_LDS
#define VGPU_POSEIDON_LDS 0
#endif
// call at the top of a kernel, by every thread of the workgroup, before any early return
__device__ __forceinline__ PoseidonTab poseidon_tab_in_lds(const PoseidonTab& tab, uint32_t* s_opt) {
#if VGPU_POSEIDON_LDS
    if (tab.opt == nullptr) return tab;
    for (int i = (int)threadIdx.x; i < POPT_WORDS; i += (int)blockDim.x) s_opt[i] = tab.opt[i];
    __syncthreads();
    return PoseidonTab{tab.rc, tab.mds, s_opt};
#else
    return tab;
#endif
}

// (forcing 6 / 7 / 8 waves per SIMD on the two thread-per-node kernels was measured in round 6: nothing at 6, slower beyond — profiles/r06_ab_quotient_acc2_poseidon_waves.txt)
template <class Cols>
__global__ void __launch_bounds__(256) k_poseidon_leaves(const Cols cols, int n_elems, uint64_t n_rows, PoseidonTab gtab, uint32_t* __restrict__ digests) {
    __shared__ uint32_t s_opt[POPT_WORDS];
    const PoseidonTab tab = poseidon_tab_in_lds(gtab, s_opt);
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    uint32_t d[8];
    poseidon_hash_row(cols, n_elems, r, tab, d);
    p_store_digest(digests + 8 * r, d);
}

__device__ __forceinline__ void poseidon_node(const uint32_t* __restrict__ prev, const uint32_t* const* cols, int n_elems, uint64_t i, const PoseidonTab& tab,
                                              uint32_t* __restrict__ next) {
    uint32_t l[8], r[8], d[8];
    p_load_digest(prev + 16 * i, l);
    p_load_digest(prev + 16 * i + 8, r);
    poseidon_compress2(l, r, tab, d);
    if (n_elems > 0) {
        uint32_t h[8], d2[8];
        poseidon_hash_row(PPtrCols{cols}, n_elems, i, tab, h);
        poseidon_compress2(d, h, tab, d2);
        p_store_digest(next + 8 * i, d2);
    } else {
        p_store_digest(next + 8 * i, d);
    }
}

__global__ void __launch_bounds__(256) k_poseidon_compress(const uint32_t* __restrict__ prev, const uint32_t* const* __restrict__ cols, int n_elems, uint64_t n_out,
                                                           PoseidonTab gtab, uint32_t* __restrict__ next) {
    __shared__ uint32_t s_opt[POPT_WORDS];
    const PoseidonTab tab = poseidon_tab_in_lds(gtab, s_opt);
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    poseidon_node(prev, cols, n_elems, i, tab, next);
}

// ---- ONE PERMUTATION PER 16-LANE ROW (round 4): for the one-permutation-deep levels of a tree top --------------------------------------------
// A tree top is a chain: level after level of at most a few dozen nodes, each a permutation deep (three with an injected matrix).  With a
// thread per node a lone wave issues the permutation's ~9 200 dependent instructions one after the other (~19 us per level); spread over the 16
// lanes of a DPP row — lane l holds state[l] — a round is an S-box (every lane, or lane 0 in the partial rounds), the round constant of the
// lane, and the circulant MDS layer as 16 row rotations (v_mov_dpp row_ror:d brings x[(l - d) & 15] to lane l) times the wave-uniform
// coefficient mds[d], accumulated lazily four at a time: ~85 dependent instructions per round, ~2 600 per permutation, four permutations per
// wave.  Plain rounds (no sparse form: the partial rounds' saving is in the products the other lanes do in parallel anyway).  Levels of more
// than 64 nodes stay with the thread-per-node permutation (1024 threads / 16 lanes = 64 rows).  VGPU_POSEIDON_ROWS=0: A/B builds.
#ifndef VGPU_POSEIDON_ROWS
#define VGPU_POSEIDON_ROWS 1
#endif
template <int D> __device__ __forceinline__ uint32_t p_row_ror(uint32_t v) {
    if (D == 0) return v;
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + D, 0xF, 0xF, true);  // DPP row_ror:D — lane l of each row reads lane (l - D) & 15
}
template <int D0> __device__ __forceinline__ Fp p_mds4(uint32_t x, const uint32_t (&m)[16]) {
    uint64_t t = (uint64_t)m[D0] * p_row_ror<D0>(x);
    t += (uint64_t)m[D0 + 1] * p_row_ror<D0 + 1>(x);
    t += (uint64_t)m[D0 + 2] * p_row_ror<D0 + 2>(x);
    t += (uint64_t)m[D0 + 3] * p_row_ror<D0 + 3>(x);
    return Fp::raw(vg::monty_reduce_wide(t));
}
// st = this lane's coordinate; all 16 lanes of the row run it together
__device__ __forceinline__ Fp poseidon16_row(Fp st, const PoseidonTab& tab, const uint32_t (&m)[16], int l16) {
    uint32_t c_next = tab.rc[l16];  // the lane's round constant, fetched one round ahead (a vector load in the dependent chain would cost more than the round)
#pragma unroll 1
    for (int r = 0; r < 30; r++) {
        const uint32_t c = c_next;
        c_next = tab.rc[16 * (r < 29 ? r + 1 : 0) + l16];
        const bool full = r < 4 || r >= 26;
        const Fp x = (full || l16 == 0) ? poseidon_sbox_plus(st, c) : st + Fp::raw(c);
        st = (p_mds4<0>(x.v, m) + p_mds4<4>(x.v, m)) + (p_mds4<8>(x.v, m) + p_mds4<12>(x.v, m));  // y[l] = sum_d mds[d] x[(l - d) & 15]
    }
    return st;
}
// one tree node by the 16 lanes of a row: compress of the two child digests, plus the injected rows' hash and a second compress
__device__ __forceinline__ void poseidon_node_row(const uint32_t* __restrict__ prev, const uint32_t* const* cols, int n_elems, uint64_t node, int l16, const PoseidonTab& tab,
                                                  const uint32_t (&m)[16], uint32_t* __restrict__ next) {
    Fp st = poseidon16_row(Fp::from_canonical(prev[16 * node + l16]), tab, m, l16);  // lanes 0..7: left digest, 8..15: right — as they lie in the layer below
    if (n_elems > 0) {
        const Fp d = st;
        Fp h = Fp::zero();
        for (int base = 0; base < n_elems; base += 8) {  // PaddingFreeSponge: each chunk of 8 overwrites the head of the state
            if (l16 < 8 && base + l16 < n_elems) h = Fp::raw(cols[base + l16][node]);
            h = poseidon16_row(h, tab, m, l16);
        }
        // compress(d, h): the canonical words of both digests re-enter as field elements (canonical -> Montgomery is the identity on the value)
        const uint32_t hs = p_row_ror<8>(h.v);  // lane 8 + k reads lane k
        st = poseidon16_row(l16 < 8 ? d : Fp::raw(hs), tab, m, l16);
    }
    if (l16 < 8) next[8 * node + l16] = st.canonical();
}

// A layer of the latency-bound MIDDLE of a tree (256 < parents <= POSEIDON_ROW_MAX) with one node per 16-lane row (round 5): with a thread per
// node such a layer is at most a quarter of a wave per SIMD, every wave alone with its ~9 200 dependent instructions (16-19 us per layer, 26 trees
// x up to 4 such layers per proof); a row finishes its node in ~2 600.  The price is 4.4 x the lane-instructions per permutation, so the
// threshold stays where the GPU is mostly idle anyway (thresholds 0 .. 65536 measured: profiles/r05_ab_valu_sensitivity_poseidon_rows.txt).
constexpr uint64_t POSEIDON_ROW_MAX = 16384;
__global__ void __launch_bounds__(256) k_poseidon_compress_row(const uint32_t* __restrict__ prev, const uint32_t* const* __restrict__ cols, int n_elems, uint64_t n_out,
                                                               PoseidonTab tab, uint32_t* __restrict__ next) {
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = tab.mds[i];
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, node = t >> 4;
    if (node >= n_out) return;  // whole rows leave together (256 threads = 16 rows)
    poseidon_node_row(prev, cols, n_elems, node, (int)(t & 15), tab, m, next);
}

// the last <= 11 levels of a tree in one launch (one 1024-thread workgroup, a barrier per level), as k_keccak_top
__global__ void __launch_bounds__(1024) k_poseidon_top(KeccakTopArgs a, PoseidonTab gtab) {
    __shared__ uint32_t s_opt[POPT_WORDS];
    const PoseidonTab tab = poseidon_tab_in_lds(gtab, s_opt);
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = tab.mds[i];
    const uint32_t* prev = a.prev;
    for (int l = 0; l < a.levels; l++) {
        const uint64_t len = a.first_len >> l;
        if (VGPU_POSEIDON_ROWS && len <= 64) {
            if ((threadIdx.x >> 4) < len) poseidon_node_row(prev, a.cols[l], a.n_elems[l], threadIdx.x >> 4, (int)(threadIdx.x & 15), tab, m, a.out[l]);
        } else if (threadIdx.x < len) {
            poseidon_node(prev, a.cols[l], a.n_elems[l], threadIdx.x, tab, a.out[l]);
        }
        __threadfence_block();
        __syncthreads();
        prev = a.out[l];
    }
    if (a.ch_pos && threadIdx.x < 64) fri_challenge_step((int)threadIdx.x, a.ch_pos, a.ch_state, prev, a.ch_beta5, a.ch_commit8);
}

// Algorithmic VALU work for the profiler's valu_ops column (like KECCAK_VALU_PER_PERM): the instructions of ONE permutation as these kernels run
// it, by issue class (bench.py holds the same two numbers and prices them at the measured issue rates).
//   half rate (v_mad_u64_u32 / v_mad_i64_i32 / v_mul_lo_u32 / v_mul_hi_[ui]32): a Montgomery product is 3, a lazily accumulated term 1, a reduction of <= 4 terms 2
//     S-boxes (8 x 16 + 22) x 3 products                                              1350
//     MDS layer as CRT blocks, 8 x (96 + 24 x 2)                                       1152
//     sparse partial rounds in groups of four with deferred updates (poseidon_perm.hpp):
//       5 groups x (4 x (16 + 15) + 6 cross terms + 34 reductions x 2) + (31 + 19 x 2)  1059
//     dense partial round, 16 x (16 + 4 x 2)                                            384
//   full rate (add / sub / carry / select): a signed product's subtraction 1, a reduction 5, a modular addition 3
//     round constants 8 x 16 x 1 = 128 (one addition: the signed S-box takes x + (c - p) unreduced); S-boxes 150 x (3 + 3) = 900 (signed Montgomery
//     products, one correction); MDS 8 x (48 x 3 + 24 x 5 + 8 x 3) = 2304; sparse rounds 189 reductions x 5 + (5 x 34 + 19) additions x 3 - 21 x 2 = 1470;
//     dense round 16 x (4 x 5 + 3 x 3) - 2 = 462
constexpr double POSEIDON_HALF_PER_PERM = 1350.0 + 1152.0 + 1059.0 + 384.0, POSEIDON_FULL_PER_PERM = 128.0 + 900.0 + 2304.0 + 1470.0 + 462.0;
constexpr double POSEIDON_VALU_PER_PERM = POSEIDON_HALF_PER_PERM + POSEIDON_FULL_PER_PERM;
static double p_row_perms(int n_elems) { return (double)((n_elems + 7) / 8); }
static double p_node_perms(int n_inject) { return n_inject > 0 ? 2.0 + p_row_perms(n_inject) : 1.0; }
void launch_poseidon_leaves(hipStream_t st, const uint32_t* pos_dev, bool sparse, const uint32_t* const* cols_dev, int n_elems, uint64_t n_rows, uint32_t* digests) {
    ProfScope ps("k_poseidon_leaves", st, (double)n_rows * (4.0 * n_elems + 32.0), (double)n_rows * p_row_perms(n_elems) * POSEIDON_VALU_PER_PERM / 64.0);
    VK_LAUNCH(k_poseidon_leaves<PPtrCols>, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, PPtrCols{cols_dev}, n_elems, n_rows, tab_of(pos_dev, sparse), digests);
}
void launch_poseidon_leaves_strided(hipStream_t st, const uint32_t* pos_dev, bool sparse, const uint32_t* base, uint64_t stride, int n_elems, uint64_t n_rows, uint32_t* digests) {
    ProfScope ps("k_poseidon_leaves", st, (double)n_rows * (4.0 * n_elems + 32.0), (double)n_rows * p_row_perms(n_elems) * POSEIDON_VALU_PER_PERM / 64.0);
    VK_LAUNCH(k_poseidon_leaves<PStridedCols>, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, PStridedCols{base, stride}, n_elems, n_rows,
                       tab_of(pos_dev, sparse), digests);
}
void launch_poseidon_compress(hipStream_t st, const uint32_t* pos_dev, bool sparse, const uint32_t* prev, const uint32_t* const* cols_dev, int n_elems, uint64_t n_out, uint32_t* next) {
    if (VGPU_POSEIDON_ROWS && n_out > 256 && n_out <= POSEIDON_ROW_MAX) {  // (layers of <= 256 parents belong to k_poseidon_top in a tree; the emulated-source tests launch tiny ones here)
        ProfScope ps("k_poseidon_compress_row", st, (double)n_out * (96.0 + 4.0 * n_elems), (double)n_out * p_node_perms(n_elems) * POSEIDON_VALU_PER_PERM / 64.0);
        VK_LAUNCH(k_poseidon_compress_row, dim3((unsigned)((16 * n_out + 255) / 256)), dim3(256), 0, st, prev, cols_dev, n_elems, n_out, tab_of(pos_dev, sparse), next);
        return;
    }
    ProfScope ps("k_poseidon_compress", st, (double)n_out * (96.0 + 4.0 * n_elems), (double)n_out * p_node_perms(n_elems) * POSEIDON_VALU_PER_PERM / 64.0);
    VK_LAUNCH(k_poseidon_compress, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, prev, cols_dev, n_elems, n_out, tab_of(pos_dev, sparse), next);
}
void launch_poseidon_top(hipStream_t st, const uint32_t* pos_dev, bool sparse, const KeccakTopArgs& a) {
    double bytes = 0, perms = 0;
    for (int l = 0; l < a.levels; l++) {
        bytes += (double)(a.first_len >> l) * (96.0 + 4.0 * a.n_elems[l]);
        perms += (double)(a.first_len >> l) * p_node_perms(a.n_elems[l]);
    }
    ProfScope ps("k_poseidon_top", st, bytes, perms * POSEIDON_VALU_PER_PERM / 64.0);
    VK_LAUNCH(k_poseidon_top, dim3(1), dim3(1024), 0, st, a, tab_of(pos_dev, sparse));
}

}  // namespace vk
