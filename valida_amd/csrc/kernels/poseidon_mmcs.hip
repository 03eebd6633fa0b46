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
// than 64 nodes stay with the thread-per-node permutation (1024 threads / 16 lanes = 64 rows).
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
// st = this lane's coordinate; all 16 lanes of the row run it together.  Plain form: 30 rounds with the dense circulant layer.
__device__ __forceinline__ Fp poseidon16_row_plain(Fp st, const PoseidonTab& tab, const uint32_t (&m)[16], int l16) {
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
// The SPARSE form of the 22 partial rounds by a row (round 6; host/poseidon_opt.hpp: S = [[a, u^T],[w, I]] per round, one dense matrix F at the end).
// A sparse round costs the row: the broadcast of coordinate 0 (DPP row_newbcast), its S-box in every lane (redundantly: no divergence), ONE product per
// lane — a x0 in lane 0, u[l] x[l] elsewhere — summed over the row by four rotate-and-add steps (the new coordinate 0), and x[l] += w[l] x0 beside it:
// ~50 instructions, ~40 of them dependent, against the dense round's 85.  The per-lane coefficients (two words per round) are fetched a round ahead.
__device__ __forceinline__ uint32_t p_row_bcast0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150, 0xF, 0xF, true); }  // DPP row_newbcast:0
template <int D> __device__ __forceinline__ Fp p_row_add_ror(Fp v) { return v + Fp::raw(p_row_ror<D>(v.v)); }
template <int D0> __device__ __forceinline__ Fp p_dense4(uint32_t x, const uint32_t (&f)[16]) {
    uint64_t t = (uint64_t)f[D0] * p_row_ror<D0>(x);
    t += (uint64_t)f[D0 + 1] * p_row_ror<D0 + 1>(x);
    t += (uint64_t)f[D0 + 2] * p_row_ror<D0 + 2>(x);
    t += (uint64_t)f[D0 + 3] * p_row_ror<D0 + 3>(x);
    return Fp::raw(vg::monty_reduce_wide(t));
}
__device__ __forceinline__ Fp poseidon16_row(Fp st, const PoseidonTab& tab, const uint32_t (&m)[16], int l16) {
    if (tab.opt == nullptr) return poseidon16_row_plain(st, tab, m, l16);
    const uint32_t* __restrict__ o = tab.opt;
    uint32_t c_next = o[POPT_RC_FULL + l16];
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        const uint32_t c = c_next;
        c_next = o[POPT_RC_FULL + 16 * (r < 3 ? r + 1 : 4) + l16];  // (after round 3: the constants of the first closing full round)
        const Fp x = poseidon_sbox_plus(st, c);
        st = (p_mds4<0>(x.v, m) + p_mds4<4>(x.v, m)) + (p_mds4<8>(x.v, m) + p_mds4<12>(x.v, m));
    }
    const uint32_t c_close = c_next;
    // 21 sparse rounds.  Coordinate 0 carries its round's scalar t_i as PENDING: it rides in the S-box's input (poseidon_sbox_plus), as in the thread form.
    uint32_t dot_next = o[POPT_SPARSE + l16], upd_next = l16 ? o[POPT_SPARSE + 15 + l16] : 0u;
#pragma unroll 1
    for (int i = 0; i < 21; i++) {
        const uint32_t dot = dot_next, upd = upd_next, t_i = o[POPT_T + i];
        const int nx = i < 20 ? i + 1 : 0;
        dot_next = o[POPT_SPARSE + 32 * nx + l16];
        upd_next = l16 ? o[POPT_SPARSE + 32 * nx + 15 + l16] : 0u;
        const Fp x0 = poseidon_sbox_plus(Fp::raw(p_row_bcast0(st.v)), t_i);
        Fp r = Fp::raw(dot) * (l16 == 0 ? x0 : st);   // lane 0: a x0; lane l: u[l] x[l]
        r = p_row_add_ror<8>(r); r = p_row_add_ror<4>(r); r = p_row_add_ror<2>(r); r = p_row_add_ror<1>(r);  // every lane: the row's sum = the new coordinate 0
        const Fp up = st + Fp::raw(upd) * x0;         // x[l] + w[l] x0
        st = l16 == 0 ? r : up;
    }
    {   // the last partial round: S-box on coordinate 0 (scalar t_21 pending), then the dense matrix F: y[l] = sum_d F[l][(l - d) & 15] x[(l - d) & 15]
        uint32_t f[16];
#pragma unroll
        for (int d = 0; d < 16; d++) f[d] = o[POPT_F + 16 * l16 + ((l16 - d) & 15)];
        const Fp sb = poseidon_sbox_plus(st, o[POPT_T + 21]);
        const Fp x = l16 == 0 ? sb : st;
        st = (p_dense4<0>(x.v, f) + p_dense4<4>(x.v, f)) + (p_dense4<8>(x.v, f) + p_dense4<12>(x.v, f));
    }
    c_next = c_close;
#pragma unroll 1
    for (int r = 4; r < 8; r++) {
        const uint32_t c = c_next;
        c_next = o[POPT_RC_FULL + 16 * (r < 7 ? r + 1 : 0) + l16];
        const Fp x = poseidon_sbox_plus(st, c);
        st = (p_mds4<0>(x.v, m) + p_mds4<4>(x.v, m)) + (p_mds4<8>(x.v, m) + p_mds4<12>(x.v, m));
    }
    return st;
}

// ---- THE LATENCY-BOUND PART OF A TREE (parents <= POSEIDON_ROW_MAX per layer) AS MULTI-LAYER LAUNCHES OF 16-ROW WORKGROUPS (round 6) ------------
// Round 5 ran every layer of 256 < parents <= 16384 as a launch of its own (one node per 16-lane row) and the last nine layers in ONE 1024-thread
// workgroup whose 256- and 128-parent layers fell back to a thread per node (~19 us each): 26 trees x (6 launches + a ~146 us top) = 5 ms of a lone
// proof.  Here a 256-thread workgroup owns 16 consecutive parents of a launch's first layer and walks its own sub-tree DOWN TO ONE NODE — five
// layers, every node by a 16-lane row, the children's digests handed from layer to layer through LDS (and stored for the query phase) — so that the
// layers below 16384 parents cost a tree three launches (16384 .. 1024, 512 .. 32, 16 .. 1), each five row-permutations deep.  The last launch of a
// tree (one workgroup) carries the FRI challenger step as k_keccak_top does.  Layers group by log2(parents) / 5 (DeviceTree, host/pcs.hpp).
constexpr uint64_t POSEIDON_ROW_MAX = 16384;
constexpr int POSEIDON_LEVELS_PER_LAUNCH = 5, POSEIDON_LEVELS_ROWS = 16;
#ifndef VGPU_POSEIDON_ROW_LDS
#define VGPU_POSEIDON_ROW_LDS 1  // the row permutation's per-lane tables (full-round constants, sparse rounds, dense matrix: 1080 words) staged in LDS by the workgroup; 0: read through L1 / L2 (A/B builds)
#endif
constexpr int POPT_ROW_WORDS = POPT_F + 256;  // RC_FULL | T | SPARSE | F: contiguous at the head of the table image
// call at the top of a kernel, by every thread of the workgroup, before any early return
__device__ __forceinline__ PoseidonTab poseidon_row_tab_in_lds(const PoseidonTab& tab, uint32_t* s_row) {
#if VGPU_POSEIDON_ROW_LDS
    if (tab.opt == nullptr) return tab;
    for (int i = (int)threadIdx.x; i < POPT_ROW_WORDS; i += (int)blockDim.x) s_row[i] = tab.opt[i];
    __syncthreads();
    return PoseidonTab{tab.rc, tab.mds, s_row};
#else
    return tab;
#endif
}
__global__ void __launch_bounds__(256) k_poseidon_levels_row(KeccakTopArgs a, PoseidonTab gtab) {
    __shared__ uint32_t dig[2][POSEIDON_LEVELS_ROWS * 8];  // the digests a layer hands to the next one (double-buffered: one barrier per layer)
    __shared__ uint32_t s_row[POPT_ROW_WORDS];
    const PoseidonTab tab = poseidon_row_tab_in_lds(gtab, s_row);
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = tab.mds[i];
    const int row = (int)(threadIdx.x >> 4), l16 = (int)(threadIdx.x & 15);
    const uint64_t rows0 = a.first_len < (uint64_t)POSEIDON_LEVELS_ROWS ? a.first_len : (uint64_t)POSEIDON_LEVELS_ROWS;  // nodes of the first layer in this workgroup
    const uint64_t base0 = (uint64_t)blockIdx.x * rows0;
    for (int l = 0; l < a.levels; l++) {
        const uint64_t cnt = rows0 >> l, node = (base0 >> l) + (uint64_t)row;  // this workgroup's nodes of layer l: cnt of them from base0 >> l
        if ((uint64_t)row < cnt) {
            // lanes 0..7: the left child's digest, 8..15: the right one's — from the layer below in HBM (first layer) or from the previous layer's LDS slots
            const uint32_t child = l == 0 ? a.prev[16 * node + l16] : dig[(l - 1) & 1][(2 * row + (l16 >> 3)) * 8 + (l16 & 7)];
            Fp st = poseidon16_row(Fp::from_canonical(child), tab, m, l16);
            const int n_elems = a.n_elems[l];
            if (n_elems > 0) {  // C(parent, H(rows)) at a layer that injects shorter matrices (FieldMerkleTree, App. B5)
                const uint32_t* const* cols = a.cols[l];
                const Fp d = st;
                Fp h = Fp::zero();
                for (int b = 0; b < n_elems; b += 8) {  // PaddingFreeSponge: each chunk of 8 overwrites the head of the state
                    if (l16 < 8 && b + l16 < n_elems) h = Fp::raw(cols[b + l16][node]);
                    h = poseidon16_row(h, tab, m, l16);
                }
                const uint32_t hs = p_row_ror<8>(h.v);  // lane 8 + k reads lane k
                st = poseidon16_row(l16 < 8 ? d : Fp::raw(hs), tab, m, l16);
            }
            if (l16 < 8) {
                const uint32_t w = st.canonical();
                a.out[l][8 * node + l16] = w;
                dig[l & 1][row * 8 + l16] = w;
            }
        }
        __syncthreads();
    }
    if (a.ch_pos && threadIdx.x < 64) {  // (one workgroup: the launch that ends at the root)
        __threadfence_block();
        fri_challenge_step((int)threadIdx.x, a.ch_pos, a.ch_state, a.out[a.levels - 1], a.ch_beta5, a.ch_commit8);
    }
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
// the bottom of a big tree at query time (merkle.hip, k_keccak_bottom_q: same job words).  32 lanes per job, the ROW permutation: lanes 0..15 hash row `node`
// (level 0) or row 2 node (level 1), lanes 16..31 row 2 node + 1; at level 1 the first row then compresses its digest with the second one's.
__global__ void __launch_bounds__(256) k_poseidon_bottom_q(const uint32_t* __restrict__ jobs, uint32_t n_jobs, const uint32_t* __restrict__ indices, PoseidonTab gtab, uint32_t* __restrict__ dst) {
    __shared__ uint32_t s_row[POPT_ROW_WORDS];
    const PoseidonTab tab = poseidon_row_tab_in_lds(gtab, s_row);
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = tab.mds[i];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, j = t >> 5;
    if (j >= n_jobs) return;  // whole half-waves leave together
    const int grp = (int)((t >> 4) & 1u), l16 = (int)(t & 15u);
    const uint32_t* e = jobs + 8 * j;
    const uint64_t ptr = ((uint64_t)e[1] << 32) | e[0], stride = ((uint64_t)e[3] << 32) | e[2];
    const int n_elems = (int)e[4];
    const uint32_t q = e[6] & 0xffu, level = (e[6] >> 8) & 0xffu, shift = e[6] >> 16;
    const uint64_t node = (((uint64_t)indices[q] >> shift) >> level) ^ 1u, row = level ? 2 * node + (uint64_t)grp : node;
    Fp h = Fp::zero();
    for (int b = 0; b < n_elems; b += 8) {  // PaddingFreeSponge: each chunk of 8 overwrites the head of the state
        if (l16 < 8 && b + l16 < n_elems) {
            const uint32_t* col = stride ? reinterpret_cast<const uint32_t*>(ptr) + (uint64_t)(b + l16) * stride : reinterpret_cast<const uint32_t* const*>(ptr)[b + l16];
            h = Fp::raw(col[row]);
        }
        h = poseidon16_row(h, tab, m, l16);
    }
    if (level) {  // compress(left, right): lanes 8..15 of the first row take lanes 0..7 of the second (the Montgomery words re-enter as they are)
        const uint32_t other = (uint32_t)__shfl((int)h.v, (int)((threadIdx.x & 32u) + 16u + (uint32_t)(l16 & 7)), 64);  // source lane within the wave
        h = poseidon16_row(l16 < 8 ? h : Fp::raw(other), tab, m, l16);
    }
    if (grp == 0 && l16 < 8) dst[e[5] + l16] = h.canonical();
}
void launch_poseidon_bottom_q(hipStream_t st, const uint32_t* pos_dev, bool sparse, const uint32_t* jobs_dev, uint32_t n_jobs, const uint32_t* indices_dev, uint32_t* dst) {
    if (!n_jobs) return;
    ProfScope ps("k_gather", st, 0.0);
    VK_LAUNCH(k_poseidon_bottom_q, dim3((32 * n_jobs + 255) / 256), dim3(256), 0, st, jobs_dev, n_jobs, indices_dev, tab_of(pos_dev, sparse), dst);
}
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
    ProfScope ps("k_poseidon_compress", st, (double)n_out * (96.0 + 4.0 * n_elems), (double)n_out * p_node_perms(n_elems) * POSEIDON_VALU_PER_PERM / 64.0);
    VK_LAUNCH(k_poseidon_compress, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, prev, cols_dev, n_elems, n_out, tab_of(pos_dev, sparse), next);
}
bool poseidon_levels_take(uint64_t parents) { return parents <= POSEIDON_ROW_MAX; }
int poseidon_levels_group(uint64_t parents) { return (int)(vg::log2_strict_u64(parents) / POSEIDON_LEVELS_PER_LAUNCH); }
// a.first_len parents in the first layer, a.levels <= 5 layers, first_len >> (levels - 1) >= first_len / 16 (a workgroup's sub-tree ends at one node at the latest)
void launch_poseidon_levels(hipStream_t st, const uint32_t* pos_dev, bool sparse, const KeccakTopArgs& a) {
    if (a.levels < 1 || a.levels > POSEIDON_LEVELS_PER_LAUNCH || (a.first_len & (a.first_len - 1)) || !a.first_len) throw std::logic_error("poseidon levels: 1..5 layers from a power-of-two first layer");
    const uint64_t rows0 = a.first_len < (uint64_t)POSEIDON_LEVELS_ROWS ? a.first_len : (uint64_t)POSEIDON_LEVELS_ROWS;
    if ((rows0 >> (a.levels - 1)) == 0) throw std::logic_error("poseidon levels: more layers than a workgroup's sub-tree has");
    if (a.ch_pos && a.first_len > rows0) throw std::logic_error("poseidon levels: the challenger step belongs to the launch that ends at the root");
    double bytes = 0, perms = 0;
    for (int l = 0; l < a.levels; l++) {
        bytes += (double)(a.first_len >> l) * (96.0 + 4.0 * a.n_elems[l]);
        perms += (double)(a.first_len >> l) * p_node_perms(a.n_elems[l]);
    }
    ProfScope ps(a.first_len <= rows0 ? "k_poseidon_top" : "k_poseidon_levels_row", st, bytes, perms * POSEIDON_VALU_PER_PERM / 64.0);
    VK_LAUNCH(k_poseidon_levels_row, dim3((unsigned)(a.first_len / rows0)), dim3(256), 0, st, a, tab_of(pos_dev, sparse));
}
// the top of a sharded commitment (W / 2 <= 16 parents down to the root): the same launch as a tree's last group
void launch_poseidon_top(hipStream_t st, const uint32_t* pos_dev, bool sparse, const KeccakTopArgs& a) {
    if (a.first_len > (uint64_t)POSEIDON_LEVELS_ROWS) throw std::invalid_argument("poseidon top: at most 16 parents in the first layer (32 ranks)");
    launch_poseidon_levels(st, pos_dev, sparse, a);
}

}  // namespace vk
