// Opening-phase kernels realising `pcs.open_multi_batches`
// (basic/src/lib.rs:611-619 -> Plonky3 TwoAdicFriPcs; conventions SURVEY.md App. B9/B10):
//   * barycentric evaluation of every committed column at the opening points,
//   * reduced openings per LDE height (each LDE is read ONCE for all of its points),
//   * FRI folding,
//   * query gathers (matrix rows + Merkle sibling paths).
//
// Ext5 vectors of length L (reduced openings, FRI layers) live in "pair layout": a column-major
// (L/2) x 10 matrix whose row r holds f[2r] (columns 0-4) and f[2r+1] (columns 5-9) — exactly the
// matrix ExtensionMmcs commits for a FRI layer (App. B10), so layer trees hash it without reshaping.
#include <cstdlib>
#include <stdexcept>
#include <type_traits>
#include <vector>
#include <string>
#include "launch.hpp"
#include "poseidon_perm.hpp"
#include "challenger_dev.hpp"

namespace vk {

// ---- barycentric weights --------------------------------------------------------------------------
// The first n storage rows of a bit-reversed LDE are the evaluations on s*H_n in bit-reversed order:
// row j <-> x_j = s * r_j, r_j = w_n^{bitrev(j)}.  p(z) = scale * sum_j y_j * r_j / (z - s r_j) with
// scale = (z^n - s^n) / (n s^{n-1}) applied on the host.  w: 5 columns of height n (stride n).
// 1/(z - x) = -g(x)/m(x) with m the minimal polynomial of z over the base field and g = m/(X - z) (see k_reduce_openings; the host ships
// [m0..m4][g0..g3] per point): no extension-field inversion.  Four consecutive rows per thread share ONE base-field inversion.
constexpr int BARY_ROWS = 4;
// For n >= MFMA_DOT_MIN_ROWS the kernel also writes the weights as k_col_dot_mfma's A operand (`img`, see there): 7-bit digit planes, four rows
// to a word, laid out so that a lane of that kernel fetches its four operand registers of a plane with one 16-byte load.
__device__ __forceinline__ void bary_weights_block(uint64_t block, uint64_t n, const uint32_t* __restrict__ mg, uint32_t shift, const DeviceTables& tb, uint32_t* __restrict__ w,
                                                   uint32_t* __restrict__ img) {
    const uint64_t j0 = (block * blockDim.x + threadIdx.x) * BARY_ROWS;
    if (j0 >= n) return;
    const uint32_t* g = mg + 5;
    Fp r[BARY_ROWS], mx[BARY_ROWS], pre[BARY_ROWS];
    Ext5 gx[BARY_ROWS];
    Fp run = Fp::one();
#pragma unroll
    for (int q = 0; q < BARY_ROWS; q++) {
        const bool live = j0 + q < n;
        r[q] = live ? domain_point(tb, (uint32_t)(j0 + q)) : Fp::one();
        const Fp x = Fp::raw(shift) * r[q], x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x;
        const uint64_t tm = (uint64_t)mg[1] * x.v + (uint64_t)mg[2] * x2.v + (uint64_t)mg[3] * x3.v + (uint64_t)mg[4] * x4.v;
        mx[q] = Fp::raw(vg::monty_reduce_wide(tm)) + Fp::raw(mg[0]) + x5;  // z is out of the domain: never zero
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const uint64_t tg = (uint64_t)g[5 + k] * x.v + (uint64_t)g[10 + k] * x2.v + (uint64_t)g[15 + k] * x3.v;
            gx[q].c[k] = Fp::raw(vg::monty_reduce_wide(tg)) + Fp::raw(g[k]);
        }
        gx[q].c[0] += x4;
        pre[q] = run;
        run = run * mx[q];
    }
    Fp inv = run.inv();
    Ext5 res[BARY_ROWS];
#pragma unroll
    for (int q = BARY_ROWS - 1; q >= 0; q--) {
        const Fp minv = inv * pre[q];
        inv = inv * mx[q];
        res[q] = gx[q] * (-(minv * r[q]));
    }
    if (img) {  // n is a multiple of 64 here
        const uint64_t slab = j0 >> 6, gg = (j0 >> 4) & 3, qq = (j0 >> 2) & 3;
#pragma unroll
        for (int k = 0; k < 5; k++)
#pragma unroll
            for (int a = 0; a < 5; a++) {
                uint32_t word = 0;
#pragma unroll
                for (int q = 0; q < BARY_ROWS; q++) word |= ((res[q].c[k].v >> (7 * a)) & 0x7fu) << (8 * q);
                img[((((slab * 5 + a) * 4 + gg) * 5 + k) << 2) + qq] = word;
            }
    }
    if (j0 + BARY_ROWS <= n) {  // n >= 4 is a power of two: the four rows of a limb are one aligned 16-byte store
#pragma unroll
        for (int k = 0; k < 5; k++) *reinterpret_cast<uint4*>(w + (uint64_t)k * n + j0) = make_uint4(res[0].c[k].v, res[1].c[k].v, res[2].c[k].v, res[3].c[k].v);
    } else {
#pragma unroll
        for (int q = 0; q < BARY_ROWS; q++) if (j0 + q < n) store_ext(w, n, j0 + q, res[q]);
    }
}

__global__ void __launch_bounds__(256) k_bary_weights(uint64_t n, const uint32_t* __restrict__ mg, uint32_t shift, DeviceTables tb, uint32_t* __restrict__ w,
                                                      uint32_t* __restrict__ img) {
    bary_weights_block(blockIdx.x, n, mg, shift, tb, w, img);
}
// All weight vectors of an opening in ONE launch: a proof needs one per (height, point) — seventeen for the BasicMachine, most of them a handful
// of blocks — and as launches of their own they sit one after the other at the head of the opening phase (0.27 ms of a lone proof with
// nothing else on the GPU).  jobs: n_jobs x { first block, n (u64), mg pointer (u64), w pointer (u64), img pointer (u64) }, first blocks ascending.
struct BaryJob { uint32_t first_block, shift; uint64_t n; const uint32_t* mg; uint32_t* w; uint32_t* img; };  // shift (Montgomery): 0 = the launch's shift (the sharded prover's range weights carry their own)
__global__ void __launch_bounds__(256) k_bary_weights_batch(const BaryJob* __restrict__ jobs, uint32_t n_jobs, uint32_t shift, DeviceTables tb) {
    uint32_t lo = 0, hi = n_jobs - 1;  // the job of this block: the last one whose first block is <= blockIdx.x (block-uniform)
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (jobs[mid].first_block <= blockIdx.x) lo = mid; else hi = mid - 1; }
    const BaryJob j = jobs[lo];
    bary_weights_block(blockIdx.x - j.first_block, j.n, j.mg, j.shift ? j.shift : shift, tb, j.w, j.img);
}

// ---- column dot products --------------------------------------------------------------------------
// Y[c][p] = sum_rows M[row][c] * w_p[row]  (base x Ext5): a skinny GEMM (10 x n) * (n x C) over F_p.
// A block stages a tile of DOT_TR rows of M (all C columns, coalesced column segments) and of the weights in
// LDS, both ROW-CONTIGUOUS per column; thread (c, pk) owns ONE output limb and walks the tile four rows at a time
// with one ds_read_b128 per operand (lanes sharing c or pk read the same address: broadcast).  Products are
// accumulated unreduced: four v_mad_u64_u32 into a 64-bit partial, folded into a 96-bit accumulator with three
// carry adds — no Montgomery reduction inside the loop, one per thread at the very end.  No cross-lane reduction.
// Blocks grid-stride over row tiles and emit one partial per block.
constexpr int DOT_THREADS = 256, DOT_TR = 128, DOT_TRP = DOT_TR + 4, DOT_MAX_PASSES = 4, DOT_MAX_BLOCKS = 1024;

__device__ __forceinline__ Fp wave_sum(Fp v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += Fp::raw((uint32_t)__shfl_xor((int)v.v, off, 64));
    return v;
}

struct Acc96 {
    uint32_t lo, mid, hi;
    __device__ __forceinline__ void add(uint64_t t) {
        const uint64_t s = (uint64_t)lo + (uint32_t)t;
        lo = (uint32_t)s;
        const uint64_t m = (uint64_t)mid + (uint32_t)(t >> 32) + (uint32_t)(s >> 32);
        mid = (uint32_t)m;
        hi += (uint32_t)(m >> 32);
    }
    // (lo + mid 2^32 + hi 2^64) / 2^32 mod p: the Montgomery-form value of the accumulated sum of products
    __device__ __forceinline__ Fp reduce() const {
        uint32_t m = mid;
        if (m >= vg::P) m -= vg::P;
        if (m >= vg::P) m -= vg::P;
        return Fp::raw(vg::monty_reduce((uint64_t)lo)) + Fp::raw(m) + Fp::from_canonical(hi);
    }
};

template <int NP>
__global__ void __launch_bounds__(DOT_THREADS) k_col_dot(DMatView m, uint64_t n, const uint32_t* __restrict__ w0, const uint32_t* __restrict__ w1, uint32_t* __restrict__ partial) {
    extern __shared__ uint32_t lds[];
    constexpr int PK = NP * 5;
    const int C = (int)m.width, n_out = C * PK;
    uint32_t* Mt = lds;                 // [C][DOT_TRP]
    uint32_t* Wt = lds + C * DOT_TRP;   // [PK][DOT_TRP]
    Acc96 acc[DOT_MAX_PASSES];
#pragma unroll
    for (int q = 0; q < DOT_MAX_PASSES; q++) acc[q] = Acc96{0, 0, 0};
    const uint64_t n_tiles = (n + DOT_TR - 1) / DOT_TR;
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t row0 = t * DOT_TR;
        const int rows = (int)((n - row0) < (uint64_t)DOT_TR ? (n - row0) : (uint64_t)DOT_TR);
        {   // staging: a thread's row r = tid & 127 is fixed, its columns are c0, c0 + 2, ...: four loads in flight
            const int r = threadIdx.x & (DOT_TR - 1);
            const uint32_t* src = m.data + row0 + r;
            uint32_t* dstp = Mt + r;
            const bool live = r < rows;
            int c = threadIdx.x / DOT_TR;  // 0 or 1
            for (; c + 6 < C; c += 8) {
                uint32_t v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = live ? src[(uint64_t)(c + 2 * u) * m.stride] : 0u;
#pragma unroll
                for (int u = 0; u < 4; u++) dstp[(c + 2 * u) * DOT_TRP] = v[u];
            }
            for (; c < C; c += 2) dstp[c * DOT_TRP] = live ? src[(uint64_t)c * m.stride] : 0u;
        }
        {   // weights: same fixed-row scheme, all of a thread's limb loads issued before the LDS stores
            const int r = threadIdx.x & (DOT_TR - 1);
            const bool live = r < rows;
            uint32_t v[(PK + 1) / 2];
#pragma unroll
            for (int u = 0; u < (PK + 1) / 2; u++) {
                const int pk = (int)(threadIdx.x / DOT_TR) + 2 * u;
                const uint32_t* w = pk < 5 ? w0 : w1;
                v[u] = (live && pk < PK) ? w[(uint64_t)(pk < 5 ? pk : pk - 5) * n + row0 + r] : 0u;
            }
#pragma unroll
            for (int u = 0; u < (PK + 1) / 2; u++) {
                const int pk = (int)(threadIdx.x / DOT_TR) + 2 * u;
                if (pk < PK) Wt[pk * DOT_TRP + r] = v[u];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < DOT_MAX_PASSES; q++) {
            int idx = q * DOT_THREADS + threadIdx.x;
            if (idx < n_out) {
                int c = idx / PK, pk = idx - c * PK;
                const uint4* mp = reinterpret_cast<const uint4*>(Mt + c * DOT_TRP);
                const uint4* wp = reinterpret_cast<const uint4*>(Wt + pk * DOT_TRP);
                Acc96 a = acc[q];
#pragma unroll 4
                for (int r = 0; r < DOT_TR / 4; r++) {
                    const uint4 mv = mp[r], wv = wp[r];
                    uint64_t t4 = (uint64_t)mv.x * wv.x;  // four products < 4 p^2 < 2^64
                    t4 += (uint64_t)mv.y * wv.y;
                    t4 += (uint64_t)mv.z * wv.z;
                    t4 += (uint64_t)mv.w * wv.w;
                    a.add(t4);
                }
                acc[q] = a;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < DOT_MAX_PASSES; q++) {
        int idx = q * DOT_THREADS + threadIdx.x;
        if (idx < n_out) partial[(uint64_t)blockIdx.x * n_out + idx] = acc[q].reduce().v;
    }
}

// ---- the same product on the matrix cores --------------------------------------------------------------
// Y = W^T M is a skinny GEMM ((5 NP) x n times n x C) over F_p, and exact in integers if the operands are cut into 7-bit digits:
// W = sum_a 128^a W_a, M = sum_b 128^b M_b, so Y = sum_s 128^s sum_{a+b=s} W_a^T M_b with every W_a^T M_b an i8 x i8 -> i32 MFMA
// (v_mfma_i32_16x16x64_i8: 16 output rows = the 5 NP limbs, 16 columns of M, 64 rows of the reduction per instruction; nine i32
// accumulator tiles, one per s, hold 16384 rows without overflow: 5 * 127^2 * 16384 < 2^31).  A wave owns 16 columns x one chunk of rows;
// per 64 rows it loads its 16 values (four 16-byte loads), cuts them into the five digit planes (the only VALU work left: ~9 instructions
// per element where the LDS-fed multiply-add loop above needs ~40), fetches the weights' planes ready-made (k_bary_weights' `img`) and
// issues 25 MFMAs.  Operand convention checked on the device by tools/mfma_i8_probe.hip: lane l holds row / column l % 16 and
// k = 16 (l / 16) .. + 15 in the bytes of its four registers, the same map for A and B; D at column l & 15, rows 4 (l >> 4) + r.
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr uint64_t MFMA_DOT_MIN_ROWS = 1024, MFMA_DOT_MAX_CHUNK = 16384;
__host__ __device__ inline uint64_t mfma_dot_chunk_rows(uint64_t n) {
    uint64_t c = n / 2048;  // up to 2048 row chunks x ceil(C / 16) column groups of waves
    return c < 64 ? 64 : (c > MFMA_DOT_MAX_CHUNK ? MFMA_DOT_MAX_CHUNK : c);
}

template <int NP>
__global__ void __launch_bounds__(256) k_col_dot_mfma(DMatView m, uint64_t n, const uint32_t* __restrict__ img0, const uint32_t* __restrict__ img1, uint64_t chunk_rows,
                                                      uint32_t* __restrict__ partial) {
    constexpr int PK = NP * 5;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
    const uint64_t C = m.width, n_groups = (C + 15) / 16, n_chunks = n / chunk_rows;
    const uint64_t item = (uint64_t)blockIdx.x * 4 + wave;
    if (item >= n_groups * n_chunks) return;
    const uint64_t group = item % n_groups, chunk = item / n_groups;
    const uint64_t c = group * 16 + j;
    // lanes beyond the 5 NP limbs / beyond the last column compute output rows / columns nobody reads: they load valid memory and are not masked
    const bool col_live = c < C;
    const uint32_t* img = (j < 5 || NP == 1) ? img0 : img1;
    const int limb = j % 5;
    const uint32_t* colp = m.data + (col_live ? c : 0) * m.stride + 16 * g;
    v4i acc[9];
#pragma unroll
    for (int s = 0; s < 9; s++) acc[s] = v4i{0, 0, 0, 0};
    const uint64_t slab0 = chunk * chunk_rows / 64, slab1 = slab0 + chunk_rows / 64;
    // software pipeline: the NEXT slab's operands are in flight while this one is cut into digits and multiplied
    v4i A[5], An[5];
    uint4 v[4], vn[4];
    auto fetch = [&](uint64_t slab, v4i (&a5)[5], uint4 (&v4)[4]) {
#pragma unroll
        for (int a = 0; a < 5; a++) a5[a] = *reinterpret_cast<const v4i*>(img + ((((slab * 5 + a) * 4 + g) * 5 + limb) << 2));
        const uint4* src = reinterpret_cast<const uint4*>(colp + slab * 64);
#pragma unroll
        for (int q = 0; q < 4; q++) v4[q] = src[q];
    };
    fetch(slab0, A, v);
    for (uint64_t slab = slab0; slab < slab1; slab++) {
        fetch(slab + 1 < slab1 ? slab + 1 : slab, An, vn);  // the last iteration re-reads its own slab (no branch around the loads)
        v4i B[5];
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t d0 = __builtin_amdgcn_ubfe(v[q].x, 7 * a, 7), d1 = __builtin_amdgcn_ubfe(v[q].y, 7 * a, 7);
                const uint32_t d2 = __builtin_amdgcn_ubfe(v[q].z, 7 * a, 7), d3 = __builtin_amdgcn_ubfe(v[q].w, 7 * a, 7);
                B[a][q] = (int)(d0 | (d1 << 8) | (d2 << 16) | (d3 << 24));
            }
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
            for (int b = 0; b < 5; b++) acc[a + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[a], B[b], acc[a + b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 5; a++) A[a] = An[a];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = vn[q];
    }
    // Y'(pk, c) = sum_s 128^s acc[s] is the INTEGER sum of products of the stored (Montgomery) words; the stored word of the field product
    // is Y' / 2^32 mod p: evaluate in F_p (stored = Y' 2^32), then two Montgomery reductions
    Fp pw[9];
    pw[0] = Fp::one();
    const Fp step = Fp::from_canonical(128);
#pragma unroll
    for (int s = 1; s < 9; s++) pw[s] = pw[s - 1] * step;
    const uint64_t n_out = C * PK;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int pk = 4 * g + r;
        if (pk < PK && col_live) {
            Fp y = Fp::zero();
#pragma unroll
            for (int s = 0; s < 9; s++) y += Fp::from_canonical((uint32_t)acc[s][r]) * pw[s];
            partial[chunk * n_out + c * PK + pk] = vg::monty_reduce((uint64_t)vg::monty_reduce((uint64_t)y.v));
        }
    }
}

// One wave per (col, p): out[(col * NP + p) * 5 + k] = canonical( scale_p * sum_blocks partial[block][col][p][k] )
__global__ void __launch_bounds__(64) k_col_dot_finish(const uint32_t* __restrict__ partial, uint64_t n_blocks, uint64_t width, int NP, const uint32_t* __restrict__ scale5 /* NP x 5 */,
                                 uint32_t* __restrict__ out) {
    const uint64_t idx = blockIdx.x;  // (col, p)
    const uint64_t c = idx / NP;
    const int p = (int)(idx % NP);
    const uint64_t n_out = width * NP * 5;
    Ext5 acc = Ext5::zero();
    for (uint64_t b = threadIdx.x; b < n_blocks; b += 64) acc += ext_from_words(partial + b * n_out + (c * NP + p) * 5);
#pragma unroll
    for (int k = 0; k < 5; k++) acc.c[k] = wave_sum(acc.c[k]);
    if (threadIdx.x == 0) {
        acc = acc * ext_from_words(scale5 + 5 * p);
        for (int k = 0; k < 5; k++) out[idx * 5 + k] = acc.c[k].canonical();
    }
}

// Every finish of an opening in ONE launch (round 5): behind each of a proof's 42 column-dot launches sat a finish launch of width x points waves — 5-15 us
// each, serialised on its stream, 0.3 ms of a lone proof.  jobs: first blocks ascending; block b of job j = its (column, point) pair b - first_block.
struct DotFinishJob { uint32_t first_block, np; const uint32_t* partial; uint64_t n_blocks, width; const uint32_t* scale5; uint32_t* out; };
__global__ void __launch_bounds__(64) k_col_dot_finish_batch(const DotFinishJob* __restrict__ jobs, uint32_t n_jobs) {
    uint32_t lo = 0, hi = n_jobs - 1;  // the job of this block: the last one whose first block is <= blockIdx.x (block-uniform)
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (jobs[mid].first_block <= blockIdx.x) lo = mid; else hi = mid - 1; }
    const DotFinishJob j = jobs[lo];
    const uint64_t idx = blockIdx.x - j.first_block;  // (col, p)
    const uint64_t c = idx / j.np;
    const int p = (int)(idx % j.np);
    const uint64_t n_out = j.width * j.np * 5;
    Ext5 acc = Ext5::zero();
    for (uint64_t b = threadIdx.x; b < j.n_blocks; b += 64) acc += ext_from_words(j.partial + b * n_out + (c * j.np + p) * 5);
#pragma unroll
    for (int k = 0; k < 5; k++) acc.c[k] = wave_sum(acc.c[k]);
    if (threadIdx.x == 0) {
        acc = acc * ext_from_words(j.scale5 + 5 * p);
        for (int k = 0; k < 5; k++) j.out[idx * 5 + k] = acc.c[k].canonical();
    }
}

// ---- reduced openings -----------------------------------------------------------------------------
// Descriptor (u32 words, uniform loads) for one LDE height:
//   [0] n_mats  [1] n_points  [2] max_width
//   [3 .. 3 + 25*n_points)                      per distinct opening point z_p: its minimal polynomial m_p over the base field, X^5 + m4 X^4 + .. + m0
//                                               ([m0..m4], Montgomery), and g_p = m_p / (X - z_p) = X^4 + g3 X^3 + .. + g0 ([g0 (5)] .. [g3 (5)], Ext5)
//   then alpha powers alpha^c, c < max_width      (5 words each)
//   then per matrix: [col_ptr_lo] [col_ptr_hi] [stride_lo] [stride_hi] [width] [n_pts]
//                    then n_pts x { [point slot] [coef: alpha^offset (5)] [Y = sum_c alpha^c y_c (5)] }
// ro[j] = sum_p 1/(z_p - x_j) * sum_{(mat, p)} coef * (Y - sum_c alpha^c M[j][c])      (App. B9)
// Output in pair layout: `out` is (L/2) x 10 column-major with stride L/2.
constexpr int MAX_OPEN_POINTS = MAX_OPEN_POINTS_PER_LAUNCH;

__global__ void __launch_bounds__(256) k_reduce_openings(const uint32_t* __restrict__ desc, uint64_t L, uint32_t shift, DeviceTables tb, uint32_t* __restrict__ out, int accumulate) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= L) return;
    const uint32_t n_mats = desc[0], n_points = desc[1], max_w = desc[2];
    const uint32_t* mg = desc + 3;
    const uint32_t* apow = mg + 25 * n_points;
    const uint32_t* md = apow + 5 * max_w;
    Ext5 S[MAX_OPEN_POINTS];
#pragma unroll
    for (int p = 0; p < MAX_OPEN_POINTS; p++) S[p] = Ext5::zero();
    for (uint32_t mi = 0; mi < n_mats; mi++) {
        const uint32_t* colp = reinterpret_cast<const uint32_t*>(((uint64_t)md[1] << 32) | md[0]);
        const uint64_t stride = ((uint64_t)md[3] << 32) | md[2];
        const uint32_t width = md[4], npts = md[5];
        md += 6;
        // reduced row sum_c alpha^c * M[j][c], lazily: 4 columns per Montgomery reduction and limb.  The kernel is a stream over
        // every committed LDE (2 GB per proof) and lives on memory-level parallelism: SIXTEEN independent column loads are issued
        // before the first is consumed (the alpha powers are wave-uniform scalar loads).
        Ext5 rr = Ext5::zero();
        uint32_t c = 0;
        for (; c + 16 <= width; c += 16) {
            uint32_t v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = colp[(uint64_t)(c + u) * stride + j];
#pragma unroll
            for (int g = 0; g < 16; g += 4) {
                const uint32_t* a0 = apow + 5 * (c + g);
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    uint64_t t = (uint64_t)a0[k] * v[g] + (uint64_t)a0[5 + k] * v[g + 1] + (uint64_t)a0[10 + k] * v[g + 2] + (uint64_t)a0[15 + k] * v[g + 3];
                    rr.c[k] += Fp::raw(vg::monty_reduce_wide(t));
                }
            }
        }
        for (; c + 4 <= width; c += 4) {
            const uint32_t v0 = colp[(uint64_t)c * stride + j], v1 = colp[(uint64_t)(c + 1) * stride + j];
            const uint32_t v2 = colp[(uint64_t)(c + 2) * stride + j], v3 = colp[(uint64_t)(c + 3) * stride + j];
            const uint32_t* a0 = apow + 5 * c;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                uint64_t t = (uint64_t)a0[k] * v0 + (uint64_t)a0[5 + k] * v1 + (uint64_t)a0[10 + k] * v2 + (uint64_t)a0[15 + k] * v3;
                rr.c[k] += Fp::raw(vg::monty_reduce_wide(t));
            }
        }
        for (; c < width; c++) rr += ext_from_words(apow + 5 * c) * Fp::raw(colp[(uint64_t)c * stride + j]);
        for (uint32_t q = 0; q < npts; q++, md += 11) {
            Ext5 t = ext_from_words(md + 1) * (ext_from_words(md + 6) - rr);
            const uint32_t slot = md[0];
#pragma unroll
            for (int p = 0; p < MAX_OPEN_POINTS; p++) if ((uint32_t)p == slot) S[p] += t;
        }
    }
    const Fp x = Fp::raw(shift) * domain_point(tb, (uint32_t)j);
    // 1/(z_p - x) WITHOUT an extension-field inversion: x lies in the base field, so with m_p the minimal polynomial of z_p over it
    // (m_p(X) = prod_i (X - frob^i z_p), coefficients in the base field) and g_p(X) = m_p(X) / (X - z_p) = prod_{i>=1} (X - frob^i z_p),
    //     1 / (z_p - x) = -g_p(x) / m_p(x),        m_p(x) in the base field.
    // The host ships m_p (5 words, monic) and g_p (4 Ext5, monic) per point; a row costs x^2..x^5, per point 19 multiply-adds for g_p(x) and
    // m_p(x), and ONE base-field inversion for all points (Montgomery's trick) -- about 900 instructions where the Ext5 inversion with its
    // prefix products cost 2100 (and this kernel spends most of its instructions per ROW, not per element: the tall matrices are narrow).
    const Fp x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x;
    Ext5 gx[MAX_OPEN_POINTS];
    Fp mx[MAX_OPEN_POINTS], pre[MAX_OPEN_POINTS];
    Fp run = Fp::one();
#pragma unroll
    for (int p = 0; p < MAX_OPEN_POINTS; p++)
        if ((uint32_t)p < n_points) {
            const uint32_t* q = mg + 25 * p;
            const uint64_t tm = (uint64_t)q[1] * x.v + (uint64_t)q[2] * x2.v + (uint64_t)q[3] * x3.v + (uint64_t)q[4] * x4.v;
            mx[p] = Fp::raw(vg::monty_reduce_wide(tm)) + Fp::raw(q[0]) + x5;
            const uint32_t* g = q + 5;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const uint64_t tg = (uint64_t)g[5 + k] * x.v + (uint64_t)g[10 + k] * x2.v + (uint64_t)g[15 + k] * x3.v;
                gx[p].c[k] = Fp::raw(vg::monty_reduce_wide(tg)) + Fp::raw(g[k]);
            }
            gx[p].c[0] += x4;
            pre[p] = run;
            run = run * mx[p];
        }
    Fp inv_run = run.inv();
    Ext5 ro = Ext5::zero();
#pragma unroll
    for (int p = MAX_OPEN_POINTS - 1; p >= 0; p--)
        if ((uint32_t)p < n_points) { ro -= (S[p] * gx[p]) * (inv_run * pre[p]); inv_run = inv_run * mx[p]; }
    const uint64_t half = L >> 1;
    if (accumulate) ro += load_ext(out + (j & 1) * 5 * half, half, j >> 1);
    store_ext(out + (j & 1) * 5 * half, half, j >> 1, ro);
}

// ---- the same with R = 2 or 4 consecutive rows per thread (round 5) ---------------------------------------------------------------------------
// Storage rows j0 .. j0 + 3 (j0 a multiple of 4) are the points x, -x, i x, -i x (x = shift w^bitrev(j0), i = the primitive 4th root: consecutive
// rows differ in the top bits of the natural index), so the rows of a thread share x^2 .. x^5 up to signs and factors i; ONE base-field inversion
// serves their R x n_points denominators (the thread-per-row kernel pays one per row: a third of its ~900 per-row instructions); every column is
// read with one 8- or 16-byte load per lane (the narrow, tall matrices — mem's 14 / 10 / 10 columns at 2^23 rows — had four 4-byte loads in flight
// per thread: 590 us for 1.3 GB).  R = 4 holds 4 x n_points Ext5 sums: 172 VGPRs at three points, two waves per SIMD — R is chosen per launch
// (launch_reduce_openings).  Same values as k_reduce_openings (exact field arithmetic); heights below 1024 keep the row kernel.
template <int R> struct RowVec;
template <> struct RowVec<2> { using T = uint2; static __device__ __forceinline__ void get(const T& v, uint32_t (&e)[2]) { e[0] = v.x; e[1] = v.y; } };
template <> struct RowVec<4> { using T = uint4; static __device__ __forceinline__ void get(const T& v, uint32_t (&e)[4]) { e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; } };
template <int NP, int R>
__global__ void __launch_bounds__(256) k_reduce_openings_rows(const uint32_t* __restrict__ desc, uint64_t L, uint32_t shift, DeviceTables tb, uint32_t* __restrict__ out, int accumulate) {
    using V = typename RowVec<R>::T;
    const uint64_t j0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * R;
    if (j0 >= L) return;
    const uint32_t n_mats = desc[0], n_points = desc[1], max_w = desc[2];
    const uint32_t* mg = desc + 3;
    const uint32_t* apow = mg + 25 * n_points;
    const uint32_t* md = apow + 5 * max_w;
    Ext5 S[NP][R];
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
        for (int r = 0; r < R; r++) S[p][r] = Ext5::zero();
    for (uint32_t mi = 0; mi < n_mats; mi++) {
        const uint32_t* colp = reinterpret_cast<const uint32_t*>(((uint64_t)md[1] << 32) | md[0]) + j0;
        const uint64_t stride = ((uint64_t)md[3] << 32) | md[2];
        const uint32_t width = md[4], npts = md[5];
        md += 6;
        Ext5 rr[R];
#pragma unroll
        for (int r = 0; r < R; r++) rr[r] = Ext5::zero();
        auto four_cols = [&](const V& v0, const V& v1, const V& v2, const V& v3, const uint32_t* a0) {
            uint32_t e0[R], e1[R], e2[R], e3[R];
            RowVec<R>::get(v0, e0); RowVec<R>::get(v1, e1); RowVec<R>::get(v2, e2); RowVec<R>::get(v3, e3);
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const uint64_t t = (uint64_t)a0[k] * e0[r] + (uint64_t)a0[5 + k] * e1[r] + (uint64_t)a0[10 + k] * e2[r] + (uint64_t)a0[15 + k] * e3[r];
                    rr[r].c[k] += Fp::raw(vg::monty_reduce_wide(t));
                }
        };
        uint32_t c = 0;
        for (; c + 8 <= width; c += 8) {  // eight vector loads in flight
            V v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const V*>(colp + (uint64_t)(c + u) * stride);
            four_cols(v[0], v[1], v[2], v[3], apow + 5 * c);
            four_cols(v[4], v[5], v[6], v[7], apow + 5 * (c + 4));
        }
        for (; c + 4 <= width; c += 4) {
            V v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const V*>(colp + (uint64_t)(c + u) * stride);
            four_cols(v[0], v[1], v[2], v[3], apow + 5 * c);
        }
        for (; c < width; c++) {
            uint32_t e[R];
            RowVec<R>::get(*reinterpret_cast<const V*>(colp + (uint64_t)c * stride), e);
            const Ext5 a = ext_from_words(apow + 5 * c);
#pragma unroll
            for (int r = 0; r < R; r++) rr[r] += a * Fp::raw(e[r]);
        }
        for (uint32_t q = 0; q < npts; q++, md += 11) {
            const Ext5 coef = ext_from_words(md + 1), Y = ext_from_words(md + 6);
            const uint32_t slot = md[0];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const Ext5 t = coef * (Y - rr[r]);
#pragma unroll
                for (int p = 0; p < NP; p++) if ((uint32_t)p == slot) S[p][r] += t;
            }
        }
    }
    // the points of the rows and their powers: x_r = u_r x with u = (1, -1, i, -i); x_r^e = u_r^e x^e
    const Fp x = Fp::raw(shift) * domain_point(tb, (uint32_t)j0), im = domain_point(tb, 2u);
    const Fp x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x;
    Fp pw[R][5];  // [x, x^2, x^3, x^4, x^5] of row r
    pw[0][0] = x; pw[0][1] = x2; pw[0][2] = x3; pw[0][3] = x4; pw[0][4] = x5;
    pw[1][0] = -x; pw[1][1] = x2; pw[1][2] = -x3; pw[1][3] = x4; pw[1][4] = -x5;
    if constexpr (R == 4) {
        const Fp ix = im * x, ix3 = im * x3, ix5 = im * x5;
        pw[2][0] = ix; pw[2][1] = -x2; pw[2][2] = -ix3; pw[2][3] = x4; pw[2][4] = ix5;
        pw[3][0] = -ix; pw[3][1] = -x2; pw[3][2] = ix3; pw[3][3] = x4; pw[3][4] = -ix5;
    }
    // denominators m_p(x_r) first (their prefix products feed the ONE inversion); g_p(x_r) is formed only when its product is due, so that the
    // R x n_points Ext5 values are never alive together
    Fp mx[NP][R], pre[NP][R];
    Fp run = Fp::one();
#pragma unroll
    for (int p = 0; p < NP; p++)
        if ((uint32_t)p < n_points) {
            const uint32_t* q = mg + 25 * p;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint64_t tm = (uint64_t)q[1] * pw[r][0].v + (uint64_t)q[2] * pw[r][1].v + (uint64_t)q[3] * pw[r][2].v + (uint64_t)q[4] * pw[r][3].v;
                mx[p][r] = Fp::raw(vg::monty_reduce_wide(tm)) + Fp::raw(q[0]) + pw[r][4];  // z is out of the domain: never zero
                pre[p][r] = run;
                run = run * mx[p][r];
            }
        }
    Fp inv_run = run.inv();
    Ext5 ro[R];
#pragma unroll
    for (int r = 0; r < R; r++) ro[r] = Ext5::zero();
    // (instantiated per point: as a loop the body is beyond what `#pragma unroll` accepts, and a rolled loop would index S / mx / pre dynamically)
    auto point_step = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if ((uint32_t)p < n_points) {
            const uint32_t* g = mg + 25 * p + 5;
#pragma unroll
            for (int r = R - 1; r >= 0; r--) {
                Ext5 gx;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const uint64_t tg = (uint64_t)g[5 + k] * pw[r][0].v + (uint64_t)g[10 + k] * pw[r][1].v + (uint64_t)g[15 + k] * pw[r][2].v;
                    gx.c[k] = Fp::raw(vg::monty_reduce_wide(tg)) + Fp::raw(g[k]);
                }
                gx.c[0] += pw[r][3];
                ro[r] -= (S[p][r] * gx) * (inv_run * pre[p][r]);
                inv_run = inv_run * mx[p][r];
            }
        }
    };
    if constexpr (NP > 3) point_step(std::integral_constant<int, 3>{});
    if constexpr (NP > 2) point_step(std::integral_constant<int, 2>{});
    if constexpr (NP > 1) point_step(std::integral_constant<int, 1>{});
    point_step(std::integral_constant<int, 0>{});
    // pair layout: the even rows -> columns 0..4, the odd rows -> columns 5..9, at index row / 2
    const uint64_t half = L >> 1, at = j0 >> 1;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        uint32_t* pe = out + (uint64_t)k * half + at;
        uint32_t* po = out + (uint64_t)(5 + k) * half + at;
        if constexpr (R == 4) {
            Fp e0 = ro[0].c[k], e1 = ro[2].c[k], o0 = ro[1].c[k], o1 = ro[3].c[k];
            if (accumulate) { const uint2 a = *reinterpret_cast<uint2*>(pe), b = *reinterpret_cast<uint2*>(po); e0 += Fp::raw(a.x); e1 += Fp::raw(a.y); o0 += Fp::raw(b.x); o1 += Fp::raw(b.y); }
            *reinterpret_cast<uint2*>(pe) = make_uint2(e0.v, e1.v);
            *reinterpret_cast<uint2*>(po) = make_uint2(o0.v, o1.v);
        } else {
            Fp e0 = ro[0].c[k], o0 = ro[1].c[k];
            if (accumulate) { e0 += Fp::raw(*pe); o0 += Fp::raw(*po); }
            *pe = e0.v;
            *po = o0.v;
        }
    }
}

// ---- FRI fold ---------------------------------------------------------------------------------------
// in: (L/2) x 10 pair layout (stride L/2).  out[i] = (f0 + f1)/2 + (beta/2) x_i^{-1} (f0 - f1) [+ add[i]],
// x_i^{-1} = w_L^{-bitrev(i)} (no coset shift inside FRI), written in pair layout of length L/2.
__global__ void __launch_bounds__(256) k_fri_fold(const uint32_t* __restrict__ in, uint64_t L, const uint32_t* __restrict__ beta5 /* device, Montgomery */,
                           const uint32_t* __restrict__ add, DeviceTables tb, uint32_t* __restrict__ out) {
    const uint64_t half = L >> 1;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    Ext5 f0 = load_ext(in, half, i), f1 = load_ext(in + 5 * half, half, i);
    Ext5 beta = ext_from_words(beta5);
    Fp xinv = inv_domain_point(tb, (uint32_t)(2 * i));  // f[2i], f[2i+1] sit at +-x, x = w_L^{bitrev_L(2i)}
    Ext5 r = (f0 + f1) + beta * ((f0 - f1) * xinv);
#pragma unroll
    for (int k = 0; k < 5; k++) r.c[k] = r.c[k].halve();
    const uint64_t q = half >> 1;  // rows of the output pair matrix (0 when the output has a single element)
    if (q == 0) {                   // L == 2: single output element, stored as a 1-element "vector" in slot 0
        if (add) r += load_ext(add, 1, 0);
        store_ext(out, 1, 0, r);
        return;
    }
    uint32_t* o = out + (i & 1) * 5 * q;
    if (add) r += load_ext(add + (i & 1) * 5 * q, q, i >> 1);
    store_ext(o, q, i >> 1, r);
}

// ---- gathers ----------------------------------------------------------------------------------------
// Descriptor = 6 words: [ptr_lo] [ptr_hi] [stride_lo] [stride_hi] [count | kind << 28] [dst offset (words)]
//   kind 0: Montgomery elements src[k * stride], k < count  -> canonical
//   kind 1: raw words src[k * stride]                        -> copied
__global__ void k_gather(const uint32_t* __restrict__ desc, uint64_t n_desc, uint32_t* __restrict__ dst) {
    uint64_t d = (uint64_t)blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);  // 32 lanes per descriptor
    if (d >= n_desc) return;
    const uint32_t* e = desc + 6 * d;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(((uint64_t)e[1] << 32) | e[0]);
    const uint64_t stride = ((uint64_t)e[3] << 32) | e[2];
    const uint32_t count = e[4] & 0x0fffffffu, kind = e[4] >> 28;
    uint32_t* o = dst + e[5];
    for (uint32_t k = threadIdx.x & 31; k < count; k += 32) {
        uint32_t v = src[(uint64_t)k * stride];
        o[k] = kind == 0 ? Fp::raw(v).canonical() : v;
    }
}

// The query openings from a TEMPLATE (round 4): everything of a gather descriptor except the query index is known before the FRI commit
// phase has finished, so the host builds and uploads the template while the GPU runs that phase and ships only the sampled indices afterwards
// (16 000 descriptors of a C2 proof used to be built between the proof-of-work search and the gather, with the GPU idle).
// Template = 8 words: [base_lo] [base_hi] [stride_lo] [stride_hi] [count | kind << 28] [dst offset] [query | mode << 8 | shift << 16] [aux]
//   mode 0: src = base + (index >> shift)                                     a committed row (count = width, stride = LDE height)
//   mode 1: src = base + 8 * ((index >> shift) ^ 1)                           the sibling digest of a Merkle path level
//   mode 2: src = base + (index >> (shift + 1)) + (bit shift of index set ? 0 : aux)   the sibling VALUE of a FRI layer in pair layout (aux = 5 * half)
__global__ void k_gather_q(const uint32_t* __restrict__ templ, uint64_t n_desc, const uint32_t* __restrict__ indices, uint32_t* __restrict__ dst) {
    uint64_t d = (uint64_t)blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);  // 32 lanes per descriptor
    if (d >= n_desc) return;
    const uint32_t* e = templ + 8 * d;
    const uint32_t* base = reinterpret_cast<const uint32_t*>(((uint64_t)e[1] << 32) | e[0]);
    const uint64_t stride = ((uint64_t)e[3] << 32) | e[2];
    const uint32_t count = e[4] & 0x0fffffffu, kind = e[4] >> 28;
    const uint32_t q = e[6] & 0xffu, mode = (e[6] >> 8) & 0xffu, shift = e[6] >> 16;
    const uint64_t index = indices[q];
    const uint32_t* src = mode == 0 ? base + (index >> shift) : mode == 1 ? base + 8 * ((index >> shift) ^ 1) : base + (index >> (shift + 1)) + (((index >> shift) & 1) ? 0u : e[7]);
    uint32_t* o = dst + e[5];
    for (uint32_t k = threadIdx.x & 31; k < count; k += 32) {
        uint32_t v = src[(uint64_t)k * stride];
        o[k] = kind == 0 ? Fp::raw(v).canonical() : v;
    }
}
void launch_gather_q(hipStream_t st, const uint32_t* templ_dev, uint64_t n_desc, const uint32_t* indices_dev, uint32_t* dst) {
    if (!n_desc) return;
    ProfScope ps("k_gather", st, 0.0);
    VK_LAUNCH(k_gather_q, dim3((unsigned)((n_desc + 7) / 8)), dim3(256), 0, st, templ_dev, n_desc, indices_dev, dst);
}

// ---- Fiat-Shamir on the device for the FRI commit phase: challenger_dev.hpp (the step itself), as a launch of its own here
__global__ void __launch_bounds__(64) k_fri_challenge(const uint32_t* __restrict__ pos, uint32_t* __restrict__ ch, const uint32_t* __restrict__ digest8,
                                                      uint32_t* __restrict__ beta5, uint32_t* __restrict__ commit8) {
    fri_challenge_step((int)threadIdx.x, pos, ch, digest8, beta5, commit8);
}

void launch_fri_challenge(hipStream_t st, const uint32_t* pos_dev, uint32_t* ch_dev, const uint32_t* digest8_dev, uint32_t* beta5_dev, uint32_t* commit8_dev) {
    ProfScope ps("k_fri_challenge", st, 0.0);
    VK_LAUNCH(k_fri_challenge, dim3(1), dim3(64), 0, st, pos_dev, ch_dev, digest8_dev, beta5_dev, commit8_dev);
}

// ---- proof-of-work grinding ---------------------------------------------------------
// check_witness(w): observe(w) then sample_bits(bits) == 0.  With k values pending in the input buffer the
// duplexing overwrites state[0..k] with (pending, w), permutes, and the sample is state[15].  Thread t
// tries w = first + t; the smallest passing witness is kept with atomicMin (canonical rule: smallest).
// pos: [480 round constants][16 circulant MDS coefficients m[d] = sum_k (31 w16^d)^k][16 base state], Montgomery.
__global__ void __launch_bounds__(256) k_pow_grind(const uint32_t* __restrict__ pos, int sparse, uint32_t k_pending, uint32_t first, uint32_t count, uint32_t mask,
                                                   uint32_t* __restrict__ best) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t* base = pos + 496;
    Fp st[16];
#pragma unroll
    for (int i = 0; i < 16; i++) st[i] = Fp::raw(base[i]);
    const Fp wit = Fp::from_canonical(first + t);
#pragma unroll
    for (int i = 0; i < 16; i++) if ((uint32_t)i == k_pending) st[i] = wit;
    // the permutation of the Poseidon MMCS kernels (poseidon_perm.hpp: sparse partial rounds, convolution MDS — a fifth of the plain
    // rounds' instructions; a batch is one wave per CU, i.e. latency-bound: the search costs what one permutation costs)
    poseidon16_permute(st, tab_of(pos, sparse != 0));
    if ((st[15].canonical() & mask) == 0) atomicMin(best, first + t);
}

void launch_pow_grind(hipStream_t st, const uint32_t* pos_dev, bool sparse, uint32_t k_pending, uint32_t first, uint32_t count, uint32_t bits, uint32_t* best_dev) {
    ProfScope ps("k_pow_grind", st, 0.0);
    VK_LAUNCH(k_pow_grind, dim3((count + 255) / 256), dim3(256), 0, st, pos_dev, sparse ? 1 : 0, k_pending, first, count, (1u << bits) - 1u, best_dev);
}

// ---- launchers ----------------------------------------------------------------------------------------
// w: bary_buffer_words(n) words: 5 columns of height n, then (n >= MFMA_DOT_MIN_ROWS) the digit-plane image of the same weights
uint64_t bary_buffer_words(uint64_t n) { return 5 * n + (n >= MFMA_DOT_MIN_ROWS ? (n / 64) * 400 : 0); }
void launch_bary_weights(hipStream_t st, uint64_t n, const uint32_t* min_poly_dev, Fp shift, const DeviceTables& tb, uint32_t* w) {
    ProfScope ps("k_bary_weights", st, 20.0 * n);
    const uint64_t threads = (n + BARY_ROWS - 1) / BARY_ROWS;
    VK_LAUNCH(k_bary_weights, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, n, min_poly_dev, shift.v, tb, w, n >= MFMA_DOT_MIN_ROWS ? w + 5 * n : (uint32_t*)nullptr);
}
// jobs_dev: the job table on the device (5 x u64 words per job as BaryJob lays them out); total_blocks = sum of the jobs' blocks
void launch_bary_weights_batch(hipStream_t st, const uint32_t* jobs_dev, uint32_t n_jobs, uint32_t total_blocks, double total_rows, Fp shift, const DeviceTables& tb) {
    if (!n_jobs) return;
    ProfScope ps("k_bary_weights", st, 20.0 * total_rows);
    VK_LAUNCH(k_bary_weights_batch, dim3(total_blocks), dim3(256), 0, st, (const BaryJob*)jobs_dev, n_jobs, shift.v, tb);
}
uint32_t bary_weights_blocks(uint64_t n) { return (uint32_t)(((n + BARY_ROWS - 1) / BARY_ROWS + 255) / 256); }
bool bary_weights_has_image(uint64_t n) { return n >= MFMA_DOT_MIN_ROWS; }
uint64_t col_dot_slots(uint64_t n) {
    if (n >= MFMA_DOT_MIN_ROWS) return n / mfma_dot_chunk_rows(n);
    uint64_t tiles = (n + DOT_TR - 1) / DOT_TR;
    return tiles < (uint64_t)DOT_MAX_BLOCKS ? tiles : (uint64_t)DOT_MAX_BLOCKS;
}
uint64_t col_dot_max_columns(int np) { return (uint64_t)DOT_MAX_PASSES * DOT_THREADS / (5 * (uint64_t)np); }
// m: LDE (only rows < n are read).  np = 1 or 2 points.  partial: col_dot_slots(n) * width * np * 5 words.
void launch_col_dot_finish_batch(hipStream_t st, const uint32_t* jobs_dev, uint32_t n_jobs, uint32_t total_blocks) {
    if (!n_jobs) return;
    ProfScope ps("k_col_dot_finish", st, 0.0);
    VK_LAUNCH(k_col_dot_finish_batch, dim3(total_blocks), dim3(64), 0, st, reinterpret_cast<const DotFinishJob*>(jobs_dev), n_jobs);
}
uint32_t col_dot_finish_job(std::vector<uint32_t>& jobs, uint32_t first_block, uint64_t n, uint64_t width, int np, const uint32_t* partial, const uint32_t* scale5_dev, uint32_t* out_dev) {
    auto ptr = [&](const void* q) { const uint64_t v = (uint64_t)q; jobs.push_back((uint32_t)v); jobs.push_back((uint32_t)(v >> 32)); };
    auto u64 = [&](uint64_t v) { jobs.push_back((uint32_t)v); jobs.push_back((uint32_t)(v >> 32)); };
    jobs.push_back(first_block); jobs.push_back((uint32_t)np);
    ptr(partial); u64(col_dot_slots(n)); u64(width); ptr(scale5_dev); ptr(out_dev);
    static_assert(sizeof(DotFinishJob) == 48, "job image");
    return first_block + (uint32_t)(width * (uint64_t)np);
}
void launch_col_dot(hipStream_t st, DMatView m, uint64_t n, int np, const uint32_t* w0, const uint32_t* w1, uint32_t* partial,
                    const uint32_t* scale5_dev, uint32_t* out_dev, bool finish) {
    if (np < 1 || np > 2 || m.width > col_dot_max_columns(np)) throw std::runtime_error("col_dot: one launch takes 1 or 2 points and at most 1024 / (5 points) columns (the caller chunks)");
    unsigned blocks = (unsigned)col_dot_slots(n);
    const int pk = np * 5;
    size_t lds = (size_t)DOT_TRP * (m.width + pk) * 4;
    ProfScope ps(n >= MFMA_DOT_MIN_ROWS ? "k_col_dot_mfma" : "k_col_dot", st, 4.0 * n * (m.width + 5.0 * np));
    if (n >= MFMA_DOT_MIN_ROWS) {  // matrix cores; w0 / w1 are bary_buffer_words(n) buffers: the digit planes follow the five weight columns
        const uint64_t chunk = mfma_dot_chunk_rows(n), items = ((m.width + 15) / 16) * (n / chunk);
        const unsigned grid = (unsigned)((items + 3) / 4);
        if (np == 1) VK_LAUNCH(k_col_dot_mfma<1>, dim3(grid), dim3(256), 0, st, m, n, w0 + 5 * n, w1 + 5 * n, chunk, partial);
        else VK_LAUNCH(k_col_dot_mfma<2>, dim3(grid), dim3(256), 0, st, m, n, w0 + 5 * n, w1 + 5 * n, chunk, partial);
        if (finish) VK_LAUNCH(k_col_dot_finish, dim3((unsigned)(m.width * np)), dim3(64), 0, st, partial, (uint64_t)blocks, m.width, np, scale5_dev, out_dev);
        return;
    }
    if (np == 1) VK_LAUNCH(k_col_dot<1>, dim3(blocks), dim3(DOT_THREADS), lds, st, m, n, w0, w1, partial);
    else VK_LAUNCH(k_col_dot<2>, dim3(blocks), dim3(DOT_THREADS), lds, st, m, n, w0, w1, partial);
    if (finish) VK_LAUNCH(k_col_dot_finish, dim3((unsigned)(m.width * np)), dim3(64), 0, st, partial, (uint64_t)blocks, m.width, np, scale5_dev, out_dev);
}
// ---- Y = sum_col alpha^col y_col of every (matrix, point) of an opening, on the device (round 4) --------------------------------------
// k_reduce_openings needs, per (matrix, point), the alpha-weighted sum of the OPENED VALUES of that matrix at that point.  The host used to
// form it from the downloaded values — the one thing that tied the reduced openings to a host round trip.  One wave per (matrix, point) reads
// the values where k_col_dot_finish left them (canonical words), weighs them and writes the sum (Montgomery) into its slot of the reduce
// descriptors: col_dot -> this -> k_reduce_openings run back to back, the values travel to the host beside them.
// desc: per entry at entry_off[e]: [word offset of Y in `pool`] [n_seg] then n_seg x { out_off, np, p_local, c0, cw }: the values of columns
// c0 .. c0 + cw of the matrix at this point sit at vals[out_off + (col * np + p_local) * 5 ..].  apow: alpha^c, 5 words each.
__global__ void __launch_bounds__(64) k_open_y(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ apow, const uint32_t* __restrict__ desc,
                                              const uint32_t* __restrict__ entry_off, uint32_t* __restrict__ pool) {
    const uint32_t* d = desc + entry_off[blockIdx.x];
    const uint32_t y_dst = d[0], n_seg = d[1];
    Ext5 acc = Ext5::zero();
    for (uint32_t sgi = 0; sgi < n_seg; sgi++) {
        const uint32_t* sg = d + 2 + 5 * sgi;
        const uint32_t out_off = sg[0], np = sg[1], pl = sg[2], c0 = sg[3], cw = sg[4];
        for (uint32_t col = threadIdx.x; col < cw; col += 64) {
            const uint32_t* v = vals + out_off + (col * np + pl) * 5;
            Ext5 y;
#pragma unroll
            for (int k = 0; k < 5; k++) y.c[k] = Fp::from_canonical(v[k]);
            acc += ext_from_words(apow + 5 * (c0 + col)) * y;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) acc.c[k] = wave_sum(acc.c[k]);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 5; k++) pool[y_dst + k] = acc.c[k].v;
    }
}
void launch_open_y(hipStream_t st, const uint32_t* vals_dev, const uint32_t* apow_dev, const uint32_t* desc_dev, const uint32_t* entry_off_dev, uint32_t n_entries, uint32_t* pool_dev) {
    if (!n_entries) return;
    ProfScope ps("k_open_y", st, 0.0);
    VK_LAUNCH(k_open_y, dim3(n_entries), dim3(64), 0, st, vals_dev, apow_dev, desc_dev, entry_off_dev, pool_dev);
}
void launch_reduce_openings(hipStream_t st, const uint32_t* desc_dev, uint64_t L, Fp shift, const DeviceTables& tb, uint32_t* out, uint64_t total_width, bool accumulate, int n_points, bool vec_ok) {
    ProfScope ps("k_reduce_openings", st, 4.0 * L * (total_width + (accumulate ? 10.0 : 5.0)));
    // VGPU_REDUCE_ROWS = rows per thread: 4, 2 or 1 (1: the thread-per-row kernel everywhere; A/B).  Default: 4 rows for launches of at most two distinct
    // points, 2 rows for three and four (register budget: see k_reduce_openings_rows)
    static const int rows_env = [] { const char* e = getenv("VGPU_REDUCE_ROWS"); return e ? atoi(e) : 0; }();
    // the rows kernel's invariants (8- / 16-byte column loads at j0 = R * thread, uint2 stores into `out`): see launch.hpp; not met -> the row kernel
    const bool rows_kernel_ok = vec_ok && L >= 1024 && (L & (L - 1)) == 0 && ((uintptr_t)out & 15) == 0;
    if (rows_env != 1 && rows_kernel_ok) {
        const int np = n_points >= 1 && n_points <= MAX_OPEN_POINTS ? n_points : MAX_OPEN_POINTS;
        const int R = rows_env == 2 || rows_env == 4 ? rows_env : (np <= 2 ? 4 : 2);
        const dim3 grid((unsigned)((L / R + 255) / 256)), block(256);
#define VG_RO(NP_, R_) VK_LAUNCH((k_reduce_openings_rows<NP_, R_>), grid, block, 0, st, desc_dev, L, shift.v, tb, out, accumulate ? 1 : 0)
        if (R == 4) { switch (np) { case 1: VG_RO(1, 4); break; case 2: VG_RO(2, 4); break; case 3: VG_RO(3, 4); break; default: VG_RO(4, 4); break; } }
        else { switch (np) { case 1: VG_RO(1, 2); break; case 2: VG_RO(2, 2); break; case 3: VG_RO(3, 2); break; default: VG_RO(4, 2); break; } }
#undef VG_RO
        return;
    }
    VK_LAUNCH(k_reduce_openings, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, desc_dev, L, shift.v, tb, out, accumulate ? 1 : 0);
}
void launch_fri_fold(hipStream_t st, const uint32_t* in, uint64_t L, const uint32_t* beta5_dev, const uint32_t* add, const DeviceTables& tb, uint32_t* out) {
    uint64_t half = L >> 1;
    ProfScope ps("k_fri_fold", st, 20.0 * L + 20.0 * half * (add ? 2 : 1));
    VK_LAUNCH(k_fri_fold, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, st, in, L, beta5_dev, add, tb, out);
}
void launch_gather(hipStream_t st, const uint32_t* desc_dev, uint64_t n_desc, uint32_t* dst) {
    if (!n_desc) return;
    ProfScope ps("k_gather", st, 0.0);
    VK_LAUNCH(k_gather, dim3((unsigned)((n_desc + 7) / 8)), dim3(256), 0, st, desc_dev, n_desc, dst);
}

}  // namespace vk
