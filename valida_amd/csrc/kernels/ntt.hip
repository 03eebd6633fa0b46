// Column-batched radix-2 NTT / coset low-degree extension over BabyBear for gfx950.
//
// Realises `pcs.commit_batches` / `commit_shifted_batches`'s LDE step (basic/src/lib.rs:199,223,258,599 ->
// Plonky3 TwoAdicFriPcs: coset_lde_batch(m, log_blowup, 31/shift_i) then bit_reverse_rows; SURVEY.md
// App. B3/B4, kernel K2).  Per column of height N = 2^k:
//   coefficients = iDFT_N(evals)                       (in-place DIT: bit-reversed-position in, natural out)
//   for each coset t < b = 2^log_blowup:
//       block bitrev_lb(t) of the LDE = DIF-NTT_N( c_i * (s * w_{bN}^t)^i )   (natural in, bit-reversed out)
// so the committed bit-reversed LDE is produced directly, with no permutation pass.
//
// Sizes above 2^12 points use the two-pass ("four-step") split N = N_hi * N_lo, N_lo = 2^12 (2^14 from N = 2^23):
//   inverse: [contiguous DIT over N_lo] * w_N^{-bitrev(h) r}  ->  [strided DIT over N_hi] * 1/N
//   forward: c * shift^i -> [strided DIF over N_hi] * w_N^{r bitrev(h')}  ->  [contiguous DIF over N_lo]
// Every pass stages its tile in LDS (160 KiB/CU) and reads/writes HBM once.  Inside LDS the radix-2
// stages are grouped into ROUNDS of up to 4 stages executed in registers (16 points per work item:
// 32 butterflies between one LDS read and one LDS write, one barrier per round instead of per stage);
// the twiddles of a round come from per-stage compact tables — read through L1/L2 in the contiguous passes
// (16-64 KB shared by every block), staged in LDS in the strided passes; the coset powers and four-step twiddles
// come from small host-prepared / universal root tables instead of per-thread exponentiations.
// No MFMA: 31-bit modular butterflies are VALU work.  Measured balance: DESIGN.md "Measurement".
#include "launch.hpp"
#include "butterfly.hpp"

namespace vk {

// Contiguous tiles (T = 1) are stored padded: word i at i + (i >> 4), so the stride-16 accesses of the
// last round are bank-conflict free.  Strided tiles use element (h, c) at h * LD + c with LD = T + 1.
template <bool PAD> __device__ __forceinline__ int tile_addr(int i, int c, int LD) { return PAD ? i + (i >> 4) : i * LD + c; }

// One round = stages s_lo .. s_lo + R - 1 of the radix-2 network on a tile of 2^logn points x 2^logT columns.
// tw: compact tables in LDS, stage s at offset 2^(s-1) - 1: tw[off + j] = w_{2^s}^{+-j}, j < 2^(s-1).
// LB >= 0 fixes s_lo - 1 at compile time (the 2^12-point contiguous tile: every LDS address of the 16
// points and 15 twiddles becomes base + immediate offset, no per-access address arithmetic).
template <int R>
__device__ __forceinline__ void load_round_twiddles(Fp (&held)[(1 << R) - 1], const uint32_t* tw, int low, int lowbits) {
#pragma unroll
    for (int st = 0; st < R; st++) {
        const uint32_t* t = tw + ((1 << (lowbits + st)) - 1) + low;
#pragma unroll
        for (int k = 0; k < (1 << st); k++) held[(1 << st) - 1 + k] = Fp::raw(t[k << lowbits]);
    }
}

template <int R, bool DIT, bool PAD, int LB>
__device__ __forceinline__ void ntt_round(uint32_t* buf, const uint32_t* tw, int logn, int s_lo, int logT, int LD) {
    constexpr int G = 1 << R;
    const int n_items = (1 << (logn - R)) << logT;
    const int maskT = (1 << logT) - 1, lowbits = LB >= 0 ? LB : s_lo - 1, lowmask = (1 << lowbits) - 1;
    // the 16 points of a work item are base | (g << lowbits); in both layouts their addresses are linear in g
    // unless a padded tile has the g bits straddling bit 4
    const bool linear = !PAD || lowbits >= 4 || lowbits + R <= 4;
    const int gs = PAD ? (lowbits >= 4 ? (1 << lowbits) + (1 << (lowbits - 4)) : (1 << lowbits)) : (LD << lowbits);
    for (int w = threadIdx.x; w < n_items; w += blockDim.x) {
        const int c = w & maskT, q = w >> logT;
        const int low = q & lowmask, base = ((q >> lowbits) << (lowbits + R)) | low;
        uint32_t* p0 = buf + tile_addr<PAD>(base, c, LD);
        Fp x[G];
        if (linear) {
#pragma unroll
            for (int g = 0; g < G; g++) x[g] = Fp::raw(p0[g * gs]);
        } else {
#pragma unroll
            for (int g = 0; g < G; g++) x[g] = Fp::raw(buf[tile_addr<PAD>(base | (g << lowbits), c, LD)]);
        }
        butterflies<R, DIT, LB == 0>(x, tw, low, lowbits);
        if (linear) {
#pragma unroll
            for (int g = 0; g < G; g++) p0[g * gs] = x[g].v;
        } else {
#pragma unroll
            for (int g = 0; g < G; g++) buf[tile_addr<PAD>(base | (g << lowbits), c, LD)] = x[g].v;
        }
    }
    __syncthreads();
}

template <bool DIT, bool PAD>
__device__ __forceinline__ void ntt_round_dispatch(int R, uint32_t* buf, const uint32_t* tw, int logn, int s_lo, int logT, int LD) {
    if (s_lo == 1) {  // the round that holds stage 1: compile-time twiddle exponents, the trivial ones dropped (block-uniform branch)
        switch (R) {
            case 1: ntt_round<1, DIT, PAD, 0>(buf, tw, logn, 1, logT, LD); break;
            case 2: ntt_round<2, DIT, PAD, 0>(buf, tw, logn, 1, logT, LD); break;
            case 3: ntt_round<3, DIT, PAD, 0>(buf, tw, logn, 1, logT, LD); break;
            default: ntt_round<4, DIT, PAD, 0>(buf, tw, logn, 1, logT, LD); break;
        }
        return;
    }
    switch (R) {
        case 1: ntt_round<1, DIT, PAD, -1>(buf, tw, logn, s_lo, logT, LD); break;
        case 2: ntt_round<2, DIT, PAD, -1>(buf, tw, logn, s_lo, logT, LD); break;
        case 3: ntt_round<3, DIT, PAD, -1>(buf, tw, logn, s_lo, logT, LD); break;
        default: ntt_round<4, DIT, PAD, -1>(buf, tw, logn, s_lo, logT, LD); break;
    }
}

// All logn stages, split into ceil(logn / 4) rounds of near-equal size.
// DIT: bit-reversed-position input -> natural output (inverse tables); DIF: natural -> bit-reversed.
template <bool DIT, bool PAD>
__device__ __forceinline__ void tile_transform(uint32_t* buf, const uint32_t* tw, int logn, int logT, int LD) {
    if (logn == 0) return;
    if (PAD && logn == 12) {  // the hot case: three radix-16 rounds with compile-time strides
        if (DIT) {
            ntt_round<4, true, PAD, 0>(buf, tw, 12, 1, 0, LD);
            ntt_round<4, true, PAD, 4>(buf, tw, 12, 5, 0, LD);
            ntt_round<4, true, PAD, 8>(buf, tw, 12, 9, 0, LD);
        } else {
            ntt_round<4, false, PAD, 8>(buf, tw, 12, 9, 0, LD);
            ntt_round<4, false, PAD, 4>(buf, tw, 12, 5, 0, LD);
            ntt_round<4, false, PAD, 0>(buf, tw, 12, 1, 0, LD);
        }
        return;
    }
    if (PAD && logn == 14) {  // 2^14-point contiguous tiles (heights >= 2^23): three radix-16 rounds and one radix-4
        if (DIT) {
            ntt_round<4, true, PAD, 0>(buf, tw, 14, 1, 0, LD);
            ntt_round<4, true, PAD, 4>(buf, tw, 14, 5, 0, LD);
            ntt_round<4, true, PAD, 8>(buf, tw, 14, 9, 0, LD);
            ntt_round<2, true, PAD, 12>(buf, tw, 14, 13, 0, LD);
        } else {
            ntt_round<2, false, PAD, 12>(buf, tw, 14, 13, 0, LD);
            ntt_round<4, false, PAD, 8>(buf, tw, 14, 9, 0, LD);
            ntt_round<4, false, PAD, 4>(buf, tw, 14, 5, 0, LD);
            ntt_round<4, false, PAD, 0>(buf, tw, 14, 1, 0, LD);
        }
        return;
    }
    const int rounds = (logn + 3) >> 2, small = logn / rounds, extra = logn - small * rounds;
    if (DIT) {
        int s_lo = 1;
        for (int i = 0; i < rounds; i++) { int R = small + (i < extra ? 1 : 0); ntt_round_dispatch<true, PAD>(R, buf, tw, logn, s_lo, logT, LD); s_lo += R; }
    } else {
        int s_top = logn;
        for (int i = 0; i < rounds; i++) { int R = small + (i < extra ? 1 : 0); ntt_round_dispatch<false, PAD>(R, buf, tw, logn, s_top - R + 1, logT, LD); s_top -= R; }
    }
}

// Strided tiles of the two hot shapes with everything the address arithmetic needs at compile time: heights 2^20 (2^8 rows x 64 columns) and
// 2^22 / 2^24 (2^10 rows x 16 columns).  The generic rounds compute every LDS address (row x LD + column) and twiddle offset at run time —
// about as many instructions again as the butterflies themselves; here a work item's 2^R points are base + immediate offsets.
template <int R, bool DIT, int LB, int LOGT>
__device__ __forceinline__ void ntt_round_strided(uint32_t* buf, const uint32_t* tw, int logn) {
    constexpr int G = 1 << R, T = 1 << LOGT, LD = T + 1, gs = LD << LB;
    const int n_items = (1 << (logn - R)) << LOGT;
    for (int w = threadIdx.x; w < n_items; w += blockDim.x) {
        const int c = w & (T - 1), q = w >> LOGT;
        const int low = q & ((1 << LB) - 1), base = ((q >> LB) << (LB + R)) | low;
        uint32_t* p0 = buf + base * LD + c;
        Fp x[G];
#pragma unroll
        for (int g = 0; g < G; g++) x[g] = Fp::raw(p0[g * gs]);
        butterflies<R, DIT, LB == 0>(x, tw, low, LB);
#pragma unroll
        for (int g = 0; g < G; g++) p0[g * gs] = x[g].v;
    }
    __syncthreads();
}
// K_HI = 8: two radix-16 rounds; K_HI = 10: radix-4 on the top two stages + two radix-16 rounds
template <bool DIT, int K_HI, int LOGT>
__device__ __forceinline__ void strided_transform(uint32_t* buf, const uint32_t* tw) {
    static_assert(K_HI == 8 || K_HI == 10, "hot shapes only");
    if (DIT) {
        ntt_round_strided<4, true, 0, LOGT>(buf, tw, K_HI);
        ntt_round_strided<4, true, 4, LOGT>(buf, tw, K_HI);
        if (K_HI == 10) ntt_round_strided<2, true, 8, LOGT>(buf, tw, K_HI);
    } else {
        if (K_HI == 10) ntt_round_strided<2, false, 8, LOGT>(buf, tw, K_HI);
        ntt_round_strided<4, false, 4, LOGT>(buf, tw, K_HI);
        ntt_round_strided<4, false, 0, LOGT>(buf, tw, K_HI);
    }
}
#ifndef VGPU_STRIDED_FIXED
#define VGPU_STRIDED_FIXED 1  // 0: the generic rounds for every shape (A/B builds)
#endif
template <bool DIT>
__device__ __forceinline__ void strided_tile_transform(uint32_t* buf, const uint32_t* tw, int k_hi, int logT, int LD) {
    if (VGPU_STRIDED_FIXED && k_hi == 8 && logT == 6) strided_transform<DIT, 8, 6>(buf, tw);
    else if (VGPU_STRIDED_FIXED && k_hi == 10 && logT == 4) strided_transform<DIT, 10, 4>(buf, tw);
    else tile_transform<DIT, false>(buf, tw, k_hi, logT, LD);
}

__device__ __forceinline__ void stage_twiddles(uint32_t* dst, const uint32_t* __restrict__ compact, int logn) {
    const int count = (1 << logn) - 1;
    for (int i = threadIdx.x; i < count; i += blockDim.x) dst[i] = compact[i];
}

// w_{2^m}^x and w_{2^m}^{-x} (x < 2^m) from the universal bit-reversed tables: the product over the set bits i of x of
// w_{2^(m-i)} is domain_point(bitrev_m(x)) — three table lookups and at most two multiplications instead of a pow.
__device__ __forceinline__ Fp root_pow(const DeviceTables& tb, int m, uint32_t x) { return domain_point(tb, m ? __brev(x) >> (32 - m) : 0u); }
__device__ __forceinline__ Fp inv_root_pow(const DeviceTables& tb, int m, uint32_t x) { return inv_domain_point(tb, m ? __brev(x) >> (32 - m) : 0u); }

__host__ __device__ inline int padded_words(int n) { return n + (n >> 4); }

// ---- inverse, contiguous pass: persistent blocks over (N / N_lo) x columns tiles ----------------------
// In place on `data` (column-major, height N = 2^k).  k_lo = min(k, 12) stages on each contiguous block
// of N_lo points.  If k_hi > 0 multiplies element r of block h by w_N^{-bitrev(h) * r}; else scales by 1/N.
// Blocks grid-stride over tiles; the compact twiddle tables (16 KB, shared by every block) are read through L1/L2, which
// halves the LDS footprint of a block (6 instead of 4 resident blocks per CU).
// Software pipeline shared by the two contiguous passes: a block's tiles are independent, so the NEXT tile's global
// loads are issued (into registers) before the butterfly rounds of the current one and land while it computes —
// otherwise all co-resident blocks, started together and doing identical phases, stall on HBM in lockstep.
constexpr int CONTIG_MAX_PER_THREAD = 16;  // n_lo / blockDim.x (launcher: blockDim = max(64, n_lo / 16))
__device__ __forceinline__ void prefetch_tile(uint32_t (&pre)[CONTIG_MAX_PER_THREAD], const uint32_t* __restrict__ src, int n_lo) {
#pragma unroll
    for (int u = 0; u < CONTIG_MAX_PER_THREAD; u++) {
        const int i = threadIdx.x + u * blockDim.x;
        if (i < n_lo) pre[u] = src[i];
    }
}
__device__ __forceinline__ void commit_tile(const uint32_t (&pre)[CONTIG_MAX_PER_THREAD], uint32_t* lds, int n_lo) {
#pragma unroll
    for (int u = 0; u < CONTIG_MAX_PER_THREAD; u++) {
        const int i = threadIdx.x + u * blockDim.x;
        if (i < n_lo) lds[tile_addr<true>(i, 0, 0)] = pre[u];
    }
}

__global__ void k_intt_contig(DMatView m, int k, int k_lo, DeviceTables tb, uint32_t n_inv_mont) {
    extern __shared__ uint32_t lds[];
    const int n_lo = 1 << k_lo, k_hi = k - k_lo;
    const uint32_t* tw = tb.itwc;  // compact per-stage tables straight from L1/L2: no LDS copy, more blocks per CU
    // tiles per column = 2^k_hi: column and tile of a work item by shift and mask (a 64-bit division of wave-uniform values is ~100 scalar instructions and a dozen SGPRs)
    const uint64_t tiles_per_col = m.height >> k_lo, total = tiles_per_col * m.width;
    auto tile_ptr = [&](uint64_t t) { const uint64_t cidx = t >> k_hi; return m.col(cidx) + (t & (tiles_per_col - 1)) * n_lo; };
    uint32_t pre[CONTIG_MAX_PER_THREAD];
    if (blockIdx.x < total) prefetch_tile(pre, tile_ptr(blockIdx.x), n_lo);
    for (uint64_t t = blockIdx.x; t < total; t += gridDim.x) {
        const uint64_t h = t & (tiles_per_col - 1);
        uint32_t* col = tile_ptr(t);
        commit_tile(pre, lds, n_lo);
        __syncthreads();
        if (t + gridDim.x < total) prefetch_tile(pre, tile_ptr(t + gridDim.x), n_lo);
        tile_transform<true, true>(lds, tw, k_lo, 0, 0);
        if (k_hi > 0) {
            // element r of block h is multiplied by w_N^{-e r}, e = bitrev_{k_hi}(h); a thread walks r = tid, tid + B, ...
            const uint32_t e = __brev((uint32_t)h) >> (32 - k_hi), nmask = (1u << k) - 1u;
            Fp step = inv_root_pow(tb, k, (e * blockDim.x) & nmask), cur = inv_root_pow(tb, k, (e * threadIdx.x) & nmask);
            for (int i = threadIdx.x; i < n_lo; i += blockDim.x) { col[i] = (Fp::raw(lds[tile_addr<true>(i, 0, 0)]) * cur).v; cur *= step; }
        } else {
            Fp ninv = Fp::raw(n_inv_mont);
            for (int i = threadIdx.x; i < n_lo; i += blockDim.x) col[i] = (Fp::raw(lds[tile_addr<true>(i, 0, 0)]) * ninv).v;
        }
        __syncthreads();
    }
}

// Tile order of the strided passes.  A tile's rows are T consecutive words (64 bytes at T = 16) while HBM is fetched in 128-byte
// lines (tools/microbench.hip k_copy_segments<16>: FETCH_SIZE shows 2x the bytes used), so tiles 2q and 2q+1 share every line
// they read.  Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b mod 8, each with its own L2): in dispatch
// order the two tiles would sit on different XCDs and both fetch the line.  This map gives blocks x and x + 8 — same XCD,
// adjacent in that XCD's dispatch order — the tiles 2q and 2q + 1, so the second one finds the line in its L2.
__device__ __forceinline__ unsigned strided_tile_of_block(unsigned x, unsigned n_tiles) {
    if (n_tiles & 15u) return x;
    return (x & ~15u) | ((x & 7u) << 1) | ((x >> 3) & 1u);
}

// ---- inverse, strided pass: grid = (N_lo / T, columns) -----------------------------------------------
// Tile = all N_hi values of h for T consecutive r; DIT over h; scale by 1/N.
__global__ void k_intt_strided(DMatView m, int k, int k_lo, int logT, DeviceTables tb, uint32_t n_inv_mont) {
    extern __shared__ uint32_t lds[];
    const int k_hi = k - k_lo, n_hi = 1 << k_hi, T = 1 << logT, LD = T + 1;
    const uint64_t n_lo = 1ull << k_lo;
    uint32_t* tw = lds + n_hi * LD;
    uint32_t* col = m.col(blockIdx.y) + (uint64_t)strided_tile_of_block(blockIdx.x, gridDim.x) * T;
    const int total = n_hi << logT;
    {
        int e = threadIdx.x;
        for (; e + 7 * (int)blockDim.x < total; e += 8 * blockDim.x) {  // eight independent loads in flight per thread
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int eu = e + u * (int)blockDim.x; v[u] = col[(uint64_t)(eu >> logT) * n_lo + (eu & (T - 1))]; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const int eu = e + u * (int)blockDim.x; lds[(eu >> logT) * LD + (eu & (T - 1))] = v[u]; }
        }
        for (; e < total; e += blockDim.x) { int c = e & (T - 1), h = e >> logT; lds[h * LD + c] = col[h * n_lo + c]; }
    }
    stage_twiddles(tw, tb.itwc, k_hi);
    __syncthreads();
    strided_tile_transform<true>(lds, tw, k_hi, logT, LD);
    Fp ninv = Fp::raw(n_inv_mont);
    for (int e = threadIdx.x; e < total; e += blockDim.x) { int c = e & (T - 1), h = e >> logT; col[h * n_lo + c] = (Fp::raw(lds[h * LD + c]) * ninv).v; }
}

// ---- forward, strided pass: grid = (N_lo / T, columns) -----------------------------------------------
// Reads coefficients c_i (natural, column-major `src`), multiplies by shift^i, DIF over h, multiplies the
// value at (h', r) by w_N^{r * bitrev(h')}, writes to `dst` (one N-row block of the LDE).
// Powers of the coset shift, prepared on the host per launch (kernel arguments, scalar loads):
//   shift^r = lo[r & 63] * hi[r >> 6] for r < N_lo <= 2^14;   hp[i] = shift^(i * N_lo);   step = shift^(hstep * N_lo)
struct CosetPowers { uint32_t lo[64], hi[256], hp[128], step; };

__global__ void k_ntt_strided(DMatView src, DMatView dst, uint64_t dst_row0, int k, int k_lo, int logT, DeviceTables tb, CosetPowers cp) {
    extern __shared__ uint32_t lds[];
    const int k_hi = k - k_lo, n_hi = 1 << k_hi, T = 1 << logT, LD = T + 1;
    const uint64_t n_lo = 1ull << k_lo, r0 = (uint64_t)strided_tile_of_block(blockIdx.x, gridDim.x) * T;
    uint32_t* tw = lds + n_hi * LD;
    const uint32_t* in = src.col(blockIdx.y) + r0;
    uint32_t* out = dst.col(blockIdx.y) + dst_row0 + r0;
    const int total = n_hi << logT;
    {
        // element e = h*T + c has exponent i = h*N_lo + r0 + c; a thread's c is fixed when blockDim % T == 0
        int c = threadIdx.x & (T - 1), h0 = threadIdx.x >> logT, hstep = blockDim.x >> logT;
        const uint32_t r = (uint32_t)r0 + c;
        Fp cur = Fp::raw(cp.lo[r & 63]) * Fp::raw(cp.hi[r >> 6]) * Fp::raw(cp.hp[h0]), step = Fp::raw(cp.step);
        // eight independent row loads in flight per thread before the (serial) running product consumes them
        int h = h0;
        for (; h + 7 * hstep < n_hi; h += 8 * hstep) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = in[(uint64_t)(h + u * hstep) * n_lo + c];
#pragma unroll
            for (int u = 0; u < 8; u++) { lds[(h + u * hstep) * LD + c] = (Fp::raw(v[u]) * cur).v; cur *= step; }
        }
        for (; h < n_hi; h += hstep) { lds[h * LD + c] = (Fp::raw(in[h * n_lo + c]) * cur).v; cur *= step; }
    }
    stage_twiddles(tw, tb.twc, k_hi);
    __syncthreads();
    strided_tile_transform<false>(lds, tw, k_hi, logT, LD);
    // twiddle: thread per row h', running product over the T consecutive r
    for (int h = threadIdx.x; h < n_hi; h += blockDim.x) {
        const uint32_t e = __brev((uint32_t)h) >> (32 - k_hi);  // k_hi >= 1 in this kernel
        Fp base = root_pow(tb, k, e);                                                   // w_N^{bitrev_{k_hi}(h')}
        Fp cur = root_pow(tb, k, (uint32_t)(((uint64_t)e * r0) & ((1ull << k) - 1)));  // base^r0
        for (int c = 0; c < T; c++) { lds[h * LD + c] = (Fp::raw(lds[h * LD + c]) * cur).v; cur *= base; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += blockDim.x) { int c = e & (T - 1), h = e >> logT; out[h * n_lo + c] = lds[h * LD + c]; }
}

// ---- forward, contiguous pass: persistent blocks over (N / N_lo) x columns tiles ----------------------
// DIF over each contiguous block of N_lo points of `dst` rows [dst_row0, dst_row0 + N).  When the whole
// transform fits one pass (k_hi == 0) the input is read from `src` and multiplied by shift^i first.
__global__ void k_ntt_contig(DMatView src, DMatView dst, uint64_t dst_row0, int k, int k_lo, DeviceTables tb, uint32_t shift_mont,
                             int from_src
                             ) {
    extern __shared__ uint32_t lds[];
    const int n_lo = 1 << k_lo;
    const uint32_t* tw = tb.twc;
    const uint64_t tiles_per_col = src.height >> k_lo, total = tiles_per_col * src.width;  // 2^(k - k_lo)
    auto tile_ptr = [&](uint64_t t) { const uint64_t cidx = t >> (k - k_lo); return dst.col(cidx) + dst_row0 + (t & (tiles_per_col - 1)) * n_lo; };
    Fp cur0 = Fp::one(), step = Fp::one();
    if (from_src) { Fp shift = Fp::raw(shift_mont); cur0 = shift.pow(threadIdx.x); step = shift.pow(blockDim.x); }
    uint32_t pre[CONTIG_MAX_PER_THREAD];
    if (!from_src && blockIdx.x < total) prefetch_tile(pre, tile_ptr(blockIdx.x), n_lo);
    for (uint64_t t = blockIdx.x; t < total; t += gridDim.x) {
        uint32_t* out = tile_ptr(t);
        if (from_src) {  // single-pass transform (n <= 2^12): tiles_per_col == 1, input = coefficients * shift^i
            const uint32_t* in = src.col(t);
            Fp cur = cur0;
            for (int i = threadIdx.x; i < n_lo; i += blockDim.x) { lds[tile_addr<true>(i, 0, 0)] = (Fp::raw(in[i]) * cur).v; cur *= step; }
        } else {
            commit_tile(pre, lds, n_lo);
        }
        __syncthreads();
        if (!from_src && t + gridDim.x < total) prefetch_tile(pre, tile_ptr(t + gridDim.x), n_lo);
        tile_transform<false, true>(lds, tw, k_lo, 0, 0);
        for (int i = threadIdx.x; i < n_lo; i += blockDim.x) out[i] = lds[tile_addr<true>(i, 0, 0)];
        __syncthreads();
    }
}

// =======================================================================================================
// Fused LDE for NATURAL-order input: no row bit-reversal pass, the inverse transform's second pass and the forward transforms' first
// pass of every coset fused in LDS (the coefficients never travel to HBM), the committed (bit-reversed) row order produced by the last
// pass's store.  Per column of N = n_hi n_lo evaluations A[i], i = i_hi n_lo + i_lo (index sets: tools/ntt_fused_model.py):
//   k_lde_a    tile = all i_hi x T consecutive i_lo: DIF over i_hi with inverse roots -> row p holds c_a = bitrev(p); times
//              w_N^-(i_lo c_a); written to scratch S1 at (p, i_lo).                                          read N, write N
//   k_lde_mid  block p of S1 (n_lo contiguous words): DIF over i_lo (inverse roots) -> position i holds c_b = bitrev(i): the
//              coefficient c = c_a + n_hi c_b (times N), which stays in LDS; for every coset t: times fac_t[i] = sigma_t^(n_hi c_b) / N,
//              DIT over c_b (forward roots) -> position q holds f_b = q, times sigma_t^c_a w_N^(c_a q), written to scratch S2
//              (coset t, block p).                                                                          read N, write b N
//   k_lde_c    tile = all blocks (row h = block bitrev(h), i.e. c_a = h) x T consecutive q of S2: DIF over c_a (forward roots) ->
//              row p'' holds f_a = bitrev(p''); element (p'', q) is the evaluation f = q + n_lo f_a and goes to its committed position
//              bitrev_k(f) = bitrev(q) n_hi + p'' of LDE block bitrev(t): runs of n_hi consecutive words.     read b N, write b N
// Traffic (3 + 3 b) N words per column against (1 + 2 + 2 b) 2 N of the unfused passes (b = 2: 9 N instead of 14 N; b = 4: 15 N / 22 N),
// and 3 launches per matrix instead of 3 + 2 b.  A column of at most 2^12 rows is one tile: k_lde_mid alone (one launch instead of 2 + b).
// Correctness of the index arithmetic is checked on the CPU by running these kernels under tools/hipemu (tests/test_ntt_emu_cpu.py).

// ---- the two hot strided tile shapes (2^8 rows x 64 columns: heights 2^20; 2^10 rows x 16 columns: 2^22 .. 2^24), 512 threads x 32 elements -------
// The generic load / store loops of passes A and C spend ~30 VALU instructions per ELEMENT on addresses (row = e >> logT, its bit reversal, a 64-bit
// multiply-add per global address, the LDS address) against 48 for the element's four butterflies (PMC: 2517 instructions per thread of k_lde_c, 1536
// of them butterflies).  For these shapes a thread's 32 elements are (row h0 + HSTEP u, column c) resp. (row p2, column c0 + CSTEP i) with h0, c, p2, c0
// fixed per thread: every global address is a WAVE-UNIFORM base (scalar ALU: the part that depends on u / i, bit reversals of compile-time numbers)
// plus ONE per-thread byte offset (load_at / store_at: raw buffer accesses), every LDS address the thread's base plus an immediate.
// (round 5; VGPU_STRIDED_IO=0 keeps the generic loops: A/B builds)
#ifndef VGPU_STRIDED_IO
#define VGPU_STRIDED_IO 1
#endif
// keeps a wave-uniform value as the running SGPR it is written as (the unrolled loop would otherwise become 32 hoisted constants x n_lo)
#if defined(__HIP_DEVICE_COMPILE__)
#define VGPU_OPAQUE_S(x) asm volatile("" : "+s"(x))
#else
#define VGPU_OPAQUE_S(x) ((void)0)
#endif
__host__ __device__ constexpr uint32_t brev_c(uint32_t x, int bits) { uint32_t r = 0; for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i); return r; }
template <int K_HI, int LOGT> struct StridedShape {
    static constexpr int NT = 512, T = 1 << LOGT, LD = T + 1, NH = 1 << K_HI, LH = 9 - LOGT /* log2 rows per sweep of the 512 threads */, HSTEP = 1 << LH, U = NH >> LH;
    static_assert(U == 32 && (K_HI - LH) == 5, "512 threads x 32 elements");
};
// global (row-strided source) -> LDS tile.  SRC_BITREV: tile row h lives in source block bitrev_{K_HI}(h) (pass C reads S2); else in block h (pass A).
// col0 = the tile's first word of the column (wave-uniform); blocks are n_lo words apart.
template <int K_HI, int LOGT, bool SRC_BITREV>
__device__ __forceinline__ void strided_tile_load(const uint32_t* col0, uint32_t n_lo, uint32_t* lds) {
    using S = StridedShape<K_HI, LOGT>;
    const uint32_t tid = threadIdx.x, c = tid & (S::T - 1), h0 = tid >> LOGT;  // h = h0 + HSTEP u
    const uint32_t blk0 = SRC_BITREV ? (__brev(h0) >> (32 - S::LH)) << 5 : h0;  // the thread's part of the source block index
    const uint32_t voff = (blk0 * n_lo + c) * 4u;
    uint32_t* l0 = lds + h0 * S::LD + c;
    const uint32_t step = (uint32_t)S::HSTEP * n_lo * 4u;
    uint32_t run = 0;  // natural source: the scalar offset as a running sum (32 independent multiples of n_lo would all be hoisted into SGPRs and overflow them)
#pragma unroll
    for (int u0 = 0; u0 < S::U; u0 += 8) {  // eight loads in flight
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t u = (uint32_t)(u0 + k);
            if (SRC_BITREV) v[k] = load_at(col0, voff, brev_c(u, 5) * n_lo * 4u);  // byte offsets below 2^31: at most 2^10 blocks of 2^14 words
            else { v[k] = load_at(col0, voff, run); run += step; VGPU_OPAQUE_S(run); }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) l0[(u0 + k) * S::HSTEP * S::LD] = v[k];
    }
}
// LDS tile -> global, same element map as the load (pass A: tile row h -> block h of S1)
template <int K_HI, int LOGT>
__device__ __forceinline__ void strided_tile_store(uint32_t* col0, uint32_t n_lo, const uint32_t* lds) {
    using S = StridedShape<K_HI, LOGT>;
    const uint32_t tid = threadIdx.x, c = tid & (S::T - 1), h0 = tid >> LOGT;
    const uint32_t voff = (h0 * n_lo + c) * 4u;
    const uint32_t* l0 = lds + h0 * S::LD + c;
    const uint32_t step = (uint32_t)S::HSTEP * n_lo * 4u;
    uint32_t run = 0;
#pragma unroll
    for (int u = 0; u < S::U; u++) { store_at(col0, voff, run, l0[u * S::HSTEP * S::LD]); run += step; VGPU_OPAQUE_S(run); }
}
// pass C's transposing store: tile element (p2, c) -> out[bitrev_{k_lo}(q0 + c) n_hi + p2], q0 = tile T.  Thread tid takes e = tid + 512 i:
//   K_HI = 8:  p2 = tid & 255, c = (tid >> 8) + 2 i;   K_HI = 10: p2 = tid + 512 (i & 1), c = i >> 1 (wave-uniform)
// bitrev_{k_lo}(q0 + c) = bitrev_{k_lo - LOGT}(tile) + (bitrev_LOGT(c) << (k_lo - LOGT)): the tile's part and the i-dependent part are scalar.
template <int K_HI, int LOGT>
__device__ __forceinline__ void strided_tile_store_transposed(uint32_t* out_col, uint32_t tile, int k_lo, const uint32_t* lds) {
    using S = StridedShape<K_HI, LOGT>;
    const uint32_t tid = threadIdx.x, kq = (uint32_t)(k_lo - LOGT);
    const uint32_t qb_tile = kq ? __brev(tile) >> (32 - kq) : 0u;
    if (K_HI == 8) {
        const uint32_t p2 = tid & 255u, c0 = tid >> 8;  // c = c0 + 2 i: bitrev6(c) = (c0 << 5) | bitrev5(i)
        uint32_t* base = out_col + ((uint64_t)qb_tile << K_HI);
        const uint32_t voff = ((((c0 << 5) << kq) << K_HI) + p2) * 4u;
        const uint32_t* l0 = lds + p2 * S::LD + c0;
#pragma unroll
        for (int i = 0; i < 32; i++) store_at(base, voff, ((brev_c((uint32_t)i, 5) << kq) << K_HI) * 4u, l0[2 * i]);  // an LDE column is at most 2^27 words: offsets below 2^29 bytes
    } else {
        uint32_t* base = out_col + ((uint64_t)qb_tile << K_HI);
        const uint32_t* l0 = lds + tid * S::LD;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const uint32_t c = (uint32_t)(i >> 1), half = (uint32_t)(i & 1);
            store_at(base, (tid + 512u * half) * 4u, ((brev_c(c, LOGT) << kq) << K_HI) * 4u, l0[512 * half * S::LD + c]);
        }
    }
}

// ---- pass A: inverse, strided, natural input: grid = (n_lo / T, columns) -------------------------------
// FK_HI / FLOGT: one of the two hot tile shapes fixed at compile time (8 / 6, 10 / 4: 512 threads), or 0 / 0 = any shape (run-time k_hi, logT)
template <int FK_HI, int FLOGT>
__global__ void __launch_bounds__(FK_HI ? 512 : 1024) k_lde_a(DMatView src, DMatView dst, int k, int k_lo, int logT_arg, DeviceTables tb) {
    extern __shared__ uint32_t lds[];
    const int k_hi = FK_HI ? FK_HI : k - k_lo, logT = FK_HI ? FLOGT : logT_arg, n_hi = 1 << k_hi, T = 1 << logT, LD = T + 1;
    const uint64_t n_lo = 1ull << k_lo, r0 = (uint64_t)strided_tile_of_block(blockIdx.x, gridDim.x) * T;
    uint32_t* tw = lds + n_hi * LD;
    const uint32_t* in = src.col(blockIdx.y) + r0;
    uint32_t* out = dst.col(blockIdx.y) + r0;
    const int total = n_hi << logT;
    if constexpr (FK_HI != 0) strided_tile_load<FK_HI, FLOGT, false>(in, (uint32_t)n_lo, lds);
    else {
        int e = threadIdx.x;
        for (; e + 7 * (int)blockDim.x < total; e += 8 * blockDim.x) {  // eight independent loads in flight per thread
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int eu = e + u * (int)blockDim.x; v[u] = in[(uint64_t)(eu >> logT) * n_lo + (eu & (T - 1))]; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const int eu = e + u * (int)blockDim.x; lds[(eu >> logT) * LD + (eu & (T - 1))] = v[u]; }
        }
        for (; e < total; e += blockDim.x) { int c = e & (T - 1), h = e >> logT; lds[h * LD + c] = in[(uint64_t)h * n_lo + c]; }
    }
    stage_twiddles(tw, tb.itwc, k_hi);
    __syncthreads();
    if constexpr (FK_HI != 0) strided_transform<false, FK_HI, FLOGT>(lds, tw);  // DIF with inverse roots: row p holds c_a = bitrev(p)
    else tile_transform<false, false>(lds, tw, k_hi, logT, LD);
    for (int p = threadIdx.x; p < n_hi; p += blockDim.x) {  // times w_N^-(c_a i_lo): thread per row, running product over the T consecutive i_lo
        const uint32_t ca = __brev((uint32_t)p) >> (32 - k_hi);  // k_hi >= 1 in this kernel
        const Fp base = inv_root_pow(tb, k, ca);
        Fp cur = inv_root_pow(tb, k, (uint32_t)(((uint64_t)ca * r0) & ((1ull << k) - 1)));
        for (int c = 0; c < T; c++) { lds[p * LD + c] = (Fp::raw(lds[p * LD + c]) * cur).v; cur *= base; }
    }
    __syncthreads();
    if constexpr (FK_HI != 0) strided_tile_store<FK_HI, FLOGT>(out, (uint32_t)n_lo, lds);
    else
        for (int e = threadIdx.x; e < total; e += blockDim.x) { int c = e & (T - 1), h = e >> logT; out[(uint64_t)h * n_lo + c] = lds[h * LD + c]; }
}

// ---- pass B: contiguous, persistent blocks over (n_hi x columns) tiles: inverse second half + forward first half of every coset ----
// src: S1 (or the natural-order input itself when the column is one tile).  dst: S2 (coset t at rows [t N, (t + 1) N)) — or, one-tile
// columns, the LDE itself (block bitrev(t), rows in committed order).  LDS: two padded tiles (coefficients, work).
// MAXT: the launch's thread count bound.  One-tile columns (heights <= 2^12: every small chip of a proof, 35 launches) run 64..256 threads; bounded at
// 256 the register allocator keeps everything in registers (the generic bound of 1024 threads capped it at 128 VGPRs: 12 spilled + 105 SGPRs parked in lanes).
// (What remains: ~110 SGPRs parked in VGPR lanes — the loop-invariant twiddle-table addresses of the run-time round dispatch, hoisted out of the tile
// loop; fixing the tile size at compile time, 13 instantiations, halves the code and leaves 10-90 of them: not adopted, these launches sit on the
// auxiliary stream beside the big matrices' passes.)
template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_lde_mid(DMatView src, DMatView dst, int k, int k_lo, int lb, DeviceTables tb, LdeTables lt) {
    extern __shared__ uint32_t lds[];
    const int n_lo = 1 << k_lo, k_hi = k - k_lo, n_hi = 1 << k_hi, b = 1 << lb;
    uint32_t* A = lds;
    uint32_t* W = lds + padded_words(n_lo);
    const uint64_t N = 1ull << k, tiles_per_col = (uint64_t)n_hi, total = tiles_per_col * src.width;
    const uint32_t nmask = (uint32_t)(N - 1);
    auto tile_ptr = [&](uint64_t t) { const uint64_t cidx = t >> k_hi; return src.col(cidx) + (t & (tiles_per_col - 1)) * n_lo; };
    uint32_t pre[CONTIG_MAX_PER_THREAD];
    if (blockIdx.x < total) prefetch_tile(pre, tile_ptr(blockIdx.x), n_lo);
    for (uint64_t t = blockIdx.x; t < total; t += gridDim.x) {
        const uint64_t cidx = t >> k_hi, p = t & (tiles_per_col - 1);
        commit_tile(pre, A, n_lo);
        __syncthreads();
        if (t + gridDim.x < total) prefetch_tile(pre, tile_ptr(t + gridDim.x), n_lo);
        tile_transform<false, true>(A, tb.itwc, k_lo, 0, 0);  // position i holds (N times) the coefficient with c_b = bitrev(i)
        const uint32_t ca = k_hi ? __brev((uint32_t)p) >> (32 - k_hi) : 0u;
        for (int tc = 0; tc < b; tc++) {
            const uint32_t* fac = lt.fac + (size_t)tc * n_lo;
            for (int i = threadIdx.x; i < n_lo; i += blockDim.x) W[tile_addr<true>(i, 0, 0)] = (Fp::raw(A[tile_addr<true>(i, 0, 0)]) * Fp::raw(fac[i])).v;
            __syncthreads();
            tile_transform<true, true>(W, tb.twc, k_lo, 0, 0);  // DIT with forward roots: position q holds f_b = q
            if (k_hi > 0) {
                uint32_t* out = dst.col(cidx) + (uint64_t)tc * N + p * n_lo;
                const Fp step = root_pow(tb, k, (ca * blockDim.x) & nmask);
                Fp cur = Fp::raw(lt.sig[(size_t)tc * n_hi + ca]) * root_pow(tb, k, (ca * threadIdx.x) & nmask);
                for (int q = threadIdx.x; q < n_lo; q += blockDim.x) { out[q] = (Fp::raw(W[tile_addr<true>(q, 0, 0)]) * cur).v; cur *= step; }
            } else {
                // the whole column is this tile: committed position of the evaluation q is bitrev_k(q) — permuted on the LDS side, the
                // global store stays coalesced
                const uint32_t blk = lb ? __brev((uint32_t)tc) >> (32 - lb) : 0u;
                uint32_t* out = dst.col(cidx) + (uint64_t)blk * N;
                for (int j = threadIdx.x; j < n_lo; j += blockDim.x) {
                    const int q = k ? (int)(__brev((uint32_t)j) >> (32 - k)) : 0;
                    out[j] = W[tile_addr<true>(q, 0, 0)];
                }
            }
            __syncthreads();
        }
    }
}

// ---- pass B for 2^12-point tiles (every height from 2^13 to 2^22): the same computation with the first round of the inverse transform
// fed straight from the global loads, its last round and the first round of every forward transform joined in registers (both work on the
// thread's own 16 consecutive positions: the coefficients never leave the register file), the last forward round stored straight to
// HBM: ONE LDS tile instead of two, 2 + 3 b barriers per tile instead of 4 + 5 b, half the LDS traffic.  256 threads x 16 points.
__global__ void __launch_bounds__(256) k_lde_mid12(DMatView src, DMatView dst, int k, int lb, DeviceTables tb, LdeTables lt) {
    extern __shared__ uint32_t lds[];
    constexpr int K_LO = 12, N_LO = 1 << K_LO;
    const int k_hi = k - K_LO, n_hi = 1 << k_hi, b = 1 << lb, tid = threadIdx.x;
    const uint64_t N = 1ull << k, tiles_per_col = (uint64_t)n_hi, total = tiles_per_col * src.width;
    const uint32_t nmask = (uint32_t)(N - 1);
    auto tile_ptr = [&](uint64_t t) { const uint64_t cidx = t >> k_hi; return src.col(cidx) + (t & (tiles_per_col - 1)) * N_LO; };
    Fp tw_in[15], tw_out[15];
    load_round_twiddles<4>(tw_in, tb.itwc, tid, 8);
    load_round_twiddles<4>(tw_out, tb.twc, tid, 8);
    const Fp* const held_in = tw_in;
    const Fp* const held_out = tw_out;
    uint32_t pre[16];
    if (blockIdx.x < total) {
        const uint32_t* sp = tile_ptr(blockIdx.x);
#pragma unroll
        for (int g = 0; g < 16; g++) pre[g] = sp[tid + 256 * g];
    }
    for (uint64_t t = blockIdx.x; t < total; t += gridDim.x) {
        const uint64_t cidx = t >> k_hi, p = t & (tiles_per_col - 1);
        Fp x[16];
#pragma unroll
        for (int g = 0; g < 16; g++) x[g] = Fp::raw(pre[g]);
        if (t + gridDim.x < total) {  // the next tile's loads land while this one computes
            const uint32_t* sp = tile_ptr(t + gridDim.x);
#pragma unroll
            for (int g = 0; g < 16; g++) pre[g] = sp[tid + 256 * g];
        }
        // inverse, stages 12 .. 9 on the points tid + 256 g (DIF, inverse roots)
        butterflies<4, false, false>(x, tb.itwc, tid, 8, held_in);
        {
            uint32_t* p0 = lds + tile_addr<true>(tid, 0, 0);
#pragma unroll
            for (int g = 0; g < 16; g++) p0[g * 272] = x[g].v;  // tile_addr(tid + 256 g) = tile_addr(tid) + 272 g
        }
        __syncthreads();
        ntt_round<4, false, true, 4>(lds, tb.itwc, 12, 5, 0, 0);  // stages 8 .. 5 (ends with a barrier)
        {
            const uint32_t* p0 = lds + 17 * tid;  // positions 16 tid + g
#pragma unroll
            for (int g = 0; g < 16; g++) x[g] = Fp::raw(p0[g]);
        }
        butterflies<4, false, true>(x, tb.itwc, 0, 0);  // stages 4 .. 1: x[g] = N times the coefficient with c_b = bitrev(16 tid + g)
        const uint32_t ca = __brev((uint32_t)p) >> (32 - k_hi);
        const Fp step = root_pow(tb, k, (ca * 256u) & nmask), tw0 = root_pow(tb, k, (ca * (uint32_t)tid) & nmask);
        for (int tc = 0; tc < b; tc++) {
            Fp y[16];
            {
                const uint32_t* fac = lt.fac + (size_t)tc * N_LO + 16 * tid;
#pragma unroll
                for (int g = 0; g < 16; g++) y[g] = x[g] * Fp::raw(fac[g]);
            }
            butterflies<4, true, true>(y, tb.twc, 0, 0);  // forward, stages 1 .. 4 (DIT, forward roots)
            {
                // the thread's own 16 positions: no one else reads or writes them between the inverse transform's last round and here;
                // other threads' reads of them by the PREVIOUS coset's last round are behind that coset's closing barrier
                uint32_t* p0 = lds + 17 * tid;
#pragma unroll
                for (int g = 0; g < 16; g++) p0[g] = y[g].v;
            }
            __syncthreads();
            ntt_round<4, true, true, 4>(lds, tb.twc, 12, 5, 0, 0);  // stages 5 .. 8
            {
                const uint32_t* p0 = lds + tile_addr<true>(tid, 0, 0);
#pragma unroll
                for (int g = 0; g < 16; g++) y[g] = Fp::raw(p0[g * 272]);
            }
            butterflies<4, true, false>(y, tb.twc, tid, 8, held_out);  // stages 9 .. 12: y[g] = the evaluation f_b = tid + 256 g of this block
            uint32_t* out = dst.col(cidx) + (uint64_t)tc * N + p * N_LO + tid;
            Fp cur = Fp::raw(lt.sig[(size_t)tc * n_hi + ca]) * tw0;  // sigma_t^c_a w_N^(c_a q), q = tid + 256 g
#pragma unroll
            for (int g = 0; g < 16; g++) { out[256 * g] = (y[g] * cur).v; cur *= step; }
            __syncthreads();  // the tile is free for the next coset / the next tile
        }
    }
}

// ---- pass B for 2^14-point tiles (heights from 2^23: C3's memory chip): the same fusion with one more, radix-4, round at the outer end of each
// transform (stages 14, 13).  1024 threads x 16 points; a thread's prefetched element tid + 1024 u is point g = u >> 2 of its outer-round
// item j = u & 3 (points tid + 1024 j + 4096 g).  One LDS tile (70 KiB) instead of the generic kernel's two; 128 registers (one workgroup per CU).
__global__ void __launch_bounds__(1024) k_lde_mid14(DMatView src, DMatView dst, int k, int lb, DeviceTables tb, LdeTables lt) {
    extern __shared__ uint32_t lds[];
    constexpr int K_LO = 14, N_LO = 1 << K_LO, NT = 1024;
    uint32_t* const coef = lds + (N_LO + (N_LO >> 4));  // second padded tile: the coefficients of the current column tile
    const int k_hi = k - K_LO, n_hi = 1 << k_hi, b = 1 << lb, tid = threadIdx.x;
    const uint64_t N = 1ull << k, tiles_per_col = (uint64_t)n_hi, total = tiles_per_col * src.width;
    const uint32_t nmask = (uint32_t)(N - 1);
    for (uint64_t t = blockIdx.x; t < total; t += gridDim.x) {
        const uint64_t cidx = t >> k_hi, p = t & (tiles_per_col - 1);
        const uint32_t* sp = src.col(cidx) + p * N_LO;
        // inverse, stages 14 and 13 (DIF, inverse roots), straight from the global loads
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int w = tid + NT * j;
            Fp x4[4];
#pragma unroll
            for (int g = 0; g < 4; g++) x4[g] = Fp::raw(sp[w + 4096 * g]);
            butterflies<2, false, false>(x4, tb.itwc, w, 12);
#pragma unroll
            for (int g = 0; g < 4; g++) lds[tile_addr<true>(w + 4096 * g, 0, 0)] = x4[g].v;
        }
        __syncthreads();
        ntt_round<4, false, true, 8>(lds, tb.itwc, 14, 9, 0, 0);  // stages 12 .. 9
        ntt_round<4, false, true, 4>(lds, tb.itwc, 14, 5, 0, 0);  // stages 8 .. 5
        {
            // stages 4 .. 1 on the thread's own 16 positions; the result — N times the coefficient with c_b = bitrev(16 tid + g) — is parked in a second
            // LDS tile (`coef`, the thread's own slots: no barrier) instead of 16 registers held across the cosets: at 1024 threads the kernel may use
            // 128 VGPRs and spilled 28 of them to scratch inside the coset loop (profiles/r03_isa_static.txt); the workgroup has the CU's LDS to itself anyway
            Fp x[16];
            const uint32_t* p0 = lds + 17 * tid;  // positions 16 tid + g
#pragma unroll
            for (int g = 0; g < 16; g++) x[g] = Fp::raw(p0[g]);
            butterflies<4, false, true>(x, tb.itwc, 0, 0);
            uint32_t* c0 = coef + 17 * tid;
#pragma unroll
            for (int g = 0; g < 16; g++) c0[g] = x[g].v;
        }
        const uint32_t ca = k_hi ? __brev((uint32_t)p) >> (32 - k_hi) : 0u;
        const Fp step = root_pow(tb, k, (ca * 4096u) & nmask);
        for (int tc = 0; tc < b; tc++) {
            Fp y[16];
            {
                const uint32_t* fac = lt.fac + (size_t)tc * N_LO + 16 * tid;
                const uint32_t* c0 = coef + 17 * tid;
#pragma unroll
                for (int g = 0; g < 16; g++) y[g] = Fp::raw(c0[g]) * Fp::raw(fac[g]);
            }
            butterflies<4, true, true>(y, tb.twc, 0, 0);  // forward, stages 1 .. 4
            {
                uint32_t* p0 = lds + 17 * tid;  // the thread's own positions (see k_lde_mid12)
#pragma unroll
                for (int g = 0; g < 16; g++) p0[g] = y[g].v;
            }
            __syncthreads();
            ntt_round<4, true, true, 4>(lds, tb.twc, 14, 5, 0, 0);  // stages 5 .. 8
            ntt_round<4, true, true, 8>(lds, tb.twc, 14, 9, 0, 0);  // stages 9 .. 12
            uint32_t* out = dst.col(cidx) + (uint64_t)tc * N + p * N_LO;
            const Fp sg = Fp::raw(lt.sig[(size_t)tc * n_hi + ca]);
#pragma unroll 1  // (unrolled, the four items' loads and twiddles are hoisted together: 14 VGPRs beyond the 128 a 1024-thread workgroup may use were spilled to scratch)
            for (int j = 0; j < 4; j++) {  // stages 13 and 14, then sigma_t^c_a w_N^(c_a q) and out: q = w + 4096 g
                const int w = tid + NT * j;
                Fp y4[4];
#pragma unroll
                for (int g = 0; g < 4; g++) y4[g] = Fp::raw(lds[tile_addr<true>(w + 4096 * g, 0, 0)]);
                butterflies<2, true, false>(y4, tb.twc, w, 12);
                Fp cur = sg * root_pow(tb, k, (ca * (uint32_t)w) & nmask);
#pragma unroll
                for (int g = 0; g < 4; g++) { out[w + 4096 * g] = (y4[g] * cur).v; cur *= step; }
            }
            __syncthreads();  // the tile is free for the next coset / the next tile
        }
    }
}

// ---- pass C: forward, strided, transposing store: grid = (n_lo / T, columns, cosets) --------------------
template <int FK_HI, int FLOGT>  // as k_lde_a
__global__ void __launch_bounds__(FK_HI ? 512 : 1024) k_lde_c(DMatView src, DMatView dst, int k, int k_lo, int lb, int logT_arg, DeviceTables tb
                        ) {
    extern __shared__ uint32_t lds[];
    const int k_hi = FK_HI ? FK_HI : k - k_lo, logT = FK_HI ? FLOGT : logT_arg, n_hi = 1 << k_hi, T = 1 << logT, LD = T + 1;
    const uint64_t n_lo = 1ull << k_lo, N = 1ull << k, q0 = (uint64_t)strided_tile_of_block(blockIdx.x, gridDim.x) * T;
    const uint32_t tc = blockIdx.z, blk = lb ? __brev(tc) >> (32 - lb) : 0u;
    uint32_t* tw = lds + n_hi * LD;
    const uint32_t* in = src.col(blockIdx.y) + (uint64_t)tc * N + q0;
    uint32_t* out = dst.col(blockIdx.y) + (uint64_t)blk * N;
    const int total = n_hi << logT;
    if constexpr (FK_HI != 0) strided_tile_load<FK_HI, FLOGT, true>(in, (uint32_t)n_lo, lds);
    else {
        // tile row h (= c_a) lives in block bitrev(h) of S2
        int e = threadIdx.x;
        for (; e + 7 * (int)blockDim.x < total; e += 8 * blockDim.x) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int eu = e + u * (int)blockDim.x;
                const uint32_t h = (uint32_t)(eu >> logT), pb = __brev(h) >> (32 - k_hi);
                v[u] = in[(uint64_t)pb * n_lo + (eu & (T - 1))];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) { const int eu = e + u * (int)blockDim.x; lds[(eu >> logT) * LD + (eu & (T - 1))] = v[u]; }
        }
        for (; e < total; e += blockDim.x) {
            const int c = e & (T - 1);
            const uint32_t h = (uint32_t)(e >> logT), pb = __brev(h) >> (32 - k_hi);
            lds[h * LD + c] = in[(uint64_t)pb * n_lo + c];
        }
    }
    stage_twiddles(tw, tb.twc, k_hi);
    __syncthreads();
    if constexpr (FK_HI != 0) strided_transform<false, FK_HI, FLOGT>(lds, tw);  // DIF with forward roots: row p'' holds f_a = bitrev(p'')
    else tile_transform<false, false>(lds, tw, k_hi, logT, LD);
    // element (p'', q0 + c) -> out[bitrev_{k_lo}(q0 + c) n_hi + p'']: n_hi consecutive words per c (LDS read with the odd stride LD: conflict-free)
    const uint32_t tile = strided_tile_of_block(blockIdx.x, gridDim.x);
    if constexpr (FK_HI != 0) strided_tile_store_transposed<FK_HI, FLOGT>(out, tile, k_lo, lds);
    else
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int p2 = e & (n_hi - 1), c = e >> k_hi;
        const uint32_t qb = __brev((uint32_t)(q0 + c)) >> (32 - k_lo);
        out[(uint64_t)qb * n_hi + p2] = lds[p2 * LD + c];
    }
}

// ---- host launchers -----------------------------------------------------------------------------------
struct NttPlan { int k, k_lo, k_hi, logT; unsigned threads_contig, threads_strided; size_t lds_contig, lds_strided; };
static NttPlan make_plan(int k) {
    NttPlan p;
    // contiguous tile: 2^12 points; 2^14 for the largest heights, where a 2^12 tile would leave the strided pass 2^(k-12)
    // rows of at most 32 bytes each
    // (measured: below 2^23 the 2^12 tile with its compile-time-stride rounds is the faster split)
    p.k = k; p.k_lo = k < 12 ? k : (k >= 23 ? 14 : 12); p.k_hi = k - p.k_lo;
    int t = 16384 >> p.k_hi; if (t > 64) t = 64; if (t < 8) t = 8;
    if ((1 << p.k_lo) < t) t = 1 << p.k_lo;
    p.logT = 0; while ((1 << p.logT) < t) p.logT++;
    unsigned n_lo = 1u << p.k_lo;
    p.threads_contig = n_lo / 16 < 64 ? 64 : (n_lo / 16 > 1024 ? 1024 : n_lo / 16);
    unsigned tile = (1u << p.k_hi) << p.logT;
    p.threads_strided = tile / 32 < 64 ? 64 : (tile / 32 > 1024 ? 1024 : tile / 32);
    if (p.threads_strided < (1u << p.logT)) p.threads_strided = 1u << p.logT;
    p.lds_contig = (size_t)padded_words((int)n_lo) * 4;
    p.lds_strided = ((size_t)(1u << p.k_hi) * ((1u << p.logT) + 1) + (1u << p.k_hi)) * 4;
    return p;
}

static void set_lds_limit() {
    static bool done = false;
    if (done) return;
    const int lim = 160 * 1024;
    (void)hipFuncSetAttribute((const void*)k_intt_strided, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_ntt_strided, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_intt_contig, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_ntt_contig, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_lde_a<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_lde_a<8, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_lde_a<10, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_lde_mid<256>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_lde_mid<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_lde_c<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_lde_c<8, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_lde_c<10, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_lde_mid12, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    (void)hipFuncSetAttribute((const void*)k_lde_mid14, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    done = true;
}

// In-place inverse NTT of every column of `m` (height 2^k).  Input rows must be at bit-reversed positions.
void launch_intt(hipStream_t st, DMatView m, const DeviceTables& tb) {
    set_lds_limit();
    int k = (int)vg::log2_strict_u64(m.height);
    NttPlan p = make_plan(k);
    uint32_t ninv = Fp::from_canonical((uint32_t)(m.height % vg::P)).inv().v;
    const uint64_t tiles_c = (m.height >> p.k_lo) * m.width;
    dim3 gc((unsigned)(tiles_c < 4096 ? tiles_c : 4096));
    const double pass_bytes = 8.0 * m.height * m.width;
    {
        ProfScope ps("k_intt_contig", st, pass_bytes);
        VK_LAUNCH(k_intt_contig, gc, dim3(p.threads_contig), p.lds_contig, st, m, k, p.k_lo, tb, ninv);
    }
    if (p.k_hi > 0) {
        dim3 gs((unsigned)((1u << p.k_lo) >> p.logT), (unsigned)m.width);
        ProfScope ps("k_intt_strided", st, pass_bytes);
        VK_LAUNCH(k_intt_strided, gs, dim3(p.threads_strided), p.lds_strided, st, m, k, p.k_lo, p.logT, tb, ninv);
    }
}

// dst rows [dst_row0, dst_row0 + N) <- bit-reversed-order evaluations of the polynomials with natural
// coefficients `coeffs` on the coset shift * H_N.
void launch_coset_ntt(hipStream_t st, DMatView coeffs, DMatView dst, uint64_t dst_row0, Fp shift, const DeviceTables& tb) {
    set_lds_limit();
    int k = (int)vg::log2_strict_u64(coeffs.height);
    NttPlan p = make_plan(k);
    const uint64_t tiles_c = (coeffs.height >> p.k_lo) * coeffs.width;
    dim3 gc((unsigned)(tiles_c < 4096 ? tiles_c : 4096));
    const double pass_bytes = 8.0 * coeffs.height * coeffs.width;
    if (p.k_hi > 0) {
        dim3 gs((unsigned)((1u << p.k_lo) >> p.logT), (unsigned)coeffs.width);
        {
            CosetPowers cp;
            const unsigned hstep = p.threads_strided >> p.logT;  // <= 128: threads <= 1024, T >= 8
            Fp a = Fp::one(), s64 = shift.pow(64), sn = shift.pow(1ull << p.k_lo);
            for (int i = 0; i < 64; i++) { cp.lo[i] = a.v; a *= shift; }
            a = Fp::one();
            for (int i = 0; i < 256; i++) { cp.hi[i] = a.v; a *= s64; }
            a = Fp::one();
            cp.step = 0;
            for (unsigned i = 0; i < 128; i++) { cp.hp[i] = a.v; a *= sn; if (i + 1 == hstep) cp.step = a.v; }
            ProfScope ps("k_ntt_strided", st, pass_bytes);
            VK_LAUNCH(k_ntt_strided, gs, dim3(p.threads_strided), p.lds_strided, st, coeffs, dst, dst_row0, k, p.k_lo, p.logT, tb, cp);
        }
        ProfScope ps("k_ntt_contig", st, pass_bytes);
        VK_LAUNCH(k_ntt_contig, gc, dim3(p.threads_contig), p.lds_contig, st, coeffs, dst, dst_row0, k, p.k_lo, tb, shift.v, 0);
    } else {
        ProfScope ps("k_ntt_contig", st, pass_bytes);
        VK_LAUNCH(k_ntt_contig, gc, dim3(p.threads_contig), p.lds_contig, st, coeffs, dst, dst_row0, k, p.k_lo, tb, shift.v, 1);
    }
}

// The bit-reversed LDE of the natural-order evaluations `nat` (column-major, height N = 2^k) on shift * H_{N << log_blowup}, into `lde`
// (height N << log_blowup).  s1 (N x width) and s2 ((N << log_blowup) x width) are scratch, needed only when N > 2^12 (null views otherwise).
void launch_lde_natural(hipStream_t st, DMatView nat, DMatView lde, int log_blowup, const DeviceTables& tb, const LdeTables& lt, DMatView s1, DMatView s2) {
    set_lds_limit();
    const int k = (int)vg::log2_strict_u64(nat.height);
    NttPlan p = make_plan(k);
    const uint64_t tiles = ((uint64_t)1 << p.k_hi) * nat.width;
    dim3 gm((unsigned)(tiles < 4096 ? tiles : 4096));
    const size_t lds_mid = 2 * (size_t)padded_words(1 << p.k_lo) * 4;
    const double nw = 4.0 * nat.height * nat.width, b = (double)(1u << log_blowup);
    if (p.k_hi == 0) {
        ProfScope ps("k_lde_mid", st, nw * (1.0 + b));
        VK_LAUNCH(k_lde_mid<256>, gm, dim3(p.threads_contig), lds_mid, st, nat, lde, k, p.k_lo, log_blowup, tb, lt);  // k_lo <= 12: at most 256 threads
        return;
    }
    dim3 gs((unsigned)((1u << p.k_lo) >> p.logT), (unsigned)nat.width);
    {
        ProfScope ps("k_lde_a", st, 2.0 * nw);
        const int shape = VGPU_STRIDED_IO && VGPU_STRIDED_FIXED && p.threads_strided == 512 ? (p.k_hi == 8 && p.logT == 6 ? 8 : (p.k_hi == 10 && p.logT == 4 ? 10 : 0)) : 0;
        if (shape == 8) VK_LAUNCH((k_lde_a<8, 6>), gs, dim3(512), p.lds_strided, st, nat, s1, k, p.k_lo, p.logT, tb);
        else if (shape == 10) VK_LAUNCH((k_lde_a<10, 4>), gs, dim3(512), p.lds_strided, st, nat, s1, k, p.k_lo, p.logT, tb);
        else VK_LAUNCH((k_lde_a<0, 0>), gs, dim3(p.threads_strided), p.lds_strided, st, nat, s1, k, p.k_lo, p.logT, tb);
    }
    if (p.k_lo == 12) {
        if (gm.x > 4096u) gm.x = 4096u;  // persistent blocks
        ProfScope ps("k_lde_mid12", st, nw * (1.0 + b));
        VK_LAUNCH(k_lde_mid12, gm, dim3(256), (size_t)padded_words(4096) * 4, st, s1, s2, k, log_blowup, tb, lt);
    } else if (p.k_lo == 14) {
        ProfScope ps("k_lde_mid14", st, nw * (1.0 + b));
        VK_LAUNCH(k_lde_mid14, gm, dim3(1024), 2 * (size_t)padded_words(16384) * 4, st, s1, s2, k, log_blowup, tb, lt);  // work tile + coefficient tile: 139 KiB
    } else {
        ProfScope ps("k_lde_mid", st, nw * (1.0 + b));
        if (p.threads_contig <= 256) VK_LAUNCH(k_lde_mid<256>, gm, dim3(p.threads_contig), lds_mid, st, s1, s2, k, p.k_lo, log_blowup, tb, lt);
        else VK_LAUNCH(k_lde_mid<1024>, gm, dim3(p.threads_contig), lds_mid, st, s1, s2, k, p.k_lo, log_blowup, tb, lt);
    }
    dim3 gc((unsigned)((1u << p.k_lo) >> p.logT), (unsigned)nat.width, 1u << log_blowup);
    ProfScope ps("k_lde_c", st, 2.0 * nw * b);
    const int shape_c = VGPU_STRIDED_IO && VGPU_STRIDED_FIXED && p.threads_strided == 512 ? (p.k_hi == 8 && p.logT == 6 ? 8 : (p.k_hi == 10 && p.logT == 4 ? 10 : 0)) : 0;
#define VG_LDE_C(A, B, THREADS) VK_LAUNCH((k_lde_c<A, B>), gc, dim3(THREADS), p.lds_strided, st, s2, lde, k, p.k_lo, log_blowup, p.logT, tb)
    if (shape_c == 8) VG_LDE_C(8, 6, 512);
    else if (shape_c == 10) VG_LDE_C(10, 4, 512);
    else VG_LDE_C(0, 0, p.threads_strided);
#undef VG_LDE_C
}

}  // namespace vk
