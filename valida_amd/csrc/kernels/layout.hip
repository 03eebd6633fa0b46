// Layout kernels: ingest of host RowMajorMatrix data into the column-major Montgomery device layout
// (SURVEY.md K1), row bit-reversal of column-major matrices, and export back to canonical row-major
// (tests / openings).  Pure data movement: HBM-bound, tiles staged through LDS so both the row-major
// and the column-major side move full 128/256-byte segments.
#include "launch.hpp"
#include <stdexcept>

namespace vk {

thread_local Profiler* g_profiler = nullptr;
thread_local ProfScope* g_scope = nullptr;

// src: row-major canonical u32 [height x width] (as handed over by the reference's RowMajorMatrix<Val>,
// basic/src/lib.rs:223).  dst: column-major Montgomery; row r lands at position bitrev(r) if `bitrev`
// (the in-place DIT inverse NTT wants bit-reversed-position input) else at r.
// One block handles 64 destination rows x all columns.
__global__ void k_ingest(const uint32_t* __restrict__ src, DMatView dst, int log_h, int bitrev) {
    extern __shared__ uint32_t lds[];  // [64][width | 1]
    const int W = (int)dst.width, LD = W | 1;  // odd row stride: conflict-free column reads
    const uint64_t j0 = (uint64_t)blockIdx.x * 64;
    const int rows = (int)((dst.height - j0) < 64 ? (dst.height - j0) : 64);
    // each wave reads whole source rows (W contiguous words)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if (!bitrev) {
        // the 64 source rows are one contiguous run of 64 * W words: flat, fully coalesced reads whatever the width
        // (a 14-column trace would otherwise keep 14 of 64 lanes busy)
        const uint32_t* run = src + j0 * W;
        const int total = rows * W;
        int jr = (int)threadIdx.x / W, c = (int)threadIdx.x - jr * W;  // one division per thread, then incremental
        const int dj = (int)blockDim.x / W, dc = (int)blockDim.x - dj * W;
        int e = threadIdx.x;
        for (; e + 3 * (int)blockDim.x < total; e += 4 * blockDim.x) {  // four independent loads in flight per thread
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = run[e + u * (int)blockDim.x];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                lds[jr * LD + c] = Fp::from_canonical(v[u]).v;
                jr += dj; c += dc;
                if (c >= W) { c -= W; jr++; }
            }
        }
        for (; e < total; e += blockDim.x) {
            lds[jr * LD + c] = Fp::from_canonical(run[e]).v;
            jr += dj; c += dc;
            if (c >= W) { c -= W; jr++; }
        }
    } else {
        for (int jr = wave; jr < rows; jr += nwaves) {
            uint64_t r = (uint64_t)vg::reverse_bits_len((uint32_t)(j0 + jr), (unsigned)log_h);
            const uint32_t* row = src + r * W;
            for (int c = lane; c < W; c += 64) lds[jr * LD + c] = Fp::from_canonical(row[c]).v;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < W * 64; e += blockDim.x) {
        int jr = e & 63, c = e >> 6;
        if (jr < rows) dst.data[(uint64_t)c * dst.stride + j0 + jr] = lds[jr * LD + c];
    }
}

// dst[bitrev(r)] = src[r] per column (column-major, same shape).  Tiled so reads and writes are both
// 64-element segments: a block handles, for one column and one value `mid` of the middle bits, all
// (hi, lo) with hi, lo in [0, 64): r = hi << (k-6) | mid << 6 | lo.
__global__ void k_bitrev_rows(DMatView src, DMatView dst, int k) {
    __shared__ uint32_t tile[64][65];
    const uint32_t* s = src.col(blockIdx.y);
    uint32_t* d = dst.col(blockIdx.y);
    if (k < 12) {  // small: direct
        uint64_t n = 1ull << k;
        for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x)
            d[vg::reverse_bits_len((uint32_t)r, (unsigned)k)] = s[r];
        return;
    }
    const int midbits = k - 12;
    const uint32_t mid = blockIdx.x;
    for (int e = threadIdx.x; e < 4096; e += blockDim.x) {
        uint32_t lo = e & 63, hi = e >> 6;
        tile[hi][lo] = s[((uint64_t)hi << (k - 6)) | ((uint64_t)mid << 6) | lo];
    }
    __syncthreads();
    const uint32_t rmid = midbits ? vg::reverse_bits_len(mid, (unsigned)midbits) : 0;
    for (int e = threadIdx.x; e < 4096; e += blockDim.x) {
        uint32_t a = e & 63, b = e >> 6;  // destination: rev(lo)=b is the high part, rev(hi)=a the low part
        uint32_t lo = vg::reverse_bits_len(b, 6), hi = vg::reverse_bits_len(a, 6);
        d[((uint64_t)b << (k - 6)) | ((uint64_t)rmid << 6) | a] = tile[hi][lo];
    }
}

// Export rows [row0, row0 + nrows) of a column-major Montgomery matrix to canonical row-major.
__global__ void k_export_rows(DMatView src, uint64_t row0, uint64_t nrows, uint32_t* __restrict__ dst) {
    uint64_t total = nrows * src.width;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = e % nrows, c = e / nrows;
        dst[r * src.width + c] = src.get(row0 + r, c).canonical();
    }
}

// Sharded quotient round: a column arrives as Wq blocks of `rows` chunk rows, block r = the rows of the rank that evaluated the sub-coset
// eq = bitrev_Wq(r), each block in the natural order of its own range; the chunk row of global natural index j = eq + Wq m is row m of that
// block.  dst = the column in global natural order (what the fused LDE reads).
__global__ void __launch_bounds__(256) k_interleave_blocks(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t rows, uint32_t log_wq) {
    const uint64_t n = rows << log_wq, j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t eq = (uint32_t)(j & ((1u << log_wq) - 1u)), r = vg::reverse_bits_len(eq, log_wq);
    const uint64_t col = blockIdx.y;
    dst[col * n + j] = src[col * n + (uint64_t)r * rows + (j >> log_wq)];
}
void launch_interleave_blocks(hipStream_t st, const uint32_t* src, uint32_t* dst, uint64_t rows, uint32_t log_wq, uint64_t n_cols) {
    if (!log_wq) throw std::logic_error("interleave: one block is already in natural order");
    const uint64_t n = rows << log_wq;
    ProfScope ps("k_interleave_blocks", st, 8.0 * n * n_cols);
    VK_LAUNCH(k_interleave_blocks, dim3((unsigned)((n + 255) / 256), (unsigned)n_cols), dim3(256), 0, st, src, dst, rows, log_wq);
}

// vgpu_shader_clock_probe: shader cycles (s_memtime) against the constant 100 MHz clock over a chain of dependent VALU additions
__global__ void k_clock_probe(uint64_t* out, uint32_t iters) {
    const uint64_t c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    uint32_t x = threadIdx.x;
    for (uint32_t i = 0; i < iters; i++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(x));
    const uint64_t c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = x; }
}
// a plain launch: the caller is not a proving thread, the profiler's launch scope (profiler.hpp) belongs to those
void launch_clock_probe(hipStream_t st, uint64_t* out3, uint32_t iters) { hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, st, out3, iters); }

void launch_ingest(hipStream_t st, const uint32_t* src_dev, DMatView dst, bool bitrev) {
    int log_h = (int)vg::log2_strict_u64(dst.height);
    unsigned blocks = (unsigned)((dst.height + 63) / 64);
    size_t lds = (size_t)64 * (dst.width | 1) * 4;
    ProfScope ps("k_ingest", st, 8.0 * dst.height * dst.width);
    VK_LAUNCH(k_ingest, dim3(blocks), dim3(256), lds, st, src_dev, dst, log_h, bitrev ? 1 : 0);
}
void launch_bitrev_rows(hipStream_t st, DMatView src, DMatView dst) {
    int k = (int)vg::log2_strict_u64(src.height);
    ProfScope ps("k_bitrev_rows", st, 8.0 * src.height * src.width);
    unsigned bx = k < 12 ? (unsigned)(((1u << k) + 255) / 256) : (1u << (k - 12));
    VK_LAUNCH(k_bitrev_rows, dim3(bx, (unsigned)src.width), dim3(256), 0, st, src, dst, k);
}
void launch_export_rows(hipStream_t st, DMatView src, uint64_t row0, uint64_t nrows, uint32_t* dst_dev) {
    uint64_t total = nrows * src.width;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 65535u * 16) blocks = 65535u * 16;
    if (blocks == 0) blocks = 1;
    ProfScope ps("k_export_rows", st, 8.0 * total);
    VK_LAUNCH(k_export_rows, dim3(blocks), dim3(256), 0, st, src, row0, nrows, dst_dev);
}

}  // namespace vk
