// Keccak-256 Merkle commitment kernels: the device side of
// FieldMerkleTreeMmcs<BabyBear, SerializingHasher32<Keccak256Hash>, CompressionFunctionFromHasher<_,_,2,8>, 8>
// as configured at basic/tests/test_prover.rs:424-431 (conventions: SURVEY.md App. B5/B6):
//   leaf digest(row r)  = Keccak256( LE32(canonical(e)) for e in concat(row r of every tallest matrix) )
//   digest -> 8 field elements: each LE u32 word reduced mod p (from_wrapped_u32)
//   parent              = C(left, right) = Keccak256 over the 16 canonical words of (left || right)
//   injection at a layer whose length equals the height of shorter matrices: C(parent, H(rows))
// One thread per leaf / per parent; a wave reads 64 consecutive rows of each column (coalesced 256 B).
// Keccak-f[1600] is 64-bit-lane integer ALU work: this stage is VALU-bound, not HBM-bound
// (SURVEY.md §8(d) caveat) — 25 lanes = 50 VGPRs of state per thread.
#include "launch.hpp"

namespace vk {

__constant__ uint64_t KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
    0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int n) { return n == 0 ? x : (x << n) | (x >> (64 - n)); }

// rotation offsets r[x][y], index x + 5*y
__device__ __forceinline__ constexpr int keccak_rot(int i) {
    constexpr int R[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    return R[i];
}

__device__ __forceinline__ void keccak_f1600(uint64_t (&a)[25]) {
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; x++) {
            uint64_t d = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
#pragma unroll
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
        }
#pragma unroll
        for (int x = 0; x < 5; x++)
#pragma unroll
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y], keccak_rot(x + 5 * y));
#pragma unroll
        for (int y = 0; y < 5; y++)
#pragma unroll
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= KECCAK_RC[round];
    }
}

__device__ __forceinline__ uint32_t wrap_mod_p(uint32_t w) {  // from_wrapped_u32 -> canonical
    if (w >= vg::P) w -= vg::P;
    if (w >= vg::P) w -= vg::P;
    return w;
}

__device__ __forceinline__ void absorb_word(uint64_t (&a)[25], int k, uint32_t w) { a[k >> 1] ^= (uint64_t)w << (32 * (k & 1)); }

// Hash of one row of the column list `cols` (n_elems Montgomery columns, element r of each).
__device__ __forceinline__ void hash_row(const uint32_t* const* __restrict__ cols, int n_elems, uint64_t r, uint32_t (&out)[8]) {
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = 0;
    int base = 0;
    for (; base + 34 <= n_elems; base += 34) {
#pragma unroll
        for (int k = 0; k < 34; k++) absorb_word(a, k, Fp::raw(cols[base + k][r]).canonical());
        keccak_f1600(a);
    }
    const int rem = n_elems - base;  // 0..33
#pragma unroll
    for (int k = 0; k < 34; k++) {
        if (k < rem) absorb_word(a, k, Fp::raw(cols[base + k][r]).canonical());
        if (k == rem) absorb_word(a, k, 0x01u);  // Keccak (not SHA-3) domain padding
    }
    absorb_word(a, 33, 0x80000000u);
    keccak_f1600(a);
#pragma unroll
    for (int i = 0; i < 4; i++) { out[2 * i] = wrap_mod_p((uint32_t)a[i]); out[2 * i + 1] = wrap_mod_p((uint32_t)(a[i] >> 32)); }
}

// C(l, r): 16 canonical words, one block.
__device__ __forceinline__ void compress2(const uint32_t (&l)[8], const uint32_t (&r)[8], uint32_t (&out)[8]) {
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = (uint64_t)l[2 * i] | ((uint64_t)l[2 * i + 1] << 32); a[4 + i] = (uint64_t)r[2 * i] | ((uint64_t)r[2 * i + 1] << 32); }
    a[8] = 0x01ull;
    a[16] ^= 0x8000000000000000ull;
    keccak_f1600(a);
#pragma unroll
    for (int i = 0; i < 4; i++) { out[2 * i] = wrap_mod_p((uint32_t)a[i]); out[2 * i + 1] = wrap_mod_p((uint32_t)(a[i] >> 32)); }
}

__device__ __forceinline__ void load_digest(const uint32_t* p, uint32_t (&d)[8]) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    d[0] = lo.x; d[1] = lo.y; d[2] = lo.z; d[3] = lo.w; d[4] = hi.x; d[5] = hi.y; d[6] = hi.z; d[7] = hi.w;
}
__device__ __forceinline__ void store_digest(uint32_t* p, const uint32_t (&d)[8]) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(d[0], d[1], d[2], d[3]);
    q[1] = make_uint4(d[4], d[5], d[6], d[7]);
}

// leaf layer: digests[r] = H(row r)
__global__ void __launch_bounds__(256) k_keccak_leaves(const uint32_t* const* __restrict__ cols, int n_elems, uint64_t n_rows, uint32_t* __restrict__ digests) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    uint32_t d[8];
    hash_row(cols, n_elems, r, d);
    store_digest(digests + 8 * r, d);
}

// next[i] = C(prev[2i], prev[2i+1]); if n_elems > 0: next[i] = C(next[i], H(row i of cols))
__global__ void __launch_bounds__(256) k_keccak_compress(const uint32_t* __restrict__ prev, const uint32_t* const* __restrict__ cols, int n_elems, uint64_t n_out,
                                  uint32_t* __restrict__ next) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    uint32_t l[8], r[8], d[8];
    load_digest(prev + 16 * i, l);
    load_digest(prev + 16 * i + 8, r);
    compress2(l, r, d);
    if (n_elems > 0) {
        uint32_t h[8], d2[8];
        hash_row(cols, n_elems, i, h);
        compress2(d, h, d2);
        store_digest(next + 8 * i, d2);
    } else {
        store_digest(next + 8 * i, d);
    }
}

void launch_keccak_leaves(hipStream_t st, const uint32_t* const* cols_dev, int n_elems, uint64_t n_rows, uint32_t* digests) {
    unsigned blocks = (unsigned)((n_rows + 255) / 256);
    ProfScope ps("k_keccak_leaves", st, (double)n_rows * (4.0 * n_elems + 32.0));
    hipLaunchKernelGGL(k_keccak_leaves, dim3(blocks), dim3(256), 0, st, cols_dev, n_elems, n_rows, digests);
}
void launch_keccak_compress(hipStream_t st, const uint32_t* prev, const uint32_t* const* cols_dev, int n_elems, uint64_t n_out, uint32_t* next) {
    unsigned blocks = (unsigned)((n_out + 255) / 256);
    ProfScope ps("k_keccak_compress", st, (double)n_out * (96.0 + 4.0 * n_elems));
    hipLaunchKernelGGL(k_keccak_compress, dim3(blocks), dim3(256), 0, st, prev, cols_dev, n_elems, n_out, next);
}

}  // namespace vk
