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

// Keccak-f[1600] on 32-bit halves.  gfx950 VALU is 32-bit: 64-bit xors are two ops anyway, but 64-bit
// SHIFTS are slow multi-pass instructions, so every lane is kept as (lo, hi) and rotated with
// v_alignbit_b32 (2 per rotation); chi and theta's column parity use gfx950's v_bitop3_b32 (any 3-input
// boolean function: one op per half for chi, two for a 5-way xor).  178 full-rate VALU instructions per round.
__constant__ uint32_t KECCAK_RC_LO[24] = {0x00000001u, 0x00008082u, 0x0000808au, 0x80008000u, 0x0000808bu, 0x80000001u, 0x80008081u, 0x00008009u,
                                          0x0000008au, 0x00000088u, 0x80008009u, 0x8000000au, 0x8000808bu, 0x0000008bu, 0x00008089u, 0x00008003u,
                                          0x00008002u, 0x00000080u, 0x0000800au, 0x8000000au, 0x80008081u, 0x00008080u, 0x80000001u, 0x80008008u};
__constant__ uint32_t KECCAK_RC_HI[24] = {0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u,
                                          0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x80000000u,
                                          0x80000000u, 0x80000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0x00000000u, 0x80000000u};

struct KState { uint32_t lo[25], hi[25]; };

// rotation offsets r[x][y], index x + 5*y
__device__ __forceinline__ constexpr int keccak_rot(int i) {
    constexpr int R[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    return R[i];
}
// (lo, hi) rotated left by the compile-time constant N
template <int N> __device__ __forceinline__ void rotl_pair(uint32_t lo, uint32_t hi, uint32_t& olo, uint32_t& ohi) {
    if (N == 0) { olo = lo; ohi = hi; }
    else if (N == 32) { olo = hi; ohi = lo; }
    else if (N < 32) { ohi = __builtin_amdgcn_alignbit(hi, lo, 32 - N); olo = __builtin_amdgcn_alignbit(lo, hi, 32 - N); }
    else { ohi = __builtin_amdgcn_alignbit(lo, hi, 64 - N); olo = __builtin_amdgcn_alignbit(hi, lo, 64 - N); }
}
// gfx950 v_bitop3_b32: any 3-input boolean function in one instruction (truth table over a=0xF0, b=0xCC, c=0xAA).
// Operands known to be zero at compile time (the capacity lanes of a freshly padded block, in the peeled first
// round) fold away instead of occupying an issue slot.
#define VK_KNOWN_ZERO(v) (__builtin_constant_p(v) && (v) == 0)
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    if (VK_KNOWN_ZERO(c)) return a ^ b;
    if (VK_KNOWN_ZERO(b)) return a ^ c;
    if (VK_KNOWN_ZERO(a)) return b ^ c;
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}
__device__ __forceinline__ uint32_t chi32(uint32_t b0, uint32_t b1, uint32_t b2) {  // b0 ^ (~b1 & b2)
    if (VK_KNOWN_ZERO(b2)) return b0;
    if (VK_KNOWN_ZERO(b1)) return b0 ^ b2;
    return __builtin_amdgcn_bitop3_b32(b0, b1, b2, 0xD2);
}

// theta folded into rho/pi: B[pi(x,y)] = rotl(A[x,y] ^ C[x-1] ^ rotl(C[x+1], 1), r[x,y]) — the column parity
// D is never materialised, its two terms ride in the xor3 that applies it (10 ops fewer per round).
template <int X, int Y> __device__ __forceinline__ void theta_rho_pi(const KState& a, const uint32_t (&cl)[5], const uint32_t (&ch)[5], const uint32_t (&rl)[5],
                                                                      const uint32_t (&rh)[5], KState& b) {
    constexpr int src = X + 5 * Y, dst = Y + 5 * ((2 * X + 3 * Y) % 5);
    const uint32_t tl = xor3(a.lo[src], cl[(X + 4) % 5], rl[(X + 1) % 5]);
    const uint32_t th = xor3(a.hi[src], ch[(X + 4) % 5], rh[(X + 1) % 5]);
    rotl_pair<keccak_rot(src)>(tl, th, b.lo[dst], b.hi[dst]);
}
template <int X> __device__ __forceinline__ void theta_rho_pi_col(const KState& a, const uint32_t (&cl)[5], const uint32_t (&ch)[5], const uint32_t (&rl)[5],
                                                                   const uint32_t (&rh)[5], KState& b) {
    theta_rho_pi<X, 0>(a, cl, ch, rl, rh, b); theta_rho_pi<X, 1>(a, cl, ch, rl, rh, b); theta_rho_pi<X, 2>(a, cl, ch, rl, rh, b);
    theta_rho_pi<X, 3>(a, cl, ch, rl, rh, b); theta_rho_pi<X, 4>(a, cl, ch, rl, rh, b);
}
__device__ __forceinline__ void column_parity(const KState& a, uint32_t (&cl)[5], uint32_t (&ch)[5], uint32_t (&rl)[5], uint32_t (&rh)[5]) {
#pragma unroll
    for (int x = 0; x < 5; x++) {
        cl[x] = xor3(xor3(a.lo[x], a.lo[x + 5], a.lo[x + 10]), a.lo[x + 15], a.lo[x + 20]);
        ch[x] = xor3(xor3(a.hi[x], a.hi[x + 5], a.hi[x + 10]), a.hi[x + 15], a.hi[x + 20]);
    }
#pragma unroll
    for (int x = 0; x < 5; x++) rotl_pair<1>(cl[x], ch[x], rl[x], rh[x]);
}

// one full round: 20 (parity) + 10 (rot1) + 50 (theta apply) + 46 (rho) + 50 (chi) + 2 (iota) = 178 VALU ops
__device__ __forceinline__ void keccak_round(KState& a, uint32_t rc_lo, uint32_t rc_hi) {
    uint32_t cl[5], ch[5], rl[5], rh[5];
    KState b;
    column_parity(a, cl, ch, rl, rh);
    theta_rho_pi_col<0>(a, cl, ch, rl, rh, b); theta_rho_pi_col<1>(a, cl, ch, rl, rh, b); theta_rho_pi_col<2>(a, cl, ch, rl, rh, b);
    theta_rho_pi_col<3>(a, cl, ch, rl, rh, b); theta_rho_pi_col<4>(a, cl, ch, rl, rh, b);
#pragma unroll
    for (int y = 0; y < 5; y++)
#pragma unroll
        for (int x = 0; x < 5; x++) {
            a.lo[x + 5 * y] = chi32(b.lo[x + 5 * y], b.lo[(x + 1) % 5 + 5 * y], b.lo[(x + 2) % 5 + 5 * y]);
            a.hi[x + 5 * y] = chi32(b.hi[x + 5 * y], b.hi[(x + 1) % 5 + 5 * y], b.hi[(x + 2) % 5 + 5 * y]);
        }
    a.lo[0] ^= rc_lo;
    a.hi[0] ^= rc_hi;
}
// last round when only the 256-bit digest (lanes 0..3) is squeezed: row 0 of the output reads B[0..4], whose
// pi-preimages are the diagonal lanes (x, x) — 58 ops instead of 178.
__device__ __forceinline__ void keccak_last_round_digest(KState& a, uint32_t rc_lo, uint32_t rc_hi) {
    uint32_t cl[5], ch[5], rl[5], rh[5];
    KState b;
    column_parity(a, cl, ch, rl, rh);
    theta_rho_pi<0, 0>(a, cl, ch, rl, rh, b); theta_rho_pi<1, 1>(a, cl, ch, rl, rh, b); theta_rho_pi<2, 2>(a, cl, ch, rl, rh, b);
    theta_rho_pi<3, 3>(a, cl, ch, rl, rh, b); theta_rho_pi<4, 4>(a, cl, ch, rl, rh, b);
#pragma unroll
    for (int x = 0; x < 4; x++) {
        a.lo[x] = chi32(b.lo[x], b.lo[x + 1], b.lo[(x + 2) % 5]);
        a.hi[x] = chi32(b.hi[x], b.hi[x + 1], b.hi[(x + 2) % 5]);
    }
    a.lo[0] ^= rc_lo;
    a.hi[0] ^= rc_hi;
}

// DIGEST_ONLY: the caller squeezes lanes 0..3 and drops the state (every permutation of this file except the
// non-final blocks of a wide row).  The first round is peeled so compile-time-zero lanes fold.
template <bool DIGEST_ONLY> __device__ __forceinline__ void keccak_f1600(KState& a) {
    keccak_round(a, 0x00000001u, 0x00000000u);
#pragma unroll 2
    for (int round = 1; round < 23; round++) keccak_round(a, KECCAK_RC_LO[round], KECCAK_RC_HI[round]);
    if (DIGEST_ONLY) keccak_last_round_digest(a, 0x80008008u, 0x80000000u);
    else keccak_round(a, 0x80008008u, 0x80000000u);
}

__device__ __forceinline__ uint32_t wrap_mod_p(uint32_t w) {  // from_wrapped_u32 -> canonical
    if (w >= vg::P) w -= vg::P;
    if (w >= vg::P) w -= vg::P;
    return w;
}

// 32-bit word k of the rate (k < 34): even words are the low halves of lane k/2
__device__ __forceinline__ void absorb_word(KState& a, int k, uint32_t w) { if (k & 1) a.hi[k >> 1] ^= w; else a.lo[k >> 1] ^= w; }
__device__ __forceinline__ void kstate_zero(KState& a) {
#pragma unroll
    for (int i = 0; i < 25; i++) { a.lo[i] = 0; a.hi[i] = 0; }
}
__device__ __forceinline__ void squeeze_digest(const KState& a, uint32_t (&out)[8]) {
#pragma unroll
    for (int i = 0; i < 4; i++) { out[2 * i] = wrap_mod_p(a.lo[i]); out[2 * i + 1] = wrap_mod_p(a.hi[i]); }
}

// Hash of one row of the column list `cols` (n_elems Montgomery columns, element r of each).
struct PtrCols {  // arbitrary column list (mixed matrices of one height)
    const uint32_t* const* p;
    __device__ __forceinline__ const uint32_t* operator[](int k) const { return p[k]; }
};
struct StridedCols {  // one column-major matrix: no pointer table needed (FRI layer trees)
    const uint32_t* base;
    uint64_t stride;
    __device__ __forceinline__ const uint32_t* operator[](int k) const { return base + (uint64_t)k * stride; }
};
template <class Cols>
__device__ __forceinline__ void hash_row(const Cols cols, int n_elems, uint64_t r, uint32_t (&out)[8]) {
    KState a;
    kstate_zero(a);
    int base = 0;
    for (; base + 34 <= n_elems; base += 34) {
#pragma unroll
        for (int k = 0; k < 34; k++) absorb_word(a, k, Fp::raw(cols[base + k][r]).canonical());
        keccak_f1600<false>(a);
    }
    const int rem = n_elems - base;  // 0..33
#pragma unroll
    for (int k = 0; k < 34; k++) {
        if (k < rem) absorb_word(a, k, Fp::raw(cols[base + k][r]).canonical());
        if (k == rem) absorb_word(a, k, 0x01u);  // Keccak (not SHA-3) domain padding
    }
    absorb_word(a, 33, 0x80000000u);
    keccak_f1600<true>(a);
    squeeze_digest(a, out);
}

// C(l, r): 16 canonical words, one block.
__device__ __forceinline__ void compress2(const uint32_t (&l)[8], const uint32_t (&r)[8], uint32_t (&out)[8]) {
    KState a;
    kstate_zero(a);
#pragma unroll
    for (int i = 0; i < 4; i++) { a.lo[i] = l[2 * i]; a.hi[i] = l[2 * i + 1]; a.lo[4 + i] = r[2 * i]; a.hi[4 + i] = r[2 * i + 1]; }
    a.lo[8] = 0x01u;
    a.hi[16] = 0x80000000u;
    keccak_f1600<true>(a);
    squeeze_digest(a, out);
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
template <class Cols>
__global__ void __launch_bounds__(256) k_keccak_leaves(const Cols cols, int n_elems, uint64_t n_rows, uint32_t* __restrict__ digests) {
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
        hash_row(PtrCols{cols}, n_elems, i, h);
        compress2(d, h, d2);
        store_digest(next + 8 * i, d2);
    } else {
        store_digest(next + 8 * i, d);
    }
}

// Top of a tree in ONE launch: a single 1024-thread workgroup walks the last `levels` layers
// (first_len <= 1024 parents down to the root), one barrier per layer, instead of one ~10 us launch per
// layer — there are ~25 trees per proof (3 commitment rounds + one per FRI layer).
__global__ void __launch_bounds__(1024) k_keccak_top(KeccakTopArgs a) {
    const uint32_t* prev = a.prev;
    for (int l = 0; l < a.levels; l++) {
        const uint64_t len = a.first_len >> l;
        if (threadIdx.x < len) {
            const uint64_t i = threadIdx.x;
            uint32_t lft[8], rgt[8], d[8];
            load_digest(prev + 16 * i, lft);
            load_digest(prev + 16 * i + 8, rgt);
            compress2(lft, rgt, d);
            if (a.n_elems[l] > 0) {
                uint32_t h[8], d2[8];
                hash_row(PtrCols{a.cols[l]}, a.n_elems[l], i, h);
                compress2(d, h, d2);
                store_digest(a.out[l] + 8 * i, d2);
            } else {
                store_digest(a.out[l] + 8 * i, d);
            }
        }
        __threadfence_block();
        __syncthreads();
        prev = a.out[l];
    }
}

// Algorithmic VALU work of the Keccak kernels, in wave64 instructions: permutations x (23 full rounds of 178 ops +
// the 58-op digest-only last round) / 64 lanes.  A row of n field elements absorbs floor(n / 34) + 1 blocks.
constexpr double KECCAK_VALU_PER_PERM = 23 * 178.0 + 58.0;
static double row_perms(int n_elems) { return (double)(n_elems / 34 + 1); }
static double node_perms(int n_inject) { return n_inject > 0 ? 2.0 + row_perms(n_inject) : 1.0; }

void launch_keccak_top(hipStream_t st, const KeccakTopArgs& a) {
    double bytes = 0, perms = 0;
    for (int l = 0; l < a.levels; l++) {
        bytes += (double)(a.first_len >> l) * (96.0 + 4.0 * a.n_elems[l]);
        perms += (double)(a.first_len >> l) * node_perms(a.n_elems[l]);
    }
    ProfScope ps("k_keccak_top", st, bytes, perms * KECCAK_VALU_PER_PERM / 64.0);
    hipLaunchKernelGGL(k_keccak_top, dim3(1), dim3(1024), 0, st, a);
}

void launch_keccak_leaves(hipStream_t st, const uint32_t* const* cols_dev, int n_elems, uint64_t n_rows, uint32_t* digests) {
    unsigned blocks = (unsigned)((n_rows + 255) / 256);
    ProfScope ps("k_keccak_leaves", st, (double)n_rows * (4.0 * n_elems + 32.0), (double)n_rows * row_perms(n_elems) * KECCAK_VALU_PER_PERM / 64.0);
    hipLaunchKernelGGL(k_keccak_leaves<PtrCols>, dim3(blocks), dim3(256), 0, st, PtrCols{cols_dev}, n_elems, n_rows, digests);
}
void launch_keccak_leaves_strided(hipStream_t st, const uint32_t* base, uint64_t stride, int n_elems, uint64_t n_rows, uint32_t* digests) {
    unsigned blocks = (unsigned)((n_rows + 255) / 256);
    ProfScope ps("k_keccak_leaves", st, (double)n_rows * (4.0 * n_elems + 32.0), (double)n_rows * row_perms(n_elems) * KECCAK_VALU_PER_PERM / 64.0);
    hipLaunchKernelGGL(k_keccak_leaves<StridedCols>, dim3(blocks), dim3(256), 0, st, StridedCols{base, stride}, n_elems, n_rows, digests);
}
void launch_keccak_compress(hipStream_t st, const uint32_t* prev, const uint32_t* const* cols_dev, int n_elems, uint64_t n_out, uint32_t* next) {
    unsigned blocks = (unsigned)((n_out + 255) / 256);
    ProfScope ps("k_keccak_compress", st, (double)n_out * (96.0 + 4.0 * n_elems), (double)n_out * node_perms(n_elems) * KECCAK_VALU_PER_PERM / 64.0);
    hipLaunchKernelGGL(k_keccak_compress, dim3(blocks), dim3(256), 0, st, prev, cols_dev, n_elems, n_out, next);
}

}  // namespace vk
