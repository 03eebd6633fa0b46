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
#include "keccak.hpp"
#include "keccak_pair.hpp"
#include "challenger_dev.hpp"
#include <cstdlib>
#include <stdexcept>

namespace vk {

__device__ __forceinline__ uint32_t wrap_mod_p(uint32_t w) {  // from_wrapped_u32 -> canonical
    if (w >= vg::P) w -= vg::P;
    if (w >= vg::P) w -= vg::P;
    return w;
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

// ---- lane-pair variants (keccak_pair.hpp) for the latency-bound layers: thread t works on node t / 2, half t & 1 ----
// A digest is 8 words d[0..7] = (lane0.lo, lane0.hi, lane1.lo, ...): the half-h thread holds d[h], d[2 + h], d[4 + h], d[6 + h].
template <class Cols>
__device__ __forceinline__ void hash_row_pair(const Cols cols, int n_elems, uint64_t r, int h, uint32_t (&out)[4]) {
    KHalf a;
#pragma unroll
    for (int i = 0; i < 25; i++) a.s[i] = 0;
    int base = 0;
    for (; base + 34 <= n_elems; base += 34) {
#pragma unroll
        for (int l = 0; l < 17; l++) a.s[l] ^= Fp::raw(cols[base + 2 * l + h][r]).canonical();
        keccak_f1600_pair<false>(a, h);
    }
    const int rem = n_elems - base;  // 0..33: words base .. base + rem - 1, then the 0x01 pad word at position rem
#pragma unroll
    for (int l = 0; l < 17; l++) {
        const int k = 2 * l + h;
        if (k < rem) a.s[l] ^= Fp::raw(cols[base + k][r]).canonical();
        if (k == rem) a.s[l] ^= 0x01u;
    }
    if (h) a.s[16] ^= 0x80000000u;  // word 33
    keccak_f1600_pair<true>(a, h);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = wrap_mod_p(a.s[i]);
}
__device__ __forceinline__ void compress2_pair(const uint32_t (&l)[4], const uint32_t (&r)[4], int h, uint32_t (&out)[4]) {
    KHalf a;
#pragma unroll
    for (int i = 0; i < 25; i++) a.s[i] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { a.s[i] = l[i]; a.s[4 + i] = r[i]; }
    if (!h) a.s[8] = 0x01u;
    else a.s[16] = 0x80000000u;
    keccak_f1600_pair<true>(a, h);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = wrap_mod_p(a.s[i]);
}
__device__ __forceinline__ void load_digest_half(const uint32_t* p, int h, uint32_t (&d)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) d[i] = p[2 * i + h];
}
__device__ __forceinline__ void store_digest_half(uint32_t* p, int h, const uint32_t (&d)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) p[2 * i + h] = d[i];
}
// one node (parent i of `prev`, optionally with the injected row i of `cols`) by the lane pair
__device__ __forceinline__ void node_pair(const uint32_t* prev, const uint32_t* const* cols, int n_elems, uint64_t i, int h, uint32_t* out) {
    uint32_t l[4], r[4], d[4];
    load_digest_half(prev + 16 * i, h, l);
    load_digest_half(prev + 16 * i + 8, h, r);
    compress2_pair(l, r, h, d);
    if (n_elems > 0) {
        uint32_t hr[4], d2[4];
        hash_row_pair(PtrCols{cols}, n_elems, i, h, hr);
        compress2_pair(d, hr, h, d2);
        store_digest_half(out + 8 * i, h, d2);
    } else {
        store_digest_half(out + 8 * i, h, d);
    }
}
template <class Cols>
__global__ void __launch_bounds__(256) k_keccak_leaves_pair(const Cols cols, int n_elems, uint64_t n_rows, uint32_t* __restrict__ digests) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, r = t >> 1;
    if (r >= n_rows) return;
    uint32_t d[4];
    hash_row_pair(cols, n_elems, r, (int)(t & 1), d);
    store_digest_half(digests + 8 * r, (int)(t & 1), d);
}
__global__ void __launch_bounds__(256) k_keccak_compress_pair(const uint32_t* __restrict__ prev, const uint32_t* const* __restrict__ cols, int n_elems, uint64_t n_out,
                                                             uint32_t* __restrict__ next) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, i = t >> 1;
    if (i >= n_out) return;
    node_pair(prev, cols, n_elems, i, (int)(t & 1), next);
}
__global__ void __launch_bounds__(1024) k_keccak_top_pair(KeccakTopArgs a) {
    const uint32_t* prev = a.prev;
    if (a.leaf_rows) {  // the leaves of a small single-matrix tree: one row per lane pair, written where the first level reads them
        if ((threadIdx.x >> 1) < a.leaf_rows) {
            uint32_t d[4];
            hash_row_pair(StridedCols{a.leaf_base, a.leaf_stride}, a.leaf_elems, threadIdx.x >> 1, (int)(threadIdx.x & 1), d);
            store_digest_half(const_cast<uint32_t*>(a.prev) + 8 * (threadIdx.x >> 1), (int)(threadIdx.x & 1), d);
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int l = 0; l < a.levels; l++) {
        const uint64_t len = a.first_len >> l;
        if ((threadIdx.x >> 1) < len) node_pair(prev, a.cols[l], a.n_elems[l], threadIdx.x >> 1, (int)(threadIdx.x & 1), a.out[l]);
        __threadfence_block();
        __syncthreads();
        prev = a.out[l];
    }
    if (a.ch_pos && threadIdx.x < 64) fri_challenge_step((int)threadIdx.x, a.ch_pos, a.ch_state, prev, a.ch_beta5, a.ch_commit8);
}

// Several layers per launch with the digests handed from layer to layer through LDS (the single-workgroup top of every tree, and the
// latency-bound middle: one workgroup per block_len parents of the first layer, each walking its own sub-tree down).  Against
// k_keccak_top_pair / one k_keccak_compress_pair launch per layer: a layer boundary costs an LDS write, `s_waitcnt lgkmcnt(0); s_barrier`
// and an LDS read instead of global stores drained to L2 + a barrier + global loads (top) or a kernel boundary (middle).  Every digest is
// still written to its layer in HBM — the openings read them — but nothing waits for those stores before the kernel ends.
// LDS: X (512 digests: the leaves of the prologue, then the outputs of the odd layers) and Y (256 digests: the outputs of the even layers).
constexpr int LEVELS_LDS_X = 512, LEVELS_LDS_Y = 256;
__device__ __forceinline__ void lds_barrier() {
#ifdef __HIPCC__
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the LDS traffic of this wave is complete; the global stores are NOT waited for
#else
    __syncthreads();  // tools/hipemu (g++) compiles this file for the thread-per-node kernels
#endif
}
__global__ void __launch_bounds__(1024) k_keccak_levels_pair(KeccakTopArgs a) {
    __shared__ uint32_t lds[(LEVELS_LDS_X + LEVELS_LDS_Y) * 8];
    const int h = (int)(threadIdx.x & 1);
    const uint32_t t = threadIdx.x >> 1;
    const uint64_t b0 = a.block_len, first = (uint64_t)blockIdx.x * b0;  // this workgroup: parents first .. first + b0 - 1 of the first layer
    bool in_lds = false;
    if (a.leaf_rows) {  // the leaves of a single strided matrix: rows 2 * first .. 2 * (first + b0) - 1, one per lane pair
        if (t < 2 * b0) {
            uint32_t d[4];
            const uint64_t row = 2 * first + t;
            hash_row_pair(StridedCols{a.leaf_base, a.leaf_stride}, a.leaf_elems, row, h, d);
            store_digest_half(const_cast<uint32_t*>(a.prev) + 8 * row, h, d);
            store_digest_half(lds + 8 * t, h, d);
        }
        lds_barrier();
        in_lds = true;
    }
    for (int l = 0; l < a.levels; l++) {
        const uint32_t* in = lds + ((l & 1) ? LEVELS_LDS_X * 8 : 0);
        uint32_t* out = lds + ((l & 1) ? 0 : LEVELS_LDS_X * 8);
        if (t < (b0 >> l)) {
            const uint64_t i = (first >> l) + t;
            uint32_t lf[4], rg[4], d[4];
            if (in_lds) { load_digest_half(in + 16 * t, h, lf); load_digest_half(in + 16 * t + 8, h, rg); }
            else { load_digest_half(a.prev + 16 * i, h, lf); load_digest_half(a.prev + 16 * i + 8, h, rg); }
            compress2_pair(lf, rg, h, d);
            if (a.n_elems[l] > 0) {
                uint32_t hr[4], d2[4];
                hash_row_pair(PtrCols{a.cols[l]}, a.n_elems[l], i, h, hr);
                compress2_pair(d, hr, h, d2);
#pragma unroll
                for (int k = 0; k < 4; k++) d[k] = d2[k];
            }
            store_digest_half(a.out[l] + 8 * i, h, d);
            store_digest_half(out + 8 * t, h, d);
        }
        in_lds = true;
        if (l + 1 < a.levels) lds_barrier();
    }
    if (a.ch_pos) {  // one workgroup (the top of a FRI layer tree): the root it has just written, read back from memory by the first wave
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x < 64) fri_challenge_step((int)threadIdx.x, a.ch_pos, a.ch_state, a.out[a.levels - 1], a.ch_beta5, a.ch_commit8);
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
    if (a.ch_pos && threadIdx.x < 64) fri_challenge_step((int)threadIdx.x, a.ch_pos, a.ch_state, prev, a.ch_beta5, a.ch_commit8);
}

// Algorithmic VALU work of the Keccak kernels, in wave64 instructions: permutations x (23 full rounds of 178 ops +
// the 58-op digest-only last round) / 64 lanes.  A row of n field elements absorbs floor(n / 34) + 1 blocks.
constexpr double KECCAK_VALU_PER_PERM = 23 * 178.0 + 58.0;
// Below this many nodes a layer cannot fill the GPU's wave slots (256 CUs x 4 SIMDs x >= 2 waves x 64 lanes = 131072 threads): it is
// latency-bound and the lane-pair permutation (2 threads per node, 0.67 x the instructions per thread) is the faster one.
// VGPU_KECCAK_PAIRS=0 switches the pair kernels off (A/B, tools/gpu_ab_env.sh).
constexpr uint64_t KECCAK_PAIR_MAX_NODES = 32768;  // 8192 .. 131072 measure the same (tools/gpu_pair_sweep.sh)
static bool keccak_pairs_enabled() {
    static const bool on = [] { const char* e = getenv("VGPU_KECCAK_PAIRS"); return !(e && e[0] == '0'); }();
    return on;
}
static double row_perms(int n_elems) { return (double)(n_elems / 34 + 1); }
static double node_perms(int n_inject) { return n_inject > 0 ? 2.0 + row_perms(n_inject) : 1.0; }

bool keccak_top_takes_leaves(uint64_t n_rows) { return keccak_pairs_enabled() && n_rows >= 2 && n_rows <= 512; }  // 2 threads per row, first_len = n_rows / 2 <= 256
// VGPU_KECCAK_LEVELS=0: the round-3 launches (k_keccak_top_pair with its layers through HBM, one k_keccak_compress_pair launch per middle layer)
static bool keccak_levels_enabled() {
    static const bool on = [] { const char* e = getenv("VGPU_KECCAK_LEVELS"); return !(e && e[0] == '0'); }();
    return on && keccak_pairs_enabled();
}
bool keccak_levels_fused(uint64_t len) { return keccak_levels_enabled() && len >= KECCAK_LEVELS_BLOCK_LEN && len <= KECCAK_PAIR_MAX_NODES; }
bool keccak_levels_take_leaves(uint64_t n_rows) { return n_rows >= 2 * KECCAK_LEVELS_BLOCK_LEN && keccak_levels_fused(n_rows / 2); }

void launch_keccak_levels(hipStream_t st, const KeccakTopArgs& a0) {
    KeccakTopArgs a = a0;
    if (!a.block_len) a.block_len = a.first_len;
    const uint64_t b0 = a.block_len;
    if (!keccak_levels_enabled() || !b0 || (b0 & (b0 - 1)) || b0 > (uint64_t)LEVELS_LDS_Y || a.first_len % b0 || a.levels < 1 || a.levels > KECCAK_TOP_MAX_LEVELS ||
        (b0 >> (a.levels - 1)) == 0 || (a.leaf_rows && a.leaf_rows != 2 * a.first_len) || (a.ch_pos && a.first_len != b0))
        throw std::logic_error("keccak levels: a launch this kernel does not fit");
    double bytes = 0, perms = 0;
    for (int l = 0; l < a.levels; l++) {
        bytes += (double)(a.first_len >> l) * (96.0 + 4.0 * a.n_elems[l]);
        perms += (double)(a.first_len >> l) * node_perms(a.n_elems[l]);
    }
    if (a.leaf_rows) {
        bytes += (double)a.leaf_rows * (4.0 * a.leaf_elems + 32.0);
        perms += (double)a.leaf_rows * row_perms(a.leaf_elems);
    }
    unsigned threads = (unsigned)((a.leaf_rows ? 4 : 2) * b0);
    if (threads < 64) threads = 64;
    ProfScope ps("k_keccak_levels_pair", st, bytes, perms * KECCAK_VALU_PER_PERM / 64.0);
    VK_LAUNCH(k_keccak_levels_pair, dim3((unsigned)(a.first_len / b0)), dim3(threads), 0, st, a);
}

void launch_keccak_top(hipStream_t st, const KeccakTopArgs& a) {
    double bytes = 0, perms = 0;
    for (int l = 0; l < a.levels; l++) {
        bytes += (double)(a.first_len >> l) * (96.0 + 4.0 * a.n_elems[l]);
        perms += (double)(a.first_len >> l) * node_perms(a.n_elems[l]);
    }
    const bool pairs = keccak_pairs_enabled() && a.first_len <= 512;
    if (a.leaf_rows) {
        if (!pairs || a.leaf_rows != 2 * a.first_len || !keccak_top_takes_leaves(a.leaf_rows)) throw std::logic_error("keccak top: leaf prologue on a tree it does not fit");
        bytes += (double)a.leaf_rows * (4.0 * a.leaf_elems + 32.0);
        perms += (double)a.leaf_rows * row_perms(a.leaf_elems);
    }
    if (pairs && keccak_levels_enabled() && a.first_len <= (uint64_t)LEVELS_LDS_Y) { launch_keccak_levels(st, a); return; }
    ProfScope ps(pairs ? "k_keccak_top_pair" : "k_keccak_top", st, bytes, perms * KECCAK_VALU_PER_PERM / 64.0);
    if (pairs) { VK_LAUNCH(k_keccak_top_pair, dim3(1), dim3(1024), 0, st, a); return; }
    VK_LAUNCH(k_keccak_top, dim3(1), dim3(1024), 0, st, a);
}

// ---- the bottom of a big tree at QUERY time (round 6) ------------------------------------------------------------------------------------------
// A tree of 2^k leaves keeps 2 x 2^k digests: half of them the leaf layer, a quarter the layer above — 2.1 + 1.1 GB per 2^26-row commitment of C3
// (three of them plus the FRI layers: 12 GB of the 45 GB a context held).  Both layers are needed again for 40 sibling digests each, so big trees
// give them back to the pool once the layers above are built (DeviceTree::drop_bottom) and the query phase RECOMPUTES the siblings from the committed
// rows: a job = the leaf hash of one row (level 0) or the compression of two neighbouring leaf hashes (level 1; only when that layer injects nothing).
// job: [0..1] column-pointer table (PtrCols) or matrix base (StridedCols)  [2..3] stride (0: pointer table)  [4] n_elems  [5] destination word
//      [6] query | level << 8 | shift << 16  [7] unused;  node = ((index[query] >> shift) >> level) ^ 1
// FOUR lanes per job (the lane-pair permutation of keccak_pair.hpp, ~8 us deep): pair A = lanes 0, 1 hashes row `node` (level 0) or row 2 node
// (level 1), pair B = lanes 2, 3 row 2 node + 1; at level 1 pair A then compresses its digest with pair B's (one quad_perm exchange).
template <class Cols> __device__ __forceinline__ void bottom_digest_quad(const Cols cols, int n_elems, uint64_t node, uint32_t level, int pair, int h, uint32_t (&d)[4]) {
    uint32_t own[4];
    hash_row_pair(cols, n_elems, level ? 2 * node + (uint64_t)pair : node, h, own);
    if (level == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) d[i] = own[i];
        return;
    }
    uint32_t right[4];
#pragma unroll
    for (int i = 0; i < 4; i++) right[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)own[i], 0xEE, 0xF, 0xF, true);  // quad_perm [2, 3, 2, 3]: pair B's halves
    compress2_pair(own, right, h, d);  // (pair B runs it too, on its own digest twice: discarded)
}
__global__ void __launch_bounds__(256) k_keccak_bottom_q(const uint32_t* __restrict__ jobs, uint32_t n_jobs, const uint32_t* __restrict__ indices, uint32_t* __restrict__ dst) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, j = t >> 2;
    if (j >= n_jobs) return;  // whole quads leave together
    const int pair = (int)((t >> 1) & 1u), h = (int)(t & 1u);
    const uint32_t* e = jobs + 8 * j;
    const uint64_t ptr = ((uint64_t)e[1] << 32) | e[0], stride = ((uint64_t)e[3] << 32) | e[2];
    const uint32_t q = e[6] & 0xffu, level = (e[6] >> 8) & 0xffu, shift = e[6] >> 16;
    const uint64_t node = (((uint64_t)indices[q] >> shift) >> level) ^ 1u;
    uint32_t d[4];
    if (stride) bottom_digest_quad(StridedCols{reinterpret_cast<const uint32_t*>(ptr), stride}, (int)e[4], node, level, pair, h, d);
    else bottom_digest_quad(PtrCols{reinterpret_cast<const uint32_t* const*>(ptr)}, (int)e[4], node, level, pair, h, d);
    if (pair == 0) store_digest_half(dst + e[5], h, d);
}
void launch_keccak_bottom_q(hipStream_t st, const uint32_t* jobs_dev, uint32_t n_jobs, const uint32_t* indices_dev, uint32_t* dst) {
    if (!n_jobs) return;
    ProfScope ps("k_gather", st, 0.0);
    VK_LAUNCH(k_keccak_bottom_q, dim3((4 * n_jobs + 255) / 256), dim3(256), 0, st, jobs_dev, n_jobs, indices_dev, dst);
}

void launch_keccak_leaves(hipStream_t st, const uint32_t* const* cols_dev, int n_elems, uint64_t n_rows, uint32_t* digests) {
    unsigned blocks = (unsigned)((n_rows + 255) / 256);
    const bool pairs = keccak_pairs_enabled() && n_rows <= KECCAK_PAIR_MAX_NODES;
    ProfScope ps(pairs ? "k_keccak_leaves_pair" : "k_keccak_leaves", st, (double)n_rows * (4.0 * n_elems + 32.0), (double)n_rows * row_perms(n_elems) * KECCAK_VALU_PER_PERM / 64.0);
    if (pairs) {
        VK_LAUNCH(k_keccak_leaves_pair<PtrCols>, dim3((unsigned)((2 * n_rows + 255) / 256)), dim3(256), 0, st, PtrCols{cols_dev}, n_elems, n_rows, digests);
        return;
    }
    VK_LAUNCH(k_keccak_leaves<PtrCols>, dim3(blocks), dim3(256), 0, st, PtrCols{cols_dev}, n_elems, n_rows, digests);
}
void launch_keccak_leaves_strided(hipStream_t st, const uint32_t* base, uint64_t stride, int n_elems, uint64_t n_rows, uint32_t* digests) {
    unsigned blocks = (unsigned)((n_rows + 255) / 256);
    const bool pairs = keccak_pairs_enabled() && n_rows <= KECCAK_PAIR_MAX_NODES;
    ProfScope ps(pairs ? "k_keccak_leaves_pair" : "k_keccak_leaves", st, (double)n_rows * (4.0 * n_elems + 32.0), (double)n_rows * row_perms(n_elems) * KECCAK_VALU_PER_PERM / 64.0);
    if (pairs) {
        VK_LAUNCH(k_keccak_leaves_pair<StridedCols>, dim3((unsigned)((2 * n_rows + 255) / 256)), dim3(256), 0, st, StridedCols{base, stride}, n_elems, n_rows, digests);
        return;
    }
    VK_LAUNCH(k_keccak_leaves<StridedCols>, dim3(blocks), dim3(256), 0, st, StridedCols{base, stride}, n_elems, n_rows, digests);
}
void launch_keccak_compress(hipStream_t st, const uint32_t* prev, const uint32_t* const* cols_dev, int n_elems, uint64_t n_out, uint32_t* next) {
    unsigned blocks = (unsigned)((n_out + 255) / 256);
    const bool pairs = keccak_pairs_enabled() && n_out <= KECCAK_PAIR_MAX_NODES;
    ProfScope ps(pairs ? "k_keccak_compress_pair" : "k_keccak_compress", st, (double)n_out * (96.0 + 4.0 * n_elems), (double)n_out * node_perms(n_elems) * KECCAK_VALU_PER_PERM / 64.0);
    if (pairs) {
        VK_LAUNCH(k_keccak_compress_pair, dim3((unsigned)((2 * n_out + 255) / 256)), dim3(256), 0, st, prev, cols_dev, n_elems, n_out, next);
        return;
    }
    VK_LAUNCH(k_keccak_compress, dim3(blocks), dim3(256), 0, st, prev, cols_dev, n_elems, n_out, next);
}

}  // namespace vk
