// Keccak-f[1600] for gfx950 as device functions (one permutation per thread, state in VGPRs): shared by the Merkle kernels
// (merkle.hip) and by tools/microbench.hip, which measures the permutation's issue cost in isolation.
// Reference: Keccak256Hash behind SerializingHasher32 / CompressionFunctionFromHasher (basic/tests/test_prover.rs:424-431);
// tiny-keccak 2.0.2 is the reference's implementation (Cargo.lock), the algorithm is FIPS 202's Keccak-p[1600, 24].
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace vk {

// Keccak-f[1600] on 32-bit halves.  gfx950 VALU is 32-bit: 64-bit xors are two ops anyway, but 64-bit
// SHIFTS are slow multi-pass instructions, so every lane is kept as (lo, hi) and rotated with
// v_alignbit_b32 (2 per rotation); chi and theta's column parity use gfx950's v_bitop3_b32 (any 3-input
// boolean function: one op per half for chi, two for a 5-way xor).  178 VALU instructions per round: 122 full-rate
// (bitop3 / xor) and 56 half-rate (alignbit) — how they ISSUE is what VK_ALIGNBIT_NOP and VK_KECCAK_PIN below are about.
static __constant__ uint32_t KECCAK_RC_LO[24] = {0x00000001u, 0x00008082u, 0x0000808au, 0x80008000u, 0x0000808bu, 0x80000001u, 0x80008081u, 0x00008009u,
                                          0x0000008au, 0x00000088u, 0x80008009u, 0x8000000au, 0x8000808bu, 0x0000008bu, 0x00008089u, 0x00008003u,
                                          0x00008002u, 0x00000080u, 0x0000800au, 0x8000000au, 0x80008081u, 0x00008080u, 0x80000001u, 0x80008008u};
static __constant__ uint32_t KECCAK_RC_HI[24] = {0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u,
                                          0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x80000000u,
                                          0x80000000u, 0x80000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0x00000000u, 0x80000000u};

struct KState { uint32_t lo[25], hi[25]; };

// rotation offsets r[x][y], index x + 5*y
__device__ __forceinline__ constexpr int keccak_rot(int i) {
    constexpr int R[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    return R[i];
}
// v_alignbit_b32 is a half-rate instruction, and a half-rate instruction directly followed by another VALU instruction of the same wave
// makes both cost a full slot; ONE scalar no-op behind it restores the additive cost (profiles/r02_issue_patterns.txt: the compiled
// order of this permutation 3.88 -> 3.48 SIMD-cycles per instruction at 8 waves per SIMD).  VK_ALIGNBIT_NOP=1 ties the no-op to every
// rotation instruction of the thread-per-permutation kernels (the compiler keeps scheduling and register allocation; the lane-pair
// variants are latency-bound — one wave per SIMD issues an instruction every ~5 cycles whatever it is — and stay without).  Measured in the
// product, A/B in one session (profiles/r03_ab_keccak_nop.json): 58.1 / 59.4 -> 61.1 / 61.3 proofs/s with three proofs in flight
// (16.84-17.21 -> 16.32-16.38 ms/step), 21.69 -> 21.38 ms for a lone proof.  Round 2's hand-scheduled asm permutation had the no-ops too
// but cost a wave of occupancy and all of the compiler's register allocation; this form costs nothing else.  =0: A/B builds.
#ifndef VK_ALIGNBIT_NOP
#define VK_ALIGNBIT_NOP 1
#endif
#ifndef VK_KECCAK_PIN
#define VK_KECCAK_PIN 1  // 1: a scheduling barrier behind every lane of theta / rho / pi (measured: 16.36 -> 16.21 ms/step, profiles/r03_ab_keccak_pin.json); 2: row by row (below)
#endif
template <int S> __device__ __forceinline__ uint32_t alignbit_c(uint32_t a, uint32_t b) {
#if VK_ALIGNBIT_NOP
    if (!__builtin_constant_p(a) && !__builtin_constant_p(b)) {
        uint32_t o;
        asm("v_alignbit_b32 %0, %1, %2, %3\n\ts_nop 0" : "=v"(o) : "v"(a), "v"(b), "n"(S));
        return o;
    }
#endif
    return __builtin_amdgcn_alignbit(a, b, S);
}
// (lo, hi) rotated left by the compile-time constant N
template <int N> __device__ __forceinline__ void rotl_pair(uint32_t lo, uint32_t hi, uint32_t& olo, uint32_t& ohi) {
    if (N == 0) { olo = lo; ohi = hi; }
    else if (N == 32) { olo = hi; ohi = lo; }
    else if (N < 32) { ohi = alignbit_c<32 - N>(hi, lo); olo = alignbit_c<32 - N>(lo, hi); }
    else { ohi = alignbit_c<64 - N>(lo, hi); olo = alignbit_c<64 - N>(hi, lo); }
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
#if VK_KECCAK_PIN == 1  // A/B builds: keep a lane's two xors and two rotations together (hipcc otherwise groups the round's 50 xors and 46 rotations)
    __builtin_amdgcn_sched_barrier(0);
#endif
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
// The same round ROW BY ROW of its output (VK_KECCAK_PIN == 2, A/B builds): the five lanes whose pi-images form output row YD — source
// (x, y) = ((3 YD + xd) % 5, xd) for xd = 0..4 — then chi of that row, a scheduling barrier behind every lane and every row: rotations
// and xors alternate in groups of 2 + 2 and 10 instead of 50 + 46 + 50, and a row of B lives for ten instructions.
template <int YD, int XD> __device__ __forceinline__ void theta_rho_pi_to(const KState& a, const uint32_t (&cl)[5], const uint32_t (&ch)[5], const uint32_t (&rl)[5],
                                                                           const uint32_t (&rh)[5], uint32_t (&bl)[5], uint32_t (&bh)[5]) {
    constexpr int X = (3 * YD + XD) % 5, Y = XD, src = X + 5 * Y;
    static_assert((2 * X + 3 * Y) % 5 == YD, "pi^-1");
    const uint32_t tl = xor3(a.lo[src], cl[(X + 4) % 5], rl[(X + 1) % 5]);
    const uint32_t th = xor3(a.hi[src], ch[(X + 4) % 5], rh[(X + 1) % 5]);
    rotl_pair<keccak_rot(src)>(tl, th, bl[XD], bh[XD]);
    __builtin_amdgcn_sched_barrier(0);
}
template <int YD> __device__ __forceinline__ void keccak_row(const KState& a, const uint32_t (&cl)[5], const uint32_t (&ch)[5], const uint32_t (&rl)[5], const uint32_t (&rh)[5],
                                                             KState& n) {
    uint32_t bl[5], bh[5];
    theta_rho_pi_to<YD, 0>(a, cl, ch, rl, rh, bl, bh); theta_rho_pi_to<YD, 1>(a, cl, ch, rl, rh, bl, bh); theta_rho_pi_to<YD, 2>(a, cl, ch, rl, rh, bl, bh);
    theta_rho_pi_to<YD, 3>(a, cl, ch, rl, rh, bl, bh); theta_rho_pi_to<YD, 4>(a, cl, ch, rl, rh, bl, bh);
#pragma unroll
    for (int x = 0; x < 5; x++) {
        n.lo[x + 5 * YD] = chi32(bl[x], bl[(x + 1) % 5], bl[(x + 2) % 5]);
        n.hi[x + 5 * YD] = chi32(bh[x], bh[(x + 1) % 5], bh[(x + 2) % 5]);
    }
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void keccak_round_rows(KState& a, uint32_t rc_lo, uint32_t rc_hi) {
    uint32_t cl[5], ch[5], rl[5], rh[5];
    KState n;
    column_parity(a, cl, ch, rl, rh);
    keccak_row<0>(a, cl, ch, rl, rh, n); keccak_row<1>(a, cl, ch, rl, rh, n); keccak_row<2>(a, cl, ch, rl, rh, n);
    keccak_row<3>(a, cl, ch, rl, rh, n); keccak_row<4>(a, cl, ch, rl, rh, n);
    n.lo[0] ^= rc_lo;
    n.hi[0] ^= rc_hi;
    a = n;
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
    for (int round = 1; round < 23; round++) {
        if (VK_KECCAK_PIN == 2) keccak_round_rows(a, KECCAK_RC_LO[round], KECCAK_RC_HI[round]);
        else keccak_round(a, KECCAK_RC_LO[round], KECCAK_RC_HI[round]);
    }
    if (DIGEST_ONLY) keccak_last_round_digest(a, 0x80008008u, 0x80000000u);
    else keccak_round(a, 0x80008008u, 0x80000000u);
}

// 32-bit word k of the rate (k < 34): even words are the low halves of lane k/2
__device__ __forceinline__ void absorb_word(KState& a, int k, uint32_t w) { if (k & 1) a.hi[k >> 1] ^= w; else a.lo[k >> 1] ^= w; }
__device__ __forceinline__ void kstate_zero(KState& a) {
#pragma unroll
    for (int i = 0; i < 25; i++) { a.lo[i] = 0; a.hi[i] = 0; }
}

}  // namespace vk
