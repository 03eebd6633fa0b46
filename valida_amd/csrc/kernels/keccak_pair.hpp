// Keccak-f[1600] with ONE PERMUTATION PER LANE PAIR: lane 2i holds the low 32-bit halves of the 25 state lanes, lane 2i+1 the high halves.
// For the latency-bound part of every Merkle tree (the layers with fewer rows than the GPU has wave slots, and the single-workgroup top):
// there a wave issues one VALU instruction per 4 cycles whatever it is (tools/microbench_issue.hip), so a permutation costs its
// instruction count — 24 x 178 per thread in keccak.hpp, 24 x 119 here: xor and chi split cleanly between the halves, only the rotations
// need the partner's half (one DPP quad_perm move each).  Both lanes of a pair run the same instruction stream: rotl64 by n < 32 is
// alignbit(own, partner, 32 - n) for EITHER half, by n > 32 alignbit(partner, own, 64 - n).
// Reference: the same Keccak256Hash as keccak.hpp (basic/tests/test_prover.rs:424-431).
#pragma once
#include "keccak.hpp"

namespace vk {

struct KHalf { uint32_t s[25]; };

// the other lane of the pair (quad_perm [1, 0, 3, 2])
__device__ __forceinline__ uint32_t pair_partner(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }

template <int N> __device__ __forceinline__ uint32_t rotl_half(uint32_t own) {
    if (N == 0) return own;
    const uint32_t other = pair_partner(own);
    if (N == 32) return other;
    if (N < 32) return __builtin_amdgcn_alignbit(own, other, 32 - N);
    return __builtin_amdgcn_alignbit(other, own, 64 - N);
}

template <int X, int Y> __device__ __forceinline__ void pair_theta_rho_pi(const KHalf& a, const uint32_t (&c)[5], const uint32_t (&r)[5], KHalf& b) {
    constexpr int src = X + 5 * Y, dst = Y + 5 * ((2 * X + 3 * Y) % 5);
    b.s[dst] = rotl_half<keccak_rot(src)>(xor3(a.s[src], c[(X + 4) % 5], r[(X + 1) % 5]));
}
template <int X> __device__ __forceinline__ void pair_theta_rho_pi_col(const KHalf& a, const uint32_t (&c)[5], const uint32_t (&r)[5], KHalf& b) {
    pair_theta_rho_pi<X, 0>(a, c, r, b); pair_theta_rho_pi<X, 1>(a, c, r, b); pair_theta_rho_pi<X, 2>(a, c, r, b);
    pair_theta_rho_pi<X, 3>(a, c, r, b); pair_theta_rho_pi<X, 4>(a, c, r, b);
}
__device__ __forceinline__ void pair_column_parity(const KHalf& a, uint32_t (&c)[5], uint32_t (&r)[5]) {
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = xor3(xor3(a.s[x], a.s[x + 5], a.s[x + 10]), a.s[x + 15], a.s[x + 20]);
#pragma unroll
    for (int x = 0; x < 5; x++) r[x] = rotl_half<1>(c[x]);
}
// rc = this half of the round constant
__device__ __forceinline__ void pair_round(KHalf& a, uint32_t rc) {
    uint32_t c[5], r[5];
    KHalf b;
    pair_column_parity(a, c, r);
    pair_theta_rho_pi_col<0>(a, c, r, b); pair_theta_rho_pi_col<1>(a, c, r, b); pair_theta_rho_pi_col<2>(a, c, r, b);
    pair_theta_rho_pi_col<3>(a, c, r, b); pair_theta_rho_pi_col<4>(a, c, r, b);
#pragma unroll
    for (int y = 0; y < 5; y++)
#pragma unroll
        for (int x = 0; x < 5; x++) a.s[x + 5 * y] = chi32(b.s[x + 5 * y], b.s[(x + 1) % 5 + 5 * y], b.s[(x + 2) % 5 + 5 * y]);
    a.s[0] ^= rc;
}
__device__ __forceinline__ void pair_last_round_digest(KHalf& a, uint32_t rc) {
    uint32_t c[5], r[5];
    KHalf b;
    pair_column_parity(a, c, r);
    pair_theta_rho_pi<0, 0>(a, c, r, b); pair_theta_rho_pi<1, 1>(a, c, r, b); pair_theta_rho_pi<2, 2>(a, c, r, b);
    pair_theta_rho_pi<3, 3>(a, c, r, b); pair_theta_rho_pi<4, 4>(a, c, r, b);
#pragma unroll
    for (int x = 0; x < 4; x++) a.s[x] = chi32(b.s[x], b.s[x + 1], b.s[(x + 2) % 5]);
    a.s[0] ^= rc;
}
// half = 0 (this lane holds the low halves) or 1.  Every lane of the wave that is active must have its partner active.
template <bool DIGEST_ONLY> __device__ __forceinline__ void keccak_f1600_pair(KHalf& a, int half) {
    const uint32_t* rc = half ? KECCAK_RC_HI : KECCAK_RC_LO;
#pragma unroll 2
    for (int round = 0; round < 23; round++) pair_round(a, rc[round]);
    if (DIGEST_ONLY) pair_last_round_digest(a, rc[23]);
    else pair_round(a, rc[23]);
}

}  // namespace vk
