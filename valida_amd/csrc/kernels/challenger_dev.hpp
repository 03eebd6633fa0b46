// The DuplexChallenger step of the FRI commit phase as a device function of ONE wave (challenger_dev.hpp): shared by k_fri_challenge (open.hip)
// and by the single-workgroup tree-top kernels, which run it as an epilogue on the root they have just written (merkle.hip, poseidon_mmcs.hip).
#pragma once
#include "device_common.hpp"

namespace vk {

// ---- Fiat-Shamir on the device for the FRI commit phase -----------------------------------------------------------
// One DuplexChallenger step per FRI layer (basic/src/lib.rs:611-619 -> TwoAdicFriPcs commit phase, App. B8/B10):
// observe the layer's 8-word root, sample beta.  Keeping this on the device removes the per-layer D2H + host round
// trip from a chain of 21 dependent layers: the whole commit phase is enqueued without a single synchronisation.
// One wave; lane i < 16 owns state[i], input[i] and output[i] of the sponge; the Poseidon-16 MDS layer (circulant,
// coefficients m[(j - i) & 15]) broadcasts the state with v_readlane and accumulates lazily (4 products per
// Montgomery reduction).  State block `ch` (u32 words): [0,16) state  [16,32) input  [32] n_in  [33,49) output  [49] n_out.
__device__ __forceinline__ Fp poseidon16_lanes(Fp st, const uint32_t* __restrict__ rc, const uint32_t (&m)[16], int lane) {
#pragma unroll 1
    for (int r = 0; r < 30; r++) {
        st += Fp::raw(rc[r * 16 + (lane & 15)]);
        const Fp x2 = st * st, x5 = x2 * x2 * st;
        if (r < 4 || r >= 26 || lane == 0) st = x5;
        Fp acc = Fp::zero();
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += 4) {
            uint64_t t = 0;
#pragma unroll
            for (int i = i0; i < i0 + 4; i++) t += (uint64_t)m[i] * (uint32_t)__builtin_amdgcn_readlane((int)st.v, i);
            acc += Fp::raw(vg::monty_reduce_wide(t));
        }
        st = acc;
    }
    return st;
}

// lane = 0..63 of the calling wave (all 64 lanes active)
__device__ __forceinline__ void fri_challenge_step(int lane, const uint32_t* __restrict__ pos, uint32_t* __restrict__ ch, const uint32_t* __restrict__ digest8,
                                                   uint32_t* __restrict__ beta5, uint32_t* __restrict__ commit8) {
    const int l16 = lane & 15;
    const uint32_t* rc = pos;
    uint32_t m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = pos[480 + ((l16 - i) & 15)];
    Fp state = Fp::raw(ch[l16]), in = Fp::raw(ch[16 + l16]), out = Fp::raw(ch[33 + l16]);
    uint32_t n_in = ch[32], n_out = ch[49];  // wave-uniform
    auto duplexing = [&]() {
        if ((uint32_t)lane < n_in) state = in;
        n_in = 0;
        state = poseidon16_lanes(state, rc, m, lane);
        out = state;
        n_out = 16;
    };
    // observe the commitment: 8 canonical words -> field elements (from_wrapped values are already < p)
    const uint32_t dword = digest8[lane & 7];
    const Fp dval = Fp::from_canonical(dword);
    for (int k = 0; k < 8; k++) {
        const Fp x = Fp::raw((uint32_t)__builtin_amdgcn_readlane((int)dval.v, k));
        n_out = 0;
        if ((uint32_t)lane == n_in) in = x;
        n_in++;
        if (n_in == 16) duplexing();
    }
    // sample_ext_element: five base samples, popped from the END of the output buffer
    uint32_t beta[5];
    for (int k = 0; k < 5; k++) {
        if (n_in > 0 || n_out == 0) duplexing();
        uint32_t v = out.v, pick = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) if ((uint32_t)i == n_out - 1) pick = (uint32_t)__builtin_amdgcn_readlane((int)v, i);
        beta[k] = pick;
        n_out--;
    }
    if (lane < 16) { ch[lane] = state.v; ch[16 + lane] = in.v; ch[33 + lane] = out.v; }
    if (lane == 0) { ch[32] = n_in; ch[49] = n_out; }
    if (lane < 5) beta5[lane] = lane == 0 ? beta[0] : lane == 1 ? beta[1] : lane == 2 ? beta[2] : lane == 3 ? beta[3] : beta[4];
    if (lane < 8) commit8[lane] = dword;
}

}  // namespace vk
