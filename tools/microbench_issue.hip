// Follow-up issue experiments (gfx950): why does the in-register Keccak-f round cost 650 SIMD-cycles when its instructions
// sum to ~515 at their isolated rates?  Explicit-register asm loops:
//   bank_same / bank_diff : v_bitop3_b32 whose three sources sit in the same / in different VGPR banks (reg index mod 4)
//   mix21                 : bitop3, bitop3, alignbit repeating over independent chains (the Keccak ratio)
//   dep_pairs             : bitop3 feeding the next bitop3 of the same chain immediately (xor3(xor3(..)) of the column parity)
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_issue.hip -o build/microbench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "kernels/keccak.hpp"  // -I valida_amd/csrc
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(2); } } while (0)

#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"
#define INIT "v_mov_b32 v32, %1\n v_mov_b32 v33, %1\n v_mov_b32 v34, %1\n v_mov_b32 v35, %1\n v_mov_b32 v36, %1\n v_mov_b32 v37, %1\n v_mov_b32 v38, %1\n v_mov_b32 v39, %1\n" \
             "v_mov_b32 v40, %1\n v_mov_b32 v41, %1\n v_mov_b32 v42, %1\n v_mov_b32 v43, %1\n v_mov_b32 v44, %1\n v_mov_b32 v45, %1\n v_mov_b32 v46, %1\n v_mov_b32 v47, %1\n" \
             "v_mov_b32 v48, %1\n v_mov_b32 v49, %1\n v_mov_b32 v50, %1\n v_mov_b32 v51, %1\n v_mov_b32 v52, %1\n v_mov_b32 v53, %1\n v_mov_b32 v54, %1\n v_mov_b32 v55, %1\n" \
             "v_mov_b32 v56, %1\n v_mov_b32 v57, %1\n v_mov_b32 v58, %1\n v_mov_b32 v59, %1\n v_mov_b32 v60, %1\n v_mov_b32 v61, %1\n v_mov_b32 v62, %1\n v_mov_b32 v63, %1\n"
#define FINI "v_xor_b32 %0, v32, v33\n v_xor_b32 %0, %0, v34\n v_xor_b32 %0, %0, v35\n v_xor_b32 %0, %0, v36\n v_xor_b32 %0, %0, v37\n v_xor_b32 %0, %0, v38\n v_xor_b32 %0, %0, v39\n"

// 8 destination chains v32..v39; sources from v40.. ; X(d, a, b, c)
#define B3(d, a, b, c) "v_bitop3_b32 v" #d ", v" #a ", v" #b ", v" #c " bitop3:0x96\n"
#define AL(d, a, b) "v_alignbit_b32 v" #d ", v" #a ", v" #b ", 7\n"
#define XR(d, a, b) "v_xor_b32 v" #d ", v" #a ", v" #b "\n"

// same bank: d = a, sources a, a+4k
#define SAME8 B3(32,32,40,48) B3(33,33,41,49) B3(34,34,42,50) B3(35,35,43,51) B3(36,36,44,52) B3(37,37,45,53) B3(38,38,46,54) B3(39,39,47,55)
// different banks: a, a+1+4k, a+2+4k
#define DIFF8 B3(32,32,41,50) B3(33,33,42,51) B3(34,34,43,48) B3(35,35,40,49) B3(36,36,45,54) B3(37,37,46,55) B3(38,38,47,52) B3(39,39,44,53)
// two sources only (v_xor): same / different bank
#define XSAME8 XR(32,32,40) XR(33,33,41) XR(34,34,42) XR(35,35,43) XR(36,36,44) XR(37,37,45) XR(38,38,46) XR(39,39,47)
#define XDIFF8 XR(32,32,41) XR(33,33,42) XR(34,34,43) XR(35,35,40) XR(36,36,45) XR(37,37,46) XR(38,38,47) XR(39,39,44)
// Keccak ratio, independent chains: 16 bitop3 + 8 alignbit
#define MIX24 B3(32,32,41,50) B3(33,33,42,51) AL(56,56,45) B3(34,34,43,48) B3(35,35,40,49) AL(57,57,46) B3(36,36,45,54) B3(37,37,46,55) AL(58,58,47) B3(38,38,47,52) B3(39,39,44,53) AL(59,59,44) \
              B3(32,32,41,50) B3(33,33,42,51) AL(60,60,45) B3(34,34,43,48) B3(35,35,40,49) AL(61,61,46) B3(36,36,45,54) B3(37,37,46,55) AL(62,62,47) B3(38,38,47,52) B3(39,39,44,53) AL(63,63,44)
// the same with the alignbits grouped (8 in a row after 16 bitop3)
#define GRP24 DIFF8 DIFF8 AL(56,56,45) AL(57,57,46) AL(58,58,47) AL(59,59,44) AL(60,60,45) AL(61,61,46) AL(62,62,47) AL(63,63,44)
// dependent pairs: chain i issues twice back to back
#define DEP8 B3(32,32,41,50) B3(32,32,42,51) B3(33,33,43,48) B3(33,33,40,49) B3(34,34,45,54) B3(34,34,46,55) B3(35,35,47,52) B3(35,35,44,53)
// alignbit consuming the bitop3 just issued (theta -> rho)
#define DEPROT8 B3(32,32,41,50) AL(56,32,32) B3(33,33,43,48) AL(57,33,33) B3(34,34,45,54) AL(58,34,34) B3(35,35,47,52) AL(59,35,35)

#define KERNEL(NAME, BODY, N)                                                                                                         \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed, int iters) {                                               \
        uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, x, s = seed + t;                                                              \
        asm volatile(INIT "s_mov_b32 s20, %2\n 1:\n" BODY "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" FINI       \
                     : "=v"(x) : "v"(s), "s"(iters) : CLOB, "s20", "scc");                                                                \
        out[t] = x;                                                                                                                       \
    }                                                                                                                                     \
    static const int NAME##_n = N;

KERNEL(k_bank_same, SAME8 SAME8 SAME8 SAME8 SAME8 SAME8 SAME8 SAME8, 64)
KERNEL(k_bank_diff, DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8, 64)
KERNEL(k_xor_same, XSAME8 XSAME8 XSAME8 XSAME8 XSAME8 XSAME8 XSAME8 XSAME8, 64)
KERNEL(k_xor_diff, XDIFF8 XDIFF8 XDIFF8 XDIFF8 XDIFF8 XDIFF8 XDIFF8 XDIFF8, 64)
KERNEL(k_mix21, MIX24 MIX24 MIX24, 72)
KERNEL(k_grp21, GRP24 GRP24 GRP24, 72)
KERNEL(k_dep_pairs, DEP8 DEP8 DEP8 DEP8 DEP8 DEP8 DEP8 DEP8, 64)
KERNEL(k_dep_rot, DEPROT8 DEPROT8 DEPROT8 DEPROT8 DEPROT8 DEPROT8 DEPROT8 DEPROT8, 64)

// group sizes: G bitop3 then G/2 alignbit
#define A8 AL(56,56,45) AL(57,57,46) AL(58,58,47) AL(59,59,44) AL(60,60,45) AL(61,61,46) AL(62,62,47) AL(63,63,44)
KERNEL(k_grp32, DIFF8 DIFF8 DIFF8 DIFF8 A8 A8, 48)
KERNEL(k_grp64, DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 A8 A8 A8 A8, 96)
KERNEL(k_grp128, DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 A8 A8 A8 A8 A8 A8 A8 A8, 192)
// odd waves issue only alignbit (64 per iteration), even waves only bitop3 (128 per iteration): the same 2:1 mix, never inside a wave
__global__ void __launch_bounds__(256) k_split(uint32_t* out, uint32_t seed, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, x, s = seed + t;
    if ((threadIdx.x >> 6) & 1) {
        asm volatile(INIT "s_mov_b32 s20, %2\n 1:\n" A8 A8 A8 A8 A8 A8 A8 A8 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" FINI
                     : "=v"(x) : "v"(s), "s"(iters) : CLOB, "s20", "scc");
    } else {
        asm volatile(INIT "s_mov_b32 s20, %2\n 1:\n" DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8
                     "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" FINI
                     : "=v"(x) : "v"(s), "s"(iters) : CLOB, "s20", "scc");
    }
    out[t] = x;
}
static const int k_split_n = 96;
// roles by the hardware wave slot (HW_REG_HW_ID bits 3:0 = wave slot, 5:4 = SIMD): mode 0 = every wave bitop3 (128 / iteration), 1 = every wave alignbit
// (64 / iteration), 2 = odd slots alignbit, even slots bitop3 -- so both classes are resident on EVERY SIMD
__global__ void __launch_bounds__(256) k_roles(uint32_t* out, uint32_t seed, int iters, int mode, uint32_t* hwid_hist) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, x, s = seed + t, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0 && hwid_hist) atomicAdd(&hwid_hist[hw & 63], 1u);
    const bool rot = mode == 1 || (mode == 2 && (hw & 1));
    if (rot) {
        asm volatile(INIT "s_mov_b32 s20, %2\n 1:\n" A8 A8 A8 A8 A8 A8 A8 A8 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" FINI
                     : "=v"(x) : "v"(s), "s"(iters) : CLOB, "s20", "scc");
    } else {
        asm volatile(INIT "s_mov_b32 s20, %2\n 1:\n" DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8 DIFF8
                     "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" FINI
                     : "=v"(x) : "v"(s), "s"(iters) : CLOB, "s20", "scc");
    }
    out[t] = x;
}
static void roles(int cu, uint32_t* d) {
    uint32_t* hist;
    CHECK(hipMalloc(&hist, 64 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {2, 4, 8}) {
        float ms[3];
        uint32_t h[64];
        for (int mode = 0; mode < 3; mode++) {
            hipLaunchKernelGGL(k_roles, dim3(cu * wps), dim3(256), 0, 0, d, 1u, 8, mode, (uint32_t*)nullptr);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemset(hist, 0, 256));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_roles, dim3(cu * wps), dim3(256), 0, 0, d, 1u, 4096, mode, hist);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms[mode], e0, e1));
            if (mode == 2) CHECK(hipMemcpy(h, hist, 256, hipMemcpyDeviceToHost));
        }
        printf("roles, %d waves/SIMD: all-bitop3 %.3f ms, all-alignbit %.3f ms, split by slot parity %.3f ms  (additive model %.3f, full overlap %.3f)\n", wps, ms[0], ms[1], ms[2],
               (ms[0] + ms[1]) / 2, (ms[0] > ms[1] ? ms[0] : ms[1]) / 2);
        printf("   waves per (simd, slot) over the chip:");
        for (int sd = 0; sd < 4; sd++) { printf("  simd%d:", sd); for (int w = 0; w < 16; w++) if (h[sd * 16 + w]) printf(" %u", h[sd * 16 + w]); }
        printf("\n");
    }
    CHECK(hipFree(hist));
}  // average per wave per iteration: (128 + 64) / 2

// rotation with an s_nop 0 behind every v_alignbit_b32 (tools/issue_patterns.py: a half-rate instruction followed by another VALU
// instruction of the same wave costs the SIMD a full 4-cycle slot for BOTH; one scalar no-op behind it restores the additive cost)
template <int S> __device__ __forceinline__ uint32_t alignbit_nop(uint32_t hi, uint32_t lo) {
    uint32_t o;
    asm("v_alignbit_b32 %0, %1, %2, %3\n\ts_nop 0" : "=v"(o) : "v"(hi), "v"(lo), "n"(S));
    return o;
}
template <int N> __device__ __forceinline__ void rotl_pair_nop(uint32_t lo, uint32_t hi, uint32_t& olo, uint32_t& ohi) {
    if (N == 0) { olo = lo; ohi = hi; }
    else if (N == 32) { olo = hi; ohi = lo; }
    else if (N < 32) { ohi = alignbit_nop<32 - N>(hi, lo); olo = alignbit_nop<32 - N>(lo, hi); }
    else { ohi = alignbit_nop<64 - N>(lo, hi); olo = alignbit_nop<64 - N>(hi, lo); }
}
// ---- Keccak-f round orderings (vk:: helpers of kernels/keccak.hpp) ----
// ORDER 0: the product's round (theta-apply and rho interleaved lane by lane);  1: class-grouped in source order;
// 2: class-grouped with scheduling barriers between the groups
template <int ORDER> __device__ __forceinline__ void round_v(vk::KState& a, uint32_t rc_lo, uint32_t rc_hi) {
    if (ORDER == 0) { vk::keccak_round(a, rc_lo, rc_hi); return; }
    uint32_t cl[5], ch[5], rl[5], rh[5];
    if (ORDER == 3) {  // the product's source order, rotations with the trailing s_nop
#pragma unroll
        for (int x = 0; x < 5; x++) {
            cl[x] = vk::xor3(vk::xor3(a.lo[x], a.lo[x + 5], a.lo[x + 10]), a.lo[x + 15], a.lo[x + 20]);
            ch[x] = vk::xor3(vk::xor3(a.hi[x], a.hi[x + 5], a.hi[x + 10]), a.hi[x + 15], a.hi[x + 20]);
        }
#pragma unroll
        for (int x = 0; x < 5; x++) rotl_pair_nop<1>(cl[x], ch[x], rl[x], rh[x]);
        vk::KState b;
#define TRP(X, Y) { const uint32_t tl = vk::xor3(a.lo[X + 5 * Y], cl[(X + 4) % 5], rl[(X + 1) % 5]), th = vk::xor3(a.hi[X + 5 * Y], ch[(X + 4) % 5], rh[(X + 1) % 5]); \
                    rotl_pair_nop<vk::keccak_rot(X + 5 * Y)>(tl, th, b.lo[Y + 5 * ((2 * X + 3 * Y) % 5)], b.hi[Y + 5 * ((2 * X + 3 * Y) % 5)]); }
#define TRPX(X) TRP(X, 0) TRP(X, 1) TRP(X, 2) TRP(X, 3) TRP(X, 4)
        TRPX(0) TRPX(1) TRPX(2) TRPX(3) TRPX(4)
#pragma unroll
        for (int y = 0; y < 5; y++)
#pragma unroll
            for (int x = 0; x < 5; x++) {
                a.lo[x + 5 * y] = vk::chi32(b.lo[x + 5 * y], b.lo[(x + 1) % 5 + 5 * y], b.lo[(x + 2) % 5 + 5 * y]);
                a.hi[x + 5 * y] = vk::chi32(b.hi[x + 5 * y], b.hi[(x + 1) % 5 + 5 * y], b.hi[(x + 2) % 5 + 5 * y]);
            }
        a.lo[0] ^= rc_lo;
        a.hi[0] ^= rc_hi;
        return;
    }
#pragma unroll
    for (int x = 0; x < 5; x++) {
        cl[x] = vk::xor3(vk::xor3(a.lo[x], a.lo[x + 5], a.lo[x + 10]), a.lo[x + 15], a.lo[x + 20]);
        ch[x] = vk::xor3(vk::xor3(a.hi[x], a.hi[x + 5], a.hi[x + 10]), a.hi[x + 15], a.hi[x + 20]);
    }
    if (ORDER == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int x = 0; x < 5; x++) vk::rotl_pair<1>(cl[x], ch[x], rl[x], rh[x]);
    if (ORDER == 2) __builtin_amdgcn_sched_barrier(0);
    vk::KState t, b;
#pragma unroll
    for (int i = 0; i < 25; i++) {
        const int x = i % 5;
        t.lo[i] = vk::xor3(a.lo[i], cl[(x + 4) % 5], rl[(x + 1) % 5]);
        t.hi[i] = vk::xor3(a.hi[i], ch[(x + 4) % 5], rh[(x + 1) % 5]);
    }
    if (ORDER == 2) __builtin_amdgcn_sched_barrier(0);
#define ROT(X, Y) vk::rotl_pair<vk::keccak_rot(X + 5 * Y)>(t.lo[X + 5 * Y], t.hi[X + 5 * Y], b.lo[Y + 5 * ((2 * X + 3 * Y) % 5)], b.hi[Y + 5 * ((2 * X + 3 * Y) % 5)]);
#define ROTY(Y) ROT(0, Y) ROT(1, Y) ROT(2, Y) ROT(3, Y) ROT(4, Y)
    ROTY(0) ROTY(1) ROTY(2) ROTY(3) ROTY(4)
    if (ORDER == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int y = 0; y < 5; y++)
#pragma unroll
        for (int x = 0; x < 5; x++) {
            a.lo[x + 5 * y] = vk::chi32(b.lo[x + 5 * y], b.lo[(x + 1) % 5 + 5 * y], b.lo[(x + 2) % 5 + 5 * y]);
            a.hi[x + 5 * y] = vk::chi32(b.hi[x + 5 * y], b.hi[(x + 1) % 5 + 5 * y], b.hi[(x + 2) % 5 + 5 * y]);
        }
    a.lo[0] ^= rc_lo;
    a.hi[0] ^= rc_hi;
    if (ORDER == 2) __builtin_amdgcn_sched_barrier(0);
}
template <int ORDER> __global__ void __launch_bounds__(256) k_keccak_order(uint32_t* out, uint32_t seed, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    vk::KState a;
    for (int i = 0; i < 25; i++) { a.lo[i] = tid * 2654435761u + i * seed; a.hi[i] = tid * 40503u + i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll 2
        for (int round = 0; round < 24; round++) round_v<ORDER>(a, vk::KECCAK_RC_LO[round], vk::KECCAK_RC_HI[round]);
    }
    uint32_t x = 0;
    for (int i = 0; i < 25; i++) x ^= a.lo[i] ^ a.hi[i];
    out[tid] = x;
}

template <class K> static void row(const char* name, K kernel, int per_iter, int cu, uint32_t* d) {
    printf("%-12s", name);
    const int iters = per_iter > 1000 ? 128 : 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
        const int blocks = cu * wps;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1u, 8);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double wave_instr = (double)blocks * 4 * iters * per_iter;
        // SIMD-cycles per instruction at a nominal 2.35 GHz
        printf("  %7.1f G/s (%.2f cyc)", wave_instr / (best * 1e-3) / 1e9, cu * 4.0 * 2.35e9 / (wave_instr / (best * 1e-3)));
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cu = prop.multiProcessorCount;
    uint32_t* d;
    CHECK(hipMalloc(&d, (size_t)cu * 8 * 256 * 4));
    printf("G wave64-instr/s (SIMD-cycles per instruction at a nominal 2.35 GHz) at 1 / 2 / 3 / 4 / 5 / 6 / 8 waves per SIMD\n");
    row("bitop3 same", k_bank_same, k_bank_same_n, cu, d);
    row("bitop3 diff", k_bank_diff, k_bank_diff_n, cu, d);
    row("xor same", k_xor_same, k_xor_same_n, cu, d);
    row("xor diff", k_xor_diff, k_xor_diff_n, cu, d);
    row("mix 2:1", k_mix21, k_mix21_n, cu, d);
    row("grouped 2:1", k_grp21, k_grp21_n, cu, d);
    row("dep pairs", k_dep_pairs, k_dep_pairs_n, cu, d);
    row("dep b3->rot", k_dep_rot, k_dep_rot_n, cu, d);
    row("grp 32:16", k_grp32, k_grp32_n, cu, d);
    row("grp 64:32", k_grp64, k_grp64_n, cu, d);
    row("grp 128:64", k_grp128, k_grp128_n, cu, d);
    row("wave split", k_split, k_split_n, cu, d);
    roles(cu, d);
    printf("Keccak-f[1600] round orderings, 24 x 178 instructions per permutation (cycles per INSTRUCTION; x178 = per round)\n");
    row("keccak ord0", k_keccak_order<0>, 24 * 178, cu, d);
    row("keccak ord1", k_keccak_order<1>, 24 * 178, cu, d);
    row("keccak ord2", k_keccak_order<2>, 24 * 178, cu, d);
    row("keccak nop", k_keccak_order<3>, 24 * 178, cu, d);
    return 0;
}
