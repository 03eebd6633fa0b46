// Field-arithmetic microbenchmark (gfx950): the modular add / sub / Montgomery product of valida_amd/csrc/field.hpp as compiled (unsigned-min
// reductions: v_min_u32 is a half-rate instruction) against a carry-out + v_cndmask form (three full-rate instructions), in long dependent
// chains per lane with 8 independent chains, 8 waves per SIMD.  Prints ns per (add + sub) pair and per product, per wave64 — the ratio decides
// whether field.hpp switches its reductions.   hipcc --offload-arch=gfx950 -O3 tools/microbench_fp.hip -o build/mb/microbench_fp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
constexpr uint32_t P = 0x78000001u;
__device__ __forceinline__ uint32_t add_min(uint32_t a, uint32_t b) { uint32_t s = a + b, t = s - P; return s < t ? s : t; }
__device__ __forceinline__ uint32_t sub_min(uint32_t a, uint32_t b) { uint32_t d = a - b, t = d + P; return d < t ? d : t; }
__device__ __forceinline__ uint32_t add_c(uint32_t a, uint32_t b) {
    uint32_t s = a + b, t, r; unsigned long long vc;
    asm("v_subrev_co_u32_e64 %0, %1, %4, %3\n\tv_cndmask_b32_e64 %2, %0, %3, %1" : "=&v"(t), "=&s"(vc), "=v"(r) : "v"(s), "v"(P));
    return r;
}
__device__ __forceinline__ uint32_t sub_c(uint32_t a, uint32_t b) {
    uint32_t d, t, r; unsigned long long vc;
    asm("v_sub_co_u32_e64 %0, %1, %4, %5\n\tv_add_u32_e32 %2, %6, %0\n\tv_cndmask_b32_e64 %3, %0, %2, %1" : "=&v"(d), "=&s"(vc), "=&v"(t), "=v"(r) : "v"(a), "v"(b), "v"(P));
    return r;
}
// the same through VCC (VOP2 encodings, half the code bytes; sequences serialise on VCC — in-order issue makes that free within a wave)
__device__ __forceinline__ uint32_t add_v(uint32_t a, uint32_t b) {
    uint32_t s = a + b, t, r;
    asm("v_subrev_co_u32_e32 %0, vcc, %3, %2\n\tv_cndmask_b32_e32 %1, %0, %2, vcc" : "=&v"(t), "=v"(r) : "v"(s), "v"(P) : "vcc");
    return r;
}
__device__ __forceinline__ uint32_t sub_v(uint32_t a, uint32_t b) {
    uint32_t d, t, r;
    asm("v_sub_co_u32_e32 %0, vcc, %3, %4\n\tv_add_u32_e32 %1, %5, %0\n\tv_cndmask_b32_e32 %2, %0, %1, vcc" : "=&v"(d), "=&v"(t), "=v"(r) : "v"(a), "v"(b), "v"(P) : "vcc");
    return r;
}
__device__ __forceinline__ uint32_t mont_v(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b; uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    uint32_t m = lo * 0x88000001u, u = __umulhi(m, P);
    return sub_v(hi, u);
}
__device__ __forceinline__ uint32_t mont_min(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b; uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    uint32_t m = lo * 0x88000001u, u = __umulhi(m, P), r = hi - u, r2 = r + P; return r < r2 ? r : r2;
}
__device__ __forceinline__ uint32_t mont_c(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b; uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    uint32_t m = lo * 0x88000001u, u = __umulhi(m, P);
    return sub_c(hi, u);
}
// Shoup / Barrett product by a FIXED factor w with its precomputed quotient wq = floor(w 2^32 / p): q = mulhi(x, wq), r = x w - q p in [0, 2p), one
// correction — three multiply-class instructions instead of the Montgomery product's four (round 5: is it worth carrying twiddle quotients?)
__device__ __forceinline__ uint32_t red_c(uint32_t s) {
    uint32_t t, r; unsigned long long vc;
    asm("v_subrev_co_u32_e64 %0, %1, %4, %3\n\tv_cndmask_b32_e64 %2, %0, %3, %1" : "=&v"(t), "=&s"(vc), "=v"(r) : "v"(s), "v"(P));
    return r;
}
__device__ __forceinline__ uint32_t shoup_c(uint32_t x, uint32_t w, uint32_t wq) {
    const uint32_t q = __umulhi(x, wq);
    return red_c(x * w - q * P);
}
__device__ __forceinline__ uint32_t shoup_lazy(uint32_t x, uint32_t w, uint32_t wq) {  // result in [0, 2p)
    const uint32_t q = __umulhi(x, wq);
    return x * w - q * P;
}
template <int MODE> __global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
    uint32_t x[8], y = 123456789u + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = (threadIdx.x * 2654435761u + i * 40503u) % P;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) { x[i] = add_min(x[i], y); x[i] = sub_min(x[i], x[(i + 1) & 7]); }
            if (MODE == 1) { x[i] = add_c(x[i], y); x[i] = sub_c(x[i], x[(i + 1) & 7]); }
            if (MODE == 2) x[i] = mont_min(x[i], y);
            if (MODE == 3) x[i] = mont_c(x[i], y);
            if (MODE == 4) { x[i] = add_v(x[i], y); x[i] = sub_v(x[i], x[(i + 1) & 7]); }
            if (MODE == 5) x[i] = mont_v(x[i], y);
            if (MODE == 6) x[i] = shoup_c(x[i], y % P, (uint32_t)((((uint64_t)(y % P)) << 32) / P));  // (the factor and its quotient are loop-invariant: hoisted)
            if (MODE == 7) x[i] = red_c(shoup_lazy(x[i], y % P, (uint32_t)((((uint64_t)(y % P)) << 32) / P)) >> 1);  // lazy product, consumer halves the range (keeps the chain below 2^32)
            // DIF butterflies on pairs (x[i], x[i ^ 1]): Montgomery against Shoup twiddle products
            if (MODE == 8 && !(i & 1)) { uint32_t u = x[i], v = x[i + 1]; x[i] = add_c(u, v); x[i + 1] = mont_c(u + (P - v), y); }
            if (MODE == 9 && !(i & 1)) { uint32_t u = x[i], v = x[i + 1]; x[i] = add_c(u, v); x[i + 1] = shoup_c(u + (P - v), y % P, (uint32_t)((((uint64_t)(y % P)) << 32) / P)); }
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE> double run(uint32_t* d, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<2048, 256>>>(d, 16);
    hipDeviceSynchronize();
    hipEventRecord(a); k<MODE><<<2048, 256>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // 2048 blocks x 4 waves = 8192 waves on 1024 SIMDs = 8 per SIMD; ops per wave = iters * 8
    return ms * 1e6 / ((double)iters * 8.0) / 8.0;  // ns of SIMD time per op-group per wave
}
int main() {
    uint32_t* d; hipMalloc(&d, 2048 * 256 * 4);
    const int iters = 20000;
    uint32_t h0[4] = {0}, v[4];
    double t0 = run<0>(d, iters); hipMemcpy(&v[0], d, 4, hipMemcpyDeviceToHost);
    double t1 = run<1>(d, iters); hipMemcpy(&v[1], d, 4, hipMemcpyDeviceToHost);
    double t2 = run<2>(d, iters); hipMemcpy(&v[2], d, 4, hipMemcpyDeviceToHost);
    double t3 = run<3>(d, iters); hipMemcpy(&v[3], d, 4, hipMemcpyDeviceToHost);
    (void)h0;
    uint32_t v4, v5;
    double t4 = run<4>(d, iters); hipMemcpy(&v4, d, 4, hipMemcpyDeviceToHost);
    double t5 = run<5>(d, iters); hipMemcpy(&v5, d, 4, hipMemcpyDeviceToHost);
    double t6 = run<6>(d, iters), t7 = run<7>(d, iters), t8 = run<8>(d, iters), t9 = run<9>(d, iters);
    printf("product by a fixed factor: Montgomery (carry form) %.3f ns   Shoup + correction %.3f ns (ratio %.3f)   Shoup lazy + halving consumer %.3f ns\n", t3, t6, t6 / t3, t7);
    printf("DIF butterfly (4 per 8-slot group, so per butterfly x2): Montgomery %.3f ns   Shoup %.3f ns   (ratio %.3f)\n", 2 * t8, 2 * t9, t9 / t8);
    printf("through VCC (VOP2): add+sub pair %.3f ns   Montgomery mul %.3f ns   results %s\n", t4, t5, (v4 == v[0] && v5 == v[2]) ? "equal" : "DIFFER");
    printf("add+sub pair   min-form %.3f ns   carry-form %.3f ns   (ratio %.3f)  results %s\n", t0, t1, t1 / t0, v[0] == v[1] ? "equal" : "DIFFER");
    printf("Montgomery mul min-form %.3f ns   carry-form %.3f ns   (ratio %.3f)  results %s\n", t2, t3, t3 / t2, v[2] == v[3] ? "equal" : "DIFFER");
    return 0;
}
