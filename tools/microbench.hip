// Micro-benchmarks of the integer primitives the prover's kernels are built from (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I valida_amd/csrc tools/microbench.hip -o build/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "field.hpp"
using vg::Fp;

template <int MODE> __global__ void k_chain(uint32_t* out, uint32_t seed, int iters) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp a[4], m = Fp::raw(seed | 1);
    for (int i = 0; i < 4; i++) a[i] = Fp::raw((t * 2654435761u + i * 977u) % vg::P);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (MODE == 0) a[i] = a[i] * m;                       // Montgomery product
            if (MODE == 1) a[i] = a[i] + m;                       // modular add
            if (MODE == 2) a[i] = Fp::raw(a[i].v * m.v + 12345u);  // raw v_mul_lo + add
            if (MODE == 3) a[i] = Fp::raw(__umulhi(a[i].v, m.v) + a[i].v);  // raw v_mul_hi
            if (MODE == 4) { uint64_t p = (uint64_t)a[i].v * m.v; a[i] = Fp::raw((uint32_t)p ^ (uint32_t)(p >> 32)); }  // v_mad_u64_u32
            if (MODE == 5) a[i] = Fp::raw(__builtin_amdgcn_alignbit(a[i].v, m.v, 7) ^ a[i].v);  // alignbit + xor
        }
    }
    out[t] = a[0].v ^ a[1].v ^ a[2].v ^ a[3].v;
}

template <int MODE> double run(const char* name, int ops_per_iter) {
    const int blocks = 256 * 8, threads = 256, iters = 4096;
    uint32_t* d;
    hipMalloc(&d, blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 12345u, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 12345u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * threads * iters * 4;
    printf("%-28s %8.3f ms  %8.2f Gop/s  (%.2f cycles per wave-op per SIMD at 2.1 GHz)\n", name, ms, ops / ms / 1e6,
           1024.0 * 2.1e9 / (ops / 64 / (ms * 1e-3)));
    hipFree(d);
    return ms;
}
int main() {
    run<0>("montgomery mul", 1);
    run<1>("modular add", 1);
    run<2>("v_mul_lo_u32 (+add)", 1);
    run<3>("v_mul_hi_u32 (+add)", 1);
    run<4>("v_mad_u64_u32 (+xor)", 1);
    run<5>("v_alignbit + xor", 1);
    return 0;
}
