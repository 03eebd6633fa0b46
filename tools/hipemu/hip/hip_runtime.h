// hipemu: a host stand-in for the HIP execution model, enough to run this repository's plain-C++ kernels (no wave intrinsics, no inline
// asm) on a CPU: every workgroup runs as `blockDim.x` cooperative FIBERS (ucontext) on one OS thread, __syncthreads() yields to the next
// fiber and returns when every fiber of the block has arrived (or finished), workgroups run one after the other.  Dynamic LDS is one
// global array (stale between workgroups, as on the device).  What it checks: the INDEX ARITHMETIC and the barrier structure of a kernel
// (a read of data another thread writes later in the same phase sees stale data, since fiber t runs its whole phase before fiber t + 1) —
// before GPU minutes are spent on it.  What it cannot check: performance, wave-level primitives, real data races.
// Test infrastructure (tools/ + tests/); nothing in the product includes it.
#pragma once
#include <ucontext.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#ifdef HIPEMU_STATIC_SHARED  // kernels with STATIC __shared__ arrays (perm.hip): one function-local static, shared by the workgroup's fibers (they run on one OS
#define __shared__ static    // thread); kernels with `extern __shared__` dynamic LDS (ntt.hip, ..) keep the default: the array resolves to a global the test defines
#else
#define __shared__
#endif
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotReady = 600 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t*) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }

namespace hipemu {
struct State {
    dim3 threadIdx, blockIdx, blockDim, gridDim;
};
inline State& st() { static State s; return s; }
constexpr size_t STACK = 256 * 1024;
struct Fiber { ucontext_t ctx; char* stack = nullptr; bool done = false; };
struct Sched {
    ucontext_t main;
    std::vector<Fiber> fib;
    int cur = -1;
    std::function<void()> body;
    unsigned long barriers = 0;
};
inline Sched& sched() { static Sched s; return s; }
inline void trampoline() {
    Sched& s = sched();
    s.body();
    s.fib[(size_t)s.cur].done = true;
    swapcontext(&s.fib[(size_t)s.cur].ctx, &s.main);
}
inline void syncthreads() {
    Sched& s = sched();
    s.barriers++;
    swapcontext(&s.fib[(size_t)s.cur].ctx, &s.main);
}
// one workgroup: phases until every fiber has finished
inline void run_block(unsigned nthreads) {
    Sched& s = sched();
    if (s.fib.size() < nthreads) {
        size_t old = s.fib.size();
        s.fib.resize(nthreads);
        for (size_t i = old; i < nthreads; i++) s.fib[i].stack = (char*)malloc(STACK);
    }
    for (unsigned t = 0; t < nthreads; t++) {
        Fiber& f = s.fib[t];
        f.done = false;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = &s.main;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    for (;;) {
        bool any = false;
        for (unsigned t = 0; t < nthreads; t++) {
            Fiber& f = s.fib[t];
            if (f.done) continue;
            any = true;
            s.cur = (int)t;
            st().threadIdx = dim3(t, 0, 0);
            swapcontext(&s.main, &f.ctx);
        }
        if (!any) break;
    }
}
template <class F>
inline void launch(dim3 grid, dim3 block, size_t lds_bytes, F&& f) {
    if (block.y != 1 || block.z != 1) throw std::runtime_error("hipemu: one-dimensional workgroups only");
    if (lds_bytes > 160 * 1024) throw std::runtime_error("hipemu: more than 160 KiB of LDS");
    Sched& s = sched();
    s.body = f;
    st().blockDim = block; st().gridDim = grid;
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++)
            for (unsigned x = 0; x < grid.x; x++) {
                st().blockIdx = dim3(x, y, z);
                run_block(block.x);
            }
}
}  // namespace hipemu

#define threadIdx (hipemu::st().threadIdx)
#define blockIdx (hipemu::st().blockIdx)
#define blockDim (hipemu::st().blockDim)
#define gridDim (hipemu::st().gridDim)
inline void __syncthreads() { hipemu::syncthreads(); }
inline unsigned __brev(unsigned x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
}
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
// wave-level primitives cannot be emulated by fibers: code that reaches one under hipemu is a test error (the kernels that use them — the
// device challenger step behind the tree-top kernels — take that branch only when asked to)
inline int __builtin_amdgcn_readlane(int, int) { throw std::runtime_error("hipemu: wave intrinsic (v_readlane) reached"); }
inline int __builtin_amdgcn_update_dpp(int, int, int, int, int, bool) { throw std::runtime_error("hipemu: wave intrinsic (DPP) reached"); }
inline int __shfl(int, int, int = 64) { throw std::runtime_error("hipemu: wave intrinsic (__shfl) reached"); }
// per-lane integer primitives of gfx950 that ARE plain functions of their operands
inline unsigned __builtin_amdgcn_bitop3_b32(unsigned a, unsigned b, unsigned c, unsigned truth_table) {  // bit i of the result = table[(a_i, b_i, c_i)]
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((truth_table >> ((((a >> i) & 1u) << 2) | (((b >> i) & 1u) << 1) | ((c >> i) & 1u))) & 1u) << i;
    return r;
}
inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned shift) { return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (shift & 31u)); }
inline void __builtin_amdgcn_sched_barrier(int) {}
#define __constant__
struct uint4 { unsigned x, y, z, w; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline void __threadfence_block() {}
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) hipemu::launch(grid, block, lds, [=]() { kernel(__VA_ARGS__); })
#define hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, e0, e1, flags, ...) hipemu::launch(grid, block, lds, [=]() { kernel(__VA_ARGS__); })
