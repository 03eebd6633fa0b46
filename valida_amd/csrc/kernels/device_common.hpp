// Shared device-side definitions for the gfx950 kernels.
//
// HBM data layout (DESIGN.md "Data layout"):
//   * base-field matrices are COLUMN-MAJOR: column c is one contiguous u32[height] at data + c*stride,
//     elements in Montgomery form.  A wave reading 64 consecutive rows of a column is one coalesced
//     256-byte request.
//   * Ext5 matrices/vectors are stored as 5 consecutive base columns (exactly flatten_to_base,
//     basic/src/lib.rs:254-257).
//   * committed LDEs are stored in BIT-REVERSED row order (as Plonky3 commits them):
//     storage row j holds the evaluation at x = shift * w_N^{bitrev(j)}.
//   * digests are 8 canonical u32 (32 bytes), array-of-structs, one layer per array.
#pragma once
#include <hip/hip_runtime.h>
#include "../field.hpp"

namespace vk {
using vg::Ext5;
using vg::Fp;

struct DMatView {
    uint32_t* data;
    uint64_t height, width, stride;  // stride = elements between columns (>= height)
    __device__ __forceinline__ Fp get(uint64_t r, uint64_t c) const { return Fp::raw(data[c * stride + r]); }
    __device__ __forceinline__ void set(uint64_t r, uint64_t c, Fp v) const { data[c * stride + r] = v.v; }
    __device__ __forceinline__ uint32_t* col(uint64_t c) const { return data + c * stride; }
};

// Per-context constant tables (filled once by DeviceTables::init):
//   roots[i]     = primitive 2^i-th root of unity w_{2^i} (Montgomery), i = 0..27
//   inv_roots[i] = its inverse
// x(j) := w_N^{bitrev_N(j)} = prod_{bit b of j} roots[b+1] is independent of N; brt_* factor it into
// three table lookups (bits 0-10, 11-21, 22-26) so a domain point costs two multiplications.
struct DeviceTables {
    uint32_t roots[28];
    uint32_t inv_roots[28];
    const uint32_t* brt_lo;   // [2048]  prod over bits 0..10
    const uint32_t* brt_hi;   // [2048]  prod over bits 11..21
    const uint32_t* brt_top;  // [32]    prod over bits 22..26
    const uint32_t* ibrt_lo;  // same for inverse roots
    const uint32_t* ibrt_hi;
    const uint32_t* ibrt_top;
    const uint32_t* twc;      // [16383] compact per-stage tables: twc[2^(s-1) - 1 + j] = w_{2^s}^j, j < 2^(s-1), s = 1..14
    const uint32_t* itwc;     // same with inverse roots
};

__device__ __forceinline__ Fp domain_point(const DeviceTables& t, uint32_t j) {  // w^{bitrev(j)}
    Fp r = Fp::raw(t.brt_lo[j & 2047]);
    if (j >> 11) r *= Fp::raw(t.brt_hi[(j >> 11) & 2047]);
    if (j >> 22) r *= Fp::raw(t.brt_top[j >> 22]);
    return r;
}
__device__ __forceinline__ Fp inv_domain_point(const DeviceTables& t, uint32_t j) {  // w^{-bitrev(j)}
    Fp r = Fp::raw(t.ibrt_lo[j & 2047]);
    if (j >> 11) r *= Fp::raw(t.ibrt_hi[(j >> 11) & 2047]);
    if (j >> 22) r *= Fp::raw(t.ibrt_top[j >> 22]);
    return r;
}

__device__ __forceinline__ Ext5 load_ext(const uint32_t* base, uint64_t stride, uint64_t r) {
    Ext5 e;
#pragma unroll
    for (int k = 0; k < 5; k++) e.c[k] = Fp::raw(base[k * stride + r]);
    return e;
}
__device__ __forceinline__ void store_ext(uint32_t* base, uint64_t stride, uint64_t r, const Ext5& e) {
#pragma unroll
    for (int k = 0; k < 5; k++) base[k * stride + r] = e.c[k].v;
}
__device__ __forceinline__ Ext5 ext_from_words(const uint32_t* w) {
    Ext5 e;
#pragma unroll
    for (int k = 0; k < 5; k++) e.c[k] = Fp::raw(w[k]);
    return e;
}

}  // namespace vk

#define VG_HIP_CHECK(expr)                                                                                     \
    do {                                                                                                       \
        hipError_t _e = (expr);                                                                                \
        if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e));     \
    } while (0)
