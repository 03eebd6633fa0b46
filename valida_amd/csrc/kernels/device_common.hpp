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

// Loads / stores at a WAVE-UNIFORM base plus a 32-bit per-lane BYTE offset, as raw buffer accesses: the base travels in four SGPRs (scalar ALU), the
// lane's offset is one VGPR that many accesses share — instead of a 64-bit per-lane address per access (two v_mad_u64_u32 / v_lshl_add_u64 each, and a
// VGPR pair held per access in flight).  A plain pointer sum does not get there: LLVM re-associates (base + offset) into a per-lane pointer first.
// The offset must stay below the resource's 2^31 bytes.  VGPU_QUOT_SADDR=0 (A/B builds) and the host pass / hipemu: plain pointer arithmetic.
#ifndef VGPU_QUOT_SADDR
#define VGPU_QUOT_SADDR 1
#endif
__device__ __forceinline__ uint32_t load_at(const uint32_t* ubase, uint32_t byte_off) {
#if VGPU_QUOT_SADDR && defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(ubase), 0, 0x7fffffff, 0x00020000);  // raw buffer, no swizzle, dword format
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0);
#else
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ubase) + byte_off);
#endif
}
__device__ __forceinline__ void store_at(uint32_t* ubase, uint32_t byte_off, uint32_t v) {
#if VGPU_QUOT_SADDR && defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32((int)v, r, (int)byte_off, 0, 0);
#else
    *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ubase) + byte_off) = v;
#endif
}

// The same with a wave-uniform BYTE offset beside the base (the buffer instruction's scalar offset operand: one SGPR per distinct offset, the four-SGPR
// resource shared by all accesses of a column)
__device__ __forceinline__ uint32_t load_at(const uint32_t* ubase, uint32_t byte_off, uint32_t uniform_off) {
#if VGPU_QUOT_SADDR && defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(ubase), 0, 0x7fffffff, 0x00020000);
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, (int)uniform_off, 0);
#else
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ubase) + byte_off + uniform_off);
#endif
}
__device__ __forceinline__ void store_at(uint32_t* ubase, uint32_t byte_off, uint32_t uniform_off, uint32_t v) {
#if VGPU_QUOT_SADDR && defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32((int)v, r, (int)byte_off, (int)uniform_off, 0);
#else
    *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(ubase) + byte_off + uniform_off) = v;
#endif
}

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

// Host-side image of the tables (one allocation of `device_tables_words()` words; bind_device_tables points the struct into it).
// Shared by DeviceCtx::init_tables and the host emulation of the kernels (tools/hipemu, tests/emu).
constexpr int DEVICE_TABLES_TWC = 1 << 14;  // compact tables cover stages 1..14 (contiguous NTT tiles of up to 2^14 points)
inline size_t device_tables_words() { return 2048 * 4 + 64 + 2 * DEVICE_TABLES_TWC; }
inline void build_device_tables(DeviceTables& tb, uint32_t* h) {
    Fp roots[28], inv_roots[28];
    for (int i = 0; i <= 27; i++) { roots[i] = vg::two_adic_generator(i); inv_roots[i] = roots[i].inv(); }
    for (int i = 0; i < 28; i++) { tb.roots[i] = roots[i].v; tb.inv_roots[i] = inv_roots[i].v; }
    auto fill = [&](uint32_t* dst, int count, int first_bit, const Fp* r) {
        for (int j = 0; j < count; j++) {
            Fp p = Fp::one();
            for (int b = 0; (j >> b) != 0; b++) if ((j >> b) & 1) p *= r[first_bit + b + 1];
            dst[j] = p.v;
        }
    };
    for (size_t i = 0; i < device_tables_words(); i++) h[i] = 0;
    fill(h, 2048, 0, roots);              // brt_lo
    fill(h + 2048, 2048, 11, roots);      // brt_hi
    fill(h + 4096, 32, 22, roots);        // brt_top (bits 22..26 -> roots up to index 27)
    fill(h + 4096 + 32, 2048, 0, inv_roots);
    fill(h + 4096 + 32 + 2048, 2048, 11, inv_roots);
    fill(h + 4096 + 32 + 4096, 32, 22, inv_roots);
    uint32_t* twc = h + 4096 + 32 + 4096 + 32;
    // compact per-stage tables: stage s (1..14) at offset 2^(s-1) - 1 holds w_{2^s}^j, j < 2^(s-1)
    for (int s = 1; s <= 14; s++) {
        Fp ws = roots[s], wsi = inv_roots[s], a = Fp::one(), b = Fp::one();
        int off = (1 << (s - 1)) - 1;
        for (int j = 0; j < (1 << (s - 1)); j++) { twc[off + j] = a.v; twc[DEVICE_TABLES_TWC + off + j] = b.v; a *= ws; b *= wsi; }
    }
}
inline void bind_device_tables(DeviceTables& tb, const uint32_t* base) {
    tb.brt_lo = base;
    tb.brt_hi = base + 2048;
    tb.brt_top = base + 4096;
    tb.ibrt_lo = base + 4096 + 32;
    tb.ibrt_hi = base + 4096 + 32 + 2048;
    tb.ibrt_top = base + 4096 + 32 + 4096;
    tb.twc = base + 4096 + 32 + 4096 + 32;
    tb.itwc = tb.twc + DEVICE_TABLES_TWC;
}

// Per-(height, blowup, shift) tables of the fused LDE (ntt.hip, k_lde_mid): for coset t < b = 2^log_blowup, sigma_t = shift w_{bN}^t,
//   fac[t n_lo + i] = (sigma_t^n_hi)^bitrev_{k_lo}(i) / N     scale of the coefficient at tile position i before the forward transform
//   sig[t n_hi + a] = sigma_t^a                                the block's share of the coset powers
// words = b (n_lo + n_hi); built on the host (build_lde_tables), cached per context.
struct LdeTables { const uint32_t* fac; const uint32_t* sig; };
inline int lde_k_lo(int k) { return k < 12 ? k : (k >= 23 ? 14 : 12); }
inline size_t lde_tables_words(int k, int log_blowup) { const int k_lo = lde_k_lo(k); return ((size_t)1 << log_blowup) * (((size_t)1 << k_lo) + ((size_t)1 << (k - k_lo))); }
inline void build_lde_tables(int k, int log_blowup, Fp shift, uint32_t* h) {
    const int k_lo = lde_k_lo(k), k_hi = k - k_lo;
    const size_t n_lo = (size_t)1 << k_lo, n_hi = (size_t)1 << k_hi, b = (size_t)1 << log_blowup;
    const Fp w = vg::two_adic_generator((unsigned)(k + log_blowup)), ninv = Fp::from_canonical((uint32_t)(((uint64_t)1 << k) % vg::P)).inv();
    uint32_t* fac = h;
    uint32_t* sig = h + b * n_lo;
    Fp sigma = shift;
    for (size_t t = 0; t < b; t++) {
        const Fp g = sigma.exp_power_of_2((unsigned)k_hi);
        Fp pw = ninv;  // g^j / N
        for (size_t j = 0; j < n_lo; j++) { fac[t * n_lo + (k_lo ? vg::reverse_bits_len((uint32_t)j, (unsigned)k_lo) : 0)] = pw.v; pw *= g; }
        Fp a = Fp::one();
        for (size_t c = 0; c < n_hi; c++) { sig[t * n_hi + c] = a.v; a *= sigma; }
        sigma *= w;
    }
}

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
