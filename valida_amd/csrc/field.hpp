// BabyBear (p = 2^31 - 2^27 + 1) in Montgomery form (R = 2^32) and its degree-5 binomial extension
// F[X]/(X^5 - 2), usable from host and gfx950 device code.  Replaces p3-baby-bear / p3-field's
// BabyBear and BinomialExtensionField<BabyBear, 5> (reference instantiation:
// basic/tests/test_prover.rs:413-416; constants: SURVEY.md §7.1 step 0).
//
// Device notes: a Montgomery product is 2x v_mul_lo_u32 + 2x v_mul_hi_u32 + 3 VALU; there is no MFMA
// path for 31-bit modular integer arithmetic.  Field elements cross the C ABI as canonical u32 < p and
// live in HBM in Montgomery form.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VG_HD __host__ __device__ __forceinline__
#else
#define VG_HD inline
#endif

namespace vg {

constexpr uint32_t P = 0x78000001u;       // 2013265921
constexpr uint32_t P_INV_NEG = 0x77ffffffu;  // -p^{-1} mod 2^32   (p^{-1} = 0x88000001)
constexpr uint32_t R_MOD_P = 268435454u;     // 2^32 mod p  (Montgomery form of 1)
constexpr uint32_t R2_MOD_P = 1172168163u;   // 2^64 mod p
constexpr uint32_t GENERATOR = 31;
constexpr uint32_t TWO_ADIC_ROOT_27 = 0x1a427a41u;  // canonical; 31^15

// VG_MULHI_NOP=1 (A/B builds): a scalar no-op behind the v_mul_hi_u32 that ends a Montgomery reduction's run of half-rate instructions — what
// keccak.hpp does behind every v_alignbit_b32 (a half-rate instruction directly followed by a full-rate one of the same wave costs both a slot)
#ifndef VG_MULHI_NOP
#define VG_MULHI_NOP 0  // 1: behind the v_mul_hi_u32, 2: also behind the v_mul_lo_u32.  Measured (profiles/r03_ab_nops.json): 1 = no change, 2 = 5 % slower, the Poseidon leg 2.5 % slower under both
#endif
VG_HD uint32_t mul_lo_pinv(uint32_t lo) {  // lo * p^{-1} mod 2^32
#if defined(__HIP_DEVICE_COMPILE__) && VG_MULHI_NOP == 2
    uint32_t o;
    asm("v_mul_lo_u32 %0, %1, %2\n\ts_nop 0" : "=v"(o) : "v"(lo), "v"(0x88000001u));
    return o;
#else
    return lo * 0x88000001u;
#endif
}
VG_HD uint32_t mul_hi_u32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && VG_MULHI_NOP
    uint32_t o;
    asm("v_mul_hi_u32 %0, %1, %2\n\ts_nop 0" : "=v"(o) : "v"(a), "v"(b));
    return o;
#elif defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// Montgomery reduction of a 64-bit value t < p * 2^32: returns t / 2^32 mod p, in [0, p).
// Subtractive form: mul_lo, mul_hi, sub, add, min after the product.  The additive form (m = t_lo * (-p^-1); (t + m p) >> 32 through ONE
// v_mad_u64_u32; conditional subtraction) is one instruction shorter and made the NTT / quotient kernels of a lone proof 10-20 %
// faster, but with three proofs in flight — the GPU full — the bench LOST 4.5 % (48.7 vs 51.0 proofs/s, A/B in one session,
// tools/gpu_ab.sh): twice as many 64-bit multiply-adds per product; not a power effect (2.31 GHz, 1.13 of 1.4 kW under the full bench); cause
// not established.  Measured, reverted.
// The three conditional corrections of the field arithmetic (a + b, a - b and the last step of a Montgomery reduction).  As plain C++ they
// compile to add / sub + v_min_u32, and v_min_u32 issues at HALF rate on gfx950 (profiles/r02_microbench.txt); on the device they are
// written as carry-out + v_cndmask_b32 instead — three full-rate instructions, the borrow travelling in an SGPR pair the register
// allocator picks (VOP3 encodings, so independent reductions do not serialise on VCC).  Measured (tools/microbench_fp.hip,
// profiles/r03_microbench_fp.txt): add + sub pair 9.9 -> 7.1 ns per wave, Montgomery product 11.3 -> 9.8 ns, identical results.
// VG_CARRY_REDUCE=0 restores the min form (A/B builds).
#ifndef VG_CARRY_REDUCE
#define VG_CARRY_REDUCE 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && VG_CARRY_REDUCE == 2
// the same through VCC (VOP2 encodings: half the code bytes; measured 3 % faster on the add + sub pair, equal on the product)
VG_HD uint32_t reduce_once(uint32_t s) {
    uint32_t t, r;
    asm("v_subrev_co_u32_e32 %0, vcc, %3, %2\n\tv_cndmask_b32_e32 %1, %0, %2, vcc" : "=&v"(t), "=v"(r) : "v"(s), "v"(P) : "vcc");
    return r;
}
VG_HD uint32_t sub_mod(uint32_t a, uint32_t b) {
    uint32_t d, t, r;
    asm("v_sub_co_u32_e32 %0, vcc, %3, %4\n\tv_add_u32_e32 %1, %5, %0\n\tv_cndmask_b32_e32 %2, %0, %1, vcc" : "=&v"(d), "=&v"(t), "=v"(r) : "v"(a), "v"(b), "v"(P) : "vcc");
    return r;
}
#elif defined(__HIP_DEVICE_COMPILE__) && VG_CARRY_REDUCE
// s in [0, 2p) -> s mod p
VG_HD uint32_t reduce_once(uint32_t s) {
    uint32_t t, r;
    unsigned long long borrow;
    asm("v_subrev_co_u32_e64 %0, %1, %4, %3\n\tv_cndmask_b32_e64 %2, %0, %3, %1" : "=&v"(t), "=&s"(borrow), "=v"(r) : "v"(s), "v"(P));
    return r;
}
// a, b in [0, p) (or any a >= 0, b with a - b > -p) -> (a - b) mod p
VG_HD uint32_t sub_mod(uint32_t a, uint32_t b) {
    uint32_t d, t, r;
    unsigned long long borrow;
    asm("v_sub_co_u32_e64 %0, %1, %4, %5\n\tv_add_u32_e32 %2, %6, %0\n\tv_cndmask_b32_e64 %3, %0, %2, %1" : "=&v"(d), "=&s"(borrow), "=&v"(t), "=v"(r) : "v"(a), "v"(b), "v"(P));
    return r;
}
#else
VG_HD uint32_t reduce_once(uint32_t s) { uint32_t t = s - P; return s < t ? s : t; }          // s - p wraps to a huge value when s < p
VG_HD uint32_t sub_mod(uint32_t a, uint32_t b) { uint32_t d = a - b, t = d + P; return d < t ? d : t; }  // d wraps to a huge value when a < b
#endif

#ifndef VG_MONTY_ADD
#define VG_MONTY_ADD 0
#endif
VG_HD uint32_t monty_reduce(uint64_t t) {
#if VG_MONTY_ADD == 1  // additive form (A/B builds, see above): t + m p has a zero low word and a high word in [0, 2p)
    const uint32_t m = (uint32_t)t * P_INV_NEG;
    return reduce_once((uint32_t)((t + (uint64_t)m * P) >> 32));
#else
    uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    uint32_t m = mul_lo_pinv(lo);
    uint32_t u = mul_hi_u32(m, P);         // (m * p) >> 32 ; low word of m*p equals lo
    return sub_mod(hi, u);                 // hi - u in (-p, p)
#endif
}

// Montgomery reduction of a lazily accumulated sum of up to FOUR products of values < p
// (t < 4 p^2 < 2^64, high word < 2p): one conditional subtraction of p * 2^32 first, then as above.
// 7 VALU instructions for 4 multiply-adds that each cost one v_mad_u64_u32.
VG_HD uint32_t monty_reduce_wide(uint64_t t) {
    uint32_t lo = (uint32_t)t, hi = reduce_once((uint32_t)(t >> 32));
    uint32_t m = mul_lo_pinv(lo);
    uint32_t u = mul_hi_u32(m, P);
    return sub_mod(hi, u);
}

struct Fp {
    uint32_t v;  // Montgomery representation, < p
    VG_HD static Fp raw(uint32_t m) { Fp r; r.v = m; return r; }
    VG_HD static Fp zero() { return raw(0); }
    VG_HD static Fp one() { return raw(R_MOD_P); }
    VG_HD static Fp from_canonical(uint32_t x) { return raw(monty_reduce((uint64_t)x * R2_MOD_P)); }  // x < p (or any u32: result is x mod p)
    VG_HD uint32_t canonical() const { return monty_reduce((uint64_t)v); }
    VG_HD bool is_zero() const { return v == 0; }
    VG_HD bool operator==(const Fp& o) const { return v == o.v; }
    VG_HD bool operator!=(const Fp& o) const { return v != o.v; }
    // branch-free, 3 VALU instructions each (reduce_once / sub_mod above)
    VG_HD Fp operator+(const Fp& o) const { return raw(reduce_once(v + o.v)); }
    VG_HD Fp operator-(const Fp& o) const { return raw(sub_mod(v, o.v)); }
    VG_HD Fp operator-() const { return raw(v ? P - v : 0); }
#if VG_MONTY_ADD == 2 && defined(__HIP_DEVICE_COMPILE__)  // A/B: the product through v_mul_lo_u32 + v_mul_hi_u32 instead of one v_mad_u64_u32
    VG_HD Fp operator*(const Fp& o) const {
        uint32_t lo, hi;
        asm("v_mul_lo_u32 %0, %2, %3\n\tv_mul_hi_u32 %1, %2, %3" : "=&v"(lo), "=v"(hi) : "v"(v), "v"(o.v));
        return raw(sub_mod(hi, mul_hi_u32(lo * 0x88000001u, P)));
    }
#else
    VG_HD Fp operator*(const Fp& o) const { return raw(monty_reduce((uint64_t)v * o.v)); }
#endif
    VG_HD Fp& operator+=(const Fp& o) { *this = *this + o; return *this; }
    VG_HD Fp& operator-=(const Fp& o) { *this = *this - o; return *this; }
    VG_HD Fp& operator*=(const Fp& o) { *this = *this * o; return *this; }
    VG_HD Fp square() const { return *this * *this; }
    VG_HD Fp pow(uint64_t e) const {
        Fp r = one(), b = *this;
        while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
        return r;
    }
    VG_HD Fp exp_power_of_2(unsigned k) const { Fp r = *this; for (unsigned i = 0; i < k; i++) r *= r; return r; }
    // x^(p-2), p - 2 = 7 * 2^28 + (2^27 - 1), by an addition chain: t = x^(2^27 - 1) (24 squarings, 6 products), w = t * x = x^(2^27),
    // x^(7 * 2^28) = w^14 (3 squarings, 2 products): 37 products instead of square-and-multiply's 59.  0 -> 0.
    VG_HD Fp inv() const {
        const Fp x = *this;
        const Fp x3 = x.square() * x, x7 = x3.square() * x;             // 2^2 - 1, 2^3 - 1
        const Fp a6 = x7.exp_power_of_2(3) * x7;                         // 2^6 - 1
        const Fp a12 = a6.exp_power_of_2(6) * a6;                        // 2^12 - 1
        const Fp a24 = a12.exp_power_of_2(12) * a12;                     // 2^24 - 1
        const Fp t = a24.exp_power_of_2(3) * x7;                         // 2^27 - 1
        const Fp w = t * x, w2 = w.square(), w4 = w2.square(), w8 = w4.square();
        return t * (w8 * w4 * w2);
    }
    VG_HD Fp halve() const { return raw((v & 1) ? (uint32_t)(((uint64_t)v + P) >> 1) : v >> 1); }
};

VG_HD Fp two_adic_generator(unsigned bits) {  // bits <= 27
    return Fp::from_canonical(TWO_ADIC_ROOT_27).exp_power_of_2(27 - bits);
}

struct Ext5 {
    Fp c[5];
    VG_HD static Ext5 zero() { Ext5 r; for (int i = 0; i < 5; i++) r.c[i] = Fp::zero(); return r; }
    VG_HD static Ext5 one() { Ext5 r = zero(); r.c[0] = Fp::one(); return r; }
    VG_HD static Ext5 from_base(Fp b) { Ext5 r = zero(); r.c[0] = b; return r; }
    VG_HD bool is_zero() const { return (c[0].v | c[1].v | c[2].v | c[3].v | c[4].v) == 0; }
    VG_HD bool operator==(const Ext5& o) const { for (int i = 0; i < 5; i++) if (c[i].v != o.c[i].v) return false; return true; }
    VG_HD bool operator!=(const Ext5& o) const { return !(*this == o); }
    VG_HD Ext5 operator+(const Ext5& o) const { Ext5 r; for (int i = 0; i < 5; i++) r.c[i] = c[i] + o.c[i]; return r; }
    VG_HD Ext5 operator-(const Ext5& o) const { Ext5 r; for (int i = 0; i < 5; i++) r.c[i] = c[i] - o.c[i]; return r; }
    VG_HD Ext5 operator-() const { Ext5 r; for (int i = 0; i < 5; i++) r.c[i] = -c[i]; return r; }
    VG_HD Ext5 operator*(const Fp& s) const { Ext5 r; for (int i = 0; i < 5; i++) r.c[i] = c[i] * s; return r; }
    VG_HD Ext5 operator+(const Fp& s) const { Ext5 r = *this; r.c[0] += s; return r; }
    VG_HD Ext5 operator-(const Fp& s) const { Ext5 r = *this; r.c[0] -= s; return r; }
    // Schoolbook, X^5 = 2 folded by pre-doubling one operand: limb k = sum_i a_i * B_{k,i} with
    // B_{k,i} = b_{k-i} (i <= k) or 2*b_{k+5-i} (i > k).  Four products (< 4 p^2 < 2^64) share one
    // Montgomery reduction (monty_reduce_wide): 2 reductions per limb instead of 5.
    VG_HD Ext5 operator*(const Ext5& o) const {
        Fp d[5];
#pragma unroll
        for (int j = 0; j < 5; j++) d[j] = o.c[j] + o.c[j];
        Ext5 r;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            uint32_t bb[5];
#pragma unroll
            for (int i = 0; i < 5; i++) bb[i] = i <= k ? o.c[k - i].v : d[k + 5 - i].v;
            uint64_t t0123 = (uint64_t)c[0].v * bb[0] + (uint64_t)c[1].v * bb[1] + (uint64_t)c[2].v * bb[2] + (uint64_t)c[3].v * bb[3];
            uint64_t t4 = (uint64_t)c[4].v * bb[4];
            r.c[k] = Fp::raw(monty_reduce_wide(t0123)) + Fp::raw(monty_reduce(t4));
        }
        return r;
    }
    VG_HD Ext5& operator+=(const Ext5& o) { *this = *this + o; return *this; }
    VG_HD Ext5& operator-=(const Ext5& o) { *this = *this - o; return *this; }
    VG_HD Ext5& operator*=(const Ext5& o) { *this = *this * o; return *this; }
    VG_HD Ext5 square() const { return *this * *this; }
    VG_HD Ext5 pow(uint64_t e) const {
        Ext5 r = one(), b = *this;
        while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
        return r;
    }
    VG_HD Ext5 exp_power_of_2(unsigned k) const { Ext5 r = *this; for (unsigned i = 0; i < k; i++) r *= r; return r; }
    // Frobenius^K: x -> x^(p^K) multiplies the X^i coefficient by z^(iK), z = 2^((p-1)/5) (canonical 815036133).
    // Constants below are z^j in Montgomery form, j = 1..4.
    template <int K> VG_HD Ext5 frobenius_k() const {
        constexpr uint32_t Z[5] = {0, 1079828539u, 847078768u, 1597816133u, 233372948u};
        Ext5 r;
        r.c[0] = c[0];
        r.c[1] = c[1] * Fp::raw(Z[(1 * K) % 5]);
        r.c[2] = c[2] * Fp::raw(Z[(2 * K) % 5]);
        r.c[3] = c[3] * Fp::raw(Z[(3 * K) % 5]);
        r.c[4] = c[4] * Fp::raw(Z[(4 * K) % 5]);
        return r;
    }
    VG_HD Ext5 frobenius() const { return frobenius_k<1>(); }
    // a^{-1} = (prod_{k=1..4} frob^k(a)) / Norm(a), Norm(a) = a * prod in the base field.  0 -> 0.
    VG_HD Ext5 inv() const {
        Ext5 f1 = frobenius_k<1>(), f2 = frobenius_k<2>(), f3 = frobenius_k<3>(), f4 = frobenius_k<4>();
        Ext5 prod = (f1 * f2) * (f3 * f4);
        // only the constant coefficient of a*prod is needed
        Fp hi = c[1] * prod.c[4] + c[2] * prod.c[3] + c[3] * prod.c[2] + c[4] * prod.c[1];
        Fp norm = c[0] * prod.c[0] + hi + hi;
        return prod * norm.inv();
    }
};

VG_HD uint32_t reverse_bits_len(uint32_t x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}

inline unsigned log2_strict_u64(uint64_t n) { unsigned k = 0; while ((1ull << k) < n) k++; return k; }

}  // namespace vg
