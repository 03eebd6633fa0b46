// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path.
// PARITY UNPINNED: the arithmetic restated here lives in Plonky3 (valida-xyz/Plonky3 @ bdd338d6, an
// un-vendored git dependency of the reference, Cargo.toml:24-41 / Cargo.lock:651-871) whose source is
// not in /root/reference and cannot be built here (no Rust toolchain).  Every convention below is a
// restatement of the published algorithm, anchored on the reference's call sites.
//
// BabyBear prime field and its degree-5 binomial extension, in *canonical* form (plain u32 < p,
// 64-bit products reduced with %).  Deliberately NOT Montgomery: the product path uses Montgomery
// arithmetic on the device, so agreement between the two is a real check of the arithmetic.
//
// Reference anchors: Val = BabyBear, Challenge = BinomialExtensionField<BabyBear, 5>
// (basic/tests/test_prover.rs:413-416); generator 31, two-adicity 27.
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace oracle {

constexpr uint32_t P = 2013265921u;  // 2^31 - 2^27 + 1

struct Fp {
    uint32_t v;  // canonical, < P
    Fp() : v(0) {}
    explicit Fp(uint32_t x) : v(x % P) {}
    static Fp from_u64(uint64_t x) { Fp r; r.v = (uint32_t)(x % P); return r; }
    static Fp from_i32(int32_t x) {  // Operands::from_i32_slice (machine/src/program.rs:157-165)
        uint32_t a = (uint32_t)(x < 0 ? -(int64_t)x : (int64_t)x) % P;
        Fp r; r.v = (x < 0 && a != 0) ? P - a : a; return r;
    }
    static Fp zero() { return Fp(); }
    static Fp one() { Fp r; r.v = 1; return r; }
    bool is_zero() const { return v == 0; }
    bool operator==(const Fp& o) const { return v == o.v; }
    bool operator!=(const Fp& o) const { return v != o.v; }
    Fp operator+(const Fp& o) const { Fp r; uint32_t s = v + o.v; r.v = s >= P ? s - P : s; return r; }
    Fp operator-(const Fp& o) const { Fp r; r.v = v >= o.v ? v - o.v : v + P - o.v; return r; }
    Fp operator-() const { Fp r; r.v = v ? P - v : 0; return r; }
    Fp operator*(const Fp& o) const { Fp r; r.v = (uint32_t)(((uint64_t)v * o.v) % P); return r; }
    Fp& operator+=(const Fp& o) { *this = *this + o; return *this; }
    Fp& operator-=(const Fp& o) { *this = *this - o; return *this; }
    Fp& operator*=(const Fp& o) { *this = *this * o; return *this; }
    Fp pow(uint64_t e) const {
        Fp r = one(), b = *this;
        while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
        return r;
    }
    Fp inv() const {
        if (v == 0) { fprintf(stderr, "oracle: inverse of zero\n"); abort(); }
        return pow(P - 2);
    }
    Fp exp_power_of_2(unsigned k) const { Fp r = *this; while (k--) r *= r; return r; }
};

constexpr uint32_t GENERATOR = 31;               // BabyBear::generator()
constexpr uint32_t TWO_ADIC_ROOT_27 = 0x1a427a41;  // 31^15, generator of the 2^27 subgroup
constexpr unsigned TWO_ADICITY = 27;

// TwoAdicField::two_adic_generator(bits)  (call sites: machine/src/quotient.rs:95-96)
inline Fp two_adic_generator(unsigned bits) {
    if (bits > TWO_ADICITY) { fprintf(stderr, "oracle: two_adic_generator(%u)\n", bits); abort(); }
    return Fp(TWO_ADIC_ROOT_27).exp_power_of_2(TWO_ADICITY - bits);
}

// F[X]/(X^5 - 2)
struct Ext5 {
    std::array<Fp, 5> c;
    Ext5() {}
    explicit Ext5(Fp b) { c[0] = b; }
    static Ext5 zero() { return Ext5(); }
    static Ext5 one() { return Ext5(Fp::one()); }
    static Ext5 monomial(int i) { Ext5 r; r.c[i] = Fp::one(); return r; }  // machine/src/verify.rs:42-44
    bool is_zero() const { for (auto& x : c) if (!x.is_zero()) return false; return true; }
    bool operator==(const Ext5& o) const { for (int i = 0; i < 5; i++) if (c[i] != o.c[i]) return false; return true; }
    bool operator!=(const Ext5& o) const { return !(*this == o); }
    Ext5 operator+(const Ext5& o) const { Ext5 r; for (int i = 0; i < 5; i++) r.c[i] = c[i] + o.c[i]; return r; }
    Ext5 operator-(const Ext5& o) const { Ext5 r; for (int i = 0; i < 5; i++) r.c[i] = c[i] - o.c[i]; return r; }
    Ext5 operator-() const { Ext5 r; for (int i = 0; i < 5; i++) r.c[i] = -c[i]; return r; }
    Ext5 operator*(const Fp& s) const { Ext5 r; for (int i = 0; i < 5; i++) r.c[i] = c[i] * s; return r; }
    Ext5 operator+(const Fp& s) const { Ext5 r = *this; r.c[0] += s; return r; }
    Ext5 operator-(const Fp& s) const { Ext5 r = *this; r.c[0] -= s; return r; }
    Ext5 operator*(const Ext5& o) const {
        // schoolbook, reduce X^5 = 2
        uint64_t t[9] = {0};
        for (int i = 0; i < 5; i++)
            for (int j = 0; j < 5; j++) t[i + j] = (t[i + j] + (uint64_t)c[i].v * o.c[j].v) % P;
        Ext5 r;
        for (int k = 0; k < 5; k++) {
            uint64_t hi = k + 5 < 9 ? t[k + 5] : 0;
            r.c[k] = Fp::from_u64(t[k] + 2 * hi);
        }
        return r;
    }
    Ext5& operator+=(const Ext5& o) { *this = *this + o; return *this; }
    Ext5& operator-=(const Ext5& o) { *this = *this - o; return *this; }
    Ext5& operator*=(const Ext5& o) { *this = *this * o; return *this; }
    Ext5 pow(uint64_t e) const {
        Ext5 r = one(), b = *this;
        while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
        return r;
    }
    Ext5 exp_power_of_2(unsigned k) const { Ext5 r = *this; while (k--) r *= r; return r; }
    // Inverse by solving the 5x5 linear system a*x = 1 (Gaussian elimination over Fp).  Chosen to be
    // independent of the Frobenius-norm method the product path uses.
    Ext5 inv() const {
        if (is_zero()) { fprintf(stderr, "oracle: Ext5 inverse of zero\n"); abort(); }
        Fp m[5][6];
        // column j of the multiplication-by-a matrix is a * X^j
        for (int j = 0; j < 5; j++) {
            Ext5 col = *this * monomial(j);
            for (int i = 0; i < 5; i++) m[i][j] = col.c[i];
        }
        for (int i = 0; i < 5; i++) m[i][5] = i == 0 ? Fp::one() : Fp::zero();
        for (int col = 0; col < 5; col++) {
            int piv = col;
            while (piv < 5 && m[piv][col].is_zero()) piv++;
            if (piv == 5) { fprintf(stderr, "oracle: singular\n"); abort(); }
            if (piv != col) for (int k = 0; k < 6; k++) std::swap(m[piv][k], m[col][k]);
            Fp iv = m[col][col].inv();
            for (int k = 0; k < 6; k++) m[col][k] *= iv;
            for (int r = 0; r < 5; r++) if (r != col && !m[r][col].is_zero()) {
                Fp f = m[r][col];
                for (int k = 0; k < 6; k++) m[r][k] -= f * m[col][k];
            }
        }
        Ext5 r;
        for (int i = 0; i < 5; i++) r.c[i] = m[i][5];
        return r;
    }
};

inline unsigned log2_strict(size_t n) {
    unsigned k = 0;
    while ((size_t(1) << k) < n) k++;
    if ((size_t(1) << k) != n) { fprintf(stderr, "oracle: %zu not a power of two\n", n); abort(); }
    return k;
}
inline unsigned log2_ceil(size_t n) { unsigned k = 0; while ((size_t(1) << k) < n) k++; return k; }
inline size_t reverse_bits_len(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}
template <class T> void reverse_slice_index_bits(std::vector<T>& v) {
    unsigned k = log2_strict(v.size());
    for (size_t i = 0; i < v.size(); i++) { size_t j = reverse_bits_len(i, k); if (i < j) std::swap(v[i], v[j]); }
}

// Row-major matrix of base-field elements (p3_matrix::dense::RowMajorMatrix<Val>).
struct Matrix {
    size_t height = 0, width = 0;
    std::vector<Fp> v;
    Matrix() {}
    Matrix(size_t h, size_t w) : height(h), width(w), v(h * w) {}
    Fp& at(size_t r, size_t c) { return v[r * width + c]; }
    const Fp& at(size_t r, size_t c) const { return v[r * width + c]; }
    const Fp* row(size_t r) const { return &v[r * width]; }
};

}  // namespace oracle
