// ORACLE — TEST INFRASTRUCTURE ONLY (see field.hpp header).  PARITY UNPINNED vs Plonky3@bdd338d6.
//
// Keccak-256 (original Keccak padding 0x01, rate 136; tiny-keccak 2.0.2 `Keccak::v256`, Cargo.lock:1348),
// SerializingHasher32<Keccak256Hash>, CompressionFunctionFromHasher<_,_,2,8>, Poseidon<BabyBear,
// CosetMds<16>, 16, 5> and DuplexChallenger<_, _, 16> as instantiated at
// basic/tests/test_prover.rs:418-439.
#pragma once
#include <cstring>
#include "field.hpp"

namespace oracle {

// ---------------------------------------------------------------- Keccak-f[1600]
inline uint64_t rotl64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

inline void keccak_f1600(uint64_t s[25]) {
    // Textbook formulation (FIPS-202 §3.2): theta, rho, pi, chi, iota with LFSR-generated constants.
    static uint64_t RC[24];
    static int ROT[5][5];
    static bool init = false;
    if (!init) {
        // rc(t) LFSR x^8 + x^6 + x^5 + x^4 + 1
        uint8_t lfsr = 1;
        for (int r = 0; r < 24; r++) {
            uint64_t c = 0;
            for (int j = 0; j < 7; j++) {
                if (lfsr & 1) c ^= 1ull << ((1 << j) - 1);
                uint8_t hi = lfsr & 0x80;
                lfsr <<= 1;
                if (hi) lfsr ^= 0x71;
            }
            RC[r] = c;
        }
        int x = 1, y = 0;
        ROT[0][0] = 0;
        for (int t = 0; t < 24; t++) {
            ROT[x][y] = ((t + 1) * (t + 2) / 2) % 64;
            int nx = y, ny = (2 * x + 3 * y) % 5;
            x = nx; y = ny;
        }
        init = true;
    }
    for (int round = 0; round < 24; round++) {
        uint64_t C[5], D[5], B[25];
        for (int x = 0; x < 5; x++) C[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rotl64(C[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) s[i] ^= D[i % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(s[x + 5 * y], ROT[x][y]);
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) s[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
        s[0] ^= RC[round];
    }
}

// Keccak-256 of a byte string; `pad` = 0x01 (Keccak, what the reference uses) or 0x06 (SHA3, test hook).
inline void keccak256(const uint8_t* data, size_t len, uint8_t out[32], uint8_t pad = 0x01) {
    const size_t RATE = 136;
    uint64_t s[25] = {0};
    uint8_t block[136];
    while (len >= RATE) {
        for (size_t i = 0; i < RATE / 8; i++) { uint64_t w; memcpy(&w, data + 8 * i, 8); s[i] ^= w; }
        keccak_f1600(s);
        data += RATE; len -= RATE;
    }
    memset(block, 0, RATE);
    memcpy(block, data, len);
    block[len] ^= pad;
    block[RATE - 1] ^= 0x80;
    for (size_t i = 0; i < RATE / 8; i++) { uint64_t w; memcpy(&w, block + 8 * i, 8); s[i] ^= w; }
    keccak_f1600(s);
    memcpy(out, s, 32);
}

using Digest = std::array<Fp, 8>;

// SerializingHasher32<Keccak256Hash>::hash_iter: canonical u32 LE bytes in, digest bytes mapped to 8
// field elements by from_wrapped_u32 of each LE word (App. B6).
inline Digest keccak_hash_elems(const Fp* e, size_t n) {
    std::vector<uint8_t> bytes(4 * n);
    for (size_t i = 0; i < n; i++) { uint32_t v = e[i].v; memcpy(&bytes[4 * i], &v, 4); }  // little-endian host
    uint8_t out[32];
    keccak256(bytes.data(), bytes.size(), out);
    Digest d;
    for (int i = 0; i < 8; i++) { uint32_t w; memcpy(&w, out + 4 * i, 4); d[i] = Fp(w); }
    return d;
}
// CompressionFunctionFromHasher<Val, MyHash, 2, 8>: C(a, b) = H(a || b)
inline Digest keccak_compress(const Digest& a, const Digest& b) {
    Fp buf[16];
    for (int i = 0; i < 8; i++) { buf[i] = a[i]; buf[8 + i] = b[i]; }
    return keccak_hash_elems(buf, 16);
}

// ---------------------------------------------------------------- Poseidon-16 / CosetMds
// 30 rounds (4 full + 22 partial + 4 full), x^5 S-box, 480 round constants supplied by the caller
// (the reference draws them from an RNG: basic/tests/test_prover.rs:422; they are config input here).
struct Poseidon16 {
    Fp rc[30][16];
    Fp mds[16][16];
    explicit Poseidon16(const uint32_t* constants /*480 canonical values*/) {
        for (int r = 0; r < 30; r++) for (int i = 0; i < 16; i++) rc[r][i] = Fp(constants[r * 16 + i]);
        // CosetMds<16>: y = DFT_16( diag(31^k) * (N * iDFT_16)(x) ), no 1/N: the linear map taking the
        // evaluations of a degree<16 polynomial on H (times 16) to its evaluations on 31*H.
        //   c_k = sum_i x_i w^{-ik};  y_j = sum_k c_k (31 w^j)^k  =>  M[j][i] = sum_k (31 w^{j-i})^k
        Fp w = two_adic_generator(4);
        for (int j = 0; j < 16; j++)
            for (int i = 0; i < 16; i++) {
                Fp base = Fp(GENERATOR) * w.pow((uint64_t)((j - i + 16) % 16));
                Fp acc = Fp::zero(), pw = Fp::one();
                for (int k = 0; k < 16; k++) { acc += pw; pw *= base; }
                mds[j][i] = acc;
            }
    }
    void permute(Fp st[16]) const {
        for (int r = 0; r < 30; r++) {
            for (int i = 0; i < 16; i++) st[i] += rc[r][i];
            bool full = r < 4 || r >= 26;
            for (int i = 0; i < (full ? 16 : 1); i++) { Fp x2 = st[i] * st[i]; st[i] = x2 * x2 * st[i]; }
            Fp out[16];
            for (int j = 0; j < 16; j++) {
                Fp acc = Fp::zero();
                for (int i = 0; i < 16; i++) acc += mds[j][i] * st[i];
                out[j] = acc;
            }
            for (int i = 0; i < 16; i++) st[i] = out[i];
        }
    }
};

// ---------------------------------------------------------------- MMCS hash selection
// kind 0 (default): the reference's Keccak MMCS above.  kind 1: the north-star Poseidon variant, which the reference never
// instantiates — PaddingFreeSponge<Perm16, 16, 8, 8> (hash_iter: for each chunk of RATE = 8 inputs OVERWRITE state[0..len) with
// the chunk, permute; output state[0..8)) and TruncatedPermutation<Perm16, 2, 8, 16> (permute left || right, keep the first 8),
// as p3-symmetric defines them ([P3-RECALL], unpinned).  A process-wide switch: this is test infrastructure.
struct MmcsHash { int kind = 0; const Poseidon16* poseidon = nullptr; };
inline MmcsHash& mmcs_hash() { static MmcsHash h; return h; }

inline Digest hash_elems(const Fp* e, size_t n) {
    const MmcsHash& h = mmcs_hash();
    if (h.kind == 0) return keccak_hash_elems(e, n);
    Fp st[16];
    for (size_t base = 0; base < n; base += 8) {
        for (size_t k = 0; k < 8 && base + k < n; k++) st[k] = e[base + k];
        h.poseidon->permute(st);
    }
    Digest d;
    for (int i = 0; i < 8; i++) d[i] = st[i];
    return d;
}
inline Digest hash_elems(const std::vector<Fp>& e) { return hash_elems(e.data(), e.size()); }
inline Digest compress(const Digest& a, const Digest& b) {
    const MmcsHash& h = mmcs_hash();
    if (h.kind == 0) return keccak_compress(a, b);
    Fp st[16];
    for (int i = 0; i < 8; i++) { st[i] = a[i]; st[8 + i] = b[i]; }
    h.poseidon->permute(st);
    Digest d;
    for (int i = 0; i < 8; i++) d[i] = st[i];
    return d;
}

// DuplexChallenger<Val, Perm16, 16> (App. B8): rate = full width, outputs popped from the END.
struct Challenger {
    const Poseidon16* perm;
    Fp state[16];
    std::vector<Fp> in, out;
    explicit Challenger(const Poseidon16* p) : perm(p) {}
    void duplexing() {
        for (size_t i = 0; i < in.size(); i++) state[i] = in[i];
        in.clear();
        perm->permute(state);
        out.assign(state, state + 16);
    }
    void observe(Fp x) {
        out.clear();
        in.push_back(x);
        if (in.size() == 16) duplexing();
    }
    void observe(const Digest& d) { for (auto& x : d) observe(x); }
    void observe_ext(const Ext5& e) { for (auto& x : e.c) observe(x); }
    Fp sample() {
        if (!in.empty() || out.empty()) duplexing();
        Fp r = out.back(); out.pop_back();
        return r;
    }
    Ext5 sample_ext() { Ext5 e; for (int i = 0; i < 5; i++) e.c[i] = sample(); return e; }
    size_t sample_bits(unsigned bits) { return (size_t)sample().v & ((size_t(1) << bits) - 1); }
    bool check_witness(unsigned bits, Fp w) { observe(w); return sample_bits(bits) == 0; }
    // grind: canonical rule = smallest witness (the reference's rayon find_any is
    // nondeterministic with >1 thread).
    Fp grind(unsigned bits) {
        for (uint32_t i = 0; i < P; i++) {
            Challenger c = *this;
            if (c.check_witness(bits, Fp(i))) { check_witness(bits, Fp(i)); return Fp(i); }
        }
        fprintf(stderr, "oracle: grind failed\n"); abort();
    }
};

}  // namespace oracle
