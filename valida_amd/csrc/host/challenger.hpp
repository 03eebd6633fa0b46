// Host-side Fiat-Shamir transcript: Poseidon<BabyBear, CosetMds<16>, 16, 5> (8 full + 22 partial
// rounds) inside DuplexChallenger<BabyBear, _, 16>, as instantiated by the reference at
// basic/tests/test_prover.rs:418-422, :439, :454 and driven from basic/src/lib.rs:185-263,601-619.
// Sequential, microsecond-scale, stays on the host; the 480 round constants are
// configuration input.  Conventions: SURVEY.md App. B7/B8.
#pragma once
#include <vector>
#include "../field.hpp"

namespace vhost {
using vg::Ext5;
using vg::Fp;

struct Poseidon16 {
    Fp rc[30][16];
    Fp fwd_tw[8], inv_tw[8];  // powers of the 16th root / its inverse
    Fp weights[16];           // 31^k
    explicit Poseidon16(const uint32_t* canonical480) {
        for (int r = 0; r < 30; r++) for (int i = 0; i < 16; i++) rc[r][i] = Fp::from_canonical(canonical480[r * 16 + i]);
        Fp w = vg::two_adic_generator(4), wi = w.inv(), g = Fp::from_canonical(vg::GENERATOR);
        fwd_tw[0] = inv_tw[0] = Fp::one();
        for (int i = 1; i < 8; i++) { fwd_tw[i] = fwd_tw[i - 1] * w; inv_tw[i] = inv_tw[i - 1] * wi; }
        weights[0] = Fp::one();
        for (int i = 1; i < 16; i++) weights[i] = weights[i - 1] * g;
    }
    // In-place radix-2 DIT on 16 points, natural in / natural out, no scaling.
    static void dft16(Fp* a, const Fp* tw) {
        for (int i = 0; i < 16; i++) { int j = ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); if (i < j) { Fp t = a[i]; a[i] = a[j]; a[j] = t; } }
        for (int len = 2; len <= 16; len <<= 1) {
            int half = len >> 1, step = 16 / len;
            for (int base = 0; base < 16; base += len)
                for (int j = 0; j < half; j++) {
                    Fp t = tw[j * step] * a[base + j + half], u = a[base + j];
                    a[base + j] = u + t;
                    a[base + j + half] = u - t;
                }
        }
    }
    // CosetMds<16>: evaluations on H (times 16) -> evaluations on 31*H of the same polynomial.
    void mds(Fp* st) const {
        dft16(st, inv_tw);
        for (int i = 0; i < 16; i++) st[i] *= weights[i];
        dft16(st, fwd_tw);
    }
    void permute(Fp* st) const {
        for (int r = 0; r < 30; r++) {
            for (int i = 0; i < 16; i++) st[i] += rc[r][i];
            int n_sbox = (r < 4 || r >= 26) ? 16 : 1;
            for (int i = 0; i < n_sbox; i++) { Fp x2 = st[i] * st[i]; st[i] = x2 * x2 * st[i]; }
            mds(st);
        }
    }
};

struct Challenger {
    const Poseidon16* perm;
    Fp state[16];
    std::vector<Fp> in, out;
    explicit Challenger(const Poseidon16* p) : perm(p) { for (auto& s : state) s = Fp::zero(); }
    void duplexing() {
        for (size_t i = 0; i < in.size(); i++) state[i] = in[i];
        in.clear();
        perm->permute(state);
        out.assign(state, state + 16);
    }
    void observe(Fp x) { out.clear(); in.push_back(x); if (in.size() == 16) duplexing(); }
    void observe_canonical(uint32_t x) { observe(Fp::from_canonical(x)); }
    void observe_digest(const uint32_t* canonical8) { for (int i = 0; i < 8; i++) observe_canonical(canonical8[i]); }
    void observe_ext(const Ext5& e) { for (int i = 0; i < 5; i++) observe(e.c[i]); }
    Fp sample() { if (!in.empty() || out.empty()) duplexing(); Fp r = out.back(); out.pop_back(); return r; }
    Ext5 sample_ext() { Ext5 e; for (int i = 0; i < 5; i++) e.c[i] = sample(); return e; }
    uint64_t sample_bits(unsigned bits) { return (uint64_t)sample().canonical() & ((1ull << bits) - 1); }
    bool check_witness(unsigned bits, Fp w) { observe(w); return sample_bits(bits) == 0; }
    // Canonical PoW rule: smallest witness.  Returns the canonical witness.
    uint32_t grind(unsigned bits) {
        for (uint32_t i = 0; i < vg::P; i++) {
            Challenger c = *this;
            Fp w = Fp::from_canonical(i);
            if (c.check_witness(bits, w)) { check_witness(bits, w); return i; }
        }
        return 0xffffffffu;
    }
};

}  // namespace vhost
