// Poseidon-16's 22 partial rounds in their sparse-matrix form (the standard optimisation of the Poseidon paper, Appendix B, derived
// here for the reference's instance: Poseidon<BabyBear, CosetMds<16>, 16, 5>, 4 + 22 + 4 rounds, basic/tests/test_prover.rs:418-422).
// A partial round is x <- M S0(x + c) with S0 the S-box on coordinate 0 only.  Anything that fixes coordinate 0 commutes with S0:
//   * the constants of coordinates 1..15 (a vector tau with tau[0] = 0) pass through S0 and are carried to the next round
//     (tau' = M tau + c' with its coordinate 0 split off as the scalar t' that round adds to coordinate 0);
//   * a matrix T = diag(1, T^) passes through S0, so with D = M T = [[d00, dv],[dw, D^]] = diag(1, D^) . [[d00, dv],[D^-1 dw, I]]
//     the dense part diag(1, D^) moves on to the next round and the round itself multiplies by the SPARSE matrix
//     S = [[a, u^T],[w, I]] (a = d00, u = dv, w = D^-1 dw): 31 products instead of 256.
// The recursion runs forward and leaves one dense matrix F = M T_21 and one vector f = M tau_21 for the LAST partial round; f is
// merged into the constants of the full round that follows.  Identical outputs by construction; the constructor checks it on
// random states against the plain permutation and the kernels fall back to the plain form if a block D^ is singular.
#pragma once
#include <cstring>
#include <vector>
#include "challenger.hpp"

namespace vhost {

struct PoseidonOptTables {
    // device image (Montgomery words), offsets in words
    static constexpr int RC_FULL = 0;                 // [8][16]: rounds 0..3, then 26..29 (26's with f merged)
    static constexpr int T_SCALARS = RC_FULL + 128;   // [22 (+2 pad)]: t_0 .. t_21
    static constexpr int SPARSE = T_SCALARS + 24;     // [21][32]: a, u[1..15], w[1..15], pad
    static constexpr int F_DENSE = SPARSE + 21 * 32;  // [16][16] row-major
    // the MDS layer of the FULL rounds as a 16-point cyclic convolution (CosetMds<16> is circulant: y_j = sum_i c[(j - i) & 15] x_i):
    // y = iDFT(lambda . DFT(x)), lambda = DFT(c) / 16 — 50 products and 128 additions instead of 256 products (butterfly.hpp)
    static constexpr int FFT_FWD = F_DENSE + 256;     // [15 (+1 pad)]: stage s = 1..4 at offset 2^(s-1) - 1: w_{2^s}^j, j < 2^(s-1)
    static constexpr int FFT_INV = FFT_FWD + 16;      // the same with w^-j
    static constexpr int FFT_LAM = FFT_INV + 16;      // [16]: lambda in bit-reversed order (the order the DIF transform leaves)
    // ... and as the CRT split of that convolution, x^16 - 1 = (x^8 + 1)(x^4 + 1)(x^4 - 1): two stages of additions / subtractions, three
    // SMALL dense products with lazily accumulated terms (negacyclic 8 x 8, negacyclic 4 x 4, cyclic 4 x 4: 96 terms instead of 256, no
    // twiddle products at all), two stages back.  On gfx950 a multiply-add into a 64-bit accumulator is ONE half-rate instruction and a
    // modular addition three full-rate ones: this form (~430 instructions) beats both the dense product (~900) and the transforms (~650).
    static constexpr int BLK_N8 = FFT_LAM + 16;       // [8][8]  y-[k] = sum_j N8[k][j] (a[j] - a[j + 8])
    static constexpr int BLK_C4 = BLK_N8 + 64;        // [4][4]  on a++[j] = (a[j] + a[j + 8]) + (a[j + 4] + a[j + 12])
    static constexpr int BLK_N4 = BLK_C4 + 16;        // [4][4]  on a+-[j] = (a[j] + a[j + 8]) - (a[j + 4] + a[j + 12])
    // The sparse partial rounds in groups of four with DEFERRED updates of coordinates 1..15 (kernels/poseidon_perm.hpp): inside a group a round
    // reads the coordinates as they stood at the group's start and corrects for the group's earlier rounds through the scalars
    // cross[r][j] = u_r . w_(4 (r / 4) + j), j < r mod 4.
    static constexpr int CROSS = BLK_N4 + 16;         // [21 (+3 pad)][4]
    static constexpr int WORDS = CROSS + 24 * 4;
    std::vector<uint32_t> words;
    bool valid = false;

    static bool invert(std::vector<std::vector<Fp>>& a, std::vector<std::vector<Fp>>& inv) {
        const size_t n = a.size();
        inv.assign(n, std::vector<Fp>(n, Fp::zero()));
        for (size_t i = 0; i < n; i++) inv[i][i] = Fp::one();
        for (size_t col = 0; col < n; col++) {
            size_t piv = col;
            while (piv < n && a[piv][col].is_zero()) piv++;
            if (piv == n) return false;
            std::swap(a[piv], a[col]); std::swap(inv[piv], inv[col]);
            const Fp s = a[col][col].inv();
            for (size_t j = 0; j < n; j++) { a[col][j] *= s; inv[col][j] *= s; }
            for (size_t r = 0; r < n; r++) {
                if (r == col || a[r][col].is_zero()) continue;
                const Fp f = a[r][col];
                for (size_t j = 0; j < n; j++) { a[r][j] -= f * a[col][j]; inv[r][j] -= f * inv[col][j]; }
            }
        }
        return true;
    }

    explicit PoseidonOptTables(const Poseidon16& p) : words(WORDS, 0) {
        // the MDS as a matrix: column j = image of the unit vector e_j
        Fp M[16][16];
        for (int j = 0; j < 16; j++) {
            Fp e[16];
            for (auto& x : e) x = Fp::zero();
            e[j] = Fp::one();
            p.mds(e);
            for (int i = 0; i < 16; i++) M[i][j] = e[i];
        }
        auto mat_vec = [&](const Fp (&A)[16][16], const Fp* v, Fp* out) {
            for (int i = 0; i < 16; i++) { Fp acc = Fp::zero(); for (int j = 0; j < 16; j++) acc += A[i][j] * v[j]; out[i] = acc; }
        };
        for (int r = 0; r < 4; r++) for (int i = 0; i < 16; i++) words[RC_FULL + 16 * r + i] = p.rc[r][i].v;
        Fp T[16][16];  // T_i = diag(1, T^)
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) T[i][j] = i == j ? Fp::one() : Fp::zero();
        Fp tau[16];
        for (int i = 0; i < 16; i++) tau[i] = i ? p.rc[4][i] : Fp::zero();
        words[T_SCALARS] = p.rc[4][0].v;  // t_0
        for (int i = 0; i < 22; i++) {    // partial round i = global round 4 + i
            Fp D[16][16];
            for (int a = 0; a < 16; a++) for (int b = 0; b < 16; b++) { Fp acc = Fp::zero(); for (int k = 0; k < 16; k++) acc += M[a][k] * T[k][b]; D[a][b] = acc; }
            Fp mt[16];
            mat_vec(M, tau, mt);
            if (i == 21) {  // last partial round: dense F = M T_21, f = M tau_21 merged into round 26's constants
                for (int a = 0; a < 16; a++) for (int b = 0; b < 16; b++) words[F_DENSE + 16 * a + b] = D[a][b].v;
                for (int a = 0; a < 16; a++) words[RC_FULL + 16 * 4 + a] = (p.rc[26][a] + mt[a]).v;
                for (int r = 27; r < 30; r++) for (int a = 0; a < 16; a++) words[RC_FULL + 16 * (r - 22) + a] = p.rc[r][a].v;
                break;
            }
            std::vector<std::vector<Fp>> Dh(15, std::vector<Fp>(15)), Dinv;
            for (int a = 0; a < 15; a++) for (int b = 0; b < 15; b++) Dh[a][b] = D[a + 1][b + 1];
            auto Dh_copy = Dh;
            if (!invert(Dh_copy, Dinv)) return;  // valid stays false: plain rounds are used
            uint32_t* s = &words[SPARSE + 32 * i];
            s[0] = D[0][0].v;
            for (int b = 1; b < 16; b++) s[b] = D[0][b].v;                                     // u
            for (int a = 0; a < 15; a++) { Fp acc = Fp::zero(); for (int k = 0; k < 15; k++) acc += Dinv[a][k] * D[k + 1][0]; s[16 + a] = acc.v; }  // w = D^-1 dw
            for (int a = 0; a < 16; a++) for (int b = 0; b < 16; b++) T[a][b] = (a == 0 || b == 0) ? (a == b ? Fp::one() : Fp::zero()) : Dh[a - 1][b - 1];
            // constants of the next round: M tau + c_{i+1}; its coordinate 0 is that round's scalar, the rest is carried on
            for (int a = 0; a < 16; a++) mt[a] += p.rc[5 + i][a];
            words[T_SCALARS + i + 1] = mt[0].v;
            tau[0] = Fp::zero();
            for (int a = 1; a < 16; a++) tau[a] = mt[a];
        }
        {   // the convolution form of the MDS layer
            for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) if (M[j][i] != M[(j - i) & 15][0]) return;  // not circulant: plain rounds
            const Fp w16 = vg::two_adic_generator(4), inv16 = Fp::from_canonical(16).inv();
            for (int st = 0; st < 4; st++) {
                const Fp ws = w16.pow((uint64_t)(16 >> (st + 1))), wsi = ws.inv();
                for (int k = 0; k < (1 << st); k++) { words[FFT_FWD + (1 << st) - 1 + k] = ws.pow((uint64_t)k).v; words[FFT_INV + (1 << st) - 1 + k] = wsi.pow((uint64_t)k).v; }
            }
            for (int i = 0; i < 16; i++) {
                const int k = ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3);
                Fp acc = Fp::zero();
                for (int d = 0; d < 16; d++) acc += M[d][0] * w16.pow((uint64_t)(d * k));
                words[FFT_LAM + i] = (acc * inv16).v;
            }
            // CRT blocks: kernels c- = (c[i] - c[i + 8]) / 2, c+ = (c[i] + c[i + 8]) / 2, then c+- / c++ from c+ the same way (the halves undo
            // the two reconstruction steps y = (y+ + y-) / 2)
            const Fp half = Fp::from_canonical(2).inv();
            Fp cm[8], cp[8], cpm[4], cpp[4];
            for (int i = 0; i < 8; i++) { cm[i] = (M[i][0] - M[i + 8][0]) * half; cp[i] = (M[i][0] + M[i + 8][0]) * half; }
            for (int i = 0; i < 4; i++) { cpm[i] = (cp[i] - cp[i + 4]) * half; cpp[i] = (cp[i] + cp[i + 4]) * half; }
            for (int k = 0; k < 8; k++) for (int j = 0; j < 8; j++) words[BLK_N8 + 8 * k + j] = (j <= k ? cm[k - j] : -cm[8 + k - j]).v;
            for (int k = 0; k < 4; k++) for (int j = 0; j < 4; j++) {
                words[BLK_C4 + 4 * k + j] = cpp[(k - j) & 3].v;
                words[BLK_N4 + 4 * k + j] = (j <= k ? cpm[k - j] : -cpm[4 + k - j]).v;
            }
        }
        for (int r = 0; r < 21; r++)
            for (int j = 0; j < r % 4; j++) {
                const int q = 4 * (r / 4) + j;
                Fp acc = Fp::zero();
                for (int b = 1; b < 16; b++) acc += Fp::raw(words[SPARSE + 32 * r + b]) * Fp::raw(words[SPARSE + 32 * q + 15 + b]);
                words[CROSS + 4 * r + j] = acc.v;
            }
        valid = true;
        // self-check against the plain permutation
        uint64_t seed = 0x9E3779B97F4A7C15ull;
        for (int trial = 0; trial < 8 && valid; trial++) {
            Fp x[16], y[16];
            for (int i = 0; i < 16; i++) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; x[i] = Fp::from_canonical((uint32_t)((seed >> 33) % vg::P)); y[i] = x[i]; }
            p.permute(x);
            permute(p, y);
            for (int i = 0; i < 16; i++) valid = valid && x[i] == y[i];
        }
    }

    // the MDS layer from the convolution tables, by the definition of the transforms (the kernels run the butterfly network of butterfly.hpp)
    void mds_convolution(Fp* st) const {
        const Fp w16 = vg::two_adic_generator(4), w16i = w16.inv();
        Fp X[16];
        for (int i = 0; i < 16; i++) {  // X[i] = lambda_k DFT(x)_k, k = bitrev(i)
            const int k = ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3);
            Fp acc = Fp::zero();
            for (int j = 0; j < 16; j++) acc += st[j] * w16.pow((uint64_t)(j * k));
            X[k] = acc * Fp::raw(words[FFT_LAM + i]);
        }
        for (int j = 0; j < 16; j++) { Fp acc = Fp::zero(); for (int k = 0; k < 16; k++) acc += X[k] * w16i.pow((uint64_t)(j * k)); st[j] = acc; }
    }
    // the MDS layer from the CRT block tables, step for step as the kernels do it
    void mds_blocks(Fp* st) const {
        auto W = [&](int off) { return Fp::raw(words[off]); };
        Fp am[8], ap[8], app[4], apm[4], ym[8], ypp[4], ypm[4], yp[8];
        for (int i = 0; i < 8; i++) { am[i] = st[i] - st[i + 8]; ap[i] = st[i] + st[i + 8]; }
        for (int i = 0; i < 4; i++) { apm[i] = ap[i] - ap[i + 4]; app[i] = ap[i] + ap[i + 4]; }
        for (int k = 0; k < 8; k++) { Fp acc = Fp::zero(); for (int j = 0; j < 8; j++) acc += W(BLK_N8 + 8 * k + j) * am[j]; ym[k] = acc; }
        for (int k = 0; k < 4; k++) {
            Fp a = Fp::zero(), b = Fp::zero();
            for (int j = 0; j < 4; j++) { a += W(BLK_C4 + 4 * k + j) * app[j]; b += W(BLK_N4 + 4 * k + j) * apm[j]; }
            ypp[k] = a; ypm[k] = b;
        }
        for (int i = 0; i < 4; i++) { yp[i] = ypp[i] + ypm[i]; yp[i + 4] = ypp[i] - ypm[i]; }
        for (int i = 0; i < 8; i++) { st[i] = yp[i] + ym[i]; st[i + 8] = yp[i] - ym[i]; }
    }
    // the optimised schedule on the host (what the kernels do), for the self-check and the C-ABI test hook
    void permute(const Poseidon16& p, Fp* st) const {
        auto W = [&](int off) { return Fp::raw(words[off]); };
        auto sbox = [](Fp x) { Fp x2 = x * x; return x2 * x2 * x; };
        for (int r = 0; r < 4; r++) { for (int i = 0; i < 16; i++) st[i] = sbox(st[i] + W(RC_FULL + 16 * r + i)); mds_blocks(st); }  // both table sets are exercised:
        st[0] += W(T_SCALARS);
        for (int g0 = 0; g0 < 21; g0 += 4) {  // groups of four rounds: coordinates 1..15 are updated at the group's end (as the kernels do)
            Fp x0s[4], upd[16];
            for (auto& u : upd) u = Fp::zero();
            for (int k = 0; k < 4 && g0 + k < 21; k++) {
                const int i = g0 + k, s = SPARSE + 32 * i;
                const Fp x0 = sbox(st[0]);
                x0s[k] = x0;
                Fp n0 = W(s) * x0;
                for (int b = 1; b < 16; b++) n0 += W(s + b) * st[b];
                for (int j = 0; j < k; j++) n0 += W(CROSS + 4 * i + j) * x0s[j];
                for (int a = 1; a < 16; a++) upd[a] += W(s + 15 + a) * x0;
                st[0] = n0 + W(T_SCALARS + i + 1);
            }
            for (int a = 1; a < 16; a++) st[a] += upd[a];
        }
        st[0] = sbox(st[0]);
        Fp out[16];
        for (int a = 0; a < 16; a++) { Fp acc = Fp::zero(); for (int b = 0; b < 16; b++) acc += W(F_DENSE + 16 * a + b) * st[b]; out[a] = acc; }
        for (int a = 0; a < 16; a++) st[a] = out[a];
        for (int r = 4; r < 8; r++) { for (int i = 0; i < 16; i++) st[i] = sbox(st[i] + W(RC_FULL + 16 * r + i)); mds_convolution(st); }  // blocks above, transforms here
    }
};

// The Poseidon table a context keeps on the device: [480 round constants][16 circulant MDS coefficients][.. 1023: scratch of the device
// challenger / proof-of-work search] and from word 1024 the PoseidonOptTables image.  sparse = those tables are valid.
inline std::vector<uint32_t> poseidon_device_image(const uint32_t* rc480_canonical, const Poseidon16& p, bool& sparse) {
    PoseidonOptTables popt(p);
    std::vector<uint32_t> pos(1024 + PoseidonOptTables::WORDS, 0);
    memcpy(pos.data() + 1024, popt.words.data(), popt.words.size() * 4);
    for (int i = 0; i < 480; i++) pos[i] = Fp::from_canonical(rc480_canonical[i]).v;
    const Fp w16 = vg::two_adic_generator(4), g = Fp::from_canonical(vg::GENERATOR);
    for (int d = 0; d < 16; d++) {  // column 0 of CosetMds<16>: sum_k (31 w^d)^k
        Fp base = g * w16.pow((uint64_t)d), acc = Fp::zero(), pw = Fp::one();
        for (int k = 0; k < 16; k++) { acc += pw; pw *= base; }
        pos[480 + d] = acc.v;
    }
    sparse = popt.valid;
    return pos;
}

}  // namespace vhost
