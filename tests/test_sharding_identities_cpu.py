"""The identities the sharded prover (valida_amd/csrc/host/sharded_prover.cpp) rests on, checked in plain integer arithmetic
mod p on small domains — the design's arithmetic, independent of any device code:

  * the row range [r L/W, (r+1) L/W) of a bit-reversed LDE on s H_L is the bit-reversed LDE on the sub-coset s w_L^e H_{L/W}, e = bitrev_W(r);
  * the successor x g_n of a point of shard r lies in the shard of sub-coset (e + 2) mod W, at natural index m + (e + 2) div W;
  * x^n = s^n (-1)^e on the whole shard (the zerofier the quotient kernel is handed);
  * p(z) by the barycentric formula over the whole LDE domain splits into per-shard sums with the factor w_L^e;
  * a FRI fold of a shard with beta w^-e is the shard of the folded vector.
"""
import random

P = 2013265921
G27 = pow(31, 15, P)  # element of order 2^27


def inv(a):
    return pow(a, P - 2, P)


def root(bits):
    return pow(G27, 1 << (27 - bits), P)


def brev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def domain_point(j):
    """device_common.hpp: w^bitrev(j) as a product over the set bits of j, independent of the domain size"""
    r, b = 1, 0
    while j >> b:
        if (j >> b) & 1:
            r = r * root(b + 1) % P
        b += 1
    return r


def test_row_ranges_of_a_committed_lde_are_ldes_on_sub_cosets():
    rng = random.Random(1)
    s = 31
    for logn, W in [(3, 2), (3, 4), (4, 8), (4, 1)]:
        n, logL, logW = 1 << logn, logn + 1, W.bit_length() - 1
        L, Lp, logLp = 2 * n, 2 * n // W, logn + 1 - logW
        coef = [rng.randrange(P) for _ in range(n)]

        def ev(x):
            r = 0
            for c in reversed(coef):
                r = (r * x + c) % P
            return r

        wL = root(logL)
        lde = [ev(s * pow(wL, brev(j, logL), P) % P) for j in range(L)]  # committed order
        z = rng.randrange(P)
        full_scale = (pow(z, L, P) - pow(s, L, P)) * inv(L * pow(s, L - 1, P) % P) % P
        total = 0
        for r in range(W):
            e = brev(r, logW) if logW else 0
            rho = pow(wL, e, P)
            sp = s * rho % P
            part = 0
            for jl in range(Lp):
                x = sp * domain_point(jl) % P
                assert lde[r * Lp + jl] == ev(x)                      # the shard is the LDE on s' H_{L/W}
                m = brev(jl, logLp)
                assert brev(r * Lp + jl, logL) == e + W * m          # natural index of the row
                if W > 1:
                    assert pow(x, n, P) == pow(s, n, P) * (P - 1 if e & 1 else 1) % P
                d, e2 = (e + 2) // W, (e + 2) % W                    # successor: sub-coset e2, natural index m + d
                r2 = brev(e2, logW) if logW else 0
                assert lde[r2 * Lp + brev((m + d) % Lp, logLp)] == ev(x * root(logn) % P)
                part = (part + lde[r * Lp + jl] * domain_point(jl) % P * inv((z - x) % P)) % P
            total = (total + full_scale * rho % P * part) % P
        assert total == ev(z)


def test_a_fold_of_a_shard_is_the_shard_of_the_fold():
    rng = random.Random(2)
    for logL, W in [(4, 2), (5, 4), (6, 8), (4, 1)]:
        L, logW = 1 << logL, W.bit_length() - 1
        Lp = L // W
        wL, i2 = root(logL), inv(2)
        f = [rng.randrange(P) for _ in range(L)]
        beta = rng.randrange(P)
        ref = [((f[2 * i] + f[2 * i + 1]) * i2 + beta * (f[2 * i] - f[2 * i + 1]) % P * i2 % P * inv(pow(wL, brev(2 * i, logL), P))) % P for i in range(L // 2)]
        for r in range(W):
            e = brev(r, logW) if logW else 0
            bp = beta * inv(pow(wL, e, P)) % P
            sh = f[r * Lp:(r + 1) * Lp]
            out = [((sh[2 * i] + sh[2 * i + 1]) * i2 + bp * (sh[2 * i] - sh[2 * i + 1]) % P * i2 % P * inv(domain_point(2 * i))) % P for i in range(Lp // 2)]
            assert out == ref[r * Lp // 2:(r + 1) * Lp // 2]


def test_model_of_the_ntt_pipeline_without_a_bit_reversal_pass():
    """tools/ntt_fused_model.py: the pass structure planned for folding k_bitrev_rows into the transforms (profiles/HISTORY.md section 8; built in round 3: the fused k_lde_* passes), as integers mod p,
    gives the oracle's committed LDE."""
    import os
    import subprocess
    import sys

    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root_dir, "tools", "ntt_fused_model.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("gives the committed LDE") == 4
