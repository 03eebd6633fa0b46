#!/usr/bin/env python3
"""Executable model (plain integers mod p) of the NEXT NTT pipeline — the row bit-reversal folded into the passes (profiles/HISTORY.md section 8; built in round 3: the fused k_lde_* passes):

  today      bitrev copy -> [inverse: contiguous DIT, strided DIT] -> per coset [forward: strided DIF, contiguous DIF]      1 + 2 + 2b passes
  modelled   [inverse: strided DIF, contiguous DIF (natural store inside the block)] -> per coset [forward: contiguous DIF over the
             HIGH index bits, strided DIF with a TRANSPOSING store]                                                              2 + 2b passes

Index sets (N = n_lo n_hi points of one column, evaluations A[i], i = i_hi n_lo + i_lo, natural order, where the device keeps them):
  I1  tile = all i_hi x T consecutive i_lo (64-byte row segments, as k_ntt_strided loads them today): DIF over i_hi with inverse roots,
      value at (p, i_lo), p = bitrev(c_a), times w_N^-(i_lo c_a); in place.
  I2  block p (n_lo contiguous words): DIF over i_lo, stored in NATURAL order inside the block, times 1/N: address p n_lo + c_b holds the
      coefficient c = c_a + n_hi c_b.
  F1  block p again: times shift^c, DIF over c_b -> position q = bitrev(f_b), times w_N^(c_a f_b); out of place (block of the LDE as scratch).
  F2  tile = all blocks (row h of the tile = block bitrev(h), so that row h is c_a = h) x T consecutive q: DIF over c_a -> p'' = bitrev(f_a);
      element (p'', q) goes to address q n_hi + p'' — runs of n_hi consecutive words (1-4 KB): the committed (bit-reversed) order of
      f = f_b + n_lo f_a, with no separate permutation pass.
Checked against the CPU oracle's committed LDE (oracle.pyoracle.committed_lde).  Design aid; nothing in the product imports it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = 2013265921
G27 = pow(31, 15, P)


def root(bits):
    return pow(G27, 1 << (27 - bits), P)


def brev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def dif(v, w):
    """Radix-2 decimation in frequency, natural order in, bit-reversed positions out: out[brev(f)] = sum_i v[i] w^(i f)."""
    v = list(v)
    n = len(v)
    span = n // 2
    wm = w
    while span >= 1:
        for base in range(0, n, 2 * span):
            t = 1
            for j in range(span):
                a, b = v[base + j], v[base + j + span]
                v[base + j] = (a + b) % P
                v[base + j + span] = (a - b) * t % P
                t = t * wm % P
        wm = wm * wm % P
        span //= 2
    return v


def fused_lde(col, k_lo, shift=31, log_blowup=1):
    n = len(col)
    k = n.bit_length() - 1
    k_hi = k - k_lo
    n_lo, n_hi = 1 << k_lo, 1 << k_hi
    wN, w_lo, w_hi = root(k), root(k_lo), root(k_hi)
    inv = lambda x: pow(x, P - 2, P)
    mem = [int(x) for x in col]  # address i_hi n_lo + i_lo
    # I1: strided tiles (all i_hi for one i_lo; T columns share a tile on the device)
    for i_lo in range(n_lo):
        y = dif([mem[h * n_lo + i_lo] for h in range(n_hi)], inv(w_hi))
        for p in range(n_hi):
            c_a = brev(p, k_hi)
            mem[p * n_lo + i_lo] = y[p] * pow(inv(wN), i_lo * c_a, P) % P
    # I2: contiguous blocks, natural store, 1/N
    ninv = inv(n % P)
    for p in range(n_hi):
        y = dif(mem[p * n_lo:(p + 1) * n_lo], inv(w_lo))
        for c_b in range(n_lo):
            mem[p * n_lo + c_b] = y[brev(c_b, k_lo)] * ninv % P
    coeff_layout = list(mem)  # address p n_lo + c_b  <->  coefficient c_a + n_hi c_b, c_a = brev(p)
    b = 1 << log_blowup
    out = [0] * (n * b)
    w_ext = root(k + log_blowup)
    for t in range(b):
        s = shift * pow(w_ext, t, P) % P
        tmp = [0] * n
        # F1: contiguous blocks over the high index bits c_b
        for p in range(n_hi):
            c_a = brev(p, k_hi)
            blk = [coeff_layout[p * n_lo + c_b] * pow(s, c_a + n_hi * c_b, P) % P for c_b in range(n_lo)]
            z = dif(blk, w_lo)
            for q in range(n_lo):
                f_b = brev(q, k_lo)
                tmp[p * n_lo + q] = z[q] * pow(wN, c_a * f_b, P) % P
        # F2: strided tiles over c_a with the transposing store into block brev(t) of the LDE
        row0 = brev(t, log_blowup) * n
        for q in range(n_lo):
            x = dif([tmp[brev(h, k_hi) * n_lo + q] for h in range(n_hi)], w_hi)  # tile row h = c_a = h lives in block brev(h)
            for p2 in range(n_hi):
                out[row0 + q * n_hi + p2] = x[p2]
    return out


def main():
    from oracle import pyoracle as po

    rng = np.random.default_rng(0)
    for k, k_lo, lb in [(6, 3, 1), (8, 5, 1), (8, 3, 1), (7, 4, 2)]:
        col = rng.integers(0, P, 1 << k, dtype=np.uint32)
        want = po.committed_lde(col.reshape(-1, 1), lb, 31)[:, 0]
        got = fused_lde(col, k_lo, 31, lb)
        assert [int(x) for x in want] == got, (k, k_lo, lb)
        print("2^%d points, blocks of 2^%d, blowup %d: the fused pipeline gives the committed LDE" % (k, k_lo, 1 << lb))


if __name__ == "__main__":
    main()
