"""The LDE kernels of valida_amd/csrc/kernels/ntt.hip — the very kernel source — run on the CPU under tools/hipemu (every workgroup thread a
fiber, __syncthreads() a yield) and compared element for element with the oracle's committed LDE (App. B3/B4).  This pins the index arithmetic
and the barrier structure of the fused pipeline (k_lde_a / k_lde_mid / k_lde_c) and of the unfused passes without a GPU; the -m gpu suite
compares the same kernels' device results with the same oracle (tests/test_gpu_parity.py::test_large_lde_matches_oracle and every proof test).
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2013265921
c_u32p = ctypes.POINTER(ctypes.c_uint32)


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(ROOT, "tests", "emu", "ntt_emu.cpp")
    out = os.path.join(ROOT, "build", "libnttemu.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    deps = [src, os.path.join(ROOT, "tools", "hipemu", "hip", "hip_runtime.h")] + [
        os.path.join(ROOT, "valida_amd", "csrc", "kernels", f) for f in ("ntt.hip", "butterfly.hpp", "launch.hpp", "device_common.hpp", "profiler.hpp")] + [
        os.path.join(ROOT, "valida_amd", "csrc", "field.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-I", os.path.join(ROOT, "tools", "hipemu"), src, "-o", out], check=True)
    return ctypes.CDLL(out)


def run(fn, m, log_blowup, shift):
    n, w = m.shape
    out = np.zeros((n << log_blowup, w), dtype=np.uint32)
    assert fn(m.ctypes.data_as(c_u32p), ctypes.c_uint64(n), ctypes.c_uint64(w), ctypes.c_int(log_blowup), ctypes.c_uint32(shift), out.ctypes.data_as(c_u32p)) == 0
    return out


# one-tile columns (every height up to 2^12, height 1 included), the four-step split with 2^12-point tiles (k_hi = 1 .. 8: every round
# split of the strided tile), several widths (the persistent blocks' tile -> column map), both blowups, a shifted coset (the quotient round's)
CASES = [(0, 3, 1, 31), (1, 1, 1, 31), (2, 2, 2, 31), (3, 5, 1, 31), (4, 1, 1, 31), (5, 3, 2, 31), (6, 2, 1, 31), (7, 1, 1, 31), (8, 2, 1, 961), (9, 1, 2, 31),
         (10, 2, 1, 31), (11, 1, 1, 31), (12, 3, 1, 31), (13, 2, 1, 31), (14, 1, 2, 31), (15, 3, 1, 961), (16, 2, 1, 31), (17, 1, 1, 31), (18, 1, 2, 31), (20, 2, 1, 31), (22, 1, 1, 31)]  # 2^20 / 2^22: the strided rounds with compile-time shapes


@pytest.mark.parametrize("k,w,log_blowup,shift", CASES)
def test_fused_lde_kernels_under_emulation_match_the_oracle(emu, k, w, log_blowup, shift):
    rng = np.random.default_rng(100 + k)
    m = np.ascontiguousarray(rng.integers(0, P, (1 << k, w), dtype=np.uint32))
    want = po.committed_lde(m, log_blowup, shift)
    got = run(emu.emu_lde_natural, m, log_blowup, shift)
    assert np.array_equal(got, want), "first mismatch at %s" % (np.argwhere(got != want)[0],)


@pytest.mark.parametrize("k,w,log_blowup,shift", [(0, 2, 1, 31), (5, 3, 2, 31), (11, 2, 1, 31), (12, 1, 1, 31), (13, 2, 1, 31), (16, 1, 2, 961), (20, 1, 1, 31), (22, 1, 1, 961)])
def test_unfused_passes_under_emulation_match_the_oracle(emu, k, w, log_blowup, shift):
    """The emulator's own credentials: the passes whose DEVICE results the -m gpu suite has matched to the oracle since round 1 give the
    same LDE when their source runs under it."""
    rng = np.random.default_rng(200 + k)
    m = np.ascontiguousarray(rng.integers(0, P, (1 << k, w), dtype=np.uint32))
    assert np.array_equal(run(emu.emu_lde_unfused, m, log_blowup, shift), po.committed_lde(m, log_blowup, shift))


def test_the_2_14_point_tiles_of_the_largest_heights(emu):
    """Heights from 2^23 use 2^14-point contiguous tiles (k_lde_mid14: 1024 threads x 16 points, the outer radix-4 rounds straight from / to HBM): one column of 2^23 rows."""
    rng = np.random.default_rng(323)
    m = np.ascontiguousarray(rng.integers(0, P, (1 << 23, 1), dtype=np.uint32))
    assert np.array_equal(run(emu.emu_lde_natural, m, 1, 31), po.committed_lde(m, 1, 31))
    assert np.array_equal(run(emu.emu_lde_natural, m, 2, 961), po.committed_lde(m, 2, 961))  # four cosets (C3's blowup), a shifted coset
