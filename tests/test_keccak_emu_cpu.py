"""The Keccak-256 MMCS kernels of valida_amd/csrc/kernels/merkle.hip — the very source: k_keccak_leaves, k_keccak_compress (plain and injecting),
k_keccak_top, with keccak.hpp's permutation (peeled first round, digest-only last round, 32-bit halves, bitop3 / alignbit) — run on the CPU under
tools/hipemu and compared with the oracle's Keccak MMCS (FieldMerkleTreeMmcs<SerializingHasher32<Keccak256Hash>, CompressionFunctionFromHasher<_, 2, 8>>,
basic/tests/test_prover.rs:424-431).  The lane-pair variants (DPP) and the no-op behind v_alignbit_b32 (inline asm) exist on the device only;
the -m gpu suite compares all of them with the same oracle."""
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
    before = os.environ.get("VGPU_KECCAK_PAIRS")
    os.environ["VGPU_KECCAK_PAIRS"] = "0"  # read ONCE by the emulated launchers (first launch below): every layer on the thread-per-node kernels
    src = os.path.join(ROOT, "tests", "emu", "keccak_emu.cpp")
    out = os.path.join(ROOT, "build", "libkeccakemu.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    csrc = os.path.join(ROOT, "valida_amd", "csrc")
    deps = [src, os.path.join(ROOT, "tools", "hipemu", "hip", "hip_runtime.h"), os.path.join(csrc, "field.hpp")] + [
        os.path.join(csrc, "kernels", f) for f in ("merkle.hip", "keccak.hpp", "keccak_pair.hpp", "challenger_dev.hpp", "launch.hpp", "device_common.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-DVK_ALIGNBIT_NOP=0", "-x", "c++", "-I", os.path.join(ROOT, "tools", "hipemu"), src, "-o", out], check=True)
    lib = ctypes.CDLL(out)
    one = np.ones((2, 1), dtype=np.uint32)
    root = np.zeros(8, dtype=np.uint32)
    lib.emu_keccak_root(one.ctypes.data_as(c_u32p), ctypes.c_uint64(2), ctypes.c_int(1), None, ctypes.c_uint64(0), ctypes.c_int(0), ctypes.c_uint64(256), root.ctypes.data_as(c_u32p))
    if before is None:  # the switch is latched inside the emulation library now; the process (and the product library in it) gets its environment back
        os.environ.pop("VGPU_KECCAK_PAIRS", None)
    else:
        os.environ["VGPU_KECCAK_PAIRS"] = before
    return lib


# (rows of the tall matrix, its width, rows of the injected matrix (0: none), its width, parents handled by launches of their own above this length)
CASES = [(1, 3, 0, 0, 256), (2, 1, 0, 0, 256), (16, 14, 0, 0, 4), (16, 33, 4, 10, 4), (16, 34, 8, 35, 2), (32, 67, 16, 69, 8), (64, 10, 1, 5, 256), (8, 70, 8, 1, 1)]


@pytest.mark.parametrize("h0,w0,h1,w1,top_first_len", CASES)
def test_keccak_mmcs_kernels_under_emulation_match_the_oracle(emu, h0, w0, h1, w1, top_first_len):
    rng = np.random.default_rng(h0 * 1000 + w0)
    tall = np.ascontiguousarray(rng.integers(0, P, (h0, w0), dtype=np.uint32))
    low = np.ascontiguousarray(rng.integers(0, P, (max(h1, 1), max(w1, 1)), dtype=np.uint32))
    if h1 == h0:  # same height: the oracle concatenates the rows of equally tall matrices — the device kernels get them as ONE column list
        want = po.mmcs_root([tall, low])
        both = np.ascontiguousarray(np.concatenate([tall, low], axis=1))
        root = np.zeros(8, dtype=np.uint32)
        assert emu.emu_keccak_root(both.ctypes.data_as(c_u32p), ctypes.c_uint64(h0), ctypes.c_int(w0 + w1), None, ctypes.c_uint64(0), ctypes.c_int(0), ctypes.c_uint64(top_first_len),
                                   root.ctypes.data_as(c_u32p)) == 0
        assert list(root) == list(want)
        return
    want = po.mmcs_root([tall, low] if h1 else [tall])
    root = np.zeros(8, dtype=np.uint32)
    assert emu.emu_keccak_root(tall.ctypes.data_as(c_u32p), ctypes.c_uint64(h0), ctypes.c_int(w0), low.ctypes.data_as(c_u32p), ctypes.c_uint64(h1), ctypes.c_int(w1),
                               ctypes.c_uint64(top_first_len), root.ctypes.data_as(c_u32p)) == 0
    assert list(root) == list(want)
