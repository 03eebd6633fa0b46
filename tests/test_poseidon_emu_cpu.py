"""The Poseidon-16 MMCS device code of valida_amd/csrc/kernels/poseidon_mmcs.hip — the very source — run on the CPU under tools/hipemu and
compared with the oracle's Poseidon (oracle/, the restatement of Poseidon<BabyBear, CosetMds<16>, 16, 5>, basic/tests/test_prover.rs:418-422):
the permutation in both of its device forms (plain rounds; sparse partial rounds + the MDS layer as a cyclic convolution through the butterfly
network the NTT kernels use), and the leaf / parent kernels launched through the emulator.  The -m gpu suite compares the same kernels' device
results with the same oracle (tests/test_gpu_parity.py: the Poseidon-MMCS roots, proofs and the two full-size fixtures)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import valida_amd as va
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2013265921
c_u32p = ctypes.POINTER(ctypes.c_uint32)


# the MDS layer of the full rounds: 2 = CRT blocks (what the library is built with), 1 = transforms through the butterfly network (the A/B
# alternative, kept buildable with -DVGPU_POSEIDON_MDS=1)
@pytest.fixture(scope="module", params=[2, 1], ids=["mds_blocks", "mds_transforms"])
def emu(request):
    src = os.path.join(ROOT, "tests", "emu", "poseidon_emu.cpp")
    out = os.path.join(ROOT, "build", "libposeidonemu%d.so" % request.param)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    csrc = os.path.join(ROOT, "valida_amd", "csrc")
    deps = [src, os.path.join(ROOT, "tools", "hipemu", "hip", "hip_runtime.h"), os.path.join(csrc, "field.hpp"), os.path.join(csrc, "host", "poseidon_opt.hpp"),
            os.path.join(csrc, "host", "challenger.hpp")] + [os.path.join(csrc, "kernels", f) for f in ("poseidon_mmcs.hip", "poseidon_perm.hpp", "butterfly.hpp", "launch.hpp", "device_common.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-DVGPU_POSEIDON_MDS=%d" % request.param, "-x", "c++", "-I", os.path.join(ROOT, "tools", "hipemu"), src, "-o", out],
                       check=True)
    return ctypes.CDLL(out)


def constants(seed):
    return np.ascontiguousarray(np.random.default_rng(seed).integers(0, P, 480, dtype=np.uint32))


@pytest.mark.parametrize("form", [0, 1])
def test_device_permutation_under_emulation_matches_the_oracle(emu, form):
    for seed, rc in [(0, np.ascontiguousarray(va.poseidon_round_constants(), dtype=np.uint32)), (1, constants(11)), (2, constants(12))]:
        rng = np.random.default_rng(1000 + seed)
        states = [np.zeros(16, dtype=np.uint32), np.full(16, P - 1, dtype=np.uint32)] + [rng.integers(0, P, 16, dtype=np.uint32) for _ in range(6)]
        for st in states:
            got = np.ascontiguousarray(st.copy())
            used = emu.emu_poseidon16_permute(rc.ctypes.data_as(c_u32p), got.ctypes.data_as(c_u32p), ctypes.c_int(form))
            assert used == form  # the sparse / convolution tables of these constants are valid: form 1 really ran the kernels' schedule
            assert list(got) == list(po.poseidon_permute(rc, st))


@pytest.mark.parametrize("width", [1, 7, 8, 9, 16, 29])
def test_leaf_and_parent_kernels_under_emulation_match_the_oracle(emu, width):
    rc = constants(21)
    rng = np.random.default_rng(width)
    n = 4
    m = np.ascontiguousarray(rng.integers(0, P, (n, width), dtype=np.uint32))
    leaves = np.zeros((n, 8), dtype=np.uint32)
    parents = np.zeros((n // 2, 8), dtype=np.uint32)
    assert emu.emu_poseidon_leaves_and_parents(rc.ctypes.data_as(c_u32p), m.ctypes.data_as(c_u32p), ctypes.c_uint64(n), ctypes.c_int(width), leaves.ctypes.data_as(c_u32p),
                                               parents.ctypes.data_as(c_u32p)) == 1
    po.set_mmcs_hash(1, rc)
    try:
        for i in range(n):
            assert list(leaves[i]) == list(po.mmcs_root([m[i:i + 1]]))
        for j in range(n // 2):
            assert list(parents[j]) == list(po.mmcs_root([m[2 * j:2 * j + 2]]))
    finally:
        po.set_mmcs_hash(0)
