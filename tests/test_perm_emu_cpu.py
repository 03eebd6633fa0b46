"""The permutation-trace kernels of valida_amd/csrc/kernels/perm.hip — the very source: k_perm_recip (reciprocals of the random linear combinations,
per-row contribution) and the three-phase block scan of the running sum — run on the CPU under tools/hipemu and compared with the oracle's
generate_permutation_trace (machine/src/chip.rs:121-208).  The interactions reach the kernels in the device encoding of kernels/interactions.hpp,
rebuilt here from the machine's neutral interaction image (vgpu_machine_interaction_words).  The -m gpu suite compares the same kernels' device
results with the same oracle (tests/test_gpu_parity.py::test_perm_trace_*)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import valida_amd as va
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2013265921
R_MOD_P = (1 << 32) % P
c_u32p = ctypes.POINTER(ctypes.c_uint32)


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(ROOT, "tests", "emu", "perm_emu.cpp")
    out = os.path.join(ROOT, "build", "libpermemu.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    csrc = os.path.join(ROOT, "valida_amd", "csrc")
    deps = [src, os.path.join(ROOT, "tools", "hipemu", "hip", "hip_runtime.h"), os.path.join(csrc, "field.hpp")] + [
        os.path.join(csrc, "kernels", f) for f in ("perm.hip", "interactions.hpp", "launch.hpp", "device_common.hpp")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIPCC__", "-x", "c++", "-I", os.path.join(ROOT, "tools", "hipemu"), src, "-o", out], check=True)
    return ctypes.CDLL(out)


def device_words(its):
    """kernels/interactions.hpp, encode_interactions: [M] [max_fields] [offset table] then per interaction [is_send] [n_fields] count fields.."""
    mont = lambda x: (x * R_MOD_P) % P

    def vcol(w, v):
        const, terms = v
        w += [len(terms), mont(const)]
        for is_prep, col, weight in terms:
            w += [col | (0x80000000 if is_prep else 0), mont(weight)]

    w = [len(its), max([len(i["fields"]) for i in its] + [0])]
    table = len(w)
    w += [0] * len(its)
    for m, it in enumerate(its):
        w[table + m] = len(w)
        w += [1 if it["send"] else 0, len(it["fields"])]
        vcol(w, it["count"])
        for f in it["fields"]:
            vcol(w, f)
    return np.array(w, dtype=np.uint32)


# chips without preprocessed columns (as the device tests): cpu, memory, the ALU chips; heights beyond one scan block (1024 rows) and beyond one
# block of block sums (2^18 rows would be; 2^13 crosses several scan blocks)
@pytest.mark.parametrize("chip,width,log_n", [(0, None, 6), (2, 14, 13), (3, 16, 6), (5, None, 5), (7, 28, 6), (8, 45, 4), (10, 79, 3), (4, None, 0), (13, None, 5)])
def test_permutation_trace_kernels_under_emulation_match_the_oracle(emu, machine, chip, width, log_n):
    info = machine.chip_info(chip)
    if info["preprocessed_width"]:
        pytest.skip("chip with preprocessed columns")
    w = width or info["width"]
    assert w == info["width"]
    its = machine.interactions(chip)
    rng = np.random.default_rng(300 + chip)
    n = 1 << log_n
    main = np.ascontiguousarray(rng.integers(0, P, (n, w), dtype=np.uint32))  # random rows define the same function
    ch = np.ascontiguousarray(rng.integers(0, P, 15, dtype=np.uint32))
    iw = device_words(its)
    glob = np.array([1 if i["global"] else 0 for i in its] + [0], dtype=np.uint32)
    bus = np.array([i["bus"] for i in its] + [0], dtype=np.uint32)
    got = np.zeros((n, 5 * (len(its) + 1)), dtype=np.uint32)
    assert emu.emu_perm_trace(main.ctypes.data_as(c_u32p), ctypes.c_uint64(n), ctypes.c_uint64(w), iw.ctypes.data_as(c_u32p), ch.ctypes.data_as(c_u32p), glob.ctypes.data_as(c_u32p),
                              bus.ctypes.data_as(c_u32p), got.ctypes.data_as(c_u32p)) == 0
    want = po.perm_trace(chip, main, ch)
    assert got.shape == want.shape
    assert np.array_equal(got, want), "first mismatch at %s" % (np.argwhere(got != want)[0],)
