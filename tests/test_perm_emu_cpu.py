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
        os.path.join(csrc, "kernels", f) for f in ("perm.hip", "interactions.hpp", "launch.hpp", "device_common.hpp")] + [os.path.join(csrc, "chips", "basic_machine.hpp")]
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
# native: the kernel the prover launches for the in-tree chips since round 5 — the chip's interactions compiled in (chips/basic_machine.hpp: visit_interactions) and
# the row's M combinations inverted together; walk: the encoded interactions, an inversion each (captured AIRs, VGPU_PERM_NATIVE=0)
@pytest.mark.parametrize("native", [False, True], ids=["walk", "native"])
@pytest.mark.parametrize("chip,width,log_n", [(0, None, 6), (2, 14, 13), (3, 16, 6), (5, None, 5), (7, 28, 6), (8, 45, 4), (10, 79, 3), (4, None, 0), (13, None, 5)])
def test_permutation_trace_kernels_under_emulation_match_the_oracle(emu, machine, chip, width, log_n, native):
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
                              bus.ctypes.data_as(c_u32p), got.ctypes.data_as(c_u32p), ctypes.c_int(chip if native else -2)) == 0
    want = po.perm_trace(chip, main, ch)
    assert got.shape == want.shape
    assert np.array_equal(got, want), "first mismatch at %s" % (np.argwhere(got != want)[0],)


@pytest.mark.parametrize("native", [False, True], ids=["walk", "native"])
def test_a_zero_combination_stays_zero_in_both_kernels(emu, machine, native):
    """batch_multiplicative_inverse_allowing_zero (util/src/lib.rs:21-43): a row whose combination alpha_bus + sum_j beta^j f_j is ZERO gets reciprocal 0.
    The native kernel inverts the row's combinations together (Montgomery's trick), where a zero must be taken out of the product and put back as 0.
    The add chip's bus receive has 13 fields: five of its columns (input words) are solved for so that the combination of ONE row vanishes; that row's
    other four combinations (the range sends) are ordinary, so the batched inversion sees a zero among non-zeros."""
    chip = 3  # add: four range sends (one field each) + the ALU-bus receive (opcode constant, 12 column fields)
    its = machine.interactions(chip)
    w = machine.chip_info(chip)["width"]
    rng = np.random.default_rng(77)
    n = 8
    main = np.ascontiguousarray(rng.integers(0, P, (n, w), dtype=np.uint32))
    ch = np.ascontiguousarray(rng.integers(0, P, 15, dtype=np.uint32))
    recv = its[4]
    assert not recv["send"] and len(recv["fields"]) == 13
    rnd = [[int(x) for x in ch[5 * i:5 * i + 5]] for i in range(3)]
    one = [1, 0, 0, 0, 0]
    alpha = one
    for _ in range(recv["bus"] + 1):
        alpha = [int(x) for x in po.ext5_mul(alpha, rnd[1] if recv["global"] else rnd[0])]
    betas, bp = [], one
    for _ in range(13):
        betas.append(bp)
        bp = [int(x) for x in po.ext5_mul(bp, rnd[2])]
    row = 3
    # fields 1..5 are bare columns 0..4 (input_1 and the first word of input_2): unknowns x_1..x_5 with sum_j betas[j][k] x_j = -(rest)[k] for the five limbs k
    cols = [f[1][0][1] for f in recv["fields"][1:6]]
    assert all(len(f[1]) == 1 and f[1][0][2] == 1 and f[0] == 0 for f in recv["fields"][1:6]) and len(set(cols)) == 5
    rest = list(alpha)
    for j, f in enumerate(recv["fields"]):
        if 1 <= j <= 5:
            continue
        const, terms = f
        val = (const + sum(wt * int(main[row, c]) for _, c, wt in terms)) % P
        rest = [(rest[k] + betas[j][k] * val) % P for k in range(5)]
    A = [[betas[1 + j][k] for j in range(5)] for k in range(5)]
    b = [(-rest[k]) % P for k in range(5)]
    for i in range(5):  # Gaussian elimination mod p
        piv = next(r for r in range(i, 5) if A[r][i] % P)
        A[i], A[piv], b[i], b[piv] = A[piv], A[i], b[piv], b[i]
        inv = pow(A[i][i], P - 2, P)
        A[i] = [(x * inv) % P for x in A[i]]
        b[i] = (b[i] * inv) % P
        for r in range(5):
            if r != i and A[r][i]:
                f_ = A[r][i]
                A[r] = [(x - f_ * y) % P for x, y in zip(A[r], A[i])]
                b[r] = (b[r] - f_ * b[i]) % P
    for j, c in enumerate(cols):
        main[row, c] = b[j]
    iw = device_words(its)
    glob = np.array([1 if i["global"] else 0 for i in its] + [0], dtype=np.uint32)
    bus = np.array([i["bus"] for i in its] + [0], dtype=np.uint32)
    got = np.zeros((n, 5 * (len(its) + 1)), dtype=np.uint32)
    assert emu.emu_perm_trace(main.ctypes.data_as(c_u32p), ctypes.c_uint64(n), ctypes.c_uint64(w), iw.ctypes.data_as(c_u32p), ch.ctypes.data_as(c_u32p), glob.ctypes.data_as(c_u32p),
                              bus.ctypes.data_as(c_u32p), got.ctypes.data_as(c_u32p), ctypes.c_int(chip if native else -2)) == 0
    want = po.perm_trace(chip, main, ch)
    assert not want[row, 20:25].any() and want[row, 0:20].any()  # the receive's reciprocal is zero on that row, the sends' are not
    assert np.array_equal(got, want)
