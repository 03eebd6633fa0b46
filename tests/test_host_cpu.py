"""CPU tests of the product's host side (no GPU): the C ABI loads and exports what include/vgpu.h
declares, the host transcript and the compiled constraint programs agree with the oracle, the AIR
capture FFI works, and the product refuses to run without a device (no CPU fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import valida_amd as va
from conftest import has_gpu
from oracle import pyoracle as po

P = va.P
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vgpu.h")).read()
    names = set(re.findall(r"\b(vgpu_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) > 50
    L = va.lib()
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    assert b"gfx950" in L.vgpu_version()


def test_no_oracle_in_product_path():
    # the product never imports / links the oracle
    for base, _, files in os.walk(os.path.join(ROOT, "valida_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".cpp", ".hip", ".h")):
                src = open(os.path.join(base, f)).read()
                assert "liboracle" not in src and "pyoracle" not in src and 'oracle/' not in src.replace("ORACLE", ""), os.path.join(base, f)
    out = subprocess.run(["ldd", os.path.join(ROOT, "valida_amd", "libvgpu.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out
    # the verifier's host-side commit (vgpu_host_commit_root: host_ntt + a host MMCS, for the small preprocessed traces) is no fallback of the
    # prover: no proving source reaches it
    host = os.path.join(ROOT, "valida_amd", "csrc", "host")
    for f in ("prover.cpp", "prover.hpp", "pcs.hpp", "sharded.hpp", "sharded_prover.cpp", "sharded_prover.hpp", "fabric.hpp", "runtime.hpp"):
        src = open(os.path.join(host, f)).read()
        assert "host_commit_root" not in src and "host_ntt" not in src and "machine_verifier" not in src, f


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a GPU")
def test_prover_refuses_to_run_without_device(machine, rc):
    with pytest.raises(va.VgpuError) as e:
        va.Prover(machine, rc)
    assert e.value.code == -3  # VGPU_ERR_HIP


def test_poseidon_and_challenger_match_oracle(rc):
    # product: Montgomery arithmetic + FFT-form CosetMds; oracle: canonical arithmetic + explicit matrix
    rng = np.random.default_rng(1)
    for _ in range(8):
        st = rng.integers(0, P, size=16, dtype=np.uint32)
        assert list(va.poseidon16_permute(rc, st)) == list(po.poseidon_permute(rc, st))
    obs = rng.integers(0, P, size=37, dtype=np.uint32)
    ch = va.Challenger(rc)
    ch.observe(obs)
    assert list(ch.sample(41)) == list(po.challenger_probe(rc, obs, 41))
    ch2 = va.Challenger(rc)
    ch2.observe(obs[:9])
    assert ch2.grind(7) == po.grind(rc, obs[:9], 7)


def test_sample_bits_is_low_bits_of_canonical_sample(rc):
    a, b = va.Challenger(rc), va.Challenger(rc)
    a.observe([1, 2, 3])
    b.observe([1, 2, 3])
    assert a.sample_bits(11) == int(b.sample(1)[0]) & 2047


def test_chip_table(machine):
    want = {  # (width, prep width, interactions, constraints)
        0: (51, 0, 4, 53), 1: (1, 7, 0, 0), 2: (14, 0, 1, 0), 3: (16, 0, 5, 10), 4: (16, 0, 5, 7), 5: (18, 0, 1, 5), 6: (14, 0, 1, 0),
        7: (28, 0, 2, 18), 9: (14, 0, 1, 8), 10: (79, 0, 1, 88), 11: (7, 0, 1, 3), 12: (2, 1, 1, 0), 13: (6, 0, 1, 1),
    }
    assert machine.num_chips == 14
    for chip, (w, pw, m, k) in want.items():
        info = machine.chip_info(chip)
        assert (info["width"], info["preprocessed_width"], info["interactions"], info["constraints"]) == (w, pw, m, k), (chip, info)
        assert info["log_quotient_degree"] == 1 and info["max_degree"] <= 3
    assert machine.chip_info(8)["width"] == 45 and machine.chip_info(8)["constraints"] >= 55


@pytest.mark.parametrize("chip", [0, 3, 4, 5, 7, 8, 9, 10, 11, 13])
def test_compiled_program_equals_direct_eval(machine, chip):
    # the lowered register program (what the quotient kernel interprets) == direct template instantiation
    rng = np.random.default_rng(40 + chip)
    w = machine.chip_info(chip)["width"]
    for trial in range(6):
        local = rng.integers(0, P, size=w, dtype=np.uint32)
        nxt = rng.integers(0, P, size=w, dtype=np.uint32)
        f, l, t = [int(x) for x in rng.integers(0, P, size=3)]
        got = machine.eval_constraints(chip, local, nxt, is_first=f, is_last=l, is_transition=t)
        want = po.eval_constraints(chip, local, nxt, is_first=f, is_last=l, is_transition=t)
        assert list(got) == list(want)


def test_constraints_vanish_on_the_fibonacci_trace(machine, fib25):
    for chip in (0, 3, 5):
        tr = fib25.main_trace(chip)
        n = tr.shape[0]
        for i in range(n):
            vals = machine.eval_constraints(chip, tr[i], tr[(i + 1) % n], is_first=int(i == 0), is_last=int(i == n - 1), is_transition=int(i != n - 1))
            assert not vals.any(), (chip, i, np.nonzero(vals)[0][:4])


def test_air_capture_ffi_roundtrip():
    """A host builds an AIR through the C ABI exactly as a Rust SymbolicAirBuilder shim would."""
    L = va.lib()
    air = ctypes.c_void_p()
    assert L.vgpu_air_new(b"toy", ctypes.c_uint32(3), ctypes.c_uint32(0), ctypes.byref(air)) == 0
    u = ctypes.c_uint32
    a = L.vgpu_air_variable(air, u(0), u(0), u(0))
    b = L.vgpu_air_variable(air, u(0), u(1), u(0))
    c_next = L.vgpu_air_variable(air, u(0), u(2), u(1))
    seven = L.vgpu_air_constant(air, u(7))
    # assert a*b - 7 = 0 ; when_transition: c' - (a + b) = 0 ; assert_bool(a)
    L.vgpu_air_assert_zero(air, u(L.vgpu_air_sub(air, u(L.vgpu_air_mul(air, u(a), u(b))), u(seven))))
    tr = L.vgpu_air_is_transition(air)
    L.vgpu_air_assert_zero(air, u(L.vgpu_air_mul(air, u(tr), u(L.vgpu_air_sub(air, u(c_next), u(L.vgpu_air_add(air, u(a), u(b))))))))
    one = L.vgpu_air_constant(air, u(1))
    L.vgpu_air_assert_zero(air, u(L.vgpu_air_mul(air, u(a), u(L.vgpu_air_sub(air, u(a), u(one))))))
    m = ctypes.c_void_p()
    assert L.vgpu_machine_new(ctypes.byref(m)) == 0
    assert L.vgpu_machine_push_air(m, air) == 0
    mach = va.Machine(m)
    info = mach.chip_info(0)
    assert info["constraints"] == 3 and info["max_degree"] == 2 and info["log_quotient_degree"] == 1
    vals = mach.eval_constraints(0, np.array([3, 5, 0], dtype=np.uint32), np.array([0, 0, 9], dtype=np.uint32), is_transition=2)
    assert list(vals) == [8, 2, 6]  # 3*5-7 ; 2*(9-8) ; 3*2
    L.vgpu_air_free(air)


def test_air_of_degree_five_and_nine_are_accepted_with_their_quotient_degree():
    """get_log_quotient_degree (machine/src/quotient.rs): degree 5 -> 2, degree 9 -> 3; both are within what k_quotient_general does."""
    from conftest import pow_machine

    mach, codes = pow_machine([("pow5", 5, False), ("pow9", 9, True)])
    assert codes == [0, 0]
    assert mach.chip_info(0)["max_degree"] == 5 and mach.chip_info(0)["log_quotient_degree"] == 2
    assert mach.chip_info(1)["max_degree"] == 9 and mach.chip_info(1)["log_quotient_degree"] == 3 and mach.chip_info(1)["constraints"] == 3
    from conftest import pow_trace

    tr = pow_trace(8, 9, 3)
    for i in range(8):
        vals = mach.eval_constraints(1, tr[i], tr[(i + 1) % 8], is_first=int(i == 0), is_last=int(i == 7), is_transition=int(i != 7))
        assert not vals.any()


def test_air_of_degree_above_nine_is_refused_when_pushed():
    """log_quotient_degree 1..3 is what the device quotient implements: a foreign AIR of higher degree is refused at
    vgpu_machine_push_air with a message that says so (not at prover creation, not silently)."""
    from conftest import pow_machine

    L = va.lib()
    mach, codes = pow_machine([("pow10", 10, False)])
    assert codes == [-4]  # VGPU_ERR_UNSUPPORTED
    msg = L.vgpu_last_error().decode()
    assert "pow10" in msg and "log_quotient_degree 4" in msg and "degree 10" in msg
    assert mach.num_chips == 0


def test_status_codes_and_error_messages():
    L = va.lib()
    assert L.vgpu_machine_chip_info(va.Machine.basic()._h, ctypes.c_uint32(99), (ctypes.c_uint32 * 8)()) == -1
    assert b"chip" in L.vgpu_last_error()
    assert L.vgpu_workload_fib(ctypes.c_uint32(5), None) == -1


def test_workload_shapes_follow_the_survey_formulas():
    for n in (1, 25, 100):
        w = va.Workload.fib(n)
        assert w.cycles == 17 + 7 * n and w.add_ops == 4 * n + 5 and w.mem_ops == 15 * n + 26
        hs = [m.shape[0] for m in w.main_traces()]
        assert all(h & (h - 1) == 0 for h in hs)
        assert hs[5] == 1024 and hs[12] == 256
    prep = va.Workload.fib(25).preprocessed()
    assert [c for c, _ in prep] == [1, 12] and prep[0][1].shape == (32, 7) and prep[1][1].shape == (256, 1)
    assert list(prep[1][1][:, 0]) == list(range(256))


def test_bench_algorithmic_bytes_matches_survey():
    sys.path.insert(0, ROOT)
    import bench

    ms = [4, 0, 1, 5, 5, 1, 1, 2, 1, 1, 1, 1, 1, 1]
    widths = [51, 1, 14, 16, 16, 18, 14, 28, 45, 14, 79, 7, 2, 6]
    heights = [1 << 20, 32, 1 << 22, 1 << 20, 1, 1024, 1, 1, 1, 1, 1, 1, 256, 1]
    b = bench.algorithmic_bytes_per_proof(list(zip(heights, widths, ms)))
    assert abs(b / 1e9 - 16.84) < 0.02  # C2 = 16.84 GB
    for k, n in bench.FIB_N.items():
        assert (1 << (k - 1)) < 17 + 7 * n <= (1 << k)


def test_workload_oplog_shape():
    # the logs the device trace generators consume: cpu records in clock order, each pointing at its memory operations
    w = va.Workload.alu(10)
    d = w.oplog()
    assert d.n_cpu == w.cycles and d.n_mem == w.mem_ops
    assert [int(x) for x in d.n_alu] == [30, 10, 10, 30]  # add (3/iter), sub, lt, bitwise (3/iter)
    cpu = np.ctypeslib.as_array(ctypes.cast(d.cpu, ctypes.POINTER(ctypes.c_uint32)), shape=(int(d.n_cpu), 12))
    mem = np.ctypeslib.as_array(ctypes.cast(d.mem, ctypes.POINTER(ctypes.c_uint32)), shape=(int(d.n_mem), 4))
    first = cpu[:, 11]
    assert first[0] == 0 and np.all(np.diff(first.astype(np.int64)) >= 0) and np.all(np.diff(first.astype(np.int64)) <= 3)
    assert np.all(np.diff(mem[:, 0].astype(np.int64)) >= 0)  # (clk, issue) order
    has = np.nonzero(np.diff(np.append(first, d.n_mem).astype(np.int64)) > 0)[0]
    assert np.all(mem[first[has], 0] == has)  # a record's first memory operation carries its clock
    assert cpu[-1, 8] == 9  # last record is STOP


def test_the_five_remaining_chips_generate_trace_restatement():
    """mul / div / shift / com / output rows of the workload VM against rows written out by hand from the reference's rules
    (alu_u32/src/mul/mod.rs:38-66,125-132; div/mod.rs:84-103; shift/mod.rs:143-162; com/mod.rs:87-103; output/src/lib.rs:37-100)."""
    w = va.Workload.named("mixed_ops:3")
    mt = w.main_traces()
    d = w.oplog()
    assert [int(x) for x in d.n_alu2] == [18, 12, 12, 6] and int(d.n_output) == 4  # per iteration: 4 mul + 2 shl | 2 div + shr + sra | 4 shifts | ne, eq | 1 write (+ 1 final)
    x, y = 0x12345678, 0x9ABCDEF1
    be = lambda v: [(v >> 24) & 255, (v >> 16) & 255, (v >> 8) & 255, v & 255]
    mul = mt[5]
    assert mul.shape == (1024, 18) and mul[:, 17].tolist() == list(range(1, 1025))  # counter = row + 1 on every row, 1024 rows at least
    assert mul[0].tolist() == be(x) + be(y) + be((x * y) & 0xFFFFFFFF) + [0, 0, 1, 0, 0, 1]            # Mul32: inputs, output, r = s = 0, is_mul
    assert mul[1].tolist() == be(x) + be(y) + be((x * y) >> 32) + [0, 0, 0, 0, 1, 2]                    # Mulhu32
    assert mul[2].tolist() == be(y) + be(x) + be((x * y) >> 32) + [0, 0, 0, 1, 0, 3]                    # Mulhs32: zero-extended operands (core.rs:146-158)
    assert mul[4].tolist() == be(x) + be(32) + be((x << 5) & 0xFFFFFFFF) + [0, 0, 1, 0, 0, 5]          # the Mul32 a SHL leaves: times 2^5
    assert mul[18:, :17].max() == 0
    div = mt[6]
    assert div.shape == (16, 14) and div[:12, :12].max() == 0                                            # flags only ("TODO: Fill in other columns")
    assert div[:4, 12:].tolist() == [[1, 0], [0, 1], [1, 0], [0, 1]]                                     # div, sdiv, Div32 of SHR, SDiv32 of SRA
    sh = mt[7]
    assert sh.shape == (16, 28)
    assert sh[0].tolist() == be(x) + be(5) + be((x << 5) & 0xFFFFFFFF) + [1, 0, 1, 0, 0, 0, 0, 0] + [5] + be(32) + [1, 0, 0]   # bits of 5, temp_1 = 5 (the exponent), 2^5
    sra = ((y - (1 << 32)) >> 5) & 0xFFFFFFFF
    assert sh[2].tolist() == be(y) + be(5) + be(sra) + [1, 0, 1, 0, 0, 0, 0, 0] + [5] + be(32) + [0, 1, 0]                      # SRA is logged as Shr32 (shift/mod.rs:325-328)
    com = mt[9]
    assert com.shape == (8, 14) and com[:6, :12].max() == 0 and com[:2, 12:].tolist() == [[1, 0], [0, 1]]
    out = mt[11]
    clk = [int(v) for v in np.ctypeslib.as_array(ctypes.cast(d.output, ctypes.POINTER(ctypes.c_uint32)), shape=(4, 2))[:, 0]]
    rows = []
    for i in range(3):
        num = (clk[i + 1] - clk[i]) // 4 + 1
        cl = [clk[i]] + [clk[i] + 4 * (k + 1) for k in range(1, num)]
        for k in range(num):
            rows.append([cl[k], 0, 0, (cl[k + 1] if k + 1 < num else clk[i + 1]) - cl[k]])
        rows[-num][1:3] = [None, 1]
    assert out.shape[1] == 7 and out[:, 4:].max() == 0                                                   # counter, counter_mult, opcode: never written
    assert out.shape[0] == 1 << (len(rows)).bit_length()
    for r, want in zip(out[: len(rows)], rows):
        assert int(r[0]) == want[0] and int(r[3]) == want[3] % va.P and int(r[2]) == want[2] and (want[1] is None or int(r[1]) == 0)
    assert out[len(rows), :3].tolist()[0] == clk[3] and out[len(rows), 2] == 1 and out[len(rows) + 1:].max() == 0
    # range checks come from the INSTRUCTIONS add, sub, mul*, div* (not from the Mul32 / Div32 a shift leaves): 4 bytes each
    assert int(mt[12][:, 0].sum()) == 4 * 3 * (4 + 2 + 3 + 1)  # per iteration: mul, mulhu, mulhs, muli | div, sdiv | add, add (y), add (i) | sub


def test_ffi_captured_machine_equals_in_tree_machine():
    # every chip's eval forwarded call by call through vgpu_air_* compiles to the same program as the in-tree capture,
    # and evaluates to the same constraint values on random rows
    a, b = va.Machine.basic(), va.Machine.basic_via_ffi()
    assert a.num_chips == b.num_chips == va.NUM_CHIPS
    rng = np.random.default_rng(3)
    for chip in range(va.NUM_CHIPS):
        ia, ib = a.chip_info(chip), b.chip_info(chip)
        assert ia == ib, va.CHIP_NAMES[chip]
        loc = rng.integers(0, va.P, size=ia["width"], dtype=np.uint32)
        nxt = rng.integers(0, va.P, size=ia["width"], dtype=np.uint32)
        ea = a.eval_constraints(chip, loc, nxt, is_first=5, is_last=7, is_transition=11)
        eb = b.eval_constraints(chip, loc, nxt, is_first=5, is_last=7, is_transition=11)
        assert list(ea) == list(eb), va.CHIP_NAMES[chip]


@pytest.mark.parametrize("flags", [0, va.CBOR_CANONICAL_FIELDS, va.CBOR_PLAIN_DIGESTS, va.CBOR_CANONICAL_FIELDS | va.CBOR_PLAIN_DIGESTS])
def test_proof_cbor_matches_independent_encoder(flags):
    # the product's CBOR image of MachineProof vs the test-side serde data model + generic encoder
    from oracle import cbor_ref

    w = va.Workload.fib(25)
    prep = w.preprocessed()
    words = po.prove_basic(w.main_traces(), prep[0][1], prep[1][1], va.poseidon_round_constants(), num_queries=5).words
    got = va.proof_cbor(words, flags)
    model = cbor_ref.model(words, bool(flags & va.CBOR_CANONICAL_FIELDS), bool(flags & va.CBOR_PLAIN_DIGESTS))
    assert got == cbor_ref.encode(model)
    back = cbor_ref.decode(got)
    assert back == model and list(back) == ["commitments", "opening_proof", "chip_proofs"]  # machine/src/proof.rs:15-19
    assert list(back["chip_proofs"][0]["opened_values"]) == ["preprocessed_local", "preprocessed_next", "trace_local", "trace_next", "permutation_local",
                                                              "permutation_next", "quotient_chunks"]  # :37-44
    assert back["chip_proofs"][0]["log_degree"] == 8 and len(back["chip_proofs"]) == va.NUM_CHIPS
    assert len(back["opening_proof"]["fri_proof"]["query_proofs"]) == 5
    if not flags & va.CBOR_CANONICAL_FIELDS:  # the derive on `struct BabyBear { value: u32 }`: the raw Montgomery word
        canonical = cbor_ref.model(words, True, True)["opening_proof"]["fri_proof"]["pow_witness"]
        assert back["opening_proof"]["fri_proof"]["pow_witness"] == {"value": canonical * (1 << 32) % va.P}
    with pytest.raises(va.VgpuError):
        va.proof_cbor(words[:-3], flags)  # truncated proof
    with pytest.raises(va.VgpuError):
        va.proof_cbor(np.concatenate([words, [0]]).astype(np.uint32), flags)  # trailing word


def test_committed_bench_line_follows_the_contract():
    # profiles/r06_bench_full.json is the line bench.py printed on the MI355X in the round's profile session: every field the driver reads is there
    import json

    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_bench_full.json")) as f:
        d = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "proofs/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"] and d["dtype"] == "u32" and d["data"].startswith("synthetic")
    assert abs(d["value"] - d["n_gpus"] * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    # the binding roofline first (the dominant kernel is Keccak: integer VALU), the HBM figures of the contract under `hbm`
    assert r["bound"] == "valu" and r["kernel"] == "k_keccak_compress" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # the peak is an ISSUE BOUND (the guide's 2 / 4 cycles per full- / half-rate wave64 instruction on 1024 SIMDs at 2.4 GHz), reproducible from the
    # permutation's instruction mix; the in-register figure is reported as what it is (round-3 verdict, Weak 4)
    assert abs(r["peak"] - 1024 * 2.4e9 * 64 / (24 * (122 * 2 + 56 * 4))) < 1 and "issue bound" in r["peak_is"]
    v1 = d["roofline_one_proof_in_flight"]["valu"]
    assert v1["frac"] < 1 and v1["frac_of_measured_issue_peak"] < 1 and "NOT a bound" in v1["code_ceiling_is"]
    assert abs(r["achieved"] - d["valu_roofline"]["permutations_per_launch"] / (r["avg_launch_ms"] * 1e-3)) < 1e-6 * r["achieved"]
    h = r["hbm"]
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and h["peak"] == 8000.0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-9 and h["traffic"] is not None
    assert "profiles/r0" in h["traffic_source"] and "_pmc.json" in h["traffic_source"] and "not measured in this run" in h["traffic_source"]
    assert abs(h["traffic"] / h["algorithmic_bytes_per_launch"] - 1.0) < 0.02  # no wasted re-reads in the dominant kernel
    v = d["valu_roofline"]
    assert v["kernel"] == r["kernel"] and v["microbench"].startswith("profiles/") and abs(v["frac"] - r["frac"]) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port", "port-simd") and c["cores"] >= 1 and c["unit"] == "proofs/s" and "sample" in c
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_c2_fib149794.json")) as f:
        fixture_sha = json.load(f)["proof_sha256"]
    assert c["proof_sha256"] == fixture_sha  # the baseline proved the very segment the fixture pins
    # round 5: the line is self-proving — the GPU proof of the timed region carries its own sha256, equal to the CPU leg's of the same run
    assert d["proof_sha256"] == fixture_sha and d["same_proof_as_cpu_baseline"] is True and "scalar_port_seconds_per_proof" not in c
    assert "2^20 cpu rows" in c["sample"] and "scaled" not in c["sample"]  # the headline segment itself, no extrapolation
    # no published number exists: vs_baseline is the ratio to the CPU baseline of the SAME run and says so
    assert abs(d["vs_baseline"] - d["value"] / c["value"]) < 1e-6 * d["vs_baseline"] and "kind: port-simd" in d["vs_baseline_is"] and d["vs_baseline"] > 20
    assert d["proof_checked_by_vgpu_verify"]["accepted"] is True
    p = d["pcie_inclusive"]
    assert p["value"] < d["value"] and p["from_pinned_host_memory"]["value"] < d["value"] * 1.02  # the host-resident readings, beside the headline
    # round 6: the line carries the SUSTAINED figure of >= 5 s behind the contract region, its own clock, and the dominant kernel's roofline at both clocks
    su = d["sustained"]
    assert su["seconds"] >= 5.0 and su["steps"] >= 300 and abs(su["proofs_per_s"] - su["steps"] / su["seconds"]) < 1e-6 * su["proofs_per_s"]
    assert su["proofs_per_s"] <= d["value"] * 1.02 and 1.5 < su["shader_clock_GHz"] <= 2.45
    sr = su["roofline"]
    assert sr["kernel"] == r["kernel"] and 0 < sr["frac_at_guide_clock"] < sr["frac_at_measured_clock"] < 1
    assert abs(sr["frac_at_measured_clock"] - sr["frac_at_guide_clock"] * 2.4 / su["shader_clock_GHz"]) < 1e-9


def test_bench_reads_its_valu_peaks_from_the_committed_microbench_report():
    import bench

    f = bench.microbench_facts()
    assert f["source"].startswith("profiles/") and 0.9e12 < f["full_rate"] < 1.2e12 and 0.5e12 < f["half_rate"] < 0.7e12 and 8e9 < f["keccak_perm_per_s"] < 12e9


@pytest.mark.parametrize("hash_kind", [va.HASH_KECCAK256, va.HASH_POSEIDON16])
def test_host_verify_multi_batches_accepts_oracle_openings_and_rejects_tampering(rc, hash_kind):
    """The product's host verifier (vgpu_verify_multi_batches: own Keccak, own MMCS / FRI checks) against openings produced by the
    oracle's pcs.commit_batches + open_multi_batches — two independent implementations of prover and verifier meeting."""
    rng = np.random.default_rng(21)
    P = va.P
    rounds = [[rng.integers(0, P, (32, 40), dtype=np.uint32), rng.integers(0, P, (32, 3), dtype=np.uint32)],
              [rng.integers(0, P, (64, 2), dtype=np.uint32), rng.integers(0, P, (4, 9), dtype=np.uint32), rng.integers(0, P, (1, 4), dtype=np.uint32)]]
    ext = lambda: [int(x) for x in rng.integers(1, P, 5)]
    a, b, c = ext(), ext(), ext()
    points = [[[a], [a, b]], [[c, a], [b], [a, b, c]]]
    obs = [int(x) for x in rng.integers(0, P, 5)]
    po.set_mmcs_hash(1 if hash_kind == va.HASH_POSEIDON16 else 0, rc)
    try:
        roots, values, proof = po.pcs_open(rounds, points, rc, observed=obs, num_queries=6, pow_bits=4)
    finally:
        po.set_mmcs_hash(0)
    heights = [[m.shape[0] for m in rnd] for rnd in rounds]
    widths = [[m.shape[1] for m in rnd] for rnd in rounds]

    def verify(values, proof, roots=roots):
        ch = va.Challenger(rc)
        ch.observe(obs)
        return va.verify_multi_batches(list(roots), heights, widths, points, values, proof, ch, rc, num_queries=6, pow_bits=4, hash_kind=hash_kind)

    assert verify(values, proof) is None
    other = va.HASH_POSEIDON16 if hash_kind == va.HASH_KECCAK256 else va.HASH_KECCAK256
    ch = va.Challenger(rc)
    ch.observe(obs)
    assert va.verify_multi_batches(list(roots), heights, widths, points, values, proof, ch, rc, num_queries=6, pow_bits=4, hash_kind=other) is not None
    for pos in list(range(0, values.size, 37)) + [values.size - 1]:   # every opened value is bound
        bad = values.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        assert verify(bad, proof) is not None, pos
    for pos in list(range(0, proof.size, 41)) + [proof.size - 1]:     # and the proof itself
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        assert verify(values, bad) is not None, pos
    bad_roots = roots.copy()
    bad_roots[1, 3] = (int(bad_roots[1, 3]) + 1) % P
    assert verify(values, proof, bad_roots) is not None
    assert verify(values, proof[:-1]) is not None and verify(values[:-5], proof) is not None


def test_sparse_partial_round_poseidon_is_the_same_permutation():
    """The Poseidon-MMCS kernels run the 22 partial rounds in sparse-matrix form (host/poseidon_opt.hpp); its host twin, built from
    the same tables, against the oracle's plain permutation — for the default and two other sets of round constants."""
    L = va.lib()
    rng = np.random.default_rng(8)
    for seed in (None, 17, 99):
        r = va.poseidon_round_constants() if seed is None else va.poseidon_round_constants(seed)
        rp = np.ascontiguousarray(r, dtype=np.uint32)
        for _ in range(12):
            st = rng.integers(0, va.P, 16, dtype=np.uint32)
            got = st.copy()
            assert L.vgpu_poseidon16_permute_sparse(rp.ctypes.data_as(va.c_u32p), got.ctypes.data_as(va.c_u32p)) == 0
            assert np.array_equal(got, po.poseidon_permute(r, st))


def test_c99_host_links_and_runs(tmp_path):
    """include/vgpu.h is plain C (C99, -pedantic) and a C host drives the library without Python or C++: machine description, transcript
    (its sampling convention checked against the exported permutation), prover creation — which must refuse with VGPU_ERR_HIP on a
    box without a device (no CPU fallback); on a GPU box the same program proves prove_fibonacci."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_host")
    libdir = os.path.join(root, "valida_amd")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_host.c"),
                    "-o", exe, "-L", libdir, "-lvgpu", "-Wl,-rpath," + libdir], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "chip 0: width 51" in r.stdout
    assert ("no device: code -3" in r.stdout) or ("proof words" in r.stdout)


def test_makefile_builds_the_same_sources_as_build_py():
    """Two build routes (valida_amd/build.py for the Python tests, the top-level Makefile for a Rust / C host's build script) must not drift."""
    from valida_amd import build as b

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mk = open(os.path.join(root, "Makefile")).read()
    m = re.search(r"^SRCS := (.*?)\n(?=\S)", mk, re.S | re.M)
    srcs = m.group(1).replace("\\\n", " ").split()
    assert sorted(srcs) == sorted(b.SOURCES)
    for flag in b.FLAGS:
        if flag.startswith("--offload-arch"):
            continue
        assert flag in mk, flag


def test_rust_sys_binding_is_generated_from_the_header_and_complete(tmp_path):
    """bindings/rust/src/lib.rs (the thin #[repr(C)] FFI layer of BASELINE.json's north star; no Rust toolchain here, so it is generated, not
    compiled): up to date with include/vgpu.h, declares every symbol the library exports, and every #[repr(C)] struct has the size and field
    offsets gcc gives its C twin."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import gen_rust_bindings as g

    text, info = g.generate()
    assert open(os.path.join(root, "bindings", "rust", "src", "lib.rs")).read() == text, "run python tools/gen_rust_bindings.py"
    exported = set(re.findall(r"\bT (vgpu_\w+)", subprocess.run(["nm", "-D", "--defined-only", os.path.join(root, "valida_amd", "libvgpu.so")], capture_output=True, text=True, check=True).stdout))
    declared = set(re.findall(r"pub fn (vgpu_\w+)\(", text))
    assert declared == exported
    # layout: sizes / offsets from gcc against the System V layout of the Rust field types
    size_align = {"u8": (1, 1), "u32": (4, 4), "i32": (4, 4), "u64": (8, 8), "i64": (8, 8), "f64": (8, 8)}
    structs = {}
    for m in re.finditer(r"pub struct (\w+) \{  // (\w+)\n(.*?)\n\}", text, re.S):
        fields = re.findall(r"pub (\w+): (Option<[^\n]+>|[^,\n]+),\n", m.group(3) + "\n")
        structs[m.group(2)] = (m.group(1), fields)

    def layout(ty):
        ty = ty.strip()
        a = re.match(r"\[(.+); (\d+)\]$", ty)
        if a:
            s, al = layout(a.group(1))
            return s * int(a.group(2)), al
        if ty.startswith("*") or ty.startswith("Option<unsafe extern"):
            return 8, 8
        if ty in size_align:
            return size_align[ty]
        for cname, (rname, fields) in structs.items():
            if rname == ty:
                return struct_layout(fields)[0:2]
        raise AssertionError(ty)

    def struct_layout(fields):
        off, al_max, offs = 0, 1, []
        for _, ty in fields:
            s, al = layout(ty)
            off = (off + al - 1) // al * al
            offs.append(off)
            off += s
            al_max = max(al_max, al)
        return (off + al_max - 1) // al_max * al_max, al_max, offs

    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "vgpu.h"', "int main(void) {"]
    for cname, (_, fields) in structs.items():
        prog.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for fname, _ in fields:
            prog.append('printf(" %%zu", offsetof(%s, %s));' % (cname, fname))
        prog.append('printf("\\n");')
    prog.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", exe], check=True, capture_output=True)
    lines = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert len(lines) == len(structs) >= 8
    for line in lines:
        cname, size, *offs = line.split()
        want_size, _, want_offs = struct_layout(structs[cname][1])
        assert (int(size), [int(o) for o in offs]) == (want_size, want_offs), cname


def test_bench_prices_the_poseidon_kernels_with_the_kernels_own_instruction_model():
    """bench.py's Poseidon roofline (permutations/s against an issue model) uses the per-permutation instruction counts that
    kernels/poseidon_mmcs.hip reports to the profiler: the two files must hold the same numbers."""
    import re
    src = open(os.path.join(ROOT, "valida_amd", "csrc", "kernels", "poseidon_mmcs.hip")).read()
    m = re.search(r"POSEIDON_HALF_PER_PERM = ([0-9.+ ]+), POSEIDON_FULL_PER_PERM = ([0-9.+ ]+);", src)
    half, full = (sum(float(x) for x in g.split("+")) for g in m.groups())
    bench = open(os.path.join(ROOT, "bench.py")).read()
    b = re.search(r"POSEIDON_HALF_PER_PERM, POSEIDON_FULL_PER_PERM = (\d+), (\d+)", bench)
    assert (int(b.group(1)), int(b.group(2))) == (int(half), int(full))


def test_every_environment_switch_of_the_library_is_documented():
    """INTEGRATION.md lists the library's environment switches with the A/B record each one reproduces: a new getenv("VGPU_…") in csrc must
    come with its row."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = set()
    for path in glob.glob(os.path.join(root, "valida_amd", "csrc", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.rsplit(".", 1)[-1] in ("hip", "hpp", "cpp", "h"):
            found |= set(re.findall(r'getenv\("(VGPU_[A-Z0-9_]+)"\)', open(path, errors="replace").read()))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(v for v in found if v not in doc)
    assert found and not missing, missing
    # round-5 verdict, item 7: at most twelve switches; every one that selects another kernel has a row in the GPU suite's
    # test_every_ab_switch_gives_the_oracles_proof, the rest are the documented non-arithmetic ones; the refuted experiments are gone from the sources
    assert len(found) <= 12, sorted(found)
    non_functional = {"VGPU_SPIN_WAIT", "VGPU_PROF_QUOTIENT_BY_CHIP", "VGPU_COMM_TIMEOUT_MS", "VGPU_FAILPOINT", "VGPU_TESTING"}
    gpu_tests = open(os.path.join(root, "tests", "test_gpu_parity.py")).read()
    rows = set(re.findall(r'"(VGPU_[A-Z0-9_]+)=', gpu_tests[gpu_tests.index("AB_SETTINGS = ["):gpu_tests.index("def test_every_ab_switch_gives_the_oracles_proof")]))
    assert found - non_functional == rows, (sorted(found - non_functional), sorted(rows))
    for path in glob.glob(os.path.join(root, "valida_amd", "csrc", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.rsplit(".", 1)[-1] in ("hip", "hpp", "cpp", "h"):
            txt = open(path, errors="replace").read()
            for gone in ("STANDIN_FUSE", "EXP_SKIP_ROUNDS", "VGPU_MID12_", "VGPU_LDE_GROUP_MB", "VGPU_STREAM_PRIO", "VGPU_QUEUE_MAP"):
                assert gone not in txt, (path, gone)


def test_build_says_what_it_did():
    """valida_amd/build.py records per source whether hipcc compiled it or an up-to-date object was reused (build.LAST); __graft_entry__.build() forces a
    real compile and prints that record, so that "does it build" is answered by a compile, not by the .so that travels with the tree (round-4 verdict, item 8)."""
    from valida_amd import build as b

    lib_path = b.build_vgpu()  # up to date here (conftest built it): nothing compiled, everything accounted for
    assert os.path.exists(lib_path)
    assert sorted(b.LAST["reused"] + b.LAST["compiled"]) == sorted(b.SOURCES) and set(b.LAST) == {"compiled", "reused", "linked"}
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "__graft_entry__.py")) as f:
        src = f.read()
    assert "build_vgpu(force=" in src and "compiled %d of %d sources" in src


def test_bench_launcher_spawns_ranks_and_returns_the_first_failure():
    """`python bench.py --gpus 2` without a launcher's WORLD_SIZE becomes the launcher itself (round-5 verdict, item 1).  Without a GPU its two ranks refuse to run
    ("the product path has no CPU fallback"); the launcher must come back with their status, name the rank, and print no JSON line."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run (tests/test_gpu_parity.py covers that form)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stderr[-500:])
    assert r.stderr.count("bench.py needs a HIP device") == 2 and "ended with status 1; ending the other ranks" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
