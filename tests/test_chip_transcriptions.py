"""The chips are INPUT to the prover (Chip::eval, all_interactions, the column maps).  Two independent transcriptions of them
exist here — the product's (valida_amd/csrc/chips/basic_machine.hpp: enum column constants, templates compiled to the device
program) and the oracle's (oracle/chips.hpp: the reference's column STRUCTS borrowed over a row) — and one extraction straight
from the reference's Rust sources (tests/golden/reference_shapes.json, made by tools/extract_reference_shapes.py).  These tests
hold the three against each other; none needs a GPU."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import valida_amd as va
from oracle import pyoracle as po

P = va.P
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def shapes():
    with open(os.path.join(HERE, "golden", "reference_shapes.json")) as f:
        return json.load(f)


def test_extractor_reproduces_the_committed_fixture(shapes):
    """Where the reference is present (this container, not the GPU box) the fixture must be what the extractor produces now."""
    if not os.path.isdir("/root/reference"):
        pytest.skip("reference sources are not on this machine")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "shapes.json")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "extract_reference_shapes.py"), "/root/reference", out], check=True, capture_output=True)
        with open(out) as f:
            assert json.load(f) == shapes


def test_shapes_counts_and_buses_match_the_reference(machine, shapes):
    assert shapes["constants"] == {"OPERAND_ELEMENTS": 5, "INSTRUCTION_ELEMENTS": 6, "CPU_MEMORY_CHANNELS": 3, "MEMORY_CELL_BYTES": 4, "LOOKUP_DEGREE_BOUND": 3}
    order = shapes["chip_order"]
    assert len(order) == machine.num_chips == 14
    for i, name in enumerate(order):
        ref = shapes["chips"][name]
        info = machine.chip_info(i)
        assert (info["width"], info["preprocessed_width"]) == (ref["width"], ref["preprocessed_width"]), name
        assert po.chip_shape(i) == (ref["width"], ref["preprocessed_width"]), name
        assert info["constraints"] == ref["num_constraints"], name
        assert len(po.eval_constraints(i, np.zeros(ref["width"], np.uint32), np.zeros(ref["width"], np.uint32))) == ref["num_constraints"], name
        want = [(it["send"], True, shapes["buses"][it["bus"]][1], it["n_fields"]) for it in ref["interactions"]]
        assert all(shapes["buses"][it["bus"]][0] == "Global" for it in ref["interactions"])
        for side, its in (("product", machine.interactions(i)), ("oracle", va.decode_interaction_words(po.interactions(i)))):
            assert [(x["send"], x["global"], x["bus"], len(x["fields"])) for x in its] == want, (name, side)
        # max constraint degree 3 everywhere -> log_quotient_degree 1 (machine/src/lib.rs:36 LOOKUP_DEGREE_BOUND)
        assert info["log_quotient_degree"] == 1 and info["max_degree"] <= shapes["constants"]["LOOKUP_DEGREE_BOUND"]


def test_two_transcriptions_agree_on_random_rows(machine):
    rng = np.random.default_rng(2024)
    for chip in range(14):
        w = machine.chip_info(chip)["width"]
        for _ in range(6):
            local, nxt = rng.integers(0, P, w, dtype=np.uint32), rng.integers(0, P, w, dtype=np.uint32)
            f, l, t = (int(x) for x in rng.integers(0, P, 3))
            got = machine.eval_constraints(chip, local, nxt, is_first=f, is_last=l, is_transition=t)
            want = po.eval_constraints(chip, local, nxt, is_first=f, is_last=l, is_transition=t)
            assert np.array_equal(got, want), chip
        assert machine.interactions(chip) == va.decode_interaction_words(po.interactions(chip)), chip


def test_row_selectors_filter_the_constraints_the_reference_filters(machine, shapes):
    """Constraint k vanishes identically when selector s is zero iff the reference wraps it in when_<s>: pins the ORDER of the
    selector-filtered constraints of both transcriptions to the Rust source (the order fixes the alpha powers)."""
    rng = np.random.default_rng(7)
    for i, name in enumerate(shapes["chip_order"]):
        ref = shapes["chips"][name]["constraints"]
        if not ref:
            continue
        w = shapes["chips"][name]["width"]
        for sel in ("first", "last", "transition"):
            zero_always = None
            for _ in range(3):
                local, nxt = rng.integers(1, P, w, dtype=np.uint32), rng.integers(1, P, w, dtype=np.uint32)
                vals = {s: int(rng.integers(1, P)) for s in ("first", "last", "transition")}
                vals[sel] = 0
                for ev in (machine.eval_constraints, po.eval_constraints):
                    v = ev(i, local, nxt, is_first=vals["first"], is_last=vals["last"], is_transition=vals["transition"])
                    z = v == 0
                    zero_always = z if zero_always is None else (zero_always & z)
            want = np.array([sel in c.split(":")[0].split("+") if ":" in c else False for c in ref])
            assert np.array_equal(zero_always, want), (name, sel)


def test_oracle_shares_no_source_with_the_product():
    """oracle/ includes nothing under valida_amd/ (its chips are its own transcription), and the product nothing under oracle/."""
    import re

    for d, banned in (("oracle", r"valida_amd|vchips::|vair::"), (os.path.join("valida_amd", "csrc"), r"oracle/|oracle::")):
        for base, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith((".hpp", ".cpp", ".hip", ".h")) or f == "Makefile":
                    with open(os.path.join(base, f)) as fh:
                        for n, line in enumerate(fh, 1):
                            code = line.split("//")[0]
                            assert not (re.search(r"#include", code) and re.search(banned, code)), (f, n, line)
                            assert not re.search(r"\b(vchips|vair)::", code) or d != "oracle", (f, n, line)


def test_oracle_column_structs_have_the_reference_offsets(shapes):
    """Every field of every column struct of oracle/chips.hpp sits at the offset the reference's struct gives it (by NAME)."""
    struct_of = {"cpu": "CpuCols", "program": "ProgramCols", "mem": "MemoryCols", "add_u32": "Add32Cols", "sub_u32": "Sub32Cols", "mul_u32": "Mul32Cols",
                 "div_u32": "Div32Cols", "shift_u32": "Shift32Cols", "lt_u32": "Lt32Cols", "com_u32": "Com32Cols", "bitwise_u32": "Bitwise32Cols",
                 "output": "OutputCols", "range": "RangeCols", "static_data": "StaticDataCols"}
    lines, expect = [], []
    for name in shapes["chip_order"]:
        ref = shapes["chips"][name]
        assert ref["struct"] == struct_of[name]
        lines.append('  { auto M = col_map<%s>(); static_assert(num_cols<%s>() == %d, "width");' % (ref["struct"], ref["struct"], ref["width"]))
        for path, (off, size) in ref["fields"].items():
            # first and last scalar of the field: raw view of the struct as an index array, addressed through the member
            lines.append('    { const size_t* base = reinterpret_cast<const size_t*>(&M); const size_t* f = reinterpret_cast<const size_t*>(&M.%s); '
                         'printf("%%zu %%zu %%zu\\n", (size_t)(f - base), *f, sizeof(M.%s) / sizeof(size_t)); }' % (path, path))
            expect.append((name, path, off, size))
        lines.append("  }")
    src = '#include <cstdio>\n#include "%s"\nusing namespace oracle::chips;\nint main() {\n%s\n  return 0;\n}\n' % (os.path.join(ROOT, "oracle", "chips.hpp"), "\n".join(lines))
    with tempfile.TemporaryDirectory() as d:
        cpp, exe = os.path.join(d, "cols.cpp"), os.path.join(d, "cols")
        with open(cpp, "w") as f:
            f.write(src)
        subprocess.run(["g++", "-std=c++17", "-O0", "-o", exe, cpp], check=True, capture_output=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split("\n")
    for (name, path, off, size), line in zip(expect, out):
        got_off, got_idx, got_size = (int(x) for x in line.split())
        assert (got_off, got_idx, got_size) == (off, off, size), (name, path)


def test_interaction_columns_are_the_reference_fields(machine, shapes):
    """The columns each interaction reads, spelled with the reference's field NAMES and resolved through the extracted offsets."""
    F = {n: shapes["chips"][n]["fields"] for n in shapes["chip_order"]}
    op = shapes["opcodes"]

    def col(chip, path, k=0):
        return (0, [(0, F[chip][path][0] + k, 1)])

    def word(chip, path):
        assert F[chip][path][1] == 4
        return [col(chip, path, k) for k in range(4)]

    def const(c):
        return (c, [])

    def lin(chip, pairs):
        return (0, [(0, F[chip][p][0], w) for p, w in pairs])

    want = {
        "cpu": [{"count": col("cpu", "mem_channels[%d].used" % i),
                 "fields": [col("cpu", "mem_channels[%d].is_read" % i), col("cpu", "clk"), col("cpu", "mem_channels[%d].addr" % i), const(0)] + word("cpu", "mem_channels[%d].value" % i)}
                for i in range(3)] +
               [{"count": col("cpu", "opcode_flags.is_bus_op"),
                 "fields": [col("cpu", "instruction.opcode")] + sum((word("cpu", "mem_channels[%d].value" % i) for i in range(3)), []) + [col("cpu", "chip_channel.clk_or_zero")]}],
        "program": [],
        "mem": [{"count": lin("mem", [("is_read", 1), ("is_write", 1)]),
                 "fields": [col("mem", "is_read"), col("mem", "clk"), col("mem", "addr"), col("mem", "is_static_initial")] + word("mem", "value")}],
        "mul_u32": [{"count": lin("mul_u32", [("is_mul", 1), ("is_mulhs", 1), ("is_mulhu", 1)]),
                     "fields": [lin("mul_u32", [("is_mul", op["MUL32"]), ("is_mulhs", op["MULHS32"]), ("is_mulhu", op["MULHU32"])])] + word("mul_u32", "input_1") + word("mul_u32", "input_2") + word("mul_u32", "output")}],
        "div_u32": [{"count": lin("div_u32", [("is_div", 1), ("is_sdiv", 1)]),
                     "fields": [lin("div_u32", [("is_div", op["DIV32"]), ("is_sdiv", op["SDIV32"])])] + word("div_u32", "input_1") + word("div_u32", "input_2") + word("div_u32", "output")}],
        "shift_u32": [{"count": lin("shift_u32", [("is_shl", 1), ("is_shr", 1), ("is_sra", 1)]),
                       "fields": [lin("shift_u32", [("is_shl", op["MUL32"]), ("is_shr", op["DIV32"]), ("is_sra", op["SDIV32"])])] + word("shift_u32", "input_1") + word("shift_u32", "power_of_two") + word("shift_u32", "output")},
                      {"count": lin("shift_u32", [("is_shl", 1), ("is_shr", 1), ("is_sra", 1)]),
                       "fields": [lin("shift_u32", [("is_shl", op["SHL32"]), ("is_shr", op["SHR32"]), ("is_sra", op["SRA32"])])] + word("shift_u32", "input_1") + word("shift_u32", "input_2") + word("shift_u32", "output")}],
        "lt_u32": [{"count": col("lt_u32", "multiplicity"),
                    "fields": [lin("lt_u32", [("is_lt", op["LT32"]), ("is_lte", op["LTE32"]), ("is_slt", op["SLT32"]), ("is_sle", op["SLE32"])])] + word("lt_u32", "input_1") + word("lt_u32", "input_2") + [const(0)] * 3 + [col("lt_u32", "output")]}],
        "com_u32": [{"count": lin("com_u32", [("is_ne", 1), ("is_eq", 1)]),
                     "fields": [lin("com_u32", [("is_ne", op["NE32"]), ("is_eq", op["EQ32"])])] + word("com_u32", "input_1") + word("com_u32", "input_2") + [const(0)] * 3 + [col("com_u32", "output")]}],
        "bitwise_u32": [{"count": lin("bitwise_u32", [("is_and", 1), ("is_or", 1), ("is_xor", 1)]),
                         "fields": [lin("bitwise_u32", [("is_and", op["AND32"]), ("is_or", op["OR32"]), ("is_xor", op["XOR32"])])] + word("bitwise_u32", "input_1") + word("bitwise_u32", "input_2") + word("bitwise_u32", "output")}],
        "output": [{"count": col("output", "is_real"), "fields": [col("output", "opcode")] + [const(0)] * 3 + [col("output", "value")] + [const(0)] * 8 + [col("output", "clk")]}],
        "range": [{"count": col("range", "mult"), "fields": [col("range", "counter")]}],
        "static_data": [{"count": col("static_data", "is_real"), "fields": [const(0), const(0), col("static_data", "addr"), const(1)] + word("static_data", "value")}],
    }
    for alu, opname in (("add_u32", "ADD32"), ("sub_u32", "SUB32")):
        want[alu] = [{"count": col(alu, "is_real"), "fields": [col(alu, "output", k)]} for k in range(4)] + \
                    [{"count": col(alu, "is_real"), "fields": [const(op[opname])] + word(alu, "input_1") + word(alu, "input_2") + word(alu, "output")}]
    for i, name in enumerate(shapes["chip_order"]):
        for side, its in (("product", machine.interactions(i)), ("oracle", va.decode_interaction_words(po.interactions(i)))):
            got = [{"count": (x["count"][0], [tuple(t) for t in x["count"][1]]), "fields": [(c, [tuple(t) for t in ts]) for c, ts in x["fields"]]} for x in its]
            assert got == want[name], (name, side)


def test_transcript_order_of_the_rust_prover_replays_the_proof(shapes, rc, fib25):
    """The Fiat-Shamir events of `fn prove` (basic/src/lib.rs:185-263,601-619), extracted from the Rust source IN SOURCE ORDER, replayed on a
    fresh challenger with the commitments of an actual proof, must give that proof's permutation challenges, alpha and zeta: the order in
    which the restated prover (and the device prover, whose proofs equal its) observes and samples is the reference's, not a recollection."""
    import valida_amd as va
    from oracle import pyoracle as po

    t = shapes["transcript"]
    # verify observes the same things in the same order (its names differ: the proof's commitments)
    assert [(e[0], e[2] if e[0] == "sample_ext" else None) for e in t["prove"]][:-1] == [(e[0], e[2] if e[0] == "sample_ext" else None) for e in t["verify"]][:-1]
    assert t["prove"][-1] == ["pcs", "open_multi_batches"] and t["verify"][-1] == ["pcs", "verify_multi_batches"]
    assert t["opened_rounds"] == [["main_data", "zeta_and_next"], ["perm_data", "zeta_and_next"], ["quotient_data", "zeta_exp_quotient_degree"]]
    assert t["points"] == {"zeta_and_next": "g_subgroups.map(|g| vec![zeta, zeta * g])",
                           "zeta_exp_quotient_degree": "log_quotient_degrees.map(|log_deg| vec![zeta.exp_power_of_2(log_deg)])"}
    mt, prep = fib25.main_traces(), fib25.preprocessed()
    proof = po.prove_basic(mt, prep[0][1], prep[1][1], rc, num_queries=4)
    commits = {"preprocessed_commit": proof.transcript[0:8], "main_commit": proof.words[2:10], "perm_commit": proof.words[10:18], "quotient_commit": proof.words[18:26]}
    ch = va.Challenger(rc)
    sampled = {}
    for ev in t["prove"]:
        if ev[0] == "observe":
            ch.observe(commits[ev[1]])
        elif ev[0] == "sample_ext":
            sampled[ev[1]] = ch.sample(5 * ev[2])
    assert sampled["perm_challenges"].tolist() == proof.transcript[8:23].tolist()
    assert sampled["alpha"].tolist() == proof.transcript[23:28].tolist()
    assert sampled["zeta"].tolist() == proof.transcript[28:33].tolist()
    # every chip is opened at two points in the main and permutation rounds and at one in the quotient round (the proof's five vectors per chip)
    pos = 26
    for chip in range(14):
        pos += 1
        lens = []
        for _ in range(5):
            lens.append(int(proof.words[pos]))
            pos += 1 + 5 * lens[-1]
        pos += 5
        assert lens[0] == lens[1] and lens[2] == lens[3] and lens[4] == 10
