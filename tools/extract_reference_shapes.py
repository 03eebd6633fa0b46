#!/usr/bin/env python3
"""Extract, from the reference's RUST SOURCES (read-only, /root/reference), the facts about the BasicMachine chips that the
prover's inputs must match, and write them to tests/golden/reference_shapes.json:

  * opcode numbers                                   opcodes/src/lib.rs
  * machine constants                                machine/src/lib.rs:32-36
  * chip order                                       basic/src/lib.rs (the BasicMachine struct)
  * per chip: the column struct flattened in declaration order (field path -> [offset, size]), the width
                                                     */src/columns.rs
  * per chip: the sequence of asserted constraints of `Air::eval` in source order, loops expanded, each with the row
    selectors that filter it (when_first_row / when_last_row / when_transition) and the assert kind
                                                     */src/stark.rs
  * per chip: the interactions by all_interactions slot (global sends, then global receives), their bus and number of fields
                                                     */src/lib.rs, alu_u32/src/*/mod.rs

This is the one pin of the oracle AND the product that really comes from the reference (everything below the chips is
Plonky3, which is not vendored).  The script only runs where /root/reference exists; the JSON travels.

    python tools/extract_reference_shapes.py [/root/reference] [tests/golden/reference_shapes.json]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "reference_shapes.json")


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return "\n".join(line.split("//")[0] for line in src.splitlines())


# ---------------------------------------------------------------- constants, opcodes, chip order
def constants():
    src = strip_comments(read("machine/src/lib.rs"))
    env = {}
    for name, expr in re.findall(r"pub const (\w+): usize = ([^;]+);", src):
        env[name] = eval(expr, {}, env)
    return env


def opcodes():
    src = strip_comments(read("opcodes/src/lib.rs"))
    body = re.search(r"pub enum Opcode \{(.*?)\}", src, re.S).group(1)
    out = {name: int(val) for name, val in re.findall(r"(\w+)\s*=\s*(\d+)", body)}
    out["BYTES_PER_INSTR"] = int(re.search(r"pub const BYTES_PER_INSTR: u32 = (\d+);", src).group(1))
    return out


CHIP_FILES = {  # chip field name in BasicMachine -> (crate dir, columns.rs, stark.rs, interactions file, column struct, preprocessed struct)
    "cpu": ("cpu/src/columns.rs", "cpu/src/stark.rs", "cpu/src/lib.rs", "CpuCols"),
    "program": ("program/src/columns.rs", "program/src/stark.rs", "program/src/lib.rs", "ProgramCols"),
    "mem": ("memory/src/columns.rs", "memory/src/stark.rs", "memory/src/lib.rs", "MemoryCols"),
    "add_u32": ("alu_u32/src/add/columns.rs", "alu_u32/src/add/stark.rs", "alu_u32/src/add/mod.rs", "Add32Cols"),
    "sub_u32": ("alu_u32/src/sub/columns.rs", "alu_u32/src/sub/stark.rs", "alu_u32/src/sub/mod.rs", "Sub32Cols"),
    "mul_u32": ("alu_u32/src/mul/columns.rs", "alu_u32/src/mul/stark.rs", "alu_u32/src/mul/mod.rs", "Mul32Cols"),
    "div_u32": ("alu_u32/src/div/columns.rs", "alu_u32/src/div/stark.rs", "alu_u32/src/div/mod.rs", "Div32Cols"),
    "shift_u32": ("alu_u32/src/shift/columns.rs", "alu_u32/src/shift/stark.rs", "alu_u32/src/shift/mod.rs", "Shift32Cols"),
    "lt_u32": ("alu_u32/src/lt/columns.rs", "alu_u32/src/lt/stark.rs", "alu_u32/src/lt/mod.rs", "Lt32Cols"),
    "com_u32": ("alu_u32/src/com/columns.rs", "alu_u32/src/com/stark.rs", "alu_u32/src/com/mod.rs", "Com32Cols"),
    "bitwise_u32": ("alu_u32/src/bitwise/columns.rs", "alu_u32/src/bitwise/stark.rs", "alu_u32/src/bitwise/mod.rs", "Bitwise32Cols"),
    "output": ("output/src/columns.rs", "output/src/stark.rs", "output/src/lib.rs", "OutputCols"),
    "range": ("range/src/columns.rs", "range/src/stark.rs", "range/src/lib.rs", "RangeCols"),
    "static_data": ("static_data/src/columns.rs", "static_data/src/stark.rs", "static_data/src/lib.rs", "StaticDataCols"),
}


def chip_order():
    """Field order of `pub struct BasicMachine<F>` restricted to chips = the order of the chips array (basic/src/lib.rs)."""
    src = strip_comments(read("basic/src/lib.rs"))
    body = re.search(r"pub struct BasicMachine<F[^>]*>\s*\{(.*?)\n\}", src, re.S).group(1)
    order = [name for name in re.findall(r"(\w+)\s*:", body) if name in CHIP_FILES]
    assert len(order) == 14, order
    return order


# ---------------------------------------------------------------- column structs
def parse_structs(src):
    structs = {}
    for name, body in re.findall(r"pub struct (\w+)(?:<\w+>)?\s*\{(.*?)\n\}", strip_comments(src), re.S):
        structs[name] = re.findall(r"pub (\w+)\s*:\s*([^,\n]+(?:\[[^\]]*\][^,\n]*)?),", body + ",")
    return structs


def flatten(ty, structs, consts, path, out, offset):
    ty = ty.strip()
    if ty in ("T", "F"):
        out.append((path, offset, 1))
        return offset + 1
    m = re.fullmatch(r"\[(.+);\s*([\w\s\*\+\-]+)\]", ty)
    if m:  # array: record the whole array as one field, then its elements when they are scalars of nested arrays
        n = eval(m.group(2), {}, consts)
        start = offset
        for i in range(n):
            offset = flatten(m.group(1), structs, consts, "%s[%d]" % (path, i), [] if m.group(1).strip() in ("T", "F") else out, offset)
        out.append((path, start, offset - start))  # the whole array (2-D arrays also list their rows: path[i])
        return offset
    m = re.fullmatch(r"(\w+)<\w+>", ty)
    if m and m.group(1) == "Word":
        out.append((path, offset, consts["MEMORY_CELL_BYTES"]))
        return offset + consts["MEMORY_CELL_BYTES"]
    if m and m.group(1) == "Operands":
        out.append((path, offset, consts["OPERAND_ELEMENTS"]))
        return offset + consts["OPERAND_ELEMENTS"]
    name = m.group(1) if m else ty
    if name in structs:
        for fname, fty in structs[name]:
            offset = flatten(fty, structs, consts, path + "." + fname if path else fname, out, offset)
        return offset
    raise ValueError("cannot flatten type %r at %r" % (ty, path))


def columns(chip, consts):
    colfile, _, _, struct = CHIP_FILES[chip]
    structs = parse_structs(read(colfile))
    out = []
    width = flatten(struct + "<T>", structs, consts, "", out, 0)
    shape = {"struct": struct, "width": width, "fields": {p: [o, n] for p, o, n in out}}
    if chip == "program":
        pre = []
        shape["preprocessed_width"] = flatten("ProgramPreprocessedCols<T>", structs, consts, "", pre, 0)
        shape["preprocessed_fields"] = {p: [o, n] for p, o, n in pre}
    elif chip == "range":  # preprocessed_trace = RowMajorMatrix::new_col(0..MAX) (range/src/stark.rs)
        assert "RowMajorMatrix::new_col" in read("range/src/stark.rs")
        shape["preprocessed_width"] = 1
    else:
        shape["preprocessed_width"] = 0
    return shape


# ---------------------------------------------------------------- Air::eval: the constraint sequence
def fn_body(src, name):
    m = re.search(r"fn %s\b[^{]*\{" % name, src)
    if not m:
        return None
    i, depth = m.end(), 1
    while depth:
        c = src[i]
        depth += c == "{"
        depth -= c == "}"
        i += 1
    return src[m.end():i - 1]


def loop_count(header, fields, consts):
    header = header.strip()
    m = re.fullmatch(r"for \w+ in (\w+)\.\.(\w+)", header)
    if m:
        return eval(m.group(2), {}, consts) - eval(m.group(1), {}, consts)
    # iteration over column arrays: for bit in local.bits_2.iter() / local.a.into_iter().chain(local.b...)...
    names = re.findall(r"local\s*\.\s*(\w+)(\[\w+\])?", header)
    if names:
        total = 0
        for n, idx in names:
            total += fields[n + "[0]"][1] if idx else fields[n][1]  # local.bits_1[i]: one row of a 2-D array
        return total
    raise ValueError("cannot size loop %r" % header)


def constraints_of(body, src, fields, consts, depth=0):
    """Sequence of (selectors, kind) in execution order; recurses into self.eval_* helpers and for loops."""
    out = []
    i = 0
    token = re.compile(r"(\bfor\b[^{]*\{)|(self\s*\.\s*(eval_\w+)\s*\()|(builder\b)")
    while True:
        m = token.search(body, i)
        if not m:
            break
        if m.group(1):  # for loop: find its block
            j, d = m.end(), 1
            while d:
                d += body[j] == "{"
                d -= body[j] == "}"
                j += 1
            inner = constraints_of(body[m.end():j - 1], src, fields, consts, depth + 1)
            out += inner * loop_count(m.group(1)[:-1], fields, consts)
            i = j
        elif m.group(2):
            helper = fn_body(src, m.group(3))
            out += constraints_of(helper, src, fields, consts, depth + 1)
            i = body.index(";", m.end()) + 1  # skip the call's arguments (they name `builder`)
        else:  # a builder statement: up to the terminating ';' at paren depth 0
            j, d = m.end(), 0
            while not (body[j] == ";" and d == 0):
                d += body[j] in "([{"
                d -= body[j] in ")]}"
                j += 1
                if j >= len(body):
                    break
            stmt = body[m.start():j]
            # method chain at paren depth 0
            chain, d, k = [], 0, 0
            for mm in re.finditer(r"[\(\)\[\]\{\}]|\.\s*(\w+)\s*\(", stmt):
                t = mm.group(0)
                if mm.group(1):
                    if d == 0:
                        chain.append(mm.group(1))
                    d += 1
                elif t in "([{":
                    d += 1
                elif t in ")]}":
                    d -= 1
            asserts = [c for c in chain if c.startswith("assert_")]
            if asserts:
                assert len(asserts) == 1 and chain[-1] == asserts[0], stmt
                sel = sorted({"when_first_row": "first", "when_last_row": "last", "when_transition": "transition"}[c] for c in chain if c in ("when_first_row", "when_last_row", "when_transition"))
                out.append({"selectors": sel, "filters": sum(c in ("when", "when_ne") for c in chain), "kind": asserts[0]})
            i = j + 1
    return out


def eval_constraints(chip, shape, consts):
    _, starkfile, _, _ = CHIP_FILES[chip]
    src = strip_comments(read(starkfile))
    body = fn_body(src, "eval_main" if chip == "static_data" else "eval")
    return constraints_of(body, src, shape["fields"], consts)


# ---------------------------------------------------------------- interactions
BUS = {"general_bus": 0, "program_bus": 1, "mem_bus": 2, "range_bus": 3}


def bus_ids():
    src = strip_comments(read("basic/src/lib.rs"))
    out = {}
    for name in BUS:
        m = re.search(r"fn %s\(&self\) -> BusArgument \{\s*BusArgument::(\w+)\((\d+)\)" % name, src)
        out[name] = [m.group(1), int(m.group(2))]
    return out


def count_top_level(s):
    d, n = 0, 1 if s.strip() else 0
    for c in s.strip().rstrip(","):
        d += c in "([{"
        d -= c in ")]}"
        n += c == "," and d == 0
    return n


def field_len(expr, body, consts):
    """Length of an iterator / array expression that extends `fields`."""
    expr = expr.strip()
    m = re.fullmatch(r"(\w+)", expr)
    if m:  # a let-bound name: find its definition
        d = re.search(r"let (?:mut )?%s(?:\s*:[^=]+)?\s*=\s*(.*?);" % m.group(1), body, re.S)
        assert d, expr
        return field_len(d.group(1), body, consts)
    if re.search(r"\.mem_channels\s*\.iter\(\)", expr):  # every channel's value word, flattened
        return consts["CPU_MEMORY_CHANNELS"] * consts["MEMORY_CELL_BYTES"]
    m = re.match(r"\(0\.\.([^)]+)\)\s*\.map\(", expr)
    if m:
        n = eval(m.group(1), {}, consts)
        n += len(re.findall(r"\.chain\(iter::once\(", expr))
        return n
    if re.search(r"\.0\s*\.map\(VirtualPairCol::single_main\)", expr) or re.search(r"\.value\.0\.map\(", expr):
        return consts["MEMORY_CELL_BYTES"]  # a Word's columns
    raise ValueError("cannot size %r" % expr)


def interactions_of(chip, consts):
    _, _, libfile, _ = CHIP_FILES[chip]
    src = strip_comments(read(libfile))
    out = []
    for slot in ("local_sends", "local_receives", "global_sends", "global_receives"):
        body = fn_body(src, slot)
        if body is None or not re.search(r"Interaction\s*\{", body):
            continue
        for m in re.finditer(r"Interaction\s*\{", body):
            lit = body[m.end():]
            d, j = 1, 0
            while d:
                d += lit[j] == "{"
                d -= lit[j] == "}"
                j += 1
            lit = lit[:j - 1]
            bus = re.search(r"argument_index:\s*machine\.(\w+)\(\)", lit).group(1)
            # fields: either an inline vec![..] or the let-bound `fields` vector built before the literal
            fm = re.search(r"fields:\s*vec!\[(.*?)\]", lit, re.S)
            prefix = body[:m.start()]
            if fm:
                nf = count_top_level(fm.group(1))
            else:
                # the LAST `let mut fields = vec![..]` before this literal, plus the extend / push calls after it
                starts = list(re.finditer(r"let mut fields = vec!\[(.*?)\];", prefix, re.S))
                st = starts[-1]
                nf = count_top_level(st.group(1))
                tail = prefix[st.end():]
                for e in re.finditer(r"fields\.extend\((.*?)\);", tail, re.S):
                    nf += field_len(e.group(1), prefix, consts)
                nf += len(re.findall(r"fields\.push\(", tail))
            # multiplicity: the literal sits inside a .map over a Word's columns or over (0..3)
            mult = 1
            before = prefix[-400:]
            mm = re.search(r"\(0\.\.(\w+)\)\s*\.map\(\|\w+\|\s*\{[^}]*$", prefix, re.S)
            if mm:
                mult = eval(mm.group(1), {}, consts)
            elif re.search(r"\.0\s*\.map\(\|\w+\|\s*\{[^}]*$", before, re.S):
                mult = consts["MEMORY_CELL_BYTES"]
            for _ in range(mult):
                out.append({"slot": slot, "send": slot.endswith("sends"), "bus": bus, "n_fields": nf})
    return out


# ---------------------------------------------------------------- transcript order of Machine::prove / verify
def transcript():
    """The Fiat-Shamir events of `fn prove` and `fn verify` of basic/src/lib.rs in source order: what is observed, what is sampled (with the
    loop count where a sample sits in a `for _ in 0..N`), where the PCS takes the challenger over, and the opening points per round."""
    src = strip_comments(read("basic/src/lib.rs"))
    out = {}
    for fn in ("prove", "verify"):
        body = fn_body(src, fn)
        events = []
        for m in re.finditer(r"challenger\.observe\((\w+)|(\w+)(?:\.push\()?\s*(?::[^=;]*)?=?\s*challenger\s*\.sample_ext_element|pcs\s*\.(open_multi_batches|verify_multi_batches)\(", body):
            if m.group(1):
                events.append(["observe", m.group(1)])
            elif m.group(3):
                events.append(["pcs", m.group(3)])
            else:
                name = m.group(2)
                count = 1
                head = body[:m.start()]
                loop = re.search(r"for _ in 0\.\.(\d+)\s*\{\s*$", head)
                if loop:
                    count = int(loop.group(1))
                events.append(["sample_ext", name, count])
        out[fn] = events
    body = fn_body(src, "prove")
    rounds = re.search(r"let prover_data_and_points = \[(.*?)\];", body, re.S).group(1)
    out["opened_rounds"] = re.findall(r"\(&(\w+), (\w+)\.as_slice\(\)\)", rounds)
    out["points"] = {k: re.sub(r"\s+", " ", v) for k, v in re.findall(r"let (zeta_and_next|zeta_exp_quotient_degree): [^=]*=\s*([^;]*);", body)}
    return out


def main():
    consts = constants()
    order = chip_order()
    shapes = {"source": "valida-xyz/valida Rust sources (tools/extract_reference_shapes.py)", "constants": consts, "opcodes": opcodes(), "chip_order": order,
              "buses": bus_ids(), "transcript": transcript(), "chips": {}}
    for chip in order:
        shape = columns(chip, consts)
        cons = eval_constraints(chip, shape, consts)
        shape["num_constraints"] = len(cons)
        shape["constraints"] = ["%s%s%s" % ("+".join(c["selectors"]) + ":" if c["selectors"] else "", "when*%d:" % c["filters"] if c["filters"] else "", c["kind"]) for c in cons]
        shape["interactions"] = interactions_of(chip, consts)
        shapes["chips"][chip] = shape
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(shapes, f, indent=1)
    for chip in order:
        s = shapes["chips"][chip]
        print("%-12s width %3d prep %d constraints %3d interactions %s" % (chip, s["width"], s["preprocessed_width"], s["num_constraints"],
                                                                           [(i["bus"], "S" if i["send"] else "R", i["n_fields"]) for i in s["interactions"]]))


if __name__ == "__main__":
    main()
