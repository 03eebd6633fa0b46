#!/usr/bin/env python3
"""Generate the Rust `-sys` binding of include/vgpu.h: bindings/rust/src/lib.rs — `#[repr(C)]` structs, opaque handle types, constants and
the `extern "C"` block, one item per header item, so that a Valida host (`impl UnivariatePcsWithLde for GpuPcs`) has the
thin FFI layer BASELINE.json's north star asks for.  There is no Rust toolchain in this environment: the file is generated, never
compiled here; tests/test_host_cpu.py checks that it is up to date with the header, that every exported symbol is declared and that
every struct has the fields of its C twin in the same order (so `#[repr(C)]` gives the same layout).

    python tools/gen_rust_bindings.py            # rewrite bindings/rust/src/lib.rs
    python tools/gen_rust_bindings.py --check    # exit 1 if the committed file differs
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vgpu.h")
OUT = os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")

SCALARS = {"uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64", "uint8_t": "u8", "double": "f64", "char": "c_char", "void": "c_void", "int": "c_int"}


def camel(name):
    base = name[:-2] if name.endswith("_t") else name
    return "".join(p.capitalize() for p in base.split("_"))


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def rust_type(ctype, names):
    """ctype: tokens of a C type without the declarator name, e.g. 'const vgpu_trace_t* const*'."""
    t = ctype.replace("*", " * ").split()
    # base type with its const
    const_base = False
    i = 0
    if t[i] == "const":
        const_base = True
        i += 1
    if t[i] == "struct":
        i += 1
    base = t[i]
    i += 1
    if i < len(t) and t[i] == "const":  # "T const"
        const_base = True
        i += 1
    rust = SCALARS.get(base) or names.get(base) or names.get(base + "_t")
    if rust is None:
        raise ValueError("unknown C type %r in %r" % (base, ctype))
    const_next = const_base
    while i < len(t):
        assert t[i] == "*", ctype
        rust = ("*const " if const_next else "*mut ") + rust
        const_next = False
        i += 1
        if i < len(t) and t[i] == "const":
            const_next = True
            i += 1
    return rust


def split_decl(decl):
    """'const uint32_t perm_challenges[15]' -> (type tokens, name, array length or None)"""
    decl = decl.strip()
    m = re.match(r"^(.*?)(\w+)\s*(\[\s*(\w*)\s*\])?$", decl, re.S)
    return m.group(1).strip(), m.group(2), (m.group(4) if m.group(3) else None)


def parse(src):
    src = strip_comments(src)
    defines = re.findall(r"^#define\s+(VGPU_\w+)\s+(\S+)\s*$", src, re.M)
    src = re.sub(r"^#.*$", "", src, flags=re.M)
    src = src.replace('extern "C" {', "").replace("extern \"C\"", "")
    items, depth, cur = [], 0, ""
    for ch in src:
        cur += ch
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
        elif ch == ";" and depth == 0:
            items.append(" ".join(cur.split()))
            cur = ""
    opaque, structs, enums, funcs = [], [], [], []
    for it in items:
        it = it.rstrip(";").strip()
        if it.startswith("}"):
            it = it.lstrip("} ")
        if not it:
            continue
        m = re.match(r"^typedef struct (\w+) (\w+)$", it)
        if m:
            opaque.append(m.group(2))
            continue
        m = re.match(r"^typedef struct (\w+) \{(.*)\} (\w+)$", it)
        if m:
            fields = []
            for f in m.group(2).split(";"):
                f = f.strip()
                if not f:
                    continue
                fp = re.match(r"^(.*?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", f, re.S)
                if fp:  # a callback: "int32_t (*all_gather)(void* user, ...)" -> Option<unsafe extern "C" fn(..) -> ..>
                    fields.append((("fn", fp.group(1).strip(), [split_decl(a) for a in fp.group(3).split(",")]), fp.group(2), None))
                    continue
                # "uint32_t clk, addr, value, is_write" and "uint32_t a[4]"
                first, *rest = [x.strip() for x in f.split(",")]
                ty, name, arr = split_decl(first)
                fields.append((ty, name, arr))
                for r in rest:
                    _, n2, a2 = split_decl("x " + r)
                    fields.append((ty, n2, a2))
            structs.append((m.group(3), fields))
            continue
        m = re.match(r"^enum \{(.*)\}$", it)
        if m:
            for e in m.group(1).split(","):
                e = e.strip()
                if e:
                    k, _, v = e.partition("=")
                    enums.append((k.strip(), v.strip()))
            continue
        m = re.match(r"^(.*?)(\bvgpu_\w+)\s*\((.*)\)$", it)
        if m:
            args = []
            a = m.group(3).strip()
            if a and a != "void":
                for p in a.split(","):
                    args.append(split_decl(p))
            funcs.append((m.group(1).strip(), m.group(2), args))
            continue
        raise ValueError("unparsed header item: %r" % it)
    return defines, opaque, structs, enums, funcs


def generate():
    defines, opaque, structs, enums, funcs = parse(open(HEADER).read())
    names = {o: camel(o) for o in opaque}
    names.update({s: camel(s) for s, _ in structs})
    out = ["// GENERATED by tools/gen_rust_bindings.py from include/vgpu.h — do not edit.  The `-sys` layer of the Valida GPU backend: raw FFI only;",
           "// the safe wrapper (`GpuPcs: UnivariatePcsWithLde`, INTEGRATION.md section 3b) sits on top of it.  Link: libvgpu.so (`make lib`).",
           "#![allow(non_camel_case_types, dead_code)]",
           "use core::ffi::{c_char, c_int, c_void};",
           ""]
    for k, v in defines:
        v = v.rstrip("u")
        out.append("pub const %s: u32 = %s;" % (k, v))
    prev = -1
    for k, v in enums:
        if v == "":
            v = str(prev + 1)
        prev = int(v)
        out.append("pub const %s: i32 = %s;" % (k, v))
    out.append("")
    for o in opaque:
        out.append("#[repr(C)] pub struct %s { _private: [u8; 0] }  // opaque: %s" % (names[o], o))
    out.append("")
    for s, fields in structs:
        out.append("#[repr(C)]\n#[derive(Clone, Copy)]\npub struct %s {  // %s" % (names[s], s))
        for ty, name, arr in fields:
            if isinstance(ty, tuple):  # nullable C function pointer: same size and alignment as a pointer
                ps = ", ".join("%s: %s" % (an, rust_type(aty, names)) for aty, an, _ in ty[2])
                rt = 'Option<unsafe extern "C" fn(%s)%s>' % (ps, "" if ty[1] == "void" else " -> " + rust_type(ty[1], names))
                out.append("    pub %s: %s," % (name, rt))
                continue
            rt = rust_type(ty, names)
            if arr is not None:
                rt = "[%s; %s]" % (rt, arr)
            out.append("    pub %s: %s," % (name, rt))
        out.append("}")
    out.append("")
    out.append('#[link(name = "vgpu")]\nextern "C" {')
    for ret, name, args in funcs:
        ps = []
        for ty, an, arr in args:
            if arr is not None:  # array parameters decay to pointers
                ty = ty + "*"
            ps.append("%s: %s" % (an, rust_type(ty, names)))
        r = "" if ret == "void" else " -> " + rust_type(ret, names)
        out.append("    pub fn %s(%s)%s;" % (name, ", ".join(ps), r))
    out.append("}")
    return "\n".join(out) + "\n", dict(opaque=opaque, structs=structs, funcs=funcs, enums=enums, defines=defines)


def main():
    text, _ = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == text else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print("wrote", os.path.relpath(OUT, ROOT), "(%d lines)" % text.count("\n"))


if __name__ == "__main__":
    main()
