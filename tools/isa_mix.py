#!/usr/bin/env python3
"""Static instruction mix of the product's gfx950 kernels (no GPU needed: hipcc cross-compiles, llvm-objdump disassembles).

    python tools/isa_mix.py [out.txt]        # default: profiles/<round>_isa_static.txt is NOT written; prints to stdout

Per kernel: code size, VGPR / SGPR / LDS / scratch from the kernel descriptor notes, instruction counts by class — VALU split into the
full-rate (~2.4 SIMD-cycles per wave64 instruction) and half-rate (~4.2) classes measured by tools/microbench.py (profiles/r02_microbench.txt),
MFMA, LDS, global / scratch memory, scalar — and the largest loop body (target of the last backward branch .. that branch) with its own
VALU count.  Static counts are not dynamic counts; they answer "what did the compiler emit": spills (scratch), the length of a
butterfly / a Keccak round, whether a hot loop carries address arithmetic."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "valida_amd", "csrc")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
HALF_RATE = ("v_alignbit", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev", "v_min_u32", "v_max_u32", "v_min_i32", "v_add3_u32", "v_mul_lo", "v_mul_hi", "v_mul_u32_u24", "v_mad_u32_u24",
             "v_mad_u64_u32", "v_mad_i64_i32", "v_lshl_add", "v_lshl_or", "v_and_or", "v_or3", "v_bfe", "v_perm", "v_lshlrev_b64", "v_lshrrev_b64", "v_mad_i32_i24", "v_xad", "v_add_lshl", "v_lshl_add_u64")


def classify(mn):
    if mn.startswith("v_mfma") or mn.startswith("v_smfma"):
        return "mfma"
    if mn.startswith("v_"):
        return "valu_half" if mn.startswith(HALF_RATE) else "valu_full"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if mn.startswith("scratch_"):
        return "scratch"
    if mn.startswith("s_waitcnt") or mn.startswith("s_nop") or mn.startswith("s_barrier"):
        return "sync"
    if mn.startswith("s_"):
        return "salu"
    return "other"


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


def short(d):
    d = re.sub(r"\(.*", "", d)
    return d.replace("vk::", "").replace("void ", "")


def main():
    files = ["kernels/ntt.hip", "kernels/layout.hip", "kernels/merkle.hip", "kernels/poseidon_mmcs.hip", "kernels/perm.hip", "kernels/quotient.hip", "kernels/open.hip", "kernels/tracegen.hip"]
    # python tools/isa_mix.py [OUT.txt] [--only kernels/x.hip] [-DFLAG=v ...]: one file and / or an A/B build's flags (the table of a variant before GPU minutes go into it)
    argv, defs = [a for a in sys.argv[1:] if not a.startswith("-D")], [a for a in sys.argv[1:] if a.startswith("-D")]
    if "--only" in argv:
        i = argv.index("--only")
        files = [argv[i + 1]]
        del argv[i:i + 2]
    sys.argv[1:] = argv
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for f in files:
            base = os.path.basename(f).replace(".hip", "")
            subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-save-temps=obj"] + defs + ["-x", "hip", "-c", os.path.join(CSRC, f), "-o",
                            os.path.join(td, base + ".o")], check=True, cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            co = [os.path.join(td, x) for x in os.listdir(td) if x.startswith(base) and x.endswith("gfx950.out")]
            if not co:
                continue
            dis = subprocess.run([OBJDUMP, "-d", co[0]], check=True, capture_output=True, text=True).stdout
            notes = subprocess.run([READELF, "--notes", co[0]], capture_output=True, text=True).stdout
            meta = {}
            for blk in re.split(r"\n\s+- \.agpr_count:", notes):
                nm = re.search(r"\.symbol:\s+(\S+)\.kd", blk)
                if nm:
                    g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
                    meta[nm.group(1)] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size"),
                                             sspill=g("sgpr_spill_count"), vspill=g("vgpr_spill_count"), wgmax=g("max_flat_workgroup_size"))
            syms = re.findall(r"^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M)
            dm = demangle([s for s, _ in syms])
            for sym, body in syms:
                ins = []
                for line in body.splitlines():
                    mm = re.match(r"\s+(\S+).*//\s*([0-9A-F]+):", line)
                    if mm:
                        ins.append((int(mm.group(2), 16), re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", mm.group(1)), line))
                if not ins or "k_" not in dm.get(sym, sym):
                    continue
                cls = {}
                for _, mn, _ in ins:
                    c = classify(mn)
                    cls[c] = cls.get(c, 0) + 1
                best = None
                for addr, mn, line in ins:
                    if mn.startswith("s_cbranch") or mn == "s_branch":
                        t = re.search(r"<\S+\+0x([0-9a-f]+)>", line)
                        if t:
                            target = ins[0][0] + int(t.group(1), 16)
                            if target < addr and (best is None or addr - target > best[1] - best[0]):
                                best = (target, addr)
                loop = {}
                if best:
                    for addr, mn, _ in ins:
                        if best[0] <= addr <= best[1]:
                            c = classify(mn)
                            loop[c] = loop.get(c, 0) + 1
                m = dict(meta.get(sym, {}))
                # SGPRs spilled into VGPR lanes show as v_writelane / v_readlane pairs (the kernel descriptor's sgpr_spill_count counts the spilled registers)
                m["lanemov"] = sum(1 for _, mn, _ in ins if mn.startswith(("v_writelane", "v_readlane")))
                # waves per SIMD the register file allows: 512 VGPRs per lane per SIMD, allocated in blocks of 8, at most 8 waves (gfx950; AGPRs unused here);
                # a workgroup of `wgmax` threads must fit as a whole (wgmax / 64 waves over 4 SIMDs)
                try:
                    v = (int(m["vgpr"]) + 7) // 8 * 8
                    m["waves"] = min(8, 512 // max(v, 8))
                except (KeyError, ValueError):
                    m["waves"] = "?"
                rows.append((base, short(dm.get(sym, sym)), len(ins), cls, loop, m))
    out = ["# sspill / vspill = the kernel descriptor's sgpr_spill_count / vgpr_spill_count; lanemov = v_writelane + v_readlane instructions (SGPRs parked in VGPR lanes);",
           "# waves = waves per SIMD the VGPR count allows (512 / round-up-to-8(vgpr), at most 8); scratch = private segment bytes per lane; lds_B = STATIC LDS only (dynamic LDS is set at launch)",
           "# tools/isa_mix.py — static instruction mix of the gfx950 code objects of valida_amd/csrc/kernels/*.hip (hipcc -O3, llvm-objdump -d)",
           "# VALU classes by issue rate as measured in profiles/r02_microbench.txt; loop = the LARGEST backward-branch loop body of the kernel",
           "%-14s %-58s %6s | %5s %5s %5s %5s %5s %5s %5s | loop: %5s %5s %5s %5s | %4s %4s %6s %7s %6s %6s %7s %5s" % ("file", "kernel", "instr", "vfull", "vhalf", "mfma", "lds", "vmem", "salu", "sync", "vfull", "vhalf", "lds", "vmem",
                                                                                                         "vgpr", "sgpr", "lds_B", "scratch", "sspill", "vspill", "lanemov", "waves")]
    for base, name, n, cls, loop, m in sorted(rows, key=lambda r: (r[0], -r[2])):
        out.append("%-14s %-58s %6d | %5d %5d %5d %5d %5d %5d %5d | loop: %5d %5d %5d %5d | %4s %4s %6s %7s %6s %6s %7s %5s" % (
            base, name[:58], n, cls.get("valu_full", 0), cls.get("valu_half", 0), cls.get("mfma", 0), cls.get("lds", 0), cls.get("vmem", 0), cls.get("salu", 0), cls.get("sync", 0),
            loop.get("valu_full", 0), loop.get("valu_half", 0), loop.get("lds", 0), loop.get("vmem", 0), m.get("vgpr", "?"), m.get("sgpr", "?"), m.get("lds", "?"), m.get("scratch", "?"),
            m.get("sspill", "?"), m.get("vspill", "?"), m.get("lanemov", "?"), m.get("waves", "?")))
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
