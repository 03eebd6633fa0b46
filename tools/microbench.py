"""Build tools/microbench.hip, count the VALU instructions of every rate kernel's loop body in the llvm-objdump disassembly
of the gfx950 code object (so cycles are per INSTRUCTION, not per source-level "op"), run it on the GPU when one is there,
and write the combined report.

    python tools/microbench.py [OUT.txt]          # on the GPU box: gpurun -- python tools/microbench.py gpurun_out/r02_microbench.txt

The report is what bench.py's valu_roofline reads its peak from (profiles/rNN_microbench.txt, copied from gpurun_out/).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def build(extra=(), name="microbench"):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, name)
    src = os.path.join(ROOT, "tools", "microbench.hip")
    co = os.path.join(BUILD, name + ".co")  # the code object of THIS build (every build overwrites the -save-temps file)
    if not (os.environ.get("MICROBENCH_PREBUILT") == "1" and os.path.exists(exe) and os.path.exists(co)):  # the GPU box runs what was built here
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-save-temps=obj", "-I", os.path.join(ROOT, "valida_amd", "csrc")] + list(extra) + [src, "-o", exe], check=True,
                       cwd=BUILD, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        import shutil
        shutil.copyfile(os.path.join(BUILD, "microbench-hip-amdgcn-amd-amdhsa-gfx950.out"), co)
    return exe, co
    return exe, os.path.join(BUILD, "microbench-hip-amdgcn-amd-amdhsa-gfx950.out")


def loop_bodies(code_object):
    """{kernel: {mnemonic: count}} for the instructions between the target of the last backward branch and that branch."""
    dis = subprocess.run([OBJDUMP, "-d", code_object], check=True, capture_output=True, text=True).stdout
    out = {}
    for m in re.finditer(r"^[0-9a-f]+ <(_Z\d+(k_\w+?)(?:P\w+|ILb\d+E\w+))>:\n(.*?)(?=^\S|\Z)", dis, re.S | re.M):
        name, body = m.group(2), m.group(3)
        ins = []  # (address, mnemonic)
        for line in body.splitlines():
            mm = re.match(r"\s+(\S+).*//\s*([0-9A-F]+):", line)
            if mm:
                ins.append((int(mm.group(2), 16), mm.group(1), line))
        # last backward branch
        best = None
        for i, (addr, mn, line) in enumerate(ins):
            if mn.startswith("s_cbranch"):
                t = re.search(r"<\S+\+0x([0-9a-f]+)>", line)
                if t:
                    target = ins[0][0] + int(t.group(1), 16) - (ins[0][0] - ins[0][0])
                    # offsets are relative to the symbol start = address of the first instruction
                    target = ins[0][0] + int(t.group(1), 16)
                    if target < addr:
                        best = (target, addr)
        counts = {}
        if best:
            for addr, mn, _ in ins:
                if best[0] <= addr <= best[1]:
                    mn = re.sub(r"_e(32|64)$", "", mn)
                    counts[mn] = counts.get(mn, 0) + 1
        out[name] = counts
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    exe, co = build()
    exe0, _ = build(["-DVK_ALIGNBIT_NOP=0"], "microbench_nonop")
    exe1, _ = build(["-DVK_KECCAK_PIN=0"], "microbench_nopin")
    exe2, _ = build(["-DVK_KECCAK_PIN=2"], "microbench_rows")
    lines = ["# tools/microbench.py — gfx950 VALU issue rates (inline-asm loops) + HBM copy calibration kernels", ""]
    lines.append("## loop-body instruction counts from llvm-objdump -d of the code object (per loop iteration)")
    bodies = loop_bodies(co)
    for k in sorted(bodies):
        c = bodies[k]
        valu = sum(v for m, v in c.items() if m.startswith("v_"))
        detail = ", ".join("%s x%d" % (m, v) for m, v in sorted(c.items(), key=lambda kv: -kv[1]))
        lines.append("%-22s VALU %3d  | %s" % (k, valu, detail))
    lines.append("")
    gpu = subprocess.run([exe, "all"], capture_output=True, text=True)
    if gpu.returncode == 0:
        lines.append("## measured on the GPU")
        lines += gpu.stdout.splitlines()
        # the same permutation WITHOUT the scalar no-op behind every v_alignbit_b32 (keccak.hpp, VK_ALIGNBIT_NOP=0): the round-2 code
        g0 = subprocess.run([exe0, "rates"], capture_output=True, text=True)
        if g0.returncode == 0:
            lines.append("## keccak.hpp built with -DVK_ALIGNBIT_NOP=0 (no s_nop behind v_alignbit_b32), same run")
            lines += ["without s_nop: " + l for l in g0.stdout.splitlines() if l.startswith("keccak_f1600")]
        g1 = subprocess.run([exe1, "rates"], capture_output=True, text=True)
        if g1.returncode == 0:
            lines.append("## keccak.hpp built with -DVK_KECCAK_PIN=0 (no scheduling barriers: hipcc groups the round's 50 xors and 46 rotations), same run")
            lines += ["order not pinned: " + l for l in g1.stdout.splitlines() if l.startswith("keccak_f1600")]
        g2 = subprocess.run([exe2, "rates"], capture_output=True, text=True)
        if g2.returncode == 0:
            lines.append("## keccak.hpp built with -DVK_KECCAK_PIN=2 (the round row by row of its output: five lanes of theta / rho / pi, then chi of the row), same run")
            lines += ["row by row: " + l for l in g2.stdout.splitlines() if l.startswith("keccak_f1600")]
    else:
        lines.append("## no GPU run (%s)" % (gpu.stderr.strip().splitlines()[-1] if gpu.stderr.strip() else "exit %d" % gpu.returncode))
    text = "\n".join(lines) + "\n"
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        with open(out_path, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
