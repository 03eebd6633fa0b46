"""Generate + build + run explicit-register VALU issue-pattern loops (gfx950): how do full-rate and half-rate instructions of ONE wave's
stream share the SIMD?  (tools/microbench_issue.hip found: cross-wave mixing is additive, intra-wave mixing is not.)

    python tools/issue_patterns.py [OUT.txt]      # on the GPU box

Pattern language: a string of op letters, optionally with repeat counts ("B16A8" = 16 bitop3 then 8 alignbit), '|' = timestamp section break.
  B v_bitop3_b32 (VOP3, full)   X v_xor_b32 (VOP2, full)   D v_add_u32 (VOP2, full)   F v_fma_f32 (VOP3, full)
  A v_alignbit_b32 (VOP3, half)  L v_lshlrev_b32 (VOP2, half)  M v_mul_lo_u32 (VOP3, half)   n s_nop 0   N s_nop 3
Each op writes one of 8 chain registers of its own class (full: v32..v39, half: v56..v63), reads v40..v55.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PATTERNS = [
    ("BBA", "BBA"), ("BBAn", "BBAn"), ("BBAN (s_nop 3)", "BBAN"), ("BAn", "BAn"),
    ("BBBAn", "BBBAn"), ("BBBBAn", "BBBBAn"), ("B8An", "B8An"), ("BBBBAnAn", "BBBBAnAn"), ("B8AnAnAnAn", "B8AnAnAnAn"),
    ("BBAnBBA", "BBAnBBA"), ("BBABBAn", "BBABBAn"), ("BBABBABBAn", "BBABBABBAn"),
    ("BAnBAnB (3:2)", "BAnBAnB"), ("AnB", "AnB"), ("AAnB", "AAnB"), ("AnAnB", "AnAnB"),
    ("keccak BBAn x58 + B4", "BBAn" * 58 + "BBBB"),
    ("keccak BBAn, x-dep", "BBAn" * 58 + "BBBB"),
    ("XXLn vop2", "XXLn"), ("DDMn", "DDMn"),
    ("BnnBnnAnn", "BnnBnnAnn"), ("BBBnAAAn", "BBBnAAAn"), ("BBBBBBnAAAn", "BBBBBBnAAAn"),
]


def expand(p):
    out = []
    for m in re.finditer(r"([A-Za-z|])(\d*)", p):
        out += [m.group(1)] * (int(m.group(2)) if m.group(2) else 1)
    return out


def body(ops):
    lines, fi, hi, srcs = [], 0, 0, 0
    for op in ops:
        s1, s2 = 40 + (srcs % 16), 40 + ((srcs + 5) % 16)
        srcs += 1
        if op in "BXDF":
            d = 32 + fi % 8
            fi += 1
            lines.append({"B": f"v_bitop3_b32 v{d}, v{d}, v{s1}, v{s2} bitop3:0x96", "X": f"v_xor_b32 v{d}, v{d}, v{s1}", "D": f"v_add_u32 v{d}, v{d}, v{s1}",
                          "F": f"v_fma_f32 v{d}, v{d}, v{s1}, v{s2}"}[op])
        elif op in "ALM":
            d = 56 + hi % 8
            hi += 1
            lines.append({"A": f"v_alignbit_b32 v{d}, v{d}, v{s1}, 7", "L": f"v_lshlrev_b32 v{d}, 3, v{d}", "M": f"v_mul_lo_u32 v{d}, v{d}, v{s1}"}[op])
        elif op == "n":
            lines.append("s_nop 0")
        elif op == "N":
            lines.append("s_nop 3")
        elif op == "s":
            lines.append("s_mov_b32 s21, 0")
        elif op == "w":
            lines.append("s_waitcnt lgkmcnt(0)")
    return lines


def source():
    regs = ",".join('"v%d"' % r for r in range(32, 64))
    init = "".join("v_mov_b32 v%d, %%1\\n " % r for r in range(32, 64))
    fini = "v_xor_b32 %0, v32, v56\\n " + "".join("v_xor_b32 %%0, %%0, v%d\\n " % r for r in list(range(33, 40)) + list(range(57, 64)))
    src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>',
           '#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\\n", #e, hipGetErrorString(_e)); exit(2); } } while (0)']
    metas = []
    for k, (name, pat) in enumerate(PATTERNS):
        ops = expand(pat)
        reps = max(1, 96 // max(1, sum(o in "BXDFALM" for o in ops)))  # at least ~96 VALU instructions per loop iteration
        lines = body(ops * reps)
        n_valu = sum(1 for ln in lines if ln.startswith("v_"))
        asm = "".join('"%s\\n"\n' % ln for ln in lines)
        src.append(f'''__global__ void __launch_bounds__(256) k_p{k}(uint32_t* out, uint32_t seed, int iters) {{
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, x, s = seed + t;
    asm volatile("{init}s_mov_b32 s20, %2\\n 1:\\n"
{asm}"s_sub_u32 s20, s20, 1\\n s_cmp_lg_u32 s20, 0\\n s_cbranch_scc1 1b\\n {fini}" : "=v"(x) : "v"(s), "s"(iters) : {regs}, "s20", "s21", "scc");
    out[t] = x;
}}''')
        nf = sum(o in "BXDF" for o in ops)
        nh = sum(o in "ALM" for o in ops)
        metas.append((k, name, pat, n_valu, nf, nh))
    src.append('''template <class K> static void row(const char* name, K kernel, int per_iter, double additive, int cu, uint32_t* d) {
    printf("%-22s", name);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {2, 4, 5, 8}) {
        const int blocks = cu * wps, iters = 2048;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1u, 8);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double wave_instr = (double)blocks * 4 * iters * per_iter;
        printf("  %5.2f", cu * 4.0 * 2.35e9 / (wave_instr / (best * 1e-3)));
    }
    printf("   additive %.2f\\n", additive);
}
int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cu = prop.multiProcessorCount;
    uint32_t* d;
    CHECK(hipMalloc(&d, (size_t)cu * 8 * 256 * 4));
    printf("SIMD-cycles per VALU instruction (nominal 2.35 GHz) at 2 / 4 / 5 / 8 waves per SIMD; additive = (2.3 full + 4.15 half) / n\\n");''')
    for k, name, pat, n_valu, nf, nh in metas:
        add = (2.3 * nf + 4.15 * nh) / max(1, nf + nh)
        src.append(f'    row("{name}", k_p{k}, {n_valu}, {add:.3f}, cu, d);')
    src.append("    return 0;\n}")
    return "\n".join(src)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    build = os.path.join(ROOT, "build")
    os.makedirs(build, exist_ok=True)
    hip = os.path.join(build, "issue_patterns.hip")
    with open(hip, "w") as f:
        f.write(source())
    exe = os.path.join(build, "issue_patterns")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", hip, "-o", exe], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True)
    text = "# tools/issue_patterns.py\n" + (r.stdout if r.returncode == 0 else "no GPU run: " + r.stderr[-300:])
    if out_path:
        with open(out_path, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
