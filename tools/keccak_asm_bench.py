"""Build + run the hand-scheduled Keccak-f[1600] variants of tools/keccak_asm.py next to the compiled one (kernels/keccak.hpp):
correctness on the device (same state in, same state out) and SIMD-cycles per round at 4 waves per SIMD.

    python tools/keccak_asm_bench.py [OUT.txt]      # on the GPU box
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import keccak_asm as ka

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (label, where the no-ops go, register placement: False = as numbered / "mod4" = round 2's bank model / "parity" = round 3's measurement, rounds per loop: -1 in-place, 0 two-buffer unrolled)
VARIANTS = [("in-place 80 regs, each", "each", False, -1), ("in-place mod4-placed, each", "each", "mod4", -1), ("in-place parity-placed, each", "each", "parity", -1),
            ("two-buffer 120 regs, each", "each", False, 0), ("two-buffer parity-placed, each", "each", "parity", 0), ("in-place parity-placed, none", "none", "parity", -1)]


def asm_block(lines, top=127):
    body = "".join('        "%s\\n"\n' % ln for ln in lines)
    clob = ", ".join('"v%d"' % r for r in range(72, top + 1))
    return ('asm volatile(\n' + body + '        : "+{v[8:23]}"(t0), "+{v[24:39]}"(t1), "+{v[40:55]}"(t2), "+{v[56:71]}"(t3) : : ' + clob + ');')


def source():
    src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>', '#include "kernels/keccak.hpp"',
           '#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\\n", #e, hipGetErrorString(_e)); exit(2); } } while (0)',
           'typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));',
           '''__global__ void __launch_bounds__(256) k_ref(uint32_t* out, uint32_t seed, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    vk::KState a;
    for (int i = 0; i < 25; i++) { a.lo[i] = tid * 2654435761u + i * seed; a.hi[i] = tid * 40503u + i; }
    for (int it = 0; it < iters; it++) vk::keccak_f1600<false>(a);
    uint32_t x = 0;
    for (int i = 0; i < 25; i++) x ^= a.lo[i] * (2 * i + 1) ^ a.hi[i] * (2 * i + 2);
    out[tid] = x;
}''']
    for k, (name, mode, bank, loop_rounds) in enumerate(VARIANTS):
        ka.PHYS.update({r: r for r in range(8, 128)})
        loc = list(range(25))
        if loop_rounds < 0:
            ins, loc = ka.inplace_permutation()
            if bank:
                before = (ka.bank_conflicts(ins), ka.parity_conflicts(ins))
                ka.bank_optimise(ka.add_deps(list(ins)), steps=200000, hi=87, model=bank)
                print(name, "(mod-4 same-bank instructions, same-parity 3-source instructions) of %d:" % len(ins), before, "->", (ka.bank_conflicts(ins), ka.parity_conflicts(ins)), file=sys.stderr)
        else:
            ins = []
            for r in range(loop_rounds if loop_rounds else 24):
                ins += ka.round_instrs(r)
            if bank:
                before = (ka.bank_conflicts(ins), ka.parity_conflicts(ins))
                ka.bank_optimise(ka.add_deps(list(ins)), steps=200000, model=bank)
                print(name, "(mod-4 same-bank instructions, same-parity 3-source instructions) of %d:" % len(ins), before, "->", (ka.bank_conflicts(ins), ka.parity_conflicts(ins)), file=sys.stderr)
        lines = ka.with_nops(ins, mode)
        top = max(ka.PHYS[r] for i in ins for r in (i.dst,) + tuple(i.srcs))
        if loop_rounds > 0:
            lines = ["s_mov_b32 s20, %d" % (24 // loop_rounds), "1:"] + lines + ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b"]
        inv = {ka.PHYS[8 + w]: w for w in range(50)}
        load = " ".join("t%d[%d] = w[%d];" % ((r - 8) // 16, (r - 8) % 16, inv[r]) for r in range(8, 58))
        store = " ".join("w[%d] = t%d[%d];" % (inv[r], (r - 8) // 16, (r - 8) % 16) for r in range(8, 58))
        block = asm_block(lines, top).replace(': : "v72"', ': : "s20", "scc", "v72"')
        src.append(f'''__global__ void __launch_bounds__(256) k_v{k}(uint32_t* out, uint32_t seed, int iters) {{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    u32x16 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    uint32_t w[64];
    for (int i = 0; i < 25; i++) {{ w[2 * i] = tid * 2654435761u + i * seed; w[2 * i + 1] = tid * 40503u + i; }}
    {load}
    for (int it = 0; it < iters; it++) {{
        {block}
    }}
    {store}
    uint32_t x = 0;
    for (int i = 0; i < 25; i++) x ^= w[2 * i] * (2 * i + 1) ^ w[2 * i + 1] * (2 * i + 2);
    out[tid] = x;
}}''')
    src.append('''template <class K> static void run(const char* name, K kernel, int cu, uint32_t* d, const uint32_t* ref, uint32_t* host) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-22s", name);
    const int n_check = cu * 4 * 256;
    hipLaunchKernelGGL(kernel, dim3(cu * 4), dim3(256), 0, 0, d, 12345u, 3);
    CHECK(hipMemcpy(host, d, (size_t)n_check * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    if (ref) for (int i = 0; i < n_check; i++) bad += host[i] != ref[i];
    for (int wps : {1, 2, 3, 4, 5, 6}) {
        const int blocks = cu * wps, iters = 256;
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double perms = (double)blocks * 256 * iters;
        printf("  %5.2f G perm/s (%4.0f cyc/round)", perms / best / 1e6, cu * 4.0 * 2.35e9 * best * 1e-3 / ((double)blocks * 4 * iters * 24));
        fflush(stdout);
    }
    printf(ref ? (bad ? "   MISMATCH in %d threads\\n" : "   == compiled permutation\\n") : "\\n", bad);
    fflush(stdout);
}
int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cu = prop.multiProcessorCount, n = cu * 4 * 256;
    uint32_t* d;
    CHECK(hipMalloc(&d, (size_t)n * 4));
    uint32_t *ref = (uint32_t*)malloc((size_t)n * 4), *host = (uint32_t*)malloc((size_t)n * 4);
    printf("Keccak-f[1600] permutations per second and SIMD-cycles per round (nominal 2.35 GHz) at 1 .. 6 waves per SIMD (as far as the VGPR count admits: beyond that the extra blocks queue)\\n");
    run("compiled (keccak.hpp)", k_ref, cu, d, nullptr, ref);''')
    for k, (name, mode, bank, loop_rounds) in enumerate(VARIANTS):
        src.append(f'    run("{name}", k_v{k}, cu, d, {"nullptr" if loop_rounds > 0 else "ref"}, host);')
    src.append("    return 0;\n}")
    return "\n".join(src)


def main():
    """`gen`: write + compile build/keccak_asm_bench (here; build/ travels to the GPU box, where only the binary is run)"""
    if len(sys.argv) > 1 and sys.argv[1] == "gen":
        build = os.path.join(ROOT, "build")
        os.makedirs(build, exist_ok=True)
        hip = os.path.join(build, "keccak_asm_bench.hip")
        with open(hip, "w") as f:
            f.write(source())
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "valida_amd", "csrc"), hip, "-o", os.path.join(build, "keccak_asm_bench")], check=True)
        return
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    build = os.path.join(ROOT, "build")
    os.makedirs(build, exist_ok=True)
    hip = os.path.join(build, "keccak_asm_bench.hip")
    with open(hip, "w") as f:
        f.write(source())
    exe = os.path.join(build, "keccak_asm_bench")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "valida_amd", "csrc"), hip, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    text = "# tools/keccak_asm_bench.py\n" + (r.stdout if r.returncode == 0 else "no GPU run: " + r.stderr[-300:])
    if out_path:
        with open(out_path, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
