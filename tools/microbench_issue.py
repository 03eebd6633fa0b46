"""VALU issue experiments with hand-assigned physical VGPRs (gfx950): does the register-file bank of the operands, the
destination, or the mix of full-rate and half-rate opcodes change what an instruction costs the SIMD?

tools/microbench.py measures one opcode at a time with compiler-assigned registers; the Keccak round built from those
opcodes costs ~650 SIMD cycles per wave where the sum of its instructions' isolated costs is ~525.  This generator writes one
kernel per experiment — the loop body is inline asm over fixed registers v8.., so that the operand placement is exactly what
is named — builds them and, on a GPU box, prints SIMD cycles per instruction at 1, 2, 4, 5 and 8 waves per SIMD.

    python tools/microbench_issue.py [OUT.txt]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build", "mb")

N = 96  # VALU instructions per loop body (a multiple of 1, 2, 3, 4, 6, 8, 12, 16)


def dst(i, pool=16):
    return 24 + (i % pool)


def body(fn, n=N):
    return [fn(i) for i in range(n)]


EXPERIMENTS = []  # (name, [asm lines])


def add(name, lines):
    EXPERIMENTS.append((name, lines))


# ---- two-operand full-rate: source banks ----
add("xor  d, v8, v9            (banks 0,1)", body(lambda i: f"v_xor_b32 v{dst(i)}, v8, v9"))
add("xor  d, v8, v12           (banks 0,0)", body(lambda i: f"v_xor_b32 v{dst(i)}, v8, v12"))
add("xor  d, v8, v8            (same reg)", body(lambda i: f"v_xor_b32 v{dst(i)}, v8, v8"))
add("xor  d, v8, v10           (banks 0,2)", body(lambda i: f"v_xor_b32 v{dst(i)}, v8, v10"))
add("xor_e64 d, v8, v9", body(lambda i: f"v_xor_b32_e64 v{dst(i)}, v8, v9"))
add("add  d, v8, v9", body(lambda i: f"v_add_u32 v{dst(i)}, v8, v9"))
# ---- three-operand full-rate ----
add("bitop3 d, v8, v9, v10     (banks 0,1,2)", body(lambda i: f"v_bitop3_b32 v{dst(i)}, v8, v9, v10 bitop3:0x96"))
add("bitop3 d, v8, v12, v16    (banks 0,0,0)", body(lambda i: f"v_bitop3_b32 v{dst(i)}, v8, v12, v16 bitop3:0x96"))
add("bitop3 d, v8, v12, v17    (banks 0,0,1)", body(lambda i: f"v_bitop3_b32 v{dst(i)}, v8, v12, v17 bitop3:0x96"))
add("bitop3 d, v8, v10, v14    (banks 0,2,2)", body(lambda i: f"v_bitop3_b32 v{dst(i)}, v8, v10, v14 bitop3:0x96"))
add("bitop3 d, v8, v9, v9      (one reg twice)", body(lambda i: f"v_bitop3_b32 v{dst(i)}, v8, v9, v9 bitop3:0x96"))
add("bitop3 d, v8, v8, v8      (one reg)", body(lambda i: f"v_bitop3_b32 v{dst(i)}, v8, v8, v8 bitop3:0x96"))
add("bitop3 d, v8, v16, v48    (mod 8 equal)", body(lambda i: f"v_bitop3_b32 v{dst(i)}, v8, v16, v48 bitop3:0x96"))
add("bitop3 d, s, s+1, s+2 rotating sources", body(lambda i: f"v_bitop3_b32 v{dst(i)}, v{8 + i % 13}, v{9 + i % 13}, v{10 + i % 13} bitop3:0x96"))
add("bitop3 d, s, s+4, s+8 rotating sources", body(lambda i: f"v_bitop3_b32 v{dst(i)}, v{8 + i % 7}, v{12 + i % 7}, v{16 + i % 7} bitop3:0x96"))
add("fma   d, v8, v9, v10", body(lambda i: f"v_fma_f32 v{dst(i)}, v8, v9, v10"))
add("fma   d, v8, v12, v16", body(lambda i: f"v_fma_f32 v{dst(i)}, v8, v12, v16"))
# ---- destination bank against the sources ----
add("bitop3 d(bank 0 only), v9, v10, v11", body(lambda i: f"v_bitop3_b32 v{24 + 4 * (i % 4)}, v9, v10, v11 bitop3:0x96"))
add("bitop3 d(bank 0 only), v8, v9, v10", body(lambda i: f"v_bitop3_b32 v{24 + 4 * (i % 4)}, v8, v9, v10 bitop3:0x96"))
add("xor  d(bank 0 only), v9, v10", body(lambda i: f"v_xor_b32 v{24 + 4 * (i % 4)}, v9, v10"))
add("xor  d(bank 0 only), v8, v12", body(lambda i: f"v_xor_b32 v{24 + 4 * (i % 4)}, v8, v12"))
# ---- half-rate ----
add("alignbit d, v8, v9, 7", body(lambda i: f"v_alignbit_b32 v{dst(i)}, v8, v9, 7"))
add("alignbit d, v8, v12, 7", body(lambda i: f"v_alignbit_b32 v{dst(i)}, v8, v12, 7"))
add("alignbit d, v8, v8, 7     (32-bit rotate)", body(lambda i: f"v_alignbit_b32 v{dst(i)}, v8, v8, 7"))
add("alignbit d, v8, v9, v10   (shift in a VGPR)", body(lambda i: f"v_alignbit_b32 v{dst(i)}, v8, v9, v10"))
add("alignbyte d, v8, v9, 1", body(lambda i: f"v_alignbyte_b32 v{dst(i)}, v8, v9, 1"))
add("perm  d, v8, v9, v10", body(lambda i: f"v_perm_b32 v{dst(i)}, v8, v9, v10"))
add("lshl_or d, v8, 7, v9", body(lambda i: f"v_lshl_or_b32 v{dst(i)}, v8, 7, v9"))
add("lshl_add d, v8, 3, v9", body(lambda i: f"v_lshl_add_u32 v{dst(i)}, v8, 3, v9"))
add("lshlrev d, 7, v8", body(lambda i: f"v_lshlrev_b32 v{dst(i)}, 7, v8"))
add("lshrrev d, 7, v8", body(lambda i: f"v_lshrrev_b32 v{dst(i)}, 7, v8"))
add("and_or d, v8, v9, v10", body(lambda i: f"v_and_or_b32 v{dst(i)}, v8, v9, v10"))
add("bfi   d, v8, v9, v10", body(lambda i: f"v_bfi_b32 v{dst(i)}, v8, v9, v10"))
add("bfe   d, v8, 3, 7", body(lambda i: f"v_bfe_u32 v{dst(i)}, v8, 3, 7"))
add("mov   d, v8", body(lambda i: f"v_mov_b32 v{dst(i)}, v8"))
add("pk_mov? lshlrev_b64 d, 7, v[8:9]", body(lambda i: f"v_lshlrev_b64 v[{24 + 2 * (i % 8)}:{25 + 2 * (i % 8)}], 7, v[8:9]"))
add("pk_add_u16 d, v8, v9", body(lambda i: f"v_pk_add_u16 v{dst(i)}, v8, v9"))
add("pk_lshlrev_b16 d, v8, v9", body(lambda i: f"v_pk_lshlrev_b16 v{dst(i)}, v8, v9"))
add("mul_lo d, v8, v9", body(lambda i: f"v_mul_lo_u32 v{dst(i)}, v8, v9"))
add("mad_u64_u32 d, v8, v9, v[10:11]", body(lambda i: f"v_mad_u64_u32 v[{24 + 2 * (i % 8)}:{25 + 2 * (i % 8)}], vcc, v8, v9, v[10:11]"))
add("sub_co d, s[20:21], v8, v9", body(lambda i: f"v_sub_co_u32_e64 v{dst(i)}, s[20:21], v8, v9"))
add("cndmask d, v8, v9, s[20:21]", body(lambda i: f"v_cndmask_b32_e64 v{dst(i)}, v8, v9, s[20:21]"))
add("cndmask d, v8, v9, vcc", body(lambda i: f"v_cndmask_b32_e32 v{dst(i)}, v8, v9, vcc"))
add("min_u32 d, v8, v9", body(lambda i: f"v_min_u32 v{dst(i)}, v8, v9"))
# ---- dependent issue: distance between an instruction and the consumer of its result ----
for dist in (1, 2, 3, 4, 6, 8):
    add(f"xor chain, dependence distance {dist}", body(lambda i, d=dist: f"v_xor_b32 v{24 + i % d}, v{24 + i % d}, v8"))
for dist in (1, 2, 4):
    add(f"alignbit chain, dependence distance {dist}", body(lambda i, d=dist: f"v_alignbit_b32 v{24 + i % d}, v{24 + i % d}, v8, 7"))
for dist in (1, 2, 4):
    add(f"bitop3 chain, dependence distance {dist}", body(lambda i, d=dist: f"v_bitop3_b32 v{24 + i % d}, v{24 + i % d}, v8, v9 bitop3:0x96"))
# ---- mixes of full-rate and half-rate opcodes, independent ----
add("mix xor, alignbit alternating", body(lambda i: (f"v_xor_b32 v{dst(i)}, v8, v9" if i % 2 == 0 else f"v_alignbit_b32 v{dst(i)}, v10, v11, 7")))
add("mix bitop3, alignbit alternating", body(lambda i: (f"v_bitop3_b32 v{dst(i)}, v8, v9, v10 bitop3:0x96" if i % 2 == 0 else f"v_alignbit_b32 v{dst(i)}, v10, v11, 7")))
add("mix 2 bitop3 : 1 alignbit", body(lambda i: (f"v_bitop3_b32 v{dst(i)}, v8, v9, v10 bitop3:0x96" if i % 3 != 2 else f"v_alignbit_b32 v{dst(i)}, v10, v11, 7")))
add("mix 3 bitop3 : 1 alignbit", body(lambda i: (f"v_bitop3_b32 v{dst(i)}, v8, v9, v10 bitop3:0x96" if i % 4 != 3 else f"v_alignbit_b32 v{dst(i)}, v10, v11, 7")))
add("mix 4 bitop3 then 4 alignbit", body(lambda i: (f"v_bitop3_b32 v{dst(i)}, v8, v9, v10 bitop3:0x96" if i % 8 < 4 else f"v_alignbit_b32 v{dst(i)}, v10, v11, 7")))
add("mix 16 bitop3 then 8 alignbit (the Keccak ratio)", body(lambda i: (f"v_bitop3_b32 v{dst(i)}, v8, v9, v10 bitop3:0x96" if i % 24 < 16 else f"v_alignbit_b32 v{dst(i)}, v10, v11, 7")))
add("mix xor, mul_lo alternating", body(lambda i: (f"v_xor_b32 v{dst(i)}, v8, v9" if i % 2 == 0 else f"v_mul_lo_u32 v{dst(i)}, v10, v11")))
# ---- a producer's destination read by the NEXT instruction of the same wave as an operand (forwarding) among independent work ----
add("xor d0 <- ..; bitop3 <- d0 (pairs)", body(lambda i: (f"v_xor_b32 v{24 + (i // 2) % 8}, v8, v9" if i % 2 == 0 else f"v_bitop3_b32 v{32 + (i // 2) % 8}, v{24 + (i // 2) % 8}, v10, v11 bitop3:0x96")))


def kernel_source(idx, lines):
    init = "".join(f'        "v_add_u32 v{r}, {(r * 2654435761) & 0xffff}, %1\\n"\n' for r in range(8, 72))
    text = "".join(f'        "{l}\\n"\n' for l in lines)
    clob = ", ".join(f'"v{r}"' for r in range(8, 72))
    return f"""
__global__ void __launch_bounds__(256) k_exp{idx}(uint32_t* out, uint32_t seed, int iters) {{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint32_t r;
    const uint64_t c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    asm volatile(
{init}        "s_mov_b64 s[20:21], 0x5555\\n"
        "s_mov_b64 vcc, 0x3333\\n"
        "s_mov_b32 s22, %2\\n"
        "L_loop_%=:\\n"
{text}        "s_sub_u32 s22, s22, 1\\n"
        "s_cmp_lg_u32 s22, 0\\n"
        "s_cbranch_scc1 L_loop_%=\\n"
        "v_xor_b32 %0, v24, v25\\n"
        : "=v"(r) : "v"(t ^ seed), "s"(iters) : {clob}, "s20", "s21", "s22", "scc", "vcc");
    const uint64_t c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[t] = r;
    if (t == 0) {{ g_clk[0] = c1 - c0; g_clk[1] = w1 - w0; }}
}}
"""


HEADER = r"""// generated by tools/microbench_issue.py — do not edit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(2); } } while (0)
__device__ uint64_t g_clk[2];
typedef void (*kern_t)(uint32_t*, uint32_t, int);
static double clock_hz() { uint64_t h[2]; CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), 16)); return h[1] ? (double)h[0] / ((double)h[1] / 1e8) : 0.0; }
static double time_kernel(kern_t k, int blocks, int iters, uint32_t* d) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 12345u, 8);
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return best;
}
static void row(const char* name, kern_t k, int per_iter, int cu, uint32_t* d) {
    printf("%-52s", name);
    const int iters = 2048;
    for (int wps : {1, 2, 4, 5, 8}) {
        const int blocks = cu * wps;
        const double ms = time_kernel(k, blocks, iters, d), clk = clock_hz();
        const double simd_cycles = cu * 4.0 * clk * ms * 1e-3, wave_instr = (double)blocks * 4 * iters * per_iter;
        printf("  %5.2f", simd_cycles / wave_instr);
    }
    printf("\n");
}
"""


def generate():
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(BUILD, "microbench_issue.hip")
    with open(src, "w") as f:
        f.write(HEADER)
        for i, (_, lines) in enumerate(EXPERIMENTS):
            f.write(kernel_source(i, lines))
        f.write("int main() {\n    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));\n    const int cu = prop.multiProcessorCount;\n")
        f.write("    uint32_t* d; CHECK(hipMalloc(&d, (size_t)cu * 8 * 256 * 4));\n")
        f.write('    printf("SIMD cycles per VALU instruction (in-kernel clock) at 1 / 2 / 4 / 5 / 8 waves per SIMD; %d instructions per loop body, destinations v24.. unless named\\n", ' + str(N) + ");\n")
        for i, (name, lines) in enumerate(EXPERIMENTS):
            f.write(f'    row("{name}", k_exp{i}, {len(lines)}, cu, d);\n')
        f.write("    return 0;\n}\n")
    exe = os.path.join(BUILD, "microbench_issue")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", src, "-o", exe], check=True)
    return exe


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    exe = generate() if not (len(sys.argv) > 2 and sys.argv[2] == "--prebuilt") else os.path.join(BUILD, "microbench_issue")
    gpu = subprocess.run([exe], capture_output=True, text=True)
    text = "# tools/microbench_issue.py — gfx950 VALU issue cost against operand placement, destination bank and opcode mix\n"
    text += gpu.stdout if gpu.returncode == 0 else "## no GPU run (%s)\n" % (gpu.stderr.strip().splitlines()[-1] if gpu.stderr.strip() else "exit %d" % gpu.returncode)
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        with open(out_path, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
