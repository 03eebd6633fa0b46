// Permutation (LogUp bus) trace generation on the device, realising
// generate_permutation_trace (machine/src/chip.rs:121-208) + reduce_row (:335-352) +
// batch_multiplicative_inverse_allowing_zero (util/src/lib.rs:21-43):
//   q[n][m] = 1 / (alpha_bus(m) + sum_j beta^j f_{m,j}(row n))        (0 stays 0)
//   phi[n]  = phi[n-1] + sum_m (+/-) mult_m(row n) * q[n][m]           (+ sends, - receives)
// Output: column-major n x 5(M+1) base matrix = flatten_to_base of the Ext5 trace, NATURAL row order.
// The reference's row-serial prefix loop (:178-201) becomes a 3-phase block scan (Ext5 addition is five
// independent base-field additions).  Ext5 inversion is per element (Frobenius norm -> one base pow):
// identical values to the reference's batch inversion, exact arithmetic.
#include <algorithm>
#include <cstdlib>
#include "launch.hpp"
#include "interactions.hpp"
#include "../chips/basic_machine.hpp"

namespace vk {

// chal: M x 5 words (alpha_bus per interaction) followed by max_fields x 5 words (beta^j), Montgomery.
__global__ void __launch_bounds__(256) k_perm_recip(DMatView main, DMatView prep, const uint32_t* __restrict__ iw, const uint32_t* __restrict__ chal, DMatView perm) {
    uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= main.height) return;
    const uint32_t M = iw[0];
    const uint32_t* betas = chal + 5 * M;
    Ext5 delta = Ext5::zero();
    for (uint32_t m = 0; m < M; m++) {
        uint32_t pos = iw[2 + m];
        const bool is_send = iw[pos] != 0;
        const uint32_t nf = iw[pos + 1];
        pos += 2;
        Fp mult = eval_vcol(iw, pos, main.data, main.stride, prep.data, prep.stride, n);
        Ext5 rlc = ext_from_words(chal + 5 * m);
        for (uint32_t j = 0; j < nf; j++) {
            Fp f = eval_vcol(iw, pos, main.data, main.stride, prep.data, prep.stride, n);
            rlc += ext_from_words(betas + 5 * j) * f;
        }
        Ext5 q = rlc.inv();  // inv(0) = 0
        store_ext(perm.data + (uint64_t)(5 * m) * perm.stride, perm.stride, n, q);
        Ext5 term = q * mult;
        delta = is_send ? delta + term : delta - term;
    }
    store_ext(perm.data + (uint64_t)(5 * M) * perm.stride, perm.stride, n, delta);
}

// ---- the same for the in-tree chips with their interactions COMPILED IN (round 5) ----------------------------------------------------------
// k_perm_recip walks the encoded interactions (a scalar load in front of every column load) and inverts every interaction's combination on its
// own: a cpu row costs four Ext5 inversions (~140 base-field products each), an add / sub row five.  The BasicMachine's interactions are static
// (chips/basic_machine.hpp: visit_interactions), so the visit is instantiated over this collector — every field a load at a compile-time
// column — and the row's M combinations are inverted TOGETHER (Montgomery's trick in Ext5: 3 (M - 1) products and ONE inversion; a zero
// stays zero, as batch_multiplicative_inverse_allowing_zero has it).  Same values (exact field arithmetic).  VGPU_PERM_NATIVE=0: the walk (A/B).
constexpr int PERM_NATIVE_MAX_M = 5;  // add / sub: four range sends and the bus receive
struct DevicePermRow {
    const uint32_t* main;
    uint64_t stride;
    uint32_t roff;
    const uint32_t* chal;
    const uint32_t* betas;
    Ext5 rl[PERM_NATIVE_MAX_M];
    Fp mult[PERM_NATIVE_MAX_M];
    uint32_t snd;  // bit m: interaction m is a send
    Ext5 rlc;
    uint64_t t[5];
    int pend, m, j;
    __device__ __forceinline__ Fp value(const vchips::Lin& f) const {
        Fp v = Fp::from_canonical(f.k);
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (i < f.n) {
                const Fp x = Fp::raw(load_at(main + (uint64_t)f.col[i] * stride, roff));
                v += f.w[i] == 1u ? x : x * Fp::from_canonical(f.w[i]);
            }
        return v;
    }
    __device__ __forceinline__ void flush() {
#pragma unroll
        for (int c = 0; c < 5; c++) { rlc.c[c] += Fp::raw(vg::monty_reduce_wide(t[c])); t[c] = 0; }
        pend = 0;
    }
    // (the per-interaction slots are written through unrolled compares, never through a run-time index: where the visit's loops are not folded to a
    // compile-time m the arrays would otherwise move to scratch memory)
    __device__ __forceinline__ void begin(bool is_send, int) {
        snd |= (is_send ? 1u : 0u) << m;
        rlc = ext_from_words(chal + 5 * m);
#pragma unroll
        for (int c = 0; c < 5; c++) t[c] = 0;
        pend = 0; j = 0;
    }
    __device__ __forceinline__ void field(const vchips::Lin& f) {
        if (!(f.n == 0 && f.k == 0)) {
            const Fp x = value(f);
#pragma unroll
            for (int c = 0; c < 5; c++) t[c] += (uint64_t)betas[5 * j + c] * x.v;
            if (++pend == 4) flush();
        }
        j++;
    }
    __device__ __forceinline__ void end(const vchips::Lin& count) {
        if (pend) flush();
        const Fp cnt = value(count);
#pragma unroll
        for (int mm = 0; mm < PERM_NATIVE_MAX_M; mm++) if (mm == m) { rl[mm] = rlc; mult[mm] = cnt; }
        m++;
    }
};
// one row of a native chip: the reciprocal columns are stored, the row's contribution to the running sum is returned
template <int CHIP>
__device__ __forceinline__ Ext5 perm_row_native(const DMatView& main, const uint32_t* __restrict__ iw, const uint32_t* __restrict__ chal, const DMatView& perm, uint64_t n, int* m_out) {
    const uint32_t M = iw[0];
    DevicePermRow v;
    v.main = main.data; v.stride = main.stride; v.roff = (uint32_t)n * 4u;  // trace heights <= 2^27 rows: the byte offset fits
    v.chal = chal; v.betas = chal + 5 * M;
    v.m = 0; v.pend = 0; v.j = 0; v.snd = 0;
    vchips::visit_interactions(CHIP, v);  // CHIP is a compile-time constant: straight-line code, v.m ends at M <= PERM_NATIVE_MAX_M
    Ext5 pre[PERM_NATIVE_MAX_M];
    Ext5 run = Ext5::one();
#pragma unroll
    for (int m = 0; m < PERM_NATIVE_MAX_M; m++)
        if (m < v.m) { pre[m] = run; run = run * (v.rl[m].is_zero() ? Ext5::one() : v.rl[m]); }
    Ext5 inv = run.inv();  // never zero: zeros were replaced by one
    Ext5 delta = Ext5::zero();
#pragma unroll
    for (int m = PERM_NATIVE_MAX_M - 1; m >= 0; m--)
        if (m < v.m) {
            const bool z = v.rl[m].is_zero();
            Ext5 q = inv * pre[m];
            inv = inv * (z ? Ext5::one() : v.rl[m]);
            if (z) q = Ext5::zero();
            store_ext(perm.data + (uint64_t)(5 * m) * perm.stride, perm.stride, n, q);
            const Ext5 term = q * v.mult[m];
            delta = ((v.snd >> m) & 1u) ? delta + term : delta - term;
        }
    *m_out = v.m;
    return delta;
}
template <int CHIP>
__global__ void __launch_bounds__(256) k_perm_recip_native(DMatView main, const uint32_t* __restrict__ iw, const uint32_t* __restrict__ chal, DMatView perm) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= main.height) return;
    int m;
    const Ext5 delta = perm_row_native<CHIP>(main, iw, chal, perm, n, &m);
    store_ext(perm.data + (uint64_t)(5 * m) * perm.stride, perm.stride, n, delta);
}

// (Round 6 measured the running sum in the SAME launch — a decoupled-look-back pass over 256-row workgroups, SURVEY.md K7 — against the three-phase scan
// below: not faster alone, 1.2 % slower with three proofs in flight; profiles/r06_ab_onepass_scan.txt, kernel in the history at commit f92d09a.)

// ---- inclusive prefix sum of a base-field column (blockIdx.y selects the column) ----------------------
constexpr int SCAN_ITEMS = 4, SCAN_THREADS = 256, SCAN_BLOCK = SCAN_ITEMS * SCAN_THREADS;  // (8 items per thread: 32-byte lane stride, slower — profiles/r06_ab_scan.txt)

// exclusive scan of one value per thread over the workgroup; *total = the workgroup's sum
#if defined(__HIPCC__) && !defined(HIPEMU_CHECKS)
// device: inclusive scan inside the wave by shuffles, the four wave totals through LDS — two barriers (round 6; the LDS Hillis-Steele form below
// spent 72 % of the scan kernels' wave cycles at its sixteen barriers: profiles/r06_pmc.json, k_scan_apply)
__device__ __forceinline__ Fp block_exclusive_scan(Fp v, uint32_t* lds, Fp* total) {
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    Fp incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Fp t = Fp::raw((uint32_t)__shfl_up((int)incl.v, off, 64));
        if (lane >= off) incl += t;
    }
    if (lane == 63) lds[wave] = incl.v;
    __syncthreads();
    Fp before = Fp::zero(), all = Fp::zero();
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        const Fp t = Fp::raw(lds[w]);
        all += t;
        if (w < wave) before += t;
    }
    __syncthreads();  // lds is reused by the caller's next scan
    *total = all;
    return before + incl - v;
}
#else
// host emulation (tools/hipemu: no wave primitives): Hillis-Steele over blockDim.x values in LDS
__device__ __forceinline__ Fp block_exclusive_scan(Fp v, uint32_t* lds, Fp* total) {
    lds[threadIdx.x] = v.v;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
        Fp t = Fp::raw(lds[threadIdx.x]);
        if ((int)threadIdx.x >= off) t += Fp::raw(lds[threadIdx.x - off]);
        __syncthreads();
        lds[threadIdx.x] = t.v;
        __syncthreads();
    }
    *total = Fp::raw(lds[SCAN_THREADS - 1]);
    Fp incl = Fp::raw(lds[threadIdx.x]);
    __syncthreads();
    return incl - v;
}
#endif

// a thread's SCAN_ITEMS consecutive elements: one 16-byte access where the column allows it (n a multiple of 4 and a 16-byte-aligned column), else word by word
static_assert(SCAN_ITEMS == 4, "scan_load4 / k_scan_apply move four consecutive elements as one uint4");
__device__ __forceinline__ void scan_load4(const uint32_t* col, uint64_t base, uint64_t n, bool vec, Fp (&v)[SCAN_ITEMS]) {
    if (vec && base + 4 <= n) {
        const uint4 q = *reinterpret_cast<const uint4*>(col + base);
        v[0] = Fp::raw(q.x); v[1] = Fp::raw(q.y); v[2] = Fp::raw(q.z); v[3] = Fp::raw(q.w);
        return;
    }
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) v[i] = base + i < n ? Fp::raw(col[base + i]) : Fp::zero();
}
// phase 1: block sums
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_block_sums(const uint32_t* __restrict__ data, uint64_t stride, uint64_t n, uint32_t* __restrict__ sums, uint64_t n_blocks) {
    __shared__ uint32_t lds[SCAN_THREADS];
    const uint32_t* col = data + (uint64_t)blockIdx.y * stride;
    const bool vec = ((n | stride) & 3) == 0 && ((uintptr_t)data & 15) == 0;
    uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    Fp v[SCAN_ITEMS];
    scan_load4(col, base, n, vec, v);
    const Fp s = (v[0] + v[1]) + (v[2] + v[3]);
    Fp total;
    block_exclusive_scan(s, lds, &total);
    if (threadIdx.x == 0) sums[(uint64_t)blockIdx.y * n_blocks + blockIdx.x] = total.v;
}
// phase 2: exclusive scan of the block sums of one column by one block (n_blocks arbitrary)
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_sums(uint32_t* __restrict__ sums, uint64_t n_blocks) {
    __shared__ uint32_t lds[SCAN_THREADS];
    uint32_t* s = sums + (uint64_t)blockIdx.y * n_blocks;
    Fp carry = Fp::zero();
    for (uint64_t base = 0; base < n_blocks; base += SCAN_THREADS) {
        uint64_t i = base + threadIdx.x;
        Fp v = i < n_blocks ? Fp::raw(s[i]) : Fp::zero();
        Fp total;
        Fp ex = block_exclusive_scan(v, lds, &total);
        if (i < n_blocks) s[i] = (carry + ex).v;
        carry += total;
    }
}
// phase 3: in-block inclusive scan + block offset, in place
__global__ void __launch_bounds__(SCAN_THREADS) k_scan_apply(uint32_t* __restrict__ data, uint64_t stride, uint64_t n, const uint32_t* __restrict__ sums, uint64_t n_blocks) {
    __shared__ uint32_t lds[SCAN_THREADS];
    uint32_t* col = data + (uint64_t)blockIdx.y * stride;
    const bool vec = ((n | stride) & 3) == 0 && ((uintptr_t)data & 15) == 0;
    uint64_t base = (uint64_t)blockIdx.x * SCAN_BLOCK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    Fp v[SCAN_ITEMS];
    scan_load4(col, base, n, vec, v);
    const Fp s = (v[0] + v[1]) + (v[2] + v[3]);
    Fp total;
    Fp run = block_exclusive_scan(s, lds, &total) + Fp::raw(sums[(uint64_t)blockIdx.y * n_blocks + blockIdx.x]);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) { run += v[i]; v[i] = run; }
    if (vec && base + 4 <= n) { *reinterpret_cast<uint4*>(col + base) = make_uint4(v[0].v, v[1].v, v[2].v, v[3].v); return; }
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) if (base + i < n) col[base + i] = v[i].v;
}

// five base-field columns (one Ext5 column of a flattened trace) += a constant: the offset of a row range's running sum (sharded prover)
__global__ void __launch_bounds__(256) k_add_ext_const(uint32_t* __restrict__ data, uint64_t stride, uint64_t n, const uint32_t* __restrict__ off5) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* col = data + (uint64_t)blockIdx.y * stride;
    col[i] = (Fp::raw(col[i]) + Fp::raw(off5[blockIdx.y])).v;
}
void launch_add_ext_const(hipStream_t st, uint32_t* data, uint64_t stride, uint64_t n, const uint32_t* off5_dev) {
    ProfScope ps("k_scan", st, 2.0 * 20.0 * n);
    VK_LAUNCH(k_add_ext_const, dim3((unsigned)((n + 255) / 256), 5), dim3(256), 0, st, data, stride, n, off5_dev);
}

uint64_t perm_scratch_words(uint64_t n) { return 5 * ((n + SCAN_BLOCK - 1) / SCAN_BLOCK); }

// main/prep: natural-order column-major.  perm: n x 5(M+1), natural order.  scratch: >= 5 * ceil(n / SCAN_BLOCK) words.
void launch_perm_trace(hipStream_t st, DMatView main, DMatView prep, const uint32_t* iw_dev, const uint32_t* chal_dev, uint32_t M, DMatView perm,
                       uint32_t* scratch, int native_chip) {
    uint64_t n = main.height;
    static const bool native_on = [] { const char* e = getenv("VGPU_PERM_NATIVE"); return !(e && e[0] == '0'); }();
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    { ProfScope ps("k_perm_recip", st, 4.0 * n * (main.width + perm.width));
    bool done = false;
    if (native_on && M <= (uint32_t)PERM_NATIVE_MAX_M) {
        done = true;
        switch (native_chip) {
#define VG_PERM_NATIVE(C) case vchips::C: VK_LAUNCH((k_perm_recip_native<vchips::C>), grid, block, 0, st, main, iw_dev, chal_dev, perm); break;
            VG_PERM_NATIVE(CHIP_CPU) VG_PERM_NATIVE(CHIP_PROGRAM) VG_PERM_NATIVE(CHIP_MEM) VG_PERM_NATIVE(CHIP_ADD) VG_PERM_NATIVE(CHIP_SUB) VG_PERM_NATIVE(CHIP_MUL)
            VG_PERM_NATIVE(CHIP_DIV) VG_PERM_NATIVE(CHIP_SHIFT) VG_PERM_NATIVE(CHIP_LT) VG_PERM_NATIVE(CHIP_COM) VG_PERM_NATIVE(CHIP_BITWISE) VG_PERM_NATIVE(CHIP_OUTPUT)
            VG_PERM_NATIVE(CHIP_RANGE) VG_PERM_NATIVE(CHIP_STATIC_DATA)
#undef VG_PERM_NATIVE
            default: done = false; break;  // a run-time captured AIR: the descriptor walk
        }
    }
    if (!done) VK_LAUNCH(k_perm_recip, grid, block, 0, st, main, prep, iw_dev, chal_dev, perm); }
    ProfScope ps("k_scan", st, 3.0 * 20.0 * n);
    uint32_t* phi = perm.data + (uint64_t)(5 * M) * perm.stride;
    uint64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    VK_LAUNCH(k_scan_block_sums, dim3((unsigned)nb, 5), dim3(SCAN_THREADS), 0, st, phi, perm.stride, n, scratch, nb);
    VK_LAUNCH(k_scan_sums, dim3(1, 5), dim3(SCAN_THREADS), 0, st, scratch, nb);
    VK_LAUNCH(k_scan_apply, dim3((unsigned)nb, 5), dim3(SCAN_THREADS), 0, st, phi, perm.stride, n, scratch, nb);
}

}  // namespace vk
