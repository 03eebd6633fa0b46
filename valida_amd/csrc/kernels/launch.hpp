// Host-callable launchers of the gfx950 kernels (definitions in the .hip files of this directory).
#pragma once
#include <vector>
#include "device_common.hpp"
#include "profiler.hpp"
#include "../air/symbolic.hpp"

namespace vk {

struct QuotientArgs {
    DMatView main_lde, perm_lde, prep_lde;  // bit-reversed LDEs (prep_lde.data may be null)
    int log_n;                               // trace height 2^log_n; quotient domain 2^(log_n + 1)
    const vair::Instr* prog;
    uint32_t n_instrs, n_regs, n_air_asserts;
    const uint32_t* iw;          // encoded interactions
    const uint32_t* consts;      // Ext5 words: [K alpha powers][M bus alphas][max_fields betas][cumulative_sum]
    uint32_t K;                  // total number of folded constraints = n_air_asserts + M + 3
    uint32_t coset_shift, coset_shift_inv;      // s = 31 (Montgomery), s^-1
    int lqd;                     // log_quotient_degree: quotient domain 2^(log_n + lqd); 1 for every reference chip
    uint32_t zh_inv[8];          // 1 / zh[r]
    uint32_t zh[8];              // Z_H on the 2^lqd cosets of the quotient domain: zh[r] = s^n w_Q^r - 1 for natural index = r mod 2^lqd
    uint32_t g_inv;              // g_n^{-1}  (subgroup_last)
    DMatView out;                // n x (5 << lqd), row for natural i stored at position bitrev_k(i) — or, out_natural (log_quotient_degree 1 only), at position i
    int out_natural;             // the chunk matrix in NATURAL row order: the following commitment round extends it through the fused LDE (no bit-reversed input path)
    // BasicMachine chip whose eval template is compiled into a native kernel (vchips::ChipId), or INTERPRET for
    // the register-program interpreter (AIRs captured at run time through vgpu_air_*)
    static constexpr int INTERPRET = -2;
    int native_chip;
    // Where the NEXT row of a point is read from: column 0 of main / perm / preprocessed LDEs with the strides of the views above, and the
    // distance of the next row in natural index inside that LDE.  One GPU: the same LDEs, 2^lqd (the launchers fill these in when they
    // are left null / zero).  A proof sharded over several GPUs (host/sharded_prover.cpp) evaluates a row range of the LDE, i.e. the
    // sub-coset s w_L^e H_{L/W}: the successors of its points form ANOTHER sub-coset, a different rank's row range, at the same or the
    // following natural index.
    const uint32_t* main_nx;
    const uint32_t* perm_nx;
    const uint32_t* prep_nx;
    uint32_t next_step_p1;  // 1 + that distance (0 = not set; the distance itself may be 0)
    // defaults of the single-GPU case
    QuotientArgs normalised() const {
        QuotientArgs b = *this;
        if (!b.main_nx) b.main_nx = b.main_lde.data;
        if (!b.perm_nx) b.perm_nx = b.perm_lde.data;
        if (!b.prep_nx) b.prep_nx = b.prep_lde.data;
        if (!b.next_step_p1) b.next_step_p1 = 1u + (1u << (b.lqd > 0 ? b.lqd : 1));
        return b;
    }
};


// layout.hip
void launch_ingest(hipStream_t st, const uint32_t* src_dev, DMatView dst, bool bitrev);
void launch_bitrev_rows(hipStream_t st, DMatView src, DMatView dst);
void launch_clock_probe(hipStream_t st, uint64_t* out3, uint32_t iters);  // out3: page-locked host memory (device-visible); see vgpu_shader_clock_probe
// n_cols columns of Wq = 2^log_wq blocks of `rows` rows (block r in the natural order of the sub-coset bitrev(r)) -> columns in global natural order
void launch_interleave_blocks(hipStream_t st, const uint32_t* src, uint32_t* dst, uint64_t rows, uint32_t log_wq, uint64_t n_cols);
void launch_export_rows(hipStream_t st, DMatView src, uint64_t row0, uint64_t nrows, uint32_t* dst_dev);
// ntt.hip
void launch_intt(hipStream_t st, DMatView m, const DeviceTables& tb);
void launch_coset_ntt(hipStream_t st, DMatView coeffs, DMatView dst, uint64_t dst_row0, Fp shift, const DeviceTables& tb);
// natural-order evaluations -> committed (bit-reversed) LDE in 3 fused passes (1 for heights <= 2^12); lt: build_lde_tables(k, log_blowup, shift)
void launch_lde_natural(hipStream_t st, DMatView nat, DMatView lde, int log_blowup, const DeviceTables& tb, const LdeTables& lt, DMatView s1, DMatView s2);
// merkle.hip
void launch_keccak_leaves(hipStream_t st, const uint32_t* const* cols_dev, int n_elems, uint64_t n_rows, uint32_t* digests);
void launch_keccak_leaves_strided(hipStream_t st, const uint32_t* base, uint64_t stride, int n_elems, uint64_t n_rows, uint32_t* digests);
void launch_keccak_compress(hipStream_t st, const uint32_t* prev, const uint32_t* const* cols_dev, int n_elems, uint64_t n_out, uint32_t* next);
// The sibling digests of the two BOTTOM layers of a tree that did not keep them (DeviceTree::drop_bottom), recomputed at query time from the committed rows.
// jobs: 8 words each (merkle.hip, k_keccak_bottom_q); indices_dev: the sampled query indices; dst: the proof tail being gathered.
void launch_keccak_bottom_q(hipStream_t st, const uint32_t* jobs_dev, uint32_t n_jobs, const uint32_t* indices_dev, uint32_t* dst);
void launch_poseidon_bottom_q(hipStream_t st, const uint32_t* pos_dev, bool sparse, const uint32_t* jobs_dev, uint32_t n_jobs, const uint32_t* indices_dev, uint32_t* dst);
constexpr int KECCAK_TOP_MAX_LEVELS = 11;  // first_len <= 1024
struct KeccakTopArgs {
    const uint32_t* prev;  // layer with 2 * first_len digests
    uint64_t first_len;    // parents in the first computed layer (power of two <= 1024)
    int levels;            // layers computed: first_len, first_len/2, ..., 1
    uint32_t* out[KECCAK_TOP_MAX_LEVELS];
    const uint32_t* const* cols[KECCAK_TOP_MAX_LEVELS];  // injected matrices' columns per layer (or null)
    int n_elems[KECCAK_TOP_MAX_LEVELS];
    // Optional epilogue (FRI commit phase): the workgroup's first wave runs the DuplexChallenger step on the root it has just written —
    // observe it, sample beta — instead of a k_fri_challenge launch of its own behind the tree (challenger_dev.hpp; arguments as
    // launch_fri_challenge's).  ch_pos == nullptr: none.
    // Optional prologue (trees over ONE strided matrix of <= 512 rows, i.e. the small FRI layers): the launch hashes the leaves itself
    // into `prev` (leaf_rows rows of leaf_elems elements, columns leaf_base + k * leaf_stride) instead of a leaf launch of its own before it.
    const uint32_t* leaf_base = nullptr;
    uint64_t leaf_stride = 0;
    int leaf_elems = 0;
    uint64_t leaf_rows = 0;  // 0: none; else 2 * first_len
    // launch_keccak_levels only: parents of the first layer per WORKGROUP (a power of two <= 256 dividing first_len; the grid is
    // first_len / block_len workgroups, each walks its own sub-tree `levels` <= log2(block_len) + 1 layers down).  0: first_len (one workgroup)
    uint64_t block_len = 0;
    const uint32_t* ch_pos = nullptr;
    uint32_t* ch_state = nullptr;
    uint32_t* ch_beta5 = nullptr;
    uint32_t* ch_commit8 = nullptr;
};
void launch_keccak_top(hipStream_t st, const KeccakTopArgs& a);
// Several consecutive layers of the latency-bound middle of a tree (256 < parents <= 32768) in ONE launch: a workgroup per 64 parents of the
// first layer, the digests handed from layer to layer through LDS (merkle.hip: k_keccak_levels_pair).  keccak_levels_fused(len): whether a
// layer of `len` parents may go into such a launch (lane-pair kernels on, VGPU_KECCAK_LEVELS != 0); keccak_levels_take_leaves(n_rows):
// whether the launch can also hash the leaves of a single strided matrix of n_rows rows itself (KeccakTopArgs::leaf_*).
constexpr uint64_t KECCAK_LEVELS_BLOCK_LEN = 64;
void launch_keccak_levels(hipStream_t st, const KeccakTopArgs& a);
bool keccak_levels_fused(uint64_t len);
bool keccak_levels_take_leaves(uint64_t n_rows);
bool keccak_top_takes_leaves(uint64_t n_rows);  // whether launch_keccak_top can hash the leaves of a tree of n_rows rows itself (KeccakTopArgs::leaf_*)
// poseidon_mmcs.hip — the same tree with PaddingFreeSponge / TruncatedPermutation over Poseidon-16 (hash kind 1).
// pos_dev: [480 round constants][16 circulant MDS coefficients], Montgomery (the table the device challenger uses)
void launch_poseidon_leaves(hipStream_t st, const uint32_t* pos_dev, bool sparse, const uint32_t* const* cols_dev, int n_elems, uint64_t n_rows, uint32_t* digests);
void launch_poseidon_leaves_strided(hipStream_t st, const uint32_t* pos_dev, bool sparse, const uint32_t* base, uint64_t stride, int n_elems, uint64_t n_rows, uint32_t* digests);
void launch_poseidon_compress(hipStream_t st, const uint32_t* pos_dev, bool sparse, const uint32_t* prev, const uint32_t* const* cols_dev, int n_elems, uint64_t n_out, uint32_t* next);
// the layers of at most 16384 parents: launches of 16-row workgroups, each walking its sub-tree up to five layers down (one permutation per 16-lane row);
// layers group by log2(parents) / 5, the group that ends at the root is one workgroup and may carry the FRI challenger step (a.ch_*)
bool poseidon_levels_take(uint64_t parents);
int poseidon_levels_group(uint64_t parents);
void launch_poseidon_levels(hipStream_t st, const uint32_t* pos_dev, bool sparse, const KeccakTopArgs& a);
void launch_poseidon_top(hipStream_t st, const uint32_t* pos_dev, bool sparse, const KeccakTopArgs& a);  // first_len <= 16: the last group alone (sharded commitments)
// perm.hip
uint64_t perm_scratch_words(uint64_t n);
void launch_add_ext_const(hipStream_t st, uint32_t* data, uint64_t stride, uint64_t n, const uint32_t* off5_dev);  // 5 columns += 5 Montgomery constants
// native_chip: vchips::ChipId when the AIR is one of the in-tree BasicMachine chips (its interactions are compiled into the kernel), anything else = the encoded walk
void launch_perm_trace(hipStream_t st, DMatView main, DMatView prep, const uint32_t* iw_dev, const uint32_t* chal_dev, uint32_t M, DMatView perm,
                       uint32_t* scratch, int native_chip = -2);
// quotient.hip
void launch_quotient(hipStream_t st, const QuotientArgs& a, const DeviceTables& tb);
// Debug check on the trace domain: `a` carries the NATURAL-order traces in main_lde / perm_lde / prep_lde (log_n = log height);
// *first_bad_dev (initialised to ~0) receives min over failures of (row << 16 | code), see quotient.hip.
void launch_check_constraints(hipStream_t st, const QuotientArgs& a, unsigned long long* first_bad_dev);
// tracegen.hip — device images of the VM's operation logs (C ABI twins: vgpu_cpu_op_t, vgpu_mem_op_t, vgpu_alu_op_t)
struct TgCpuOp { uint32_t pc, fp, opcode; int32_t operands[5]; uint32_t kind, has_imm, imm, mem_first; };
struct TgMemOp { uint32_t clk, addr, value, is_write; };
struct TgAluOp { uint32_t opcode, a, b, c; };  // a = result, b / c = inputs, as u32 values of the big-endian Words
struct TgOutOp { uint32_t clk, byte; };          // OutputChip::values entry
enum { TG_CPU_STORE32 = 0, TG_CPU_LOAD32, TG_CPU_JAL, TG_CPU_JALV, TG_CPU_BEQ, TG_CPU_BNE, TG_CPU_IMM32, TG_CPU_BUS, TG_CPU_BUS_LEFT_IMM, TG_CPU_STOP,
       TG_CPU_LOADFP };
void launch_tracegen_cpu(hipStream_t st, const TgCpuOp* ops, uint64_t n, const TgMemOp* mem, uint64_t n_mem, DMatView t);
size_t tracegen_mem_sort_scratch_bytes(uint64_t n);
hipError_t launch_tracegen_mem(hipStream_t st, const TgMemOp* mem, uint64_t n, const uint32_t* static_cells, uint64_t n_static, uint32_t* keys2, uint32_t* idx2,
                               void* sort_tmp, size_t sort_tmp_bytes, DMatView t);
void launch_tracegen_alu(hipStream_t st, int chip, const TgAluOp* ops, uint64_t n, DMatView t);
void launch_tracegen_idle(hipStream_t st, int mode, const uint32_t* static_cells, uint64_t n_static, DMatView t);  // 0 zeros, 1 mul counter, 2 static data
// range: the byte histogram of the words range_check()'ed on execute — the results of add, sub, mul, mulhs, mulhu, div, sdiv instructions
// (alu_u32/src/{add,sub,mul,div}/mod.rs), read from the cpu log's bus operations and the memory write of their cycle
hipError_t launch_tracegen_range(hipStream_t st, const TgCpuOp* ops, uint64_t n, const TgMemOp* mem, uint64_t n_mem, uint32_t* counts, DMatView t);
// mul / div / shift / com from their logs (chip = CHIP_MUL .. ); output from the tape and the host-computed first row of every window
void launch_tracegen_alu2(hipStream_t st, int chip, const TgAluOp* ops, uint64_t n, DMatView t);
void launch_tracegen_output(hipStream_t st, const TgOutOp* vals, const uint32_t* row0, uint64_t n, uint64_t n_rows, DMatView t);
hipError_t launch_tracegen_program(hipStream_t st, const TgCpuOp* ops, uint64_t n, uint64_t padded_n, uint32_t rom_len, uint32_t* counts, DMatView t);
// open.hip
void launch_bary_weights(hipStream_t st, uint64_t n, const uint32_t* min_poly_dev, Fp shift, const DeviceTables& tb, uint32_t* w);
// the same for several (height, point) pairs in one launch: job = { first block (u32), pad, n (u64), min-poly pointer, weight buffer, digit-plane image (or null) }
void launch_bary_weights_batch(hipStream_t st, const uint32_t* jobs_dev, uint32_t n_jobs, uint32_t total_blocks, double total_rows, Fp shift, const DeviceTables& tb);
uint32_t bary_weights_blocks(uint64_t n);
bool bary_weights_has_image(uint64_t n);
uint64_t col_dot_slots(uint64_t n);
uint64_t bary_buffer_words(uint64_t n);  // words of a weight vector launch_bary_weights fills (launch_col_dot takes such buffers)
uint64_t col_dot_max_columns(int np);  // widest matrix (view) one k_col_dot launch takes for np points; wider ones are opened in column chunks
// finish = false: only the partial sums; the caller collects a job per launch (col_dot_finish_job: appends its 12-word image to `jobs`, returns the next first
// block) and finishes all of them with ONE launch_col_dot_finish_batch behind the launches (jobs_dev = the uploaded images)
void launch_col_dot(hipStream_t st, DMatView m, uint64_t n, int np, const uint32_t* w0, const uint32_t* w1, uint32_t* partial,
                    const uint32_t* scale5_dev, uint32_t* out_dev, bool finish = true);
uint32_t col_dot_finish_job(std::vector<uint32_t>& jobs, uint32_t first_block, uint64_t n, uint64_t width, int np, const uint32_t* partial, const uint32_t* scale5_dev, uint32_t* out_dev);
void launch_col_dot_finish_batch(hipStream_t st, const uint32_t* jobs_dev, uint32_t n_jobs, uint32_t total_blocks);
// accumulate: add to the vector already in `out` (a height with more than MAX_OPEN_POINTS_PER_LAUNCH distinct opening points is
// reduced in several launches)
constexpr int MAX_OPEN_POINTS_PER_LAUNCH = 4;
// Y of every (matrix, point) written into the reduce descriptors on the device (open.hip, k_open_y)
void launch_open_y(hipStream_t st, const uint32_t* vals_dev, const uint32_t* apow_dev, const uint32_t* desc_dev, const uint32_t* entry_off_dev, uint32_t n_entries, uint32_t* pool_dev);
// n_points: the descriptor's number of distinct points (1 .. MAX_OPEN_POINTS_PER_LAUNCH; sizes the 4-rows-per-thread kernel's arrays; anything else = the maximum)
// vec_ok: the caller's word that every matrix of the descriptor has a 16-byte-aligned base and a column stride that is a multiple of 4 words
// (reduce_vec_ok below, evaluated where the descriptor is built — the pointers live in device memory here).  The R-rows-per-thread kernel reads the
// columns with 8- / 16-byte loads and needs that, `out` 16-byte aligned and L a power of two >= 1024; anything else takes the thread-per-row kernel.
inline bool reduce_vec_ok(const void* col0, uint64_t stride_words) { return ((uintptr_t)col0 & 15) == 0 && (stride_words & 3) == 0; }
void launch_reduce_openings(hipStream_t st, const uint32_t* desc_dev, uint64_t L, Fp shift, const DeviceTables& tb, uint32_t* out, uint64_t total_width,
                            bool accumulate, int n_points, bool vec_ok);
// beta5_dev: the folding challenge as 5 Montgomery words in device memory (written by k_fri_challenge)
void launch_fri_fold(hipStream_t st, const uint32_t* in, uint64_t L, const uint32_t* beta5_dev, const uint32_t* add, const DeviceTables& tb, uint32_t* out);
// One DuplexChallenger step on the device: observe the 8-word root at digest8_dev, sample beta into beta5_dev; the root is
// also copied to commit8_dev.  pos_dev: [480 Poseidon round constants][16 circulant MDS coefficients]; ch_dev: 50-word state.
constexpr int DEV_CHALLENGER_WORDS = 50;
void launch_fri_challenge(hipStream_t st, const uint32_t* pos_dev, uint32_t* ch_dev, const uint32_t* digest8_dev, uint32_t* beta5_dev, uint32_t* commit8_dev);
void launch_pow_grind(hipStream_t st, const uint32_t* pos_dev, bool sparse, uint32_t k_pending, uint32_t first, uint32_t count, uint32_t bits, uint32_t* best_dev);
void launch_gather(hipStream_t st, const uint32_t* desc_dev, uint64_t n_desc, uint32_t* dst);
// the same from 8-word templates and the query indices (<= 256 queries: the query id is a byte) — open.hip, k_gather_q
void launch_gather_q(hipStream_t st, const uint32_t* templ_dev, uint64_t n_desc, const uint32_t* indices_dev, uint32_t* dst);

}  // namespace vk
