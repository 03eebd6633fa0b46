// Machine::prove on one MI355X: host orchestration of the transcript-ordered phases of
// basic/src/lib.rs:147-675 (twin: derive/src/lib.rs:275-446) against the device kernels.
#pragma once
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include "challenger.hpp"
#include "machine.hpp"
#include "pcs.hpp"

namespace vhost {

constexpr uint32_t PROOF_MAGIC = 0x31465056u;  // "VPF1" flat wire format, see DESIGN.md "Proof wire format"

struct HostMatrix {  // canonical row-major, host memory (the reference's RowMajorMatrix<Val>)
    const uint32_t* data;
    uint64_t height, width;
};

// A trace resident in HBM: either in the layout the reference hands over (`raw`: row-major, canonical u32, from
// upload_trace) or already in the prover's working layout (`nat`: column-major Montgomery, natural row order, from
// device trace generation — needs no ingest pass).
inline uint64_t next_device_trace_uid() { static std::atomic<uint64_t> n{1}; return n.fetch_add(1); }
struct DeviceTrace {
    const uint64_t uid = next_device_trace_uid();  // identity of this upload / generation (the prover's preprocessed-commitment cache keys on it)
    DBuf raw;
    DMat nat;
    uint64_t height = 0, width = 0;
    bool nat_rows_bitrev = false;  // `nat` holds row i at position bitrev(i) (the quotient chunks of vgpu_quotient): commit skips its row permutation
};

// The VM's operation logs, as the chips' generate_trace reads them (cpu/src/lib.rs:79-97, memory/src/lib.rs:143-160,
// alu_u32/src/*/mod.rs), flattened: cpu records in clock order, memory operations in (clk, issue) order with each
// cpu record pointing at its first one, ALU operations per chip in issue order.
struct HostOplog {
    const vk::TgCpuOp* cpu = nullptr; uint64_t n_cpu = 0;
    const vk::TgMemOp* mem = nullptr; uint64_t n_mem = 0;
    const vk::TgAluOp* alu[4] = {nullptr, nullptr, nullptr, nullptr}; uint64_t n_alu[4] = {0, 0, 0, 0};  // add, sub, lt, bitwise
    const uint32_t* static_cells = nullptr; uint64_t n_static = 0;  // MemoryChip::static_data: (addr, value) pairs, ascending address
    uint32_t rom_len = 0;                                           // ProgramROM length (height of the program chip's trace before padding)
    const vk::TgAluOp* alu2[4] = {nullptr, nullptr, nullptr, nullptr}; uint64_t n_alu2[4] = {0, 0, 0, 0};  // mul, div, shift, com
    const vk::TgOutOp* output = nullptr; uint64_t n_output = 0;     // OutputChip::values
};
struct DeviceOplog {
    DBuf cpu, mem, alu[4], static_cells, alu2[4], output, output_row0;
    uint64_t n_cpu = 0, n_mem = 0, n_alu[4] = {0, 0, 0, 0}, n_static = 0, n_alu2[4] = {0, 0, 0, 0}, n_output = 0;
    uint64_t output_rows = 0;  // rows of the output chip's trace before padding (windows' rows + the final row)
    uint32_t rom_len = 0;
};

struct PhaseTimes {  // milliseconds, host clock around stream syncs
    double ingest = 0, commit_main = 0, perm = 0, commit_perm = 0, quotient = 0, commit_quotient = 0, open_values = 0, open_reduce = 0, fri = 0,
           queries = 0, total = 0;
};

struct ProveDebugOut {  // optional intermediate values for stage-parity tests (canonical words)
    uint32_t prep_root[8];
    uint32_t perm_challenges[15], alpha[5], zeta[5];
    bool check_constraints = false;                      // check_constraints + check_cumulative_sums before committing (debug builds of the reference)
    bool keep_matrices = false;                          // when set, the vectors below are filled (D2H copies)
    std::vector<std::vector<uint32_t>> perm_traces;      // per chip: n x 5(M+1) row-major, natural order
    std::vector<std::vector<uint32_t>> quotient_chunks;  // per chip: n x 10 row-major, natural order
};

// pcs.open_multi_batches (basic/src/lib.rs:611-619): one entry per commitment round — its ProverData and, per committed
// matrix (commit order), the points it is opened at.
struct OpenRound {
    const ProverData* pd = nullptr;
    std::vector<std::vector<Ext5>> points;
};
struct PcsOpening {
    std::vector<std::vector<std::vector<std::vector<Ext5>>>> opened;  // [round][matrix][point][column]  (lib.rs:622-645)
    std::vector<uint32_t> proof_words;                                 // TwoAdicFriPcsProof: the tail of the "VPF1" layout (App. B12)
    double ms_values = 0, ms_reduce = 0, ms_fri = 0, ms_queries = 0;
};

// [m0..m4][g0 (5)]..[g3 (5)]: the minimal polynomial m of z over the base field and g = m / (X - z), as k_bary_weights / k_reduce_openings read
// them (1 / (z - x) = -g(x) / m(x) for x in the base field)
void put_min_poly(std::vector<uint32_t>& w, const Ext5& z);

class Prover {
  public:
    Prover(int device, const MachineDesc& machine, const uint32_t* poseidon_rc480, const FriParams& fri);
    ~Prover();
    DeviceCtx& ctx() { return *ctx_; }
    const MachineDesc& machine() const { return machine_; }
    const FriParams& fri() const { return fri_; }

    // H2D of one host trace (not part of the timed prove()).
    std::unique_ptr<DeviceTrace> upload_trace(const HostMatrix& m);

    // Device trace generation: H2D of the operation logs, then Chip::generate_trace of chip
    // `chip` (cpu, program, mem, add, sub, lt, bitwise, range) as a kernel.  Other chips: host generate_trace + upload_trace.
    std::unique_ptr<DeviceOplog> upload_oplog(const HostOplog& log);
    std::unique_ptr<DeviceTrace> generate_trace(const DeviceOplog& log, int chip);
    static bool can_generate(int chip);
    // canonical row-major copy of a device trace (tests)
    void download_trace(const DeviceTrace& t, uint32_t* out);

    // main[i]: trace of chip i (chip order).  preprocessed: (chip index, trace) in chip order.
    std::vector<uint32_t> prove(const std::vector<const DeviceTrace*>& main, const std::vector<std::pair<int, const DeviceTrace*>>& preprocessed,
                                PhaseTimes* times = nullptr, ProveDebugOut* dbg = nullptr);

    // pcs.open_multi_batches: advances `ch` exactly as the reference's `&mut challenger` is advanced.
    PcsOpening open_multi_batches(const std::vector<OpenRound>& rounds, Challenger& ch);
    // generate_permutation_trace of chip `chip` kept in HBM (working layout): perm trace + cumulative sum (chip.rs:121-208)
    DMat permutation_trace(int chip, const DMat& main_nat, const DMat* prep_nat, const Ext5 rnd[3], Ext5* cumulative_sum);
    // quotient + decompose_and_flatten of chip `chip` (machine/src/quotient.rs:18-67) from its three committed LDEs:
    // n x 10 chunk matrix, rows at bit-reversed positions (what commit_shifted_batches consumes next)
    DMat quotient_chunks(int chip, const DMat& main_lde, const DMat& perm_lde, const DMat* prep_lde, const Ext5 rnd[3], const Ext5& alpha,
                         const Ext5& cumulative_sum);

  private:
    friend struct ShardedProof;  // one proof over several prover contexts / GPUs (sharded_prover.cpp) drives the same private pieces
    std::unique_ptr<DeviceCtx> ctx_;
    MachineDesc machine_;
    FriParams fri_;
    Poseidon16 perm16_;
    // The commitment to the PREPROCESSED traces (program ROM, range table: basic/src/lib.rs:189-201) depends on the machine and the program only,
    // not on the witness.  OPTIONAL (off by default: Machine::prove recomputes it in every call, and so does this prover — the bench's timed
    // region included): with set_prep_cache(true) (C ABI vgpu_prover_set_prep_cache) it is computed for the first proof that hands in a given set of preprocessed DeviceTraces and reused while the same traces (by
    // uid) come back — their working-layout copies, LDEs and tree stay on the device: one synchronisation point, the small LDE / tree
    // launches and their ingest less per proof (about 0.2 ms of a lone proof); the root is observed as ever.
    bool prep_cache_enabled_ = false;
  public:
    // Queues behind a running proof (prove_mu): the cached ProverData and working-layout copies are what that proof is reading.
    void set_prep_cache(bool on) {
        std::lock_guard<std::mutex> lk(ctx().prove_mu);
        prep_cache_enabled_ = on;
        if (!on) { prep_key_.clear(); prep_pd_cache_.reset(); prep_nat_cache_.clear(); }
    }
  private:
    std::vector<std::pair<int, uint64_t>> prep_key_;
    std::vector<DMat> prep_nat_cache_;
    std::unique_ptr<ProverData> prep_pd_cache_;
    std::vector<DBuf> prog_dev_, iw_dev_;  // per chip: program instructions, interaction words
    DBuf pow_pos_;                         // [480 rc][16 mds coefficients][16 state][1 best] for k_pow_grind
    static constexpr size_t CS_PINNED_WORDS = 5 * 256;
    uint32_t* cs_pinned_ = nullptr;        // pinned landing area of the chips' cumulative sums (asynchronous D2H inside prove)
    uint32_t* open_pinned_ = nullptr;      // pinned landing area of an opening's values (asynchronous D2H beside the reduced openings / FRI); one opening at a time per prover
    size_t open_pinned_words_ = 0;
    uint32_t grind(Challenger& ch);
    void fill_quotient_args(vk::QuotientArgs& a, int chip, vk::DMatView main_lde, vk::DMatView perm_lde, vk::DMatView prep_lde, unsigned log_n, const uint32_t* consts_dev);
};

}  // namespace vhost
