/*
 * vgpu.h — C ABI of the MI355X-native STARK prover backend for Valida (libvgpu.so).
 *
 * This is the drop-in boundary for ONE path of the reference: Machine::prove
 * (basic/src/lib.rs:147-675) and the traits it drives — StarkConfig (machine/src/config.rs:7-31),
 * Pcs: UnivariatePcsWithLde (call sites basic/src/lib.rs:199,201,223,225,258,261,594,599,619),
 * generate_permutation_trace (machine/src/chip.rs:121-130), quotient (machine/src/quotient.rs:18-37)
 * and the Fiat-Shamir challenger (basic/src/lib.rs:185-263,601-619).  The reference has no FFI of its
 * own (all Rust generics); every entry point below names the reference item it replaces.  A Rust host
 * binds it with #[repr(C)] structs — see INTEGRATION.md for the stub.
 *
 * Conventions
 *   - plain pointers and sizes only; all structs are POD.
 *   - field elements cross the ABI as canonical u32 < p = 2013265921 (BabyBear); extension elements
 *     (BinomialExtensionField<BabyBear,5>) as 5 consecutive u32; digests as 8 u32.
 *   - matrices cross as the reference's RowMajorMatrix<Val>: row-major u32[height * width].
 *   - every function returns VGPU_OK (0) or a negative status; vgpu_last_error() gives the message of
 *     the last failure on the calling thread.  Nothing unwinds across the ABI.
 *   - one vgpu_prover per device; a prover is not thread-safe; calls are host-synchronous.
 *   - the library never falls back to a CPU implementation: without a usable HIP device
 *     vgpu_prover_create fails with VGPU_ERR_HIP.
 */
#ifndef VGPU_H
#define VGPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
    VGPU_OK = 0,
    VGPU_ERR_INVALID_ARG = -1,
    VGPU_ERR_OOM = -2,
    VGPU_ERR_HIP = -3,
    VGPU_ERR_UNSUPPORTED = -4,
    VGPU_ERR_INTERNAL = -5,
    VGPU_ERR_FABRIC = -6   /* a sharded proof was given up because a peer rank failed, or the transport itself failed / ran into its deadline */
};
enum {
    VGPU_HASH_KECCAK256 = 0,  /* the reference's MMCS: SerializingHasher32<Keccak256Hash> + CompressionFunctionFromHasher (basic/tests/test_prover.rs:424-431) */
    VGPU_HASH_POSEIDON16 = 1  /* BASELINE.json north-star variant: PaddingFreeSponge<Perm16, 16, 8, 8> + TruncatedPermutation<Perm16, 2, 8, 16> over the
                                 challenger's Poseidon-16 (same round constants); the reference never instantiates it */
};

const char* vgpu_last_error(void);
const char* vgpu_version(void);

/* ---- StarkConfig (machine/src/config.rs:7-31; instantiated at basic/tests/test_prover.rs:413-455) ---- */
typedef struct vgpu_config {
    int32_t device;                /* HIP device ordinal */
    uint32_t log_blowup;           /* FriConfig.log_blowup      (test_prover.rs:443) */
    uint32_t num_queries;          /* FriConfig.num_queries     (:444) */
    uint32_t pow_bits;             /* FriConfig.proof_of_work_bits (:445) */
    uint32_t hash_kind;            /* VGPU_HASH_KECCAK256 | VGPU_HASH_POSEIDON16 */
    uint32_t observe_final_poly;   /* convention switch, default 0 */
    uint32_t poseidon_rc[480];     /* Poseidon<_,CosetMds<16>,16,5> round constants (test_prover.rs:418-422), canonical */
    uint32_t interpret_air;        /* 0: BasicMachine chips run their ahead-of-time compiled eval kernels; 1: every AIR, in-tree
                                      or captured through vgpu_air_*, runs as an interpreted register program (same values) */
} vgpu_config_t;

/* ---- AIR capture: the FFI image of SymbolicAirBuilder (machine/src/symbolic/symbolic_builder.rs:57-154).
 * A host runs its unchanged Chip::eval against a builder that forwards to these calls; node ids are
 * opaque u32 handles.  Interactions mirror Interaction/VirtualPairCol (machine/src/chip.rs:76-94). ---- */
typedef struct vgpu_air vgpu_air_t;
typedef struct vgpu_vcol_term { uint32_t is_preprocessed; uint32_t column; uint32_t weight; } vgpu_vcol_term_t;
typedef struct vgpu_vcol { const vgpu_vcol_term_t* terms; uint32_t n_terms; uint32_t constant; } vgpu_vcol_t;
typedef struct vgpu_interaction {
    const vgpu_vcol_t* fields; uint32_t n_fields;
    vgpu_vcol_t count;
    uint32_t is_global;   /* BusArgument::Global / Local */
    uint32_t bus_index;
    uint32_t is_send;     /* InteractionType::*Send / *Receive */
} vgpu_interaction_t;

int32_t vgpu_air_new(const char* name, uint32_t width, uint32_t preprocessed_width, vgpu_air_t** out);
void vgpu_air_free(vgpu_air_t* air);
uint32_t vgpu_air_constant(vgpu_air_t* air, uint32_t canonical);                                  /* SymbolicExpression::Constant */
uint32_t vgpu_air_variable(vgpu_air_t* air, uint32_t is_preprocessed, uint32_t column, uint32_t is_next); /* ::Variable */
uint32_t vgpu_air_is_first_row(vgpu_air_t* air);
uint32_t vgpu_air_is_last_row(vgpu_air_t* air);
uint32_t vgpu_air_is_transition(vgpu_air_t* air);
uint32_t vgpu_air_add(vgpu_air_t* air, uint32_t a, uint32_t b);
uint32_t vgpu_air_sub(vgpu_air_t* air, uint32_t a, uint32_t b);
uint32_t vgpu_air_mul(vgpu_air_t* air, uint32_t a, uint32_t b);
uint32_t vgpu_air_neg(vgpu_air_t* air, uint32_t a);
void vgpu_air_assert_zero(vgpu_air_t* air, uint32_t node);                                       /* AirBuilder::assert_zero */
int32_t vgpu_air_add_interaction(vgpu_air_t* air, const vgpu_interaction_t* it);                 /* Chip::all_interactions order */

/* ---- Machine: ordered chips (basic/src/lib.rs:151-166) ---- */
typedef struct vgpu_machine vgpu_machine_t;
int32_t vgpu_machine_new(vgpu_machine_t** out);
/* compiles the constraint program.  log_quotient_degree 1 (every reference chip) .. 3 (constraint degree <= 9) is implemented;
 * beyond that VGPU_ERR_UNSUPPORTED with a message naming the AIR and its degree */
int32_t vgpu_machine_push_air(vgpu_machine_t* m, const vgpu_air_t* air);
int32_t vgpu_machine_basic(vgpu_machine_t** out);                         /* the 14-chip BasicMachine from the in-tree chip definitions */
/* the same 14 chips, but captured the way a foreign host captures them: each chip's eval runs against a builder that only
 * calls vgpu_air_* / vgpu_air_add_interaction, then vgpu_machine_push_air (no native kernels: the interpreted path) */
int32_t vgpu_machine_basic_via_ffi(vgpu_machine_t** out);
void vgpu_machine_free(vgpu_machine_t* m);
uint32_t vgpu_machine_num_chips(const vgpu_machine_t* m);
/* per-chip facts: width, preprocessed width, #interactions, log_quotient_degree (get_log_quotient_degree,
 * symbolic_builder.rs:17-30), #constraints, program length, registers */
int32_t vgpu_machine_chip_info(const vgpu_machine_t* m, uint32_t chip, uint32_t out[8]);
/* Neutral word image of chip `chip`'s interactions in Chip::all_interactions order (machine/src/chip.rs:40-63; test hook):
 *   [n] then per interaction: [is_send] [is_global] [bus_index] [n_fields] count_vcol field_vcols..;
 *   vcol = [n_terms] [constant] n_terms x ([is_preprocessed] [column] [weight]).  Returns the word count (copies when out has room). */
int64_t vgpu_machine_interaction_words(const vgpu_machine_t* m, uint32_t chip, uint32_t* out, uint64_t cap);
/* Host interpretation of chip `chip`'s compiled program on one row pair (test hook; no GPU needed).
 * values out: the asserted constraint values in order; returns their count or a negative status. */
int32_t vgpu_machine_eval_constraints(const vgpu_machine_t* m, uint32_t chip, const uint32_t* main_local, const uint32_t* main_next,
                                      const uint32_t* prep_local, const uint32_t* prep_next, uint32_t is_first, uint32_t is_last,
                                      uint32_t is_transition, uint32_t* out, uint32_t cap);

/* ---- Challenger: DuplexChallenger<Val, Poseidon16, 16> (basic/src/lib.rs:185,200,224,229,...) ---- */
typedef struct vgpu_challenger vgpu_challenger_t;
int32_t vgpu_challenger_new(const uint32_t poseidon_rc[480], vgpu_challenger_t** out);
void vgpu_challenger_free(vgpu_challenger_t* ch);
void vgpu_challenger_observe(vgpu_challenger_t* ch, const uint32_t* values, uint64_t n);         /* CanObserve */
void vgpu_challenger_sample(vgpu_challenger_t* ch, uint32_t* out, uint64_t n);                    /* sample / sample_ext_element = 5 samples */
uint64_t vgpu_challenger_sample_bits(vgpu_challenger_t* ch, uint32_t bits);
uint32_t vgpu_challenger_grind(vgpu_challenger_t* ch, uint32_t bits);                             /* smallest witness */
void vgpu_poseidon16_permute(const uint32_t poseidon_rc[480], uint32_t state[16]);
/* the same permutation through the sparse-matrix form of the 22 partial rounds that the Poseidon-MMCS kernels use (test hook) */
int32_t vgpu_poseidon16_permute_sparse(const uint32_t poseidon_rc[480], uint32_t state[16]);

/* ---- Prover (one per device) ---- */
typedef struct vgpu_prover vgpu_prover_t;
typedef struct vgpu_trace vgpu_trace_t;   /* a RowMajorMatrix resident in HBM */
typedef struct vgpu_pdata vgpu_pdata_t;   /* Pcs::ProverData: committed LDEs + Merkle tree, in HBM */
typedef struct vgpu_proof vgpu_proof_t;

int32_t vgpu_prover_create(const vgpu_config_t* cfg, const vgpu_machine_t* machine, vgpu_prover_t** out);
void vgpu_prover_destroy(vgpu_prover_t* p);
/* bytes currently held / peak in the HBM pool */
void vgpu_prover_memory(const vgpu_prover_t* p, uint64_t* live_bytes, uint64_t* peak_bytes);
/* The prover caches device blocks by size and reuses them (no hipMalloc in the steady state).  Returns the cached-but-unused
 * blocks to the driver (bytes freed); also done automatically, once, when an allocation runs out of memory. */
uint64_t vgpu_prover_trim(vgpu_prover_t* p);
/* The pool's peak restarts from what is held now (a host that wants the high-water mark of ONE phase: bench.py reports the proving path's own). */
void vgpu_prover_memory_reset_peak(vgpu_prover_t* p);

/* per-kernel HIP-event timing (bench): switch on/off (resets the accumulators); the profile is text,
 * one line per kernel: "name launches total_ms total_algorithmic_bytes".  Returns the size needed. */
void vgpu_prover_set_profiling(vgpu_prover_t* p, uint32_t on);
/* Optional, off by default: keep the commitment to the PREPROCESSED traces (program ROM, range table: it depends on the machine and the program only,
 * not on the witness) across proofs while the SAME vgpu_trace_t handles are handed in again — working-layout copies, LDEs and tree stay on the
 * device; any other set (a re-upload of equal contents included) recomputes it; the root is observed into the transcript every time.  Saves one
 * synchronisation point and about 40 small launches per proof.  VGPU_PREP_CACHE=1 in the environment enables it for provers created afterwards. */
void vgpu_prover_set_prep_cache(vgpu_prover_t* p, uint32_t on);
/* restrict the timing to launches of ONE kernel name (NULL or "" = all): every timed launch carries a pair of events, which costs
 * the host and the command processor a little; a throughput measurement times only the kernel it reports a roofline for */
void vgpu_prover_set_profiling_filter(vgpu_prover_t* p, const char* kernel_name);
int64_t vgpu_prover_profile(vgpu_prover_t* p, char* out, uint64_t cap);
/* Measurement aid (bench.py's roofline): the shader clock the device sustains RIGHT NOW.  One wave on a stream of its own runs `iters` dependent
 * VALU additions and reads the shader-cycle counter and the 100 MHz wall clock before and after: out[0] = shader cycles, out[1] = wall ticks
 * (clock in Hz = out[0] / out[1] * 1e8).  Blocks the caller for the few tens of microseconds the wave runs; safe beside running proofs (no
 * allocation, no device-wide synchronisation after the first call).  Needs a HIP device. */
int32_t vgpu_shader_clock_probe(int32_t device, uint32_t iters, uint64_t out[2]);

/* Page-locked host memory for the matrices a host hands over: vgpu_trace_upload / vgpu_oplog_upload from such a buffer is one DMA at
 * PCIe rate; from ordinary (pageable) memory the runtime stages the copy through its own bounce buffers on the calling thread, which
 * at 516 MB of main traces per C2 proof is slower than the proof itself (bench.py: pcie_inclusive).  A Rust host would back its
 * RowMajorMatrix<Val> values with this allocator.  Needs a HIP device; free with vgpu_host_free. */
int32_t vgpu_host_alloc(uint64_t bytes, void** out);
void vgpu_host_free(void* ptr);

/* H2D of a host RowMajorMatrix (the reference passes these by value, basic/src/lib.rs:223) */
int32_t vgpu_trace_upload(vgpu_prover_t* p, const uint32_t* data, uint64_t height, uint64_t width, vgpu_trace_t** out);
void vgpu_trace_free(vgpu_trace_t* t);

/* pcs.commit_batches / commit_shifted_batches (basic/src/lib.rs:199,223,258,599): coset_shifts may be NULL */
int32_t vgpu_commit_batches(vgpu_prover_t* p, const vgpu_trace_t* const* mats, uint32_t n_mats, const uint32_t* coset_shifts,
                            uint32_t root[8], vgpu_pdata_t** out);
/* pcs.get_ldes (basic/src/lib.rs:201,225,261): copy LDE `idx` back as row-major canonical, rows in
 * COMMITTED (bit-reversed) order; out must hold (height << log_blowup) * width words */
int32_t vgpu_pdata_lde(vgpu_prover_t* p, const vgpu_pdata_t* pd, uint32_t idx, uint32_t* out, uint64_t cap_words);
/* pcs.get_ldes without a copy: the committed LDE `idx` where it lives in HBM.  Column-major (column c at data + c * stride),
 * Montgomery form (x * 2^32 mod p), rows in COMMITTED order: storage row j holds the evaluation at coset_shift * w^bitrev(j),
 * i.e. the reference's natural-order view row i is storage row bitrev(i); `vertically_strided(stride, offset)`
 * (machine/src/quotient.rs:41-47) = the first height / stride storage rows.  Valid until vgpu_pdata_free. */
typedef struct vgpu_lde_view {
    const uint32_t* data;   /* device pointer */
    uint64_t height, width, stride;
    uint32_t log_blowup;
} vgpu_lde_view_t;
uint32_t vgpu_pdata_num_matrices(const vgpu_pdata_t* pd);
int32_t vgpu_pdata_lde_view(const vgpu_pdata_t* pd, uint32_t idx, vgpu_lde_view_t* out);
void vgpu_pdata_free(vgpu_pdata_t* pd);

/* generate_permutation_trace (machine/src/chip.rs:121-208) for chip `chip` of the prover's machine.
 * challenges: 3 extension elements (15 words).  out: height x 5(M+1) row-major (flatten_to_base),
 * cumulative_sum: 5 words. */
int32_t vgpu_perm_trace(vgpu_prover_t* p, uint32_t chip, const vgpu_trace_t* main, const vgpu_trace_t* preprocessed_or_null,
                        const uint32_t challenges[15], uint32_t* out, uint64_t cap_words, uint32_t cumulative_sum[5]);

/* The same, leaving the permutation trace in HBM as a trace handle (flatten_to_base layout: ready for vgpu_commit_batches) —
 * the call a host makes at basic/src/lib.rs:232-258 when it drives the phases itself. */
int32_t vgpu_perm_trace_device(vgpu_prover_t* p, uint32_t chip, const vgpu_trace_t* main, const vgpu_trace_t* preprocessed_or_null,
                               const uint32_t challenges[15], vgpu_trace_t** out, uint32_t cumulative_sum[5]);

/* quotient (machine/src/quotient.rs:18-37) of chip `chip`: evaluates Air::eval + eval_permutation_constraints on the quotient
 * domain from the chip's three committed LDEs (pcs.get_ldes of the preprocessed / main / permutation rounds: matrix indices
 * within each ProverData; prep_pd may be NULL for chips without preprocessed columns), divides by the zerofier, decomposes and
 * flattens (quotient.rs:63-67).  Result: the height x (5 << log_quotient_degree) chunk matrix as a trace handle, to be committed
 * with vgpu_commit_batches(.., coset_shifts = coset_shift^(2^log_quotient_degree)) (basic/src/lib.rs:593-599). */
int32_t vgpu_quotient(vgpu_prover_t* p, uint32_t chip, const vgpu_pdata_t* prep_pd, uint32_t prep_idx, const vgpu_pdata_t* main_pd, uint32_t main_idx,
                      const vgpu_pdata_t* perm_pd, uint32_t perm_idx, const uint32_t perm_challenges[15], const uint32_t alpha[5],
                      const uint32_t cumulative_sum[5], vgpu_trace_t** out);

/* pcs.open_multi_batches (basic/src/lib.rs:611-619).  rounds[r]: ProverData of round r; points: for every round, for every
 * committed matrix of that round (commit order), n_points[k] extension elements (5 words each), all concatenated; n_points has
 * one entry per (round, matrix).  The transcript `ch` is advanced as the reference's `&mut challenger` is (batch challenge, FRI
 * betas, proof-of-work witness, query indices).  Result handle:
 *   vgpu_opening_values: openings[round][matrix][point][column] flattened in that order, 5 words per value (lib.rs:622-645);
 *   vgpu_opening_proof:  the TwoAdicFriPcsProof as words — exactly the tail of vgpu_proof_words after the per-chip section. */
typedef struct vgpu_opening vgpu_opening_t;
int32_t vgpu_open_multi_batches(vgpu_prover_t* p, const vgpu_pdata_t* const* rounds, uint32_t n_rounds, const uint32_t* n_points,
                                const uint32_t* points, vgpu_challenger_t* ch, vgpu_opening_t** out);
uint64_t vgpu_opening_values_len(const vgpu_opening_t* o);
const uint32_t* vgpu_opening_values(const vgpu_opening_t* o);
uint64_t vgpu_opening_proof_len(const vgpu_opening_t* o);
const uint32_t* vgpu_opening_proof(const vgpu_opening_t* o);
void vgpu_opening_free(vgpu_opening_t* o);

/* pcs.verify_multi_batches (basic/src/lib.rs:825-837).  Host-only (no device is touched; only log_blowup, num_queries, pow_bits,
 * hash_kind, observe_final_poly and poseidon_rc of `cfg` are read).  commits: n_rounds x 8 words; n_mats[r] matrices per round;
 * heights / widths / n_points: one entry per (round, matrix), rounds concatenated, heights = TRACE heights (Dimensions);
 * points / values: as vgpu_open_multi_batches takes / returns them; proof: vgpu_opening_proof words.  The transcript `ch`
 * must be in the state the prover's was in when it called open_multi_batches.  Returns VGPU_OK if the opening is accepted,
 * VGPU_ERR_INVALID_ARG with the reason in vgpu_last_error() if it is rejected. */
int32_t vgpu_verify_multi_batches(const vgpu_config_t* cfg, const uint32_t* commits, uint32_t n_rounds, const uint32_t* n_mats, const uint64_t* heights,
                                  const uint32_t* widths, const uint32_t* n_points, const uint32_t* points, const uint32_t* values, uint64_t n_value_words,
                                  const uint32_t* proof, uint64_t n_proof_words, vgpu_challenger_t* ch);

/* Machine::verify (basic/src/lib.rs:677-1064; verify_constraints, machine/src/verify.rs:11-107): checks a proof produced by vgpu_prove
 * (flat "VPF1" words) against `machine` on the HOST — transcript, pcs.verify_multi_batches over the three rounds, every chip's AIR and
 * permutation constraints out of domain against Z_H(zeta) * quotient(zeta), and that the chips' cumulative sums cancel.  Needs no device.
 * preprocessed_commit: the commitment of the preprocessed traces (8 words; the reference's verifier recomputes it with
 * pcs.commit_batches(preprocessed_traces), lib.rs:791-804: vgpu_host_commit_root does that on the host), or NULL for a machine without
 * preprocessed traces.  Returns VGPU_OK when the proof is accepted, VGPU_ERR_INVALID_ARG with the reason in vgpu_last_error when not. */
int32_t vgpu_verify(const vgpu_config_t* cfg, const vgpu_machine_t* machine, const uint32_t* preprocessed_commit, const uint32_t* proof_words,
                    uint64_t n_words);
/* pcs.commit_batches / commit_shifted_batches on the HOST (the same LDE and MMCS conventions as vgpu_commit_batches, plain O(n log n)
 * code): for the small matrices a verifier commits itself.  mats[i]: canonical row-major heights[i] x widths[i]. */
int32_t vgpu_host_commit_root(const vgpu_config_t* cfg, const uint32_t* const* mats, const uint64_t* heights, const uint64_t* widths, uint32_t n_mats,
                              const uint32_t* coset_shifts, uint32_t root[8]);

/* FRI fold_even_odd of an Ext5 vector (n x 5 words, bit-reversed domain order) — App. B10 */
int32_t vgpu_fri_fold(vgpu_prover_t* p, const uint32_t* f, uint64_t n, const uint32_t beta[5], uint32_t* out);

/* Machine::prove (basic/src/lib.rs:147-675).  main[i] = trace of chip i; preprocessed traces are given
 * with their chip indices in chip order (BasicMachine: program, range).
 * debug_flags: 1 = keep the per-chip intermediate matrices (vgpu_proof_debug_*); 2 = run check_constraints and
 * check_cumulative_sums on the device before committing, as debug builds of the reference do (basic/src/lib.rs:270-375):
 * a violated constraint fails the call with VGPU_ERR_INVALID_ARG and a message naming chip, constraint and row.
 * The traces may belong to ANOTHER prover context of the same device (a host that uploads or generates segment i+1 on a context of its
 * own while this one proves segment i): the call first waits for that context's queued work and keeps it alive until the proof is done.
 * A context runs one proof at a time; calls from several threads (and the workers of several vgpu_prove_async tickets) QUEUE on it and are
 * served one after the other, as several threads may call prove on the reference's `Machine: Sync` (machine/src/machine.rs:13).
 * The commitment to the preprocessed traces (basic/src/lib.rs:189-201) is recomputed in every call, as in the reference; see
 * vgpu_prover_set_prep_cache for hosts that prove many segments of one program. */
int32_t vgpu_prove(vgpu_prover_t* p, const vgpu_trace_t* const* main, uint32_t n_main, const uint32_t* prep_chips,
                   const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t debug_flags, vgpu_proof_t** out);
/* The same, asynchronously: returns at once, a host thread of its own drives this prover's streams.  Two provers on one
 * GPU, each with one ticket outstanding, keep two proofs in flight from a single caller thread (independent segments:
 * one proof's latency-bound Merkle-top / FRI chain overlaps the other's commits).  The traces must outlive the wait. */
typedef struct vgpu_ticket vgpu_ticket_t;
int32_t vgpu_prove_async(vgpu_prover_t* p, const vgpu_trace_t* const* main, uint32_t n_main, const uint32_t* prep_chips,
                         const vgpu_trace_t* const* prep, uint32_t n_prep, vgpu_ticket_t** out);
int32_t vgpu_ticket_wait(vgpu_ticket_t* t, vgpu_proof_t** out);   /* consumes the ticket */
/* CBOR image of the proof (ciborium over the serde-derived MachineProof, basic/src/bin/valida.rs:425-427; field names of
 * machine/src/proof.rs:13-44 and, for the PCS proof, SURVEY.md Appendix B12 — unpinned).  flags: VGPU_CBOR_*.  Returns the
 * length in bytes (copying when out has room) or a negative status. */
#define VGPU_CBOR_CANONICAL_FIELDS 1u  /* BabyBear as canonical u32 instead of the derived {"value": <Montgomery word>} */
#define VGPU_CBOR_PLAIN_DIGESTS 2u     /* commitments as [Val; 8] instead of Hash { value, _marker } */
int64_t vgpu_proof_cbor(const uint32_t* proof_words, uint64_t n_words, uint32_t flags, uint8_t* out, uint64_t cap_bytes);
/* The way back (ciborium::from_reader on the verifier's side; the reference's tests verify a proof after exactly this round trip,
 * basic/tests/test_prover.rs:456-469): CBOR image -> proof words.  Accepts either setting of the two switches per value.  Returns the word
 * count (copies when out has room), or a negative status with the reason in vgpu_last_error. */
int64_t vgpu_proof_from_cbor(const uint8_t* bytes, uint64_t n_bytes, uint32_t* out, uint64_t cap_words);
/* The same for a first contact with a proof file of the real `valida prove` (every encoding convention of the absent crates is recall):
 * flags & VGPU_CBOR_IN_BARE_IS_MONTGOMERY reads a BARE integer field element as the raw Montgomery word instead of the canonical value;
 * *forms_seen (may be null) reports what the image contained: VGPU_CBOR_SAW_* bits. */
#define VGPU_CBOR_IN_BARE_IS_MONTGOMERY 1u
#define VGPU_CBOR_SAW_FIELD_STRUCT 1u    /* {"value": m} */
#define VGPU_CBOR_SAW_FIELD_BARE 2u
#define VGPU_CBOR_SAW_DIGEST_STRUCT 4u   /* Hash { value, _marker } */
#define VGPU_CBOR_SAW_DIGEST_PLAIN 8u
int64_t vgpu_proof_from_cbor_ex(const uint8_t* bytes, uint64_t n_bytes, uint32_t flags, uint32_t* out, uint64_t cap_words, uint32_t* forms_seen);
uint64_t vgpu_proof_len(const vgpu_proof_t* pr);            /* u32 words of the flat "VPF1" encoding */
const uint32_t* vgpu_proof_words(const vgpu_proof_t* pr);
/* 11 doubles, ms: ingest, commit_main, perm, commit_perm, quotient, commit_quotient, open_values, open_reduce, fri, queries, total */
void vgpu_proof_phase_ms(const vgpu_proof_t* pr, double out[11]);
/* transcript probe: 8 (preprocessed root) + 15 (perm challenges) + 5 (alpha) + 5 (zeta) words */
void vgpu_proof_transcript(const vgpu_proof_t* pr, uint32_t out[33]);
/* with debug_flags & 1: per-chip intermediate matrices, row-major canonical, natural row order */
int64_t vgpu_proof_debug_perm_trace(const vgpu_proof_t* pr, uint32_t chip, uint32_t* out, uint64_t cap_words);
int64_t vgpu_proof_debug_quotient(const vgpu_proof_t* pr, uint32_t chip, uint32_t* out, uint64_t cap_words);
void vgpu_proof_free(vgpu_proof_t* pr);

/* ---- RCCL inside the library: one process per GPU; the host's launcher distributes the 128-byte id that rank 0
 * obtains from vgpu_comm_unique_id (any out-of-band channel: MPI, a file, the Rust host's own RPC), every rank then calls
 * vgpu_comm_init with its prover.  vgpu_comm_allgather_roots is the path's one collective: each segment's commitment roots
 * (3 x 8 words after vgpu_prove) to every rank over xGMI; out holds world * n_words words, rank-major. ---- */
typedef struct vgpu_comm vgpu_comm_t;
#define VGPU_COMM_ID_BYTES 128
int32_t vgpu_comm_unique_id(uint8_t id[VGPU_COMM_ID_BYTES]);
int32_t vgpu_comm_init(vgpu_prover_t* p, const uint8_t id[VGPU_COMM_ID_BYTES], uint32_t rank, uint32_t world, vgpu_comm_t** out);
int32_t vgpu_comm_allgather_roots(vgpu_comm_t* c, const uint32_t* words, uint32_t n_words, uint32_t* out);
void vgpu_comm_destroy(vgpu_comm_t* c);
/* Deadline of every collective issued through `c` (the roots all-gather, the exchanges of vgpu_prove_sharded): one that has not completed
 * within timeout_ms — a peer died or never entered it — is aborted (ncclCommAbort), the call fails (VGPU_ERR_HIP / VGPU_ERR_FABRIC) and the
 * communicator is dead: every later call on it is refused, create a new one.  0 (the default, or the environment's VGPU_COMM_TIMEOUT_MS) = wait for ever. */
int32_t vgpu_comm_set_timeout_ms(vgpu_comm_t* c, uint32_t timeout_ms);

/* ---- ONE proof sharded over several GPUs.  Its commitment round alone: pcs.commit_batches of one round sharded over the
 * ranks of `comm` — column-sharded LDEs, an all-to-all into row-range shards, a subtree per rank, an all-gather of the subtree
 * roots.  Every rank passes the SAME matrices (only its own columns are extended) and receives the SAME root that
 * vgpu_commit_batches gives on one GPU.  world must be a power of two. */
int32_t vgpu_commit_batches_sharded(vgpu_prover_t* p, vgpu_comm_t* comm, const vgpu_trace_t* const* mats, uint32_t n_mats, const uint32_t* coset_shifts,
                                    uint32_t root[8]);
/* The same phases with `world` prover contexts of THIS process standing in for the ranks (a box with one GPU): the exchanges
 * are device-to-device copies.  mats[r * n_mats + i] = matrix i as uploaded through provers[r]. */
int32_t vgpu_commit_batches_sharded_local(vgpu_prover_t* const* provers, uint32_t world, const vgpu_trace_t* const* mats, uint32_t n_mats,
                                          const uint32_t* coset_shifts, uint32_t root[8]);
/* The WHOLE of Machine::prove (basic/src/lib.rs:147-675) for one proof over the ranks of `comm`: every committed LDE, every Merkle
 * tree, the quotient evaluation, the opened values, the reduced openings and the FRI layers live and are computed in row-range shards
 * (a rank's row range of a bit-reversed LDE on s H_L is the sub-coset s w_L^e H_{L/W}, again bit-reversed: the single-GPU kernels run on
 * it with a shifted coset); the transcript is replicated.  Exchanges per proof: an all-to-all (columns -> row ranges) and a roots
 * all-gather per commitment round, one halo exchange (the successor shard) and one all-to-all (rows -> columns) around the quotient,
 * an all-gather of partial opened values, a roots all-gather per sharded FRI layer, an all-gather of the proof tail.  Every rank
 * passes the SAME traces (they are replicated; what a proof's memory goes into — LDEs, trees, FRI layers — is sharded) and
 * receives the SAME proof words vgpu_prove gives on one GPU.  Matrices whose LDE has fewer than max(4 world, 2^log_min_sharded) rows
 * are computed whole by every rank.  Any log_blowup (the quotient domain, machine/src/quotient.rs:41-47, is the first world >> (log_blowup - 1)
 * ranks' row ranges: those ranks evaluate the quotient); chips of log_quotient_degree 1; world must be a power of two.  The call
 * owns the prover context and the communicator until it returns: no other proof on `p`, no vgpu_comm_* call on `comm` from another thread
 * meanwhile (RCCL serialises the operations of one communicator), and every rank must make the same call with traces of the same shapes. */
int32_t vgpu_prove_sharded(vgpu_prover_t* p, vgpu_comm_t* comm, const vgpu_trace_t* const* main, uint32_t n_main, const uint32_t* prep_chips,
                           const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t log_min_sharded, vgpu_proof_t** out);
/* The same with `world` prover contexts of THIS process standing in for the ranks (device-to-device copies for the exchanges):
 * main[r * n_main + i] / prep[r * n_prep + k] = the traces as uploaded through provers[r]. */
int32_t vgpu_prove_sharded_local(vgpu_prover_t* const* provers, uint32_t world, const vgpu_trace_t* const* main, uint32_t n_main,
                                 const uint32_t* prep_chips, const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t log_min_sharded, vgpu_proof_t** out);

/* The same over the HOST'S OWN transport: one rank per process (or per thread with its own prover), the exchanges carried by two
 * callbacks over host buffers — what a Rust host plugs its MPI / TCP / shared-memory layer into, and how the sharded prover runs one rank
 * per process where no RCCL communicator exists.  Both callbacks are collective (every rank calls them in the same order with matching
 * sizes), block until this rank's data has arrived, and return 0 on success (anything else aborts the proof on every rank).
 *   all_gather : every rank contributes n_words words; out receives world * n_words words, rank-major.
 *   all_to_all : send[s] (send_words[s] words) goes to rank s, recv[s] (recv_words[s] words) arrives from rank s; the entries of s == rank
 *                are null / 0 (the library keeps its own block on the device).  Device blocks are staged through page-locked host memory.
 * Failure protocol: every exchange is staged first (everything that can fail on this rank alone: staging buffers, device-to-host copies),
 * then the ranks all_gather one status word, then the exchange runs.  A rank whose proof fails (bad shapes, out of memory, a HIP error)
 * reports it in the status word instead of going on, and EVERY rank returns VGPU_ERR_FABRIC (vgpu_last_error names the failing rank) instead
 * of blocking in an exchange its peer never enters.  A callback that itself returns non-zero is fatal for the transport: that rank returns
 * at once, issues no further callback, and its peers leave through their own callback's error or through the deadline.
 * Deadline: timeout_ms > 0 bounds every single callback.  The library then runs the callbacks on a helper thread of its own (they must be
 * callable from another thread than the caller's); one that has not returned in time is ABANDONED there — the library never touches its
 * buffers again, the helper thread ends if the callback ever returns — and the call fails with VGPU_ERR_FABRIC.  A peer that died costs the
 * survivors at most timeout_ms, never a hang.  timeout_ms == 0: callbacks run on the calling thread and must bound themselves.
 * struct_size must be sizeof(vgpu_fabric_t): a host compiled against another layout of this struct is refused, not misread. */
typedef struct vgpu_fabric {
    uint32_t struct_size;   /* = sizeof(vgpu_fabric_t) */
    uint32_t timeout_ms;    /* deadline of one callback; 0 = none */
    void* user;
    uint32_t rank, world;   /* world: a power of two */
    int32_t (*all_gather)(void* user, const uint32_t* words, uint64_t n_words, uint32_t* out);
    int32_t (*all_to_all)(void* user, const uint32_t* const* send, const uint64_t* send_words, uint32_t* const* recv, const uint64_t* recv_words);
} vgpu_fabric_t;
int32_t vgpu_prove_sharded_fabric(vgpu_prover_t* p, const vgpu_fabric_t* fabric, const vgpu_trace_t* const* main, uint32_t n_main, const uint32_t* prep_chips,
                                  const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t log_min_sharded, vgpu_proof_t** out);
/* The same three entry points with ROW-RANGE inputs: the traces themselves are sharded.  A chip whose LDE is sharded — at least
 * max(4 world, 2^log_min_sharded, 2 * 2^log_blowup) LDE rows: vgpu_sharded_trace_is_split says so for a trace height — hands in ONLY its rows
 * [rank n / world, (rank + 1) n / world) (a trace of n / world rows, natural order); every other chip its whole trace.  full_heights[i] = chip i's
 * whole trace height n on every rank.  Per-rank trace memory and the permutation-trace work are then 1 / world too: generate_permutation_trace's
 * running sum (machine/src/chip.rs:176-205) becomes a local scan plus ONE all-gather of the ranks' totals (5 words per chip), and every
 * commitment round deals the row ranges into whole columns with one more all-to-all.  The preprocessed traces (constants of the program) are
 * handed in whole on every rank.  The proof words are those of vgpu_prove on one GPU. */
uint32_t vgpu_sharded_trace_is_split(uint32_t world, uint32_t log_blowup, uint32_t log_min_sharded, uint64_t height);
int32_t vgpu_prove_sharded_rows(vgpu_prover_t* p, vgpu_comm_t* comm, const vgpu_trace_t* const* main, uint32_t n_main, const uint64_t* full_heights, const uint32_t* prep_chips,
                                const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t log_min_sharded, vgpu_proof_t** out);
int32_t vgpu_prove_sharded_rows_fabric(vgpu_prover_t* p, const vgpu_fabric_t* fabric, const vgpu_trace_t* const* main, uint32_t n_main, const uint64_t* full_heights,
                                       const uint32_t* prep_chips, const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t log_min_sharded, vgpu_proof_t** out);
/* main[r * n_main + i] = rank r's rows (or whole trace) of chip i as uploaded through provers[r] */
int32_t vgpu_prove_sharded_rows_local(vgpu_prover_t* const* provers, uint32_t world, const vgpu_trace_t* const* main, uint32_t n_main, const uint64_t* full_heights,
                                      const uint32_t* prep_chips, const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t log_min_sharded, vgpu_proof_t** out);
/* Host-only check of a transport before proofs depend on it (no device needed): a status round, an all_gather of n_words rank-dependent
 * words and an all_to_all of rank-pair-dependent blocks of different sizes, every received word verified.  fail_rank < world makes that
 * rank fail between two exchanges the way a failing proof would: every rank must then return a non-zero status — none may hang. */
int32_t vgpu_fabric_selftest(const vgpu_fabric_t* fabric, uint32_t n_words, uint32_t fail_rank);

/* ---- trace generation on the device: Chip::generate_trace (machine/src/chip.rs:22) of the big
 * BasicMachine chips as kernels, fed by the VM's operation logs instead of host-built RowMajorMatrix traces.
 * The logs are what the reference's chips hold after Machine::run: Cpu::operations + pc/fp/instruction per cycle
 * (cpu/src/lib.rs:56-77), MemoryChip::operations: BTreeMap<clk, Vec<Operation>> flattened in (clk, issue) order
 * (memory/src/lib.rs:25-60), and each ALU chip's Vec<Operation> (alu_u32/src/add/mod.rs:26-36).  Words cross the
 * ABI as the u32 value of the big-endian Word (machine/src/core.rs:9). ---- */
enum { VGPU_CPU_STORE32 = 0, VGPU_CPU_LOAD32, VGPU_CPU_JAL, VGPU_CPU_JALV, VGPU_CPU_BEQ, VGPU_CPU_BNE, VGPU_CPU_IMM32, VGPU_CPU_BUS,
       VGPU_CPU_BUS_LEFT_IMM, VGPU_CPU_STOP, VGPU_CPU_LOADFP };   /* cpu Operation (cpu/src/lib.rs:40-54) */
typedef struct vgpu_cpu_op {
    uint32_t pc, fp, opcode;
    int32_t operands[5];
    uint32_t kind;        /* VGPU_CPU_* */
    uint32_t has_imm;     /* Operation::*(Some(imm)) */
    uint32_t imm;
    uint32_t mem_first;   /* index of this cycle's first entry in the memory log */
} vgpu_cpu_op_t;
typedef struct vgpu_mem_op { uint32_t clk, addr, value, is_write; } vgpu_mem_op_t;   /* memory Operation::{Read,Write}(addr, value) at clk */
typedef struct vgpu_alu_op { uint32_t opcode, a, b, c; } vgpu_alu_op_t;              /* e.g. Operation::Add32(a, b, c): a = result */
typedef struct vgpu_out_op { uint32_t clk, byte; } vgpu_out_op_t;                    /* OutputChip::values entry (clk, byte) (output/src/lib.rs:21-23) */
typedef struct vgpu_oplog_desc {
    uint64_t struct_size;   /* = sizeof(vgpu_oplog_desc_t): vgpu_oplog_upload refuses a host compiled against another layout of this struct */
    const vgpu_cpu_op_t* cpu; uint64_t n_cpu;
    const vgpu_mem_op_t* mem; uint64_t n_mem;
    const vgpu_alu_op_t* alu[4]; uint64_t n_alu[4];   /* add, sub, lt, bitwise */
    const uint32_t* static_cells; uint64_t n_static;  /* MemoryChip::static_data as (addr, value) pairs, ascending address (may be null / 0) */
    uint32_t rom_len;                                 /* ProgramROM length (program chip rows before padding) */
    /* the five remaining chips, each as the reference's chip holds it after Machine::run: Mul32Chip / Div32Chip / Shift32Chip / Com32Chip
     * ::operations (opcode = the Operation variant: MUL32 | MULHS32 | MULHU32, DIV32 | SDIV32, SHL32 | SHR32 | SRA32, NE32 | EQ32 — a shift
     * instruction also leaves a Mul32 / Div32 / SDiv32 with the power of two in the mul / div log, alu_u32/src/shift/mod.rs:207-212) and
     * OutputChip::values.  All may be null / 0. */
    const vgpu_alu_op_t* alu2[4]; uint64_t n_alu2[4];  /* mul, div, shift, com */
    const vgpu_out_op_t* output; uint64_t n_output;
} vgpu_oplog_desc_t;
typedef struct vgpu_oplog vgpu_oplog_t;
int32_t vgpu_oplog_upload(vgpu_prover_t* p, const vgpu_oplog_desc_t* log, vgpu_oplog_t** out);
void vgpu_oplog_free(vgpu_oplog_t* log);
/* Any chip of the BasicMachine — all fourteen Chip::generate_trace as kernels: cpu, program, mem, add, sub, lt, bitwise, mul, div, shift,
 * com, output from their logs (the last five exactly as incomplete as the reference fills them: div / com rows carry only the opcode
 * flag, mul leaves r / s zero, output never writes counter / counter_mult / opcode), range from the result words of the instructions
 * that range-check them (add, sub, mul*, div*: the cpu log's bus operations and the cycle's memory write), static_data from the
 * initialised cells.  The returned trace is already in the prover's working layout (no ingest pass). */
int32_t vgpu_generate_trace(vgpu_prover_t* p, const vgpu_oplog_t* log, uint32_t chip, vgpu_trace_t** out);
void vgpu_trace_shape(const vgpu_trace_t* t, uint64_t* height, uint64_t* width);
/* canonical row-major copy of a device trace (what the reference's generate_trace would have returned) */
int32_t vgpu_trace_download(vgpu_prover_t* p, const vgpu_trace_t* t, uint32_t* out, uint64_t cap_words);

/* ---- synthetic workloads (bench inputs; upstream of the hot path) ---- */
typedef struct vgpu_workload vgpu_workload_t;
/* fib_program (basic/tests/test_prover.rs:35-188) with loop bound n, fp = 0x1000, run to STOP, traces generated */
int32_t vgpu_workload_fib(uint32_t n, vgpu_workload_t** out);
/* ALU-heavy loop (SURVEY.md §8 workload C4): add, sub, xor, and, or, lt, addi, addi, bne per iteration */
int32_t vgpu_workload_alu(uint32_t iters, vgpu_workload_t** out);
/* the reference's other pinned prover programs (basic/tests/test_prover.rs:190-402): "left_imm_ops", "signed_inequality", "loadfp",
 * and "static_data" (basic/tests/test_static_data.rs:31-59, cells 0x10 / 0x14 initialised through the static-data chip) */
int32_t vgpu_workload_named(const char* name, vgpu_workload_t** out);
/* final value of the 32-bit memory cell at `addr` (machine.mem().cells, test_prover.rs:483-486,494-640) */
int32_t vgpu_workload_cell(const vgpu_workload_t* w, uint32_t addr, uint32_t* value);
void vgpu_workload_free(vgpu_workload_t* w);
/* stats: [cycles, cpu ops, memory ops, add ops, result word (u32 at fp+4), program length, padded cpu height] */
void vgpu_workload_stats(const vgpu_workload_t* w, uint64_t out[8]);
int32_t vgpu_workload_main_trace(const vgpu_workload_t* w, uint32_t chip, const uint32_t** data, uint64_t* height, uint64_t* width);
/* the VM's operation logs (pointers stay valid until vgpu_workload_free) */
void vgpu_workload_oplog(const vgpu_workload_t* w, vgpu_oplog_desc_t* out);
/* k = 0: program ROM (chip 1), k = 1: range table (chip 12) */
int32_t vgpu_workload_preprocessed(const vgpu_workload_t* w, uint32_t k, uint32_t* chip, const uint32_t** data, uint64_t* height, uint64_t* width);

#ifdef __cplusplus
}
#endif
#endif /* VGPU_H */
