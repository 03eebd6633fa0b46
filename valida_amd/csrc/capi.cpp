// extern "C" surface of libvgpu.so — see include/vgpu.h for the contract and the reference item each
// entry point replaces.
#include "../../include/vgpu.h"
#include <algorithm>
#include <cstring>
#include <future>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include "host/cbor.hpp"
#include "host/comm.hpp"
#include "host/sharded.hpp"
#include "host/sharded_prover.hpp"
#include "host/verifier.hpp"
#include "host/machine_verifier.hpp"
#include "host/poseidon_opt.hpp"
#include "host/prover.hpp"
#include "workload/basic_vm.hpp"

using namespace vhost;

static thread_local std::string g_err;
static int32_t fail(int32_t code, const std::string& msg) { g_err = msg; return code; }
#define VG_TRY(...)                                                         \
    try { __VA_ARGS__; return VGPU_OK; }                                           \
    catch (const FabricPeerFailure& e) { return fail(VGPU_ERR_FABRIC, e.what()); }          \
    catch (const FabricTransportFailure& e) { return fail(VGPU_ERR_FABRIC, e.what()); }     \
    catch (const std::invalid_argument& e) { return fail(VGPU_ERR_INVALID_ARG, e.what()); } \
    catch (const std::bad_alloc& e) { return fail(VGPU_ERR_OOM, e.what()); }                \
    catch (const std::exception& e) {                                       \
        std::string m = e.what();                                           \
        return fail(m.find("hip") != std::string::npos ? VGPU_ERR_HIP : VGPU_ERR_INTERNAL, m); \
    }

struct vgpu_air { std::string name; vair::Dag dag; std::vector<vair::Interaction> interactions; };
struct vgpu_machine { MachineDesc desc; };
struct vgpu_challenger { std::unique_ptr<Poseidon16> perm; std::unique_ptr<Challenger> ch; };
// Device objects keep their prover (device context, memory pool) alive: a host may free handles in any order — a garbage
// collector does — and a buffer must never outlive the pool it returns to.  `owner` is declared first, so it is destroyed last.
struct vgpu_prover { std::shared_ptr<Prover> p; };
struct vgpu_trace { std::shared_ptr<Prover> owner; std::shared_ptr<DeviceTrace> t; };  // shared: an asynchronous prove holds its inputs
struct vgpu_pdata { std::shared_ptr<Prover> owner; std::unique_ptr<ProverData> pd; };
struct vgpu_ticket { std::future<std::pair<vgpu_proof_t*, std::pair<int32_t, std::string>>> result; };
struct vgpu_proof { std::vector<uint32_t> words; PhaseTimes tm; ProveDebugOut dbg; };
struct vgpu_oplog { std::shared_ptr<Prover> owner; std::unique_ptr<DeviceOplog> log; };
struct vgpu_opening { std::vector<uint32_t> values, proof; };
struct vgpu_comm { std::shared_ptr<Prover> owner; std::unique_ptr<Comm> comm; };  // owner first: the communicator dies before its context
static_assert(sizeof(vgpu_cpu_op_t) == sizeof(vk::TgCpuOp) && sizeof(vgpu_mem_op_t) == sizeof(vk::TgMemOp) && sizeof(vgpu_alu_op_t) == sizeof(vk::TgAluOp),
              "C ABI log records and their device images must match");
static_assert((int)VGPU_CPU_LOADFP == (int)vk::TG_CPU_LOADFP, "cpu op kinds");
static_assert(sizeof(vgpu_out_op_t) == sizeof(vk::TgOutOp), "C ABI log records and their device images must match");
struct vgpu_workload {
    std::vector<vgpu_cpu_op_t> log_cpu;
    std::vector<vgpu_mem_op_t> log_mem;
    std::vector<vgpu_alu_op_t> log_alu[4], log_alu2[4];
    std::vector<vgpu_out_op_t> log_output;
    std::vector<uint32_t> log_static;  // (addr, value) pairs, ascending address
    std::unique_ptr<vwork::BasicVm> vm;
    std::vector<vwork::RowMajor> main;
    vwork::RowMajor prep_program, prep_range;
    uint64_t result = 0;
};

static void fill_logs(vgpu_workload& w);

extern "C" {

const char* vgpu_last_error(void) { return g_err.c_str(); }
const char* vgpu_version(void) { return "valida_amd 0.1 (gfx950)"; }

// ---- AIR capture
int32_t vgpu_air_new(const char* name, uint32_t width, uint32_t preprocessed_width, vgpu_air_t** out) {
    VG_TRY({
        if (!out) throw std::invalid_argument("null out");
        auto* a = new vgpu_air();
        a->name = name ? name : "";
        a->dag.width = (int)width; a->dag.prep_width = (int)preprocessed_width;
        *out = a;
    })
}
void vgpu_air_free(vgpu_air_t* air) { delete air; }
uint32_t vgpu_air_constant(vgpu_air_t* a, uint32_t canonical) { return a->dag.constant(canonical); }
uint32_t vgpu_air_variable(vgpu_air_t* a, uint32_t is_prep, uint32_t column, uint32_t is_next) { return a->dag.var(is_prep != 0, (int)column, is_next != 0); }
uint32_t vgpu_air_is_first_row(vgpu_air_t* a) { return a->dag.selector(vair::N_FIRST); }
uint32_t vgpu_air_is_last_row(vgpu_air_t* a) { return a->dag.selector(vair::N_LAST); }
uint32_t vgpu_air_is_transition(vgpu_air_t* a) { return a->dag.selector(vair::N_TRANS); }
uint32_t vgpu_air_add(vgpu_air_t* a, uint32_t x, uint32_t y) { return a->dag.add(x, y); }
uint32_t vgpu_air_sub(vgpu_air_t* a, uint32_t x, uint32_t y) { return a->dag.sub(x, y); }
uint32_t vgpu_air_mul(vgpu_air_t* a, uint32_t x, uint32_t y) { return a->dag.mul(x, y); }
uint32_t vgpu_air_neg(vgpu_air_t* a, uint32_t x) { return a->dag.neg(x); }
void vgpu_air_assert_zero(vgpu_air_t* a, uint32_t node) { a->dag.constraints.push_back(node); }
static vair::VirtualCol to_vcol(const vgpu_vcol_t& v) {
    vair::VirtualCol c;
    c.constant = v.constant;
    for (uint32_t i = 0; i < v.n_terms; i++) c.terms.push_back({v.terms[i].is_preprocessed != 0, (int)v.terms[i].column, v.terms[i].weight});
    return c;
}
int32_t vgpu_air_add_interaction(vgpu_air_t* a, const vgpu_interaction_t* it) {
    VG_TRY({
        if (!a || !it) throw std::invalid_argument("null argument");
        vair::Interaction x;
        for (uint32_t i = 0; i < it->n_fields; i++) x.fields.push_back(to_vcol(it->fields[i]));
        x.count = to_vcol(it->count);
        x.bus_kind = it->is_global ? vair::BusKind::Global : vair::BusKind::Local;
        x.bus_index = (int)it->bus_index;
        x.type = it->is_global ? (it->is_send ? vair::InteractionType::GlobalSend : vair::InteractionType::GlobalReceive)
                               : (it->is_send ? vair::InteractionType::LocalSend : vair::InteractionType::LocalReceive);
        a->interactions.push_back(std::move(x));
    })
}

// ---- machine
int32_t vgpu_machine_new(vgpu_machine_t** out) { VG_TRY({ if (!out) throw std::invalid_argument("null out"); *out = new vgpu_machine(); }) }
int32_t vgpu_machine_push_air(vgpu_machine_t* m, const vgpu_air_t* air) {
    VG_TRY({
        if (!m || !air) throw std::invalid_argument("null argument");
        AirDesc d = MachineDesc::make_air_from_dag(air->name, air->dag, air->interactions);
        // log_quotient_degree 1 (constraint degree <= 3 = LOOKUP_DEGREE_BOUND, machine/src/lib.rs:36) is what every chip of the
        // reference has; captured AIRs of degree up to 9 (log_quotient_degree 2, 3) run through the general quotient kernel.  Beyond
        // that the AIR is refused HERE, when it is pushed.
        if (d.log_quotient_degree < 1 || d.log_quotient_degree > 3) {
            g_err = "AIR '" + air->name + "': max constraint degree " + std::to_string(d.max_constraint_degree) + " gives log_quotient_degree " +
                    std::to_string(d.log_quotient_degree) + "; the device path implements log_quotient_degree 1..3 (constraint degree <= 9)";
            return VGPU_ERR_UNSUPPORTED;
        }
        m->desc.airs.push_back(std::move(d));
    })
}
int32_t vgpu_machine_basic(vgpu_machine_t** out) {
    VG_TRY({ if (!out) throw std::invalid_argument("null out"); auto* m = new vgpu_machine(); m->desc = MachineDesc::basic(); *out = m; })
}
void vgpu_machine_free(vgpu_machine_t* m) { delete m; }

// The BasicMachine built the way a foreign host builds it: every in-tree chip's `eval` runs against an AirBuilder that
// only forwards to the public vgpu_air_* entry points (the FFI image of SymbolicAirBuilder), its interactions go
// through vgpu_air_add_interaction, and the captured AIR is pushed with vgpu_machine_push_air.  The resulting chips
// carry no native kernel: they prove through the interpreted register program.
namespace {
struct FfiExpr {
    vgpu_air_t* air;
    uint32_t id;
    FfiExpr operator+(const FfiExpr& o) const { return {air, vgpu_air_add(air, id, o.id)}; }
    FfiExpr operator-(const FfiExpr& o) const { return {air, vgpu_air_sub(air, id, o.id)}; }
    FfiExpr operator*(const FfiExpr& o) const { return {air, vgpu_air_mul(air, id, o.id)}; }
    FfiExpr operator-() const { return {air, vgpu_air_neg(air, id)}; }
};
struct FfiBuilder {
    using Expr = FfiExpr;
    vgpu_air_t* air;
    Expr constant(uint32_t c) { return {air, vgpu_air_constant(air, c)}; }
    Expr main(int col, bool next) { return {air, vgpu_air_variable(air, 0, (uint32_t)col, next ? 1 : 0)}; }
    Expr preprocessed(int col, bool next) { return {air, vgpu_air_variable(air, 1, (uint32_t)col, next ? 1 : 0)}; }
    Expr is_first_row() { return {air, vgpu_air_is_first_row(air)}; }
    Expr is_last_row() { return {air, vgpu_air_is_last_row(air)}; }
    Expr is_transition() { return {air, vgpu_air_is_transition(air)}; }
    void assert_zero(const Expr& e) { vgpu_air_assert_zero(air, e.id); }
};
void ffi_vcol(const vair::VirtualCol& v, std::vector<vgpu_vcol_term_t>& store, vgpu_vcol_t& out) {
    store.clear();
    for (auto& t : v.terms) store.push_back({t.preprocessed ? 1u : 0u, (uint32_t)t.col, t.weight});
    out.terms = store.data(); out.n_terms = (uint32_t)store.size(); out.constant = v.constant;
}
}  // namespace
int32_t vgpu_machine_basic_via_ffi(vgpu_machine_t** out) {
    if (!out) return VGPU_ERR_INVALID_ARG;
    vgpu_machine_t* m = nullptr;
    int32_t rc = vgpu_machine_new(&m);
    for (int chip = 0; rc == 0 && chip < vchips::NUM_CHIPS; chip++) {
        const auto& info = vchips::chip_info(chip);
        vgpu_air_t* air = nullptr;
        rc = vgpu_air_new(info.name, (uint32_t)info.width, (uint32_t)info.preprocessed_width, &air);
        if (rc) break;
        FfiBuilder b{air};
        vchips::eval_chip(chip, b);
        for (const auto& it : vchips::chip_interactions(chip)) {
            std::vector<std::vector<vgpu_vcol_term_t>> terms(it.fields.size() + 1);
            std::vector<vgpu_vcol_t> fields(it.fields.size());
            for (size_t f = 0; f < it.fields.size(); f++) ffi_vcol(it.fields[f], terms[f], fields[f]);
            vgpu_interaction_t x{};
            x.fields = fields.data(); x.n_fields = (uint32_t)fields.size();
            ffi_vcol(it.count, terms.back(), x.count);
            x.is_global = it.is_local() ? 0 : 1; x.bus_index = (uint32_t)it.bus_index; x.is_send = it.is_send() ? 1 : 0;
            rc = vgpu_air_add_interaction(air, &x);
            if (rc) break;
        }
        if (rc == 0) rc = vgpu_machine_push_air(m, air);
        vgpu_air_free(air);
    }
    if (rc) { vgpu_machine_free(m); return rc; }
    *out = m;
    return 0;
}
uint32_t vgpu_machine_num_chips(const vgpu_machine_t* m) { return (uint32_t)m->desc.airs.size(); }
int32_t vgpu_machine_chip_info(const vgpu_machine_t* m, uint32_t chip, uint32_t out[8]) {
    VG_TRY({
        if (!m || chip >= m->desc.airs.size()) throw std::invalid_argument("bad chip index");
        const AirDesc& a = m->desc.airs[chip];
        out[0] = a.width; out[1] = a.prep_width; out[2] = (uint32_t)a.interactions.size(); out[3] = a.log_quotient_degree;
        out[4] = a.program.num_asserts; out[5] = (uint32_t)a.program.instrs.size(); out[6] = a.program.num_regs; out[7] = (uint32_t)a.max_constraint_degree;
    })
}
// Neutral word image of chip `chip`'s interactions, in Chip::all_interactions order (test hook: compared with the checker's own
// transcription and with the shapes extracted from the reference's Rust sources):
//   [n] then per interaction: [is_send] [is_global] [bus_index] [n_fields] count_vcol field_vcols..
//   vcol = [n_terms] [constant] n_terms x ([is_preprocessed] [column] [weight]), canonical values
int64_t vgpu_machine_interaction_words(const vgpu_machine_t* m, uint32_t chip, uint32_t* out, uint64_t cap) {
    if (!m || chip >= m->desc.airs.size()) return (int64_t)fail(VGPU_ERR_INVALID_ARG, "bad chip index");
    std::vector<uint32_t> w;
    auto vcol = [&](const vair::VirtualCol& v) {
        w.push_back((uint32_t)v.terms.size()); w.push_back(v.constant);
        for (auto& t : v.terms) { w.push_back(t.preprocessed ? 1u : 0u); w.push_back((uint32_t)t.col); w.push_back(t.weight); }
    };
    const auto& its = m->desc.airs[chip].interactions;
    w.push_back((uint32_t)its.size());
    for (auto& it : its) {
        w.push_back(it.is_send() ? 1u : 0u); w.push_back(it.is_local() ? 0u : 1u); w.push_back((uint32_t)it.bus_index); w.push_back((uint32_t)it.fields.size());
        vcol(it.count);
        for (auto& f : it.fields) vcol(f);
    }
    if (out && cap >= w.size()) memcpy(out, w.data(), w.size() * 4);
    return (int64_t)w.size();
}
int32_t vgpu_machine_eval_constraints(const vgpu_machine_t* m, uint32_t chip, const uint32_t* main_local, const uint32_t* main_next,
                                      const uint32_t* prep_local, const uint32_t* prep_next, uint32_t is_first, uint32_t is_last,
                                      uint32_t is_transition, uint32_t* out, uint32_t cap) {
    try {
        if (!m || chip >= m->desc.airs.size()) throw std::invalid_argument("bad chip index");
        const AirDesc& a = m->desc.airs[chip];
        std::vector<Fp> ml(a.width), mn(a.width), pl(a.prep_width), pn(a.prep_width);
        for (uint32_t i = 0; i < a.width; i++) { ml[i] = Fp::from_canonical(main_local[i]); mn[i] = Fp::from_canonical(main_next[i]); }
        for (uint32_t i = 0; i < a.prep_width; i++) { pl[i] = Fp::from_canonical(prep_local[i]); pn[i] = Fp::from_canonical(prep_next[i]); }
        vair::HostEval ev{ml.data(), mn.data(), pl.data(), pn.data(), Fp::from_canonical(is_first), Fp::from_canonical(is_last), Fp::from_canonical(is_transition)};
        auto vals = ev.run(a.program);
        if (vals.size() > cap) throw std::invalid_argument("output buffer too small");
        for (size_t i = 0; i < vals.size(); i++) out[i] = vals[i].canonical();
        return (int32_t)vals.size();
    } catch (const std::exception& e) { return fail(VGPU_ERR_INVALID_ARG, e.what()); }
}

// ---- challenger
int32_t vgpu_challenger_new(const uint32_t rc[480], vgpu_challenger_t** out) {
    VG_TRY({
        if (!rc || !out) throw std::invalid_argument("null argument");
        auto* c = new vgpu_challenger();
        c->perm.reset(new Poseidon16(rc));
        c->ch.reset(new Challenger(c->perm.get()));
        *out = c;
    })
}
void vgpu_challenger_free(vgpu_challenger_t* ch) { delete ch; }
void vgpu_challenger_observe(vgpu_challenger_t* ch, const uint32_t* v, uint64_t n) { for (uint64_t i = 0; i < n; i++) ch->ch->observe_canonical(v[i]); }
void vgpu_challenger_sample(vgpu_challenger_t* ch, uint32_t* out, uint64_t n) { for (uint64_t i = 0; i < n; i++) out[i] = ch->ch->sample().canonical(); }
uint64_t vgpu_challenger_sample_bits(vgpu_challenger_t* ch, uint32_t bits) { return ch->ch->sample_bits(bits); }
uint32_t vgpu_challenger_grind(vgpu_challenger_t* ch, uint32_t bits) { return ch->ch->grind(bits); }
void vgpu_poseidon16_permute(const uint32_t rc[480], uint32_t state[16]) {
    Poseidon16 p(rc);
    Fp st[16];
    for (int i = 0; i < 16; i++) st[i] = Fp::from_canonical(state[i]);
    p.permute(st);
    for (int i = 0; i < 16; i++) state[i] = st[i].canonical();
}

// test hook: the SPARSE-partial-round schedule the Poseidon MMCS kernels run (host twin, same tables); returns 0 when the tables
// are valid and the permutation was applied, -4 when a block was singular (the kernels then use the plain rounds)
int32_t vgpu_poseidon16_permute_sparse(const uint32_t rc[480], uint32_t state[16]) {
    Poseidon16 p(rc);
    PoseidonOptTables t(p);
    if (!t.valid) return VGPU_ERR_UNSUPPORTED;
    Fp st[16];
    for (int i = 0; i < 16; i++) st[i] = Fp::from_canonical(state[i]);
    t.permute(p, st);
    for (int i = 0; i < 16; i++) state[i] = st[i].canonical();
    return VGPU_OK;
}

// ---- prover
int32_t vgpu_prover_create(const vgpu_config_t* cfg, const vgpu_machine_t* machine, vgpu_prover_t** out) {
    VG_TRY({
        if (!cfg || !machine || !out) throw std::invalid_argument("null argument");
        if (cfg->hash_kind != VGPU_HASH_KECCAK256 && cfg->hash_kind != VGPU_HASH_POSEIDON16) { g_err = "hash_kind must be VGPU_HASH_KECCAK256 or VGPU_HASH_POSEIDON16"; return VGPU_ERR_UNSUPPORTED; }
        if (cfg->log_blowup < 1 || cfg->log_blowup > 4) throw std::invalid_argument("log_blowup must be in 1..4");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= cfg->device) throw std::runtime_error("hip: no usable device (the product path has no CPU fallback)");
        FriParams fp;
        fp.log_blowup = cfg->log_blowup; fp.num_queries = cfg->num_queries; fp.pow_bits = cfg->pow_bits; fp.observe_final_poly = cfg->observe_final_poly != 0;
        fp.interpret_air = cfg->interpret_air != 0;
        fp.hash_kind = (int)cfg->hash_kind;
        auto* p = new vgpu_prover();
        p->p.reset(new Prover(cfg->device, machine->desc, cfg->poseidon_rc, fp));
        *out = p;
    })
}
void vgpu_prover_destroy(vgpu_prover_t* p) { delete p; }
void vgpu_prover_memory(const vgpu_prover_t* p, uint64_t* live, uint64_t* peak) {
    auto& c = const_cast<vgpu_prover_t*>(p)->p->ctx();
    if (live) *live = c.live;
    if (peak) *peak = c.peak_live;
}
void vgpu_prover_memory_reset_peak(vgpu_prover_t* p) {
    if (!p) return;
    DeviceCtx& c = p->p->ctx();
    std::lock_guard<std::mutex> lk(c.pool_mu);
    c.peak_live = c.live;
}
uint64_t vgpu_prover_trim(vgpu_prover_t* p) {
    if (!p) return 0;
    DeviceCtx& c = p->p->ctx();
    (void)hipSetDevice(c.device);
    return (uint64_t)c.trim();
}
void vgpu_prover_set_profiling(vgpu_prover_t* p, uint32_t on) {
    auto& c = p->p->ctx();
    c.profiler.reset();
    c.profiler.enabled = on != 0;
}
// One probe state per device, created on first use and kept for the life of the process: a stream of its own and three page-locked words the
// wave writes straight into — no allocation and no device-wide synchronisation on the calls that run beside proofs.
int32_t vgpu_shader_clock_probe(int32_t device, uint32_t iters, uint64_t out[2]) {
    VG_TRY({
        if (!out || !iters) throw std::invalid_argument("shader_clock_probe: null output or zero iterations");
        struct Probe { hipStream_t st = nullptr; uint64_t* buf = nullptr; };
        static std::mutex mu;
        static std::map<int, Probe> probes;
        std::lock_guard<std::mutex> lk(mu);
        VG_HIP_CHECK(hipSetDevice(device));
        Probe& pr = probes[device];
        if (!pr.st) {
            VG_HIP_CHECK(hipStreamCreateWithFlags(&pr.st, hipStreamNonBlocking));
            VG_HIP_CHECK(hipHostMalloc((void**)&pr.buf, 64));
        }
        pr.buf[0] = pr.buf[1] = 0;
        vk::launch_clock_probe(pr.st, pr.buf, iters);
        VG_HIP_CHECK(hipStreamSynchronize(pr.st));
        if (!pr.buf[1]) throw std::runtime_error("hip: the clock probe wrote nothing");
        out[0] = pr.buf[0]; out[1] = pr.buf[1];
    })
}
void vgpu_prover_set_prep_cache(vgpu_prover_t* p, uint32_t on) { if (p) p->p->set_prep_cache(on != 0); }
void vgpu_prover_set_profiling_filter(vgpu_prover_t* p, const char* kernel_name) { p->p->ctx().profiler.only = kernel_name ? kernel_name : ""; }
// "name launches total_ms total_algorithmic_bytes total_algorithmic_valu_ops\n" per kernel, accumulated since profiling was switched on
int64_t vgpu_prover_profile(vgpu_prover_t* p, char* out, uint64_t cap) {
    std::string s;
    {
        DeviceCtx& c = p->p->ctx();
        (void)hipSetDevice(c.device);
        (void)hipDeviceSynchronize();  // every recorded event has completed: launches made outside prove() are collected too
        c.profiler.collect();
    }
    for (auto& kv : p->p->ctx().profiler.stats) {
        char line[256];
        snprintf(line, sizeof line, "%s %llu %.6f %.0f %.0f\n", kv.first.c_str(), (unsigned long long)kv.second.launches, kv.second.ms, kv.second.bytes,
                 kv.second.valu_ops);
        s += line;
    }
    if (out && cap > s.size()) memcpy(out, s.c_str(), s.size() + 1);
    return (int64_t)s.size() + 1;
}
int32_t vgpu_host_alloc(uint64_t bytes, void** out) {
    VG_TRY({
        if (!out || !bytes) throw std::invalid_argument("vgpu_host_alloc: null output or zero size");
        void* ptr = nullptr;
        hipError_t e = hipHostMalloc(&ptr, bytes, hipHostMallocPortable);
        if (e == hipErrorOutOfMemory) throw std::bad_alloc();
        if (e != hipSuccess) throw std::runtime_error(std::string("hipHostMalloc: ") + hipGetErrorString(e));
        *out = ptr;
    })
}
void vgpu_host_free(void* ptr) { if (ptr) (void)hipHostFree(ptr); }
int32_t vgpu_trace_upload(vgpu_prover_t* p, const uint32_t* data, uint64_t height, uint64_t width, vgpu_trace_t** out) {
    VG_TRY({
        if (!p || !data || !out || !height || !width) throw std::invalid_argument("bad trace");
        std::unique_ptr<vgpu_trace> t(new vgpu_trace());
        t->owner = p->p;
        p->p->ctx().activate();
        t->t = p->p->upload_trace(HostMatrix{data, height, width});
        *out = t.release();
    })
}
void vgpu_trace_free(vgpu_trace_t* t) { delete t; }
void vgpu_trace_shape(const vgpu_trace_t* t, uint64_t* height, uint64_t* width) { *height = t->t->height; *width = t->t->width; }
int32_t vgpu_trace_download(vgpu_prover_t* p, const vgpu_trace_t* t, uint32_t* out, uint64_t cap_words) {
    VG_TRY({
        if (!p || !t || !out) throw std::invalid_argument("null argument");
        if (cap_words < t->t->height * t->t->width) throw std::invalid_argument("output buffer too small");
        p->p->ctx().activate();
        p->p->download_trace(*t->t, out);
    })
}
int32_t vgpu_oplog_upload(vgpu_prover_t* p, const vgpu_oplog_desc_t* log, vgpu_oplog_t** out) {
    VG_TRY({
        if (!p || !log || !out) throw std::invalid_argument("null argument");
        if (log->struct_size != sizeof(vgpu_oplog_desc_t))
            throw std::invalid_argument("oplog: struct_size " + std::to_string(log->struct_size) + " is not this library's sizeof(vgpu_oplog_desc_t) = " + std::to_string(sizeof(vgpu_oplog_desc_t)) +
                                        " (host compiled against another vgpu.h?)");
        p->p->ctx().activate();
        HostOplog h;
        h.cpu = (const vk::TgCpuOp*)log->cpu; h.n_cpu = log->n_cpu;
        h.mem = (const vk::TgMemOp*)log->mem; h.n_mem = log->n_mem;
        for (int k = 0; k < 4; k++) { h.alu[k] = (const vk::TgAluOp*)log->alu[k]; h.n_alu[k] = log->n_alu[k]; }
        h.static_cells = log->static_cells; h.n_static = log->n_static;
        h.rom_len = log->rom_len;
        for (int k = 0; k < 4; k++) { h.alu2[k] = (const vk::TgAluOp*)log->alu2[k]; h.n_alu2[k] = log->n_alu2[k]; if (h.n_alu2[k] && !h.alu2[k]) throw std::invalid_argument("oplog: null ALU log with a nonzero length"); }
        h.output = (const vk::TgOutOp*)log->output; h.n_output = log->n_output;
        if (h.n_output && !h.output) throw std::invalid_argument("oplog: null output tape with a nonzero length");
        {   // every logged operation must be a variant of its chip's Operation enum
            using namespace vchips;
            auto check = [&](int k, std::initializer_list<uint32_t> ok, const char* chip) {
                for (uint64_t i = 0; i < h.n_alu2[k]; i++) {
                    bool good = false;
                    for (uint32_t o : ok) good |= h.alu2[k][i].opcode == o;
                    if (!good) throw std::invalid_argument(std::string("oplog: ") + chip + " log entry " + std::to_string(i) + " has an opcode that is not an operation of that chip");
                    if (k == 2 && h.alu2[k][i].c >= 32) throw std::invalid_argument("oplog: shift log entry " + std::to_string(i) + ": shift amount >= 32");
                }
            };
            check(0, {OP_MUL32, OP_MULHS32, OP_MULHU32}, "mul");
            check(1, {OP_DIV32, OP_SDIV32}, "div");
            check(2, {OP_SHL32, OP_SHR32, OP_SRA32}, "shift");
            check(3, {OP_NE32, OP_EQ32}, "com");
        }
        for (uint64_t i = 1; i < h.n_static; i++)
            if (log->static_cells[2 * i] <= log->static_cells[2 * i - 2]) throw std::invalid_argument("oplog: static cells must be in ascending address order");
        if (!h.n_cpu || !log->cpu) throw std::invalid_argument("oplog: empty cpu log");
        if ((h.n_mem && !log->mem) || (h.n_static && !log->static_cells)) throw std::invalid_argument("oplog: null log with a nonzero length");
        for (int k = 0; k < 4; k++) if (h.n_alu[k] && !h.alu[k]) throw std::invalid_argument("oplog: null ALU log with a nonzero length");
        for (uint64_t i = 0; i < h.n_cpu; i++) {
            const vgpu_cpu_op_t& o = log->cpu[i];
            if (o.mem_first > h.n_mem || (i && o.mem_first < log->cpu[i - 1].mem_first) || o.kind > VGPU_CPU_LOADFP)
                throw std::invalid_argument("oplog: cpu record " + std::to_string(i) + " is malformed");
            // the program chip's multiplicity histogram indexes the ROM by pc
            if (h.rom_len && o.pc >= h.rom_len) throw std::invalid_argument("oplog: cpu record " + std::to_string(i) + ": pc beyond the ROM length");
            // Bus operations must target a chip of the BasicMachine (every one of them has a log-driven generator now); anything else —
            // an advice read, an opcode of another machine — has no chip to receive it here
            if (o.kind == VGPU_CPU_BUS || o.kind == VGPU_CPU_BUS_LEFT_IMM) {
                using namespace vchips;
                const uint32_t op = o.opcode;
                const bool known = (op >= OP_ADD32 && op <= OP_SLE32) || op == OP_WRITE;
                if (!known) throw std::invalid_argument("oplog: cpu record " + std::to_string(i) + ": opcode " + std::to_string(op) + " is no bus operation of the BasicMachine's chips");
            }
        }
        // the memory chip's stable by-address radix sort reproduces the reference's (addr, clk) order only for clk-ordered logs
        for (uint64_t i = 1; i < h.n_mem; i++)
            if (log->mem[i].clk < log->mem[i - 1].clk) throw std::invalid_argument("oplog: memory log entry " + std::to_string(i) + " is out of clock order");
        std::unique_ptr<vgpu_oplog> l(new vgpu_oplog());
        l->owner = p->p;
        l->log = p->p->upload_oplog(h);
        *out = l.release();
    })
}
void vgpu_oplog_free(vgpu_oplog_t* log) { delete log; }
int32_t vgpu_generate_trace(vgpu_prover_t* p, const vgpu_oplog_t* log, uint32_t chip, vgpu_trace_t** out) {
    VG_TRY({
        if (!p || !log || !out) throw std::invalid_argument("null argument");
        p->p->ctx().activate();
        std::unique_ptr<vgpu_trace> t(new vgpu_trace());
        t->owner = p->p;
        t->t = p->p->generate_trace(*log->log, (int)chip);
        *out = t.release();
    })
}

// private working-layout copy of a trace (row order is kept: see DeviceTrace::nat_rows_bitrev)
static DMat ingest(DeviceCtx& c, const DeviceTrace& t) {
    if (t.height & (t.height - 1)) throw std::invalid_argument("matrix height must be a power of two");
    DMat m(&c, t.height, t.width);
    if (!t.nat.empty()) VG_HIP_CHECK(hipMemcpyAsync(m.data, t.nat.data, t.height * t.width * 4, hipMemcpyDeviceToDevice, c.stream));
    else vk::launch_ingest(c.stream, t.raw.data, m.view(), false);
    return m;
}

int32_t vgpu_commit_batches(vgpu_prover_t* p, const vgpu_trace_t* const* mats, uint32_t n_mats, const uint32_t* coset_shifts, uint32_t root[8],
                            vgpu_pdata_t** out) {
    VG_TRY({
        if (!p || !mats || !n_mats || !root || !out) throw std::invalid_argument("null argument");
        DeviceCtx& c = p->p->ctx();
        c.activate();
        std::vector<DMat> nat;
        for (uint32_t i = 0; i < n_mats; i++) { if (!mats[i] || mats[i]->owner != p->p) throw std::invalid_argument("null trace or trace of another prover"); nat.push_back(ingest(c, *mats[i]->t)); }
        std::vector<CommitInput> in;
        for (uint32_t i = 0; i < n_mats; i++) in.push_back({&nat[i], mats[i]->t->nat_rows_bitrev, mats[i]->t->nat_rows_bitrev});  // our private copy: consumable
        std::vector<Fp> shifts;
        if (coset_shifts) for (uint32_t i = 0; i < n_mats; i++) shifts.push_back(Fp::from_canonical(coset_shifts[i]));
        std::unique_ptr<vgpu_pdata> pd(new vgpu_pdata());
        pd->owner = p->p;
        pd->pd = commit_batches(&c, in, coset_shifts ? &shifts : nullptr, p->p->fri());
        memcpy(root, pd->pd->tree.root, 32);
        *out = pd.release();
    })
}
int32_t vgpu_pdata_lde(vgpu_prover_t* p, const vgpu_pdata_t* pd, uint32_t idx, uint32_t* out, uint64_t cap_words) {
    VG_TRY({
        if (!p || !pd || idx >= pd->pd->ldes.size() || !out) throw std::invalid_argument("bad argument");
        DeviceCtx& c = p->p->ctx();
        c.activate();
        const DMat& l = pd->pd->ldes[idx];
        if (cap_words < l.height * l.width) throw std::invalid_argument("output buffer too small");
        DBuf tmp(&c, (size_t)(l.height * l.width));
        vk::launch_export_rows(c.stream, l.view(), 0, l.height, tmp.data);
        c.download(out, tmp.data, l.height * l.width * 4);
    })
}
uint32_t vgpu_pdata_num_matrices(const vgpu_pdata_t* pd) { return pd ? (uint32_t)pd->pd->ldes.size() : 0; }
int32_t vgpu_pdata_lde_view(const vgpu_pdata_t* pd, uint32_t idx, vgpu_lde_view_t* out) {
    VG_TRY({
        if (!pd || !out || idx >= pd->pd->ldes.size()) throw std::invalid_argument("bad argument");
        const DMat& l = pd->pd->ldes[idx];
        out->data = l.data; out->height = l.height; out->width = l.width; out->stride = l.height;
        out->log_blowup = pd->owner->fri().log_blowup;
    })
}
void vgpu_pdata_free(vgpu_pdata_t* pd) { delete pd; }

static Ext5 ext_of(const uint32_t* w) { Ext5 e; for (int k = 0; k < 5; k++) e.c[k] = Fp::from_canonical(w[k]); return e; }

int32_t vgpu_perm_trace_device(vgpu_prover_t* p, uint32_t chip, const vgpu_trace_t* main, const vgpu_trace_t* prep, const uint32_t challenges[15],
                               vgpu_trace_t** out, uint32_t cumulative_sum[5]) {
    VG_TRY({
        if (!p || !main || !challenges || !out) throw std::invalid_argument("null argument");
        const MachineDesc& md = p->p->machine();
        if (chip >= md.airs.size()) throw std::invalid_argument("bad chip index");
        if (main->t->width != md.airs[chip].width) throw std::invalid_argument("trace width mismatch");
        DeviceCtx& c = p->p->ctx();
        c.activate();
        DMat m = ingest(c, *main->t);
        DMat pm;
        if (prep) pm = ingest(c, *prep->t);
        Ext5 rnd[3] = {ext_of(challenges), ext_of(challenges + 5), ext_of(challenges + 10)};
        Ext5 cs;
        std::unique_ptr<vgpu_trace> t(new vgpu_trace());
        t->owner = p->p;
        t->t = std::make_shared<DeviceTrace>();
        t->t->nat = p->p->permutation_trace((int)chip, m, prep ? &pm : nullptr, rnd, &cs);
        t->t->height = t->t->nat.height; t->t->width = t->t->nat.width;
        if (cumulative_sum) for (int k = 0; k < 5; k++) cumulative_sum[k] = cs.c[k].canonical();
        *out = t.release();
    })
}

int32_t vgpu_quotient(vgpu_prover_t* p, uint32_t chip, const vgpu_pdata_t* prep_pd, uint32_t prep_idx, const vgpu_pdata_t* main_pd, uint32_t main_idx,
                      const vgpu_pdata_t* perm_pd, uint32_t perm_idx, const uint32_t perm_challenges[15], const uint32_t alpha[5],
                      const uint32_t cumulative_sum[5], vgpu_trace_t** out) {
    VG_TRY({
        if (!p || !main_pd || !perm_pd || !perm_challenges || !alpha || !cumulative_sum || !out) throw std::invalid_argument("null argument");
        if (chip >= p->p->machine().airs.size()) throw std::invalid_argument("bad chip index");
        if (main_pd->owner != p->p || perm_pd->owner != p->p || (prep_pd && prep_pd->owner != p->p)) throw std::invalid_argument("prover data of another prover context");
        if (main_idx >= main_pd->pd->ldes.size() || perm_idx >= perm_pd->pd->ldes.size() || (prep_pd && prep_idx >= prep_pd->pd->ldes.size()))
            throw std::invalid_argument("bad matrix index");
        p->p->ctx().activate();
        Ext5 rnd[3] = {ext_of(perm_challenges), ext_of(perm_challenges + 5), ext_of(perm_challenges + 10)};
        std::unique_ptr<vgpu_trace> t(new vgpu_trace());
        t->owner = p->p;
        t->t = std::make_shared<DeviceTrace>();
        t->t->nat = p->p->quotient_chunks((int)chip, main_pd->pd->ldes[main_idx], perm_pd->pd->ldes[perm_idx], prep_pd ? &prep_pd->pd->ldes[prep_idx] : nullptr, rnd,
                                          ext_of(alpha), ext_of(cumulative_sum));
        t->t->nat_rows_bitrev = true;
        t->t->height = t->t->nat.height; t->t->width = t->t->nat.width;
        *out = t.release();
    })
}

int32_t vgpu_open_multi_batches(vgpu_prover_t* p, const vgpu_pdata_t* const* rounds, uint32_t n_rounds, const uint32_t* n_points, const uint32_t* points,
                                vgpu_challenger_t* ch, vgpu_opening_t** out) {
    VG_TRY({
        if (!p || !rounds || !n_rounds || !n_points || !points || !ch || !out) throw std::invalid_argument("null argument");
        std::vector<OpenRound> rs(n_rounds);
        size_t k = 0, w = 0;
        for (uint32_t r = 0; r < n_rounds; r++) {
            if (!rounds[r] || rounds[r]->owner != p->p) throw std::invalid_argument("null prover data or prover data of another prover context");
            rs[r].pd = rounds[r]->pd.get();
            for (size_t i = 0; i < rs[r].pd->ldes.size(); i++, k++) {
                std::vector<Ext5> pts;
                for (uint32_t q = 0; q < n_points[k]; q++, w += 5) pts.push_back(ext_of(points + w));
                rs[r].points.push_back(std::move(pts));
            }
        }
        PcsOpening o = p->p->open_multi_batches(rs, *ch->ch);
        std::unique_ptr<vgpu_opening> res(new vgpu_opening());
        for (auto& round : o.opened) for (auto& mat : round) for (auto& pt : mat) for (auto& e : pt) for (int q = 0; q < 5; q++) res->values.push_back(e.c[q].canonical());
        res->proof = std::move(o.proof_words);
        *out = res.release();
    })
}
int32_t vgpu_verify_multi_batches(const vgpu_config_t* cfg, const uint32_t* commits, uint32_t n_rounds, const uint32_t* n_mats, const uint64_t* heights,
                                  const uint32_t* widths, const uint32_t* n_points, const uint32_t* points, const uint32_t* values, uint64_t n_value_words,
                                  const uint32_t* proof, uint64_t n_proof_words, vgpu_challenger_t* ch) {
    VG_TRY({
        if (!cfg || !commits || !n_rounds || !n_mats || !heights || !widths || !n_points || !points || !values || !proof || !ch) throw std::invalid_argument("null argument");
        Poseidon16 perm(cfg->poseidon_rc);
        HostMmcs mmcs{(int)cfg->hash_kind, &perm};
        std::vector<VerifyRoundIn> rounds(n_rounds);
        size_t k = 0, pw = 0, vw = 0;
        for (uint32_t r = 0; r < n_rounds; r++) {
            memcpy(rounds[r].commit.data(), commits + 8 * r, 32);
            for (uint32_t i = 0; i < n_mats[r]; i++, k++) {
                rounds[r].heights.push_back(heights[k]);
                rounds[r].widths.push_back(widths[k]);
                std::vector<Ext5> pts;
                std::vector<std::vector<Ext5>> vals;
                for (uint32_t q = 0; q < n_points[k]; q++, pw += 5) pts.push_back(ext_of(points + pw));
                for (uint32_t q = 0; q < n_points[k]; q++) {
                    std::vector<Ext5> ys;
                    for (uint32_t c = 0; c < widths[k]; c++, vw += 5) {
                        if (vw + 5 > n_value_words) throw std::invalid_argument("verify: too few opened values");
                        ys.push_back(ext_of(values + vw));
                    }
                    vals.push_back(std::move(ys));
                }
                rounds[r].points.push_back(std::move(pts));
                rounds[r].values.push_back(std::move(vals));
            }
        }
        if (vw != n_value_words) throw std::invalid_argument("verify: too many opened values");
        verify_multi_batches(rounds, proof, (size_t)n_proof_words, *ch->ch, cfg->log_blowup, cfg->num_queries, cfg->pow_bits, cfg->observe_final_poly != 0, mmcs);
    })
}
static FriParams fri_of(const vgpu_config_t* cfg) {
    FriParams f;
    f.log_blowup = cfg->log_blowup; f.num_queries = cfg->num_queries; f.pow_bits = cfg->pow_bits;
    f.observe_final_poly = cfg->observe_final_poly != 0; f.hash_kind = (int)cfg->hash_kind;
    return f;
}
int32_t vgpu_verify(const vgpu_config_t* cfg, const vgpu_machine_t* machine, const uint32_t* preprocessed_commit, const uint32_t* proof_words,
                    uint64_t n_words) {
    VG_TRY({
        if (!cfg || !machine || !proof_words) throw std::invalid_argument("null argument");
        if (cfg->hash_kind > 1) throw std::invalid_argument("unknown hash kind");
        Poseidon16 perm(cfg->poseidon_rc);
        verify_machine_proof(machine->desc, fri_of(cfg), perm, preprocessed_commit, proof_words, (size_t)n_words);
    })
}
int32_t vgpu_host_commit_root(const vgpu_config_t* cfg, const uint32_t* const* mats, const uint64_t* heights, const uint64_t* widths, uint32_t n_mats,
                              const uint32_t* coset_shifts, uint32_t root[8]) {
    VG_TRY({
        if (!cfg || !mats || !heights || !widths || !n_mats || !root) throw std::invalid_argument("null argument");
        if (cfg->hash_kind > 1) throw std::invalid_argument("unknown hash kind");
        Poseidon16 perm(cfg->poseidon_rc);
        HostMmcs mmcs{(int)cfg->hash_kind, &perm};
        std::vector<HostMatrixView> v;
        for (uint32_t i = 0; i < n_mats; i++) { if (!mats[i]) throw std::invalid_argument("null matrix"); v.push_back({mats[i], heights[i], widths[i]}); }
        Digest8 d = host_commit_root(v, coset_shifts, cfg->log_blowup, mmcs);
        memcpy(root, d.data(), 32);
    })
}
uint64_t vgpu_opening_values_len(const vgpu_opening_t* o) { return o->values.size(); }
const uint32_t* vgpu_opening_values(const vgpu_opening_t* o) { return o->values.data(); }
uint64_t vgpu_opening_proof_len(const vgpu_opening_t* o) { return o->proof.size(); }
const uint32_t* vgpu_opening_proof(const vgpu_opening_t* o) { return o->proof.data(); }
void vgpu_opening_free(vgpu_opening_t* o) { delete o; }

int32_t vgpu_perm_trace(vgpu_prover_t* p, uint32_t chip, const vgpu_trace_t* main, const vgpu_trace_t* prep, const uint32_t challenges[15], uint32_t* out,
                        uint64_t cap_words, uint32_t cumulative_sum[5]) {
    VG_TRY({
        if (!p || !main || !challenges || !out) throw std::invalid_argument("null argument");
        const MachineDesc& md = p->p->machine();
        if (chip >= md.airs.size()) throw std::invalid_argument("bad chip index");
        const AirDesc& air = md.airs[chip];
        if (main->t->width != air.width) throw std::invalid_argument("trace width mismatch");
        DeviceCtx& c = p->p->ctx();
        c.activate();
        DMat m = ingest(c, *main->t);
        DMat pm;
        if (prep) pm = ingest(c, *prep->t);
        Ext5 rnd[3];
        for (int i = 0; i < 3; i++) for (int k = 0; k < 5; k++) rnd[i].c[k] = Fp::from_canonical(challenges[5 * i + k]);
        uint32_t M = (uint32_t)air.interactions.size();
        std::vector<uint32_t> pool;
        size_t maxf = 0;
        for (auto& it : air.interactions) { put_ext(pool, (it.is_local() ? rnd[0] : rnd[1]).pow((uint64_t)it.bus_index + 1)); maxf = std::max(maxf, it.fields.size()); }
        Ext5 bp = Ext5::one();
        for (size_t j = 0; j < maxf; j++) { put_ext(pool, bp); bp *= rnd[2]; }
        pool.push_back(0);
        DBuf pool_dev(&c, pool), iw(&c, air.interaction_words), scratch(&c, (size_t)vk::perm_scratch_words(m.height));
        DMat perm(&c, m.height, 5 * (M + 1));
        if (cap_words < perm.height * perm.width) throw std::invalid_argument("output buffer too small");
        vk::launch_perm_trace(c.stream, m.view(), prep ? pm.view() : vk::DMatView{nullptr, 0, 0, 0}, iw.data, pool_dev.data, M, perm.view(), scratch.data);
        DBuf tmp(&c, (size_t)(perm.height * perm.width));
        vk::launch_export_rows(c.stream, perm.view(), 0, perm.height, tmp.data);
        c.check_launch("perm trace");
        c.download(out, tmp.data, perm.height * perm.width * 4);
        if (cumulative_sum) for (int k = 0; k < 5; k++) cumulative_sum[k] = out[(perm.height - 1) * perm.width + 5 * M + k];
    })
}

int32_t vgpu_fri_fold(vgpu_prover_t* p, const uint32_t* f, uint64_t n, const uint32_t beta[5], uint32_t* out) {
    VG_TRY({
        if (!p || !f || !beta || !out || n < 4 || (n & (n - 1))) throw std::invalid_argument("fri_fold: n must be a power of two >= 4");
        DeviceCtx& c = p->p->ctx();
        c.activate();
        uint64_t half = n / 2, q = half / 2;
        std::vector<uint32_t> in(5 * n);
        std::vector<uint32_t> bw(5);
        for (uint64_t i = 0; i < n; i++) for (int k = 0; k < 5; k++) in[((i & 1) * 5 + k) * half + (i >> 1)] = Fp::from_canonical(f[5 * i + k]).v;
        for (int k = 0; k < 5; k++) bw[k] = Fp::from_canonical(beta[k]).v;
        DBuf din(&c, in), dbeta(&c, bw), dout(&c, (size_t)(5 * half));
        vk::launch_fri_fold(c.stream, din.data, n, dbeta.data, nullptr, c.tables, dout.data);
        c.check_launch("fri fold");
        std::vector<uint32_t> o(5 * half);
        c.download(o.data(), dout.data, o.size() * 4);
        for (uint64_t i = 0; i < half; i++) for (int k = 0; k < 5; k++) out[5 * i + k] = Fp::raw(o[((i & 1) * 5 + k) * q + (i >> 1)]).canonical();
    })
}

// A trace handle handed to prove must exist and live on THIS prover's device.  Traces of another context of the same device (a host that
// uploads / generates segment i+1 on a context of its own while this one proves segment i) are accepted: `foreign` collects their
// contexts, which the caller drains (their queued work must have produced the traces) and keeps alive for the duration of the proof.
static void check_trace(const vgpu_prover_t* p, const vgpu_trace_t* t, std::vector<std::shared_ptr<Prover>>* foreign = nullptr) {
    if (!t || !t->t) throw std::invalid_argument("null trace");
    if (t->owner == p->p) return;
    if (!foreign || t->owner->ctx().device != p->p->ctx().device) throw std::invalid_argument("trace belongs to another prover context");
    if (std::find(foreign->begin(), foreign->end(), t->owner) == foreign->end()) foreign->push_back(t->owner);
}
static void drain(const std::vector<std::shared_ptr<Prover>>& ctxs) {
    for (auto& q : ctxs) { q->ctx().activate(); q->ctx().sync(); }
}

int32_t vgpu_prove(vgpu_prover_t* p, const vgpu_trace_t* const* main, uint32_t n_main, const uint32_t* prep_chips, const vgpu_trace_t* const* prep,
                   uint32_t n_prep, uint32_t debug_flags, vgpu_proof_t** out) {
    VG_TRY({
        if (!p || !main || !out || (n_prep && (!prep || !prep_chips))) throw std::invalid_argument("null argument");
        std::vector<const DeviceTrace*> m;
        std::vector<std::shared_ptr<Prover>> foreign;
        for (uint32_t i = 0; i < n_main; i++) { check_trace(p, main[i], &foreign); m.push_back(main[i]->t.get()); }
        std::vector<std::pair<int, const DeviceTrace*>> pr;
        for (uint32_t i = 0; i < n_prep; i++) { check_trace(p, prep[i], &foreign); pr.push_back({(int)prep_chips[i], prep[i]->t.get()}); }
        drain(foreign);
        auto proof = std::make_unique<vgpu_proof>();
        proof->dbg.keep_matrices = (debug_flags & 1) != 0;
        proof->dbg.check_constraints = (debug_flags & 2) != 0;
        proof->words = p->p->prove(m, pr, &proof->tm, &proof->dbg);
        *out = proof.release();
    })
}
// Asynchronous prove: the call returns at once; the proof is produced by a host thread of its own driving this prover's
// streams.  With two provers a single caller thread keeps two proofs in flight on one GPU — one proof's latency-bound
// Merkle-top / FRI tail overlaps the other's throughput-bound commits (DESIGN.md "Measurement").  One outstanding ticket
// per prover; the traces must stay alive until vgpu_ticket_wait returns.
int32_t vgpu_prove_async(vgpu_prover_t* p, const vgpu_trace_t* const* main, uint32_t n_main, const uint32_t* prep_chips, const vgpu_trace_t* const* prep,
                         uint32_t n_prep, vgpu_ticket_t** out) {
    VG_TRY({
        if (!p || !main || !out) throw std::invalid_argument("null argument");
        // the job owns what it works on (the prover first: destroyed last), so freeing handles meanwhile is harmless
        struct Job {
            std::shared_ptr<Prover> prover;
            std::vector<std::shared_ptr<DeviceTrace>> main, prep;
            std::vector<int> chips;
            std::vector<std::shared_ptr<Prover>> foreign;  // other contexts whose pools own some of the traces
        };
        auto job = std::make_shared<Job>();
        job->prover = p->p;
        if (n_prep && (!prep || !prep_chips)) throw std::invalid_argument("null argument");
        for (uint32_t i = 0; i < n_main; i++) { check_trace(p, main[i], &job->foreign); job->main.push_back(main[i]->t); }
        for (uint32_t i = 0; i < n_prep; i++) { check_trace(p, prep[i], &job->foreign); job->prep.push_back(prep[i]->t); job->chips.push_back((int)prep_chips[i]); }
        drain(job->foreign);
        std::unique_ptr<vgpu_ticket> t(new vgpu_ticket());
        t->result = std::async(std::launch::async, [job]() {
            std::pair<int32_t, std::string> status{VGPU_OK, ""};
            vgpu_proof_t* proof = nullptr;
            try {
                std::vector<const DeviceTrace*> m;
                for (auto& x : job->main) m.push_back(x.get());
                std::vector<std::pair<int, const DeviceTrace*>> pr;
                for (size_t i = 0; i < job->prep.size(); i++) pr.push_back({job->chips[i], job->prep[i].get()});
                auto out = std::make_unique<vgpu_proof>();
                out->words = job->prover->prove(m, pr, &out->tm, &out->dbg);
                proof = out.release();
            } catch (const std::invalid_argument& e) { status = {VGPU_ERR_INVALID_ARG, e.what()};
            } catch (const std::bad_alloc& e) { status = {VGPU_ERR_OOM, e.what()};
            } catch (const std::exception& e) {
                std::string msg = e.what();
                status = {msg.find("hip") != std::string::npos ? VGPU_ERR_HIP : VGPU_ERR_INTERNAL, msg};
            }
            return std::make_pair(proof, status);
        });
        *out = t.release();
    })
}
int32_t vgpu_ticket_wait(vgpu_ticket_t* t, vgpu_proof_t** out) {
    if (!t || !out) return fail(VGPU_ERR_INVALID_ARG, "null argument");
    auto r = t->result.get();
    delete t;
    if (r.second.first != VGPU_OK) return fail(r.second.first, r.second.second);
    *out = r.first;
    return VGPU_OK;
}
int64_t vgpu_proof_cbor(const uint32_t* proof_words, uint64_t n_words, uint32_t flags, uint8_t* out, uint64_t cap_bytes) {
    try {
        if (!proof_words) throw std::invalid_argument("null argument");
        std::vector<uint8_t> b = proof_to_cbor(proof_words, (size_t)n_words, flags);
        if (out && cap_bytes >= b.size()) memcpy(out, b.data(), b.size());
        return (int64_t)b.size();
    } catch (const std::exception& e) { return (int64_t)fail(VGPU_ERR_INVALID_ARG, e.what()); }
}
int64_t vgpu_proof_from_cbor(const uint8_t* bytes, uint64_t n_bytes, uint32_t* out, uint64_t cap_words) {
    try {
        if (!bytes) throw std::invalid_argument("null argument");
        std::vector<uint32_t> w = proof_from_cbor(bytes, (size_t)n_bytes);
        if (out && cap_words >= w.size()) memcpy(out, w.data(), w.size() * 4);
        return (int64_t)w.size();
    } catch (const std::exception& e) { return (int64_t)fail(VGPU_ERR_INVALID_ARG, e.what()); }
}
int64_t vgpu_proof_from_cbor_ex(const uint8_t* bytes, uint64_t n_bytes, uint32_t flags, uint32_t* out, uint64_t cap_words, uint32_t* forms_seen) {
    try {
        if (!bytes) throw std::invalid_argument("null argument");
        ProofCborDecoder dec(bytes, (size_t)n_bytes, (flags & 1u) != 0);
        std::vector<uint32_t> w = dec.decode();
        if (forms_seen) *forms_seen = dec.seen();
        if (out && cap_words >= w.size()) memcpy(out, w.data(), w.size() * 4);
        return (int64_t)w.size();
    } catch (const std::exception& e) { return (int64_t)fail(VGPU_ERR_INVALID_ARG, e.what()); }
}
uint64_t vgpu_proof_len(const vgpu_proof_t* pr) { return pr->words.size(); }
const uint32_t* vgpu_proof_words(const vgpu_proof_t* pr) { return pr->words.data(); }
void vgpu_proof_phase_ms(const vgpu_proof_t* pr, double out[11]) {
    const PhaseTimes& t = pr->tm;
    double v[11] = {t.ingest, t.commit_main, t.perm, t.commit_perm, t.quotient, t.commit_quotient, t.open_values, t.open_reduce, t.fri, t.queries, t.total};
    memcpy(out, v, sizeof(v));
}
void vgpu_proof_transcript(const vgpu_proof_t* pr, uint32_t out[33]) {
    memcpy(out, pr->dbg.prep_root, 32);
    memcpy(out + 8, pr->dbg.perm_challenges, 60);
    memcpy(out + 23, pr->dbg.alpha, 20);
    memcpy(out + 28, pr->dbg.zeta, 20);
}
static int64_t copy_dbg(const std::vector<std::vector<uint32_t>>& v, uint32_t chip, uint32_t* out, uint64_t cap) {
    if (chip >= v.size()) return -1;
    if (out && cap >= v[chip].size()) memcpy(out, v[chip].data(), v[chip].size() * 4);
    return (int64_t)v[chip].size();
}
int64_t vgpu_proof_debug_perm_trace(const vgpu_proof_t* pr, uint32_t chip, uint32_t* out, uint64_t cap) { return copy_dbg(pr->dbg.perm_traces, chip, out, cap); }
int64_t vgpu_proof_debug_quotient(const vgpu_proof_t* pr, uint32_t chip, uint32_t* out, uint64_t cap) { return copy_dbg(pr->dbg.quotient_chunks, chip, out, cap); }
void vgpu_proof_free(vgpu_proof_t* pr) { delete pr; }

// ---- RCCL inside the library
int32_t vgpu_comm_unique_id(uint8_t id[VGPU_COMM_ID_BYTES]) {
    VG_TRY({
        if (!id) throw std::invalid_argument("null argument");
        static_assert(sizeof(ncclUniqueId) == VGPU_COMM_ID_BYTES, "RCCL unique id size");
        ncclUniqueId u;
        VG_NCCL_CHECK(RcclApi::get().GetUniqueId(&u));
        memcpy(id, &u, sizeof u);
    })
}
int32_t vgpu_comm_init(vgpu_prover_t* p, const uint8_t id[VGPU_COMM_ID_BYTES], uint32_t rank, uint32_t world, vgpu_comm_t** out) {
    VG_TRY({
        if (!p || !id || !out) throw std::invalid_argument("null argument");
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        std::unique_ptr<vgpu_comm> c(new vgpu_comm());
        c->owner = p->p;
        c->comm.reset(new Comm(&p->p->ctx(), u, (int)rank, (int)world));
        *out = c.release();
    })
}
int32_t vgpu_comm_allgather_roots(vgpu_comm_t* c, const uint32_t* words, uint32_t n_words, uint32_t* out) {
    VG_TRY({
        if (!c || !words || !out || !n_words) throw std::invalid_argument("null argument");
        c->comm->all_gather_words(words, n_words, out);
    })
}
int32_t vgpu_comm_set_timeout_ms(vgpu_comm_t* c, uint32_t timeout_ms) {
    VG_TRY({
        if (!c) throw std::invalid_argument("null argument");
        c->comm->timeout_ms = timeout_ms;
    })
}
void vgpu_comm_destroy(vgpu_comm_t* c) { delete c; }

// ---- sharded commit
int32_t vgpu_commit_batches_sharded(vgpu_prover_t* p, vgpu_comm_t* comm, const vgpu_trace_t* const* mats, uint32_t n_mats, const uint32_t* coset_shifts,
                                    uint32_t root[8]) {
    VG_TRY({
        if (!p || !comm || !mats || !n_mats || !root) throw std::invalid_argument("null argument");
        if (comm->owner != p->p) throw std::invalid_argument("communicator of another prover context");
        DeviceCtx& c = p->p->ctx();
        c.activate();
        std::vector<DMat> nat;
        for (uint32_t i = 0; i < n_mats; i++) { if (!mats[i] || mats[i]->owner != p->p) throw std::invalid_argument("null trace or trace of another prover"); nat.push_back(ingest(c, *mats[i]->t)); }
        std::vector<const DMat*> np;
        for (auto& m : nat) np.push_back(&m);
        std::vector<Fp> shifts;
        if (coset_shifts) for (uint32_t i = 0; i < n_mats; i++) shifts.push_back(Fp::from_canonical(coset_shifts[i]));
        commit_sharded_rccl(*comm->comm, np, coset_shifts ? &shifts : nullptr, p->p->fri(), root);
    })
}
int32_t vgpu_commit_batches_sharded_local(vgpu_prover_t* const* provers, uint32_t world, const vgpu_trace_t* const* mats, uint32_t n_mats,
                                          const uint32_t* coset_shifts, uint32_t root[8]) {
    VG_TRY({
        if (!provers || !world || !mats || !n_mats || !root) throw std::invalid_argument("null argument");
        std::vector<DeviceCtx*> ctxs;
        std::vector<std::vector<DMat>> nat(world);
        std::vector<std::vector<const DMat*>> np(world);
        for (uint32_t r = 0; r < world; r++) {
            if (!provers[r]) throw std::invalid_argument("null prover");
            DeviceCtx& c = provers[r]->p->ctx();
            c.activate();
            ctxs.push_back(&c);
            for (uint32_t i = 0; i < n_mats; i++) {
                const vgpu_trace_t* t = mats[(size_t)r * n_mats + i];
                if (!t || t->owner != provers[r]->p) throw std::invalid_argument("mats[r * n_mats + i] must be uploaded through provers[r]");
                nat[r].push_back(ingest(c, *t->t));
            }
            for (auto& m : nat[r]) np[r].push_back(&m);
            c.sync();
        }
        std::vector<Fp> shifts;
        if (coset_shifts) for (uint32_t i = 0; i < n_mats; i++) shifts.push_back(Fp::from_canonical(coset_shifts[i]));
        commit_sharded_local(ctxs, np, coset_shifts ? &shifts : nullptr, provers[0]->p->fri(), root);
    })
}

// ---- one proof over several ranks
static ShardedInputs sharded_inputs(const vgpu_prover_t* p, const vgpu_trace_t* const* main, uint32_t n_main, const uint32_t* prep_chips, const vgpu_trace_t* const* prep,
                                    uint32_t n_prep, const uint64_t* full_heights = nullptr) {
    ShardedInputs in;
    if (full_heights) in.full_height.assign(full_heights, full_heights + n_main);
    for (uint32_t i = 0; i < n_main; i++) { check_trace(p, main[i]); in.main.push_back(main[i]->t.get()); }
    for (uint32_t i = 0; i < n_prep; i++) { check_trace(p, prep[i]); in.prep.push_back({(int)prep_chips[i], prep[i]->t.get()}); }
    return in;
}
int32_t vgpu_prove_sharded(vgpu_prover_t* p, vgpu_comm_t* comm, const vgpu_trace_t* const* main, uint32_t n_main, const uint32_t* prep_chips,
                           const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t log_min_sharded, vgpu_proof_t** out) {
    VG_TRY({
        if (!p || !comm || !main || !out || (n_prep && (!prep || !prep_chips))) throw std::invalid_argument("null argument");
        if (comm->owner != p->p) throw std::invalid_argument("communicator of another prover context");
        RcclFabric fabric(comm->comm.get());
        std::vector<Prover*> provers{p->p.get()};
        std::vector<ShardedInputs> in{sharded_inputs(p, main, n_main, prep_chips, prep, n_prep)};
        auto proof = std::make_unique<vgpu_proof>();
        proof->words = ShardedProof::run(fabric, provers, in, log_min_sharded);
        *out = proof.release();
    })
}
static void check_fabric_struct(const vgpu_fabric_t* fabric) {
    if (!fabric) throw std::invalid_argument("null fabric");
    if (fabric->struct_size != sizeof(vgpu_fabric_t))
        throw std::invalid_argument("fabric: struct_size " + std::to_string(fabric->struct_size) + " is not this library's sizeof(vgpu_fabric_t) = " + std::to_string(sizeof(vgpu_fabric_t)) +
                                    " (host compiled against another vgpu.h?)");
}
int32_t vgpu_prove_sharded_fabric(vgpu_prover_t* p, const vgpu_fabric_t* fabric, const vgpu_trace_t* const* main, uint32_t n_main, const uint32_t* prep_chips,
                                  const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t log_min_sharded, vgpu_proof_t** out) {
    VG_TRY({
        // the fabric first: from here on whatever this rank refuses — a null argument included — is reported to the peers, who are (or will
        // be) inside the prover waiting for this rank's first status word
        check_fabric_struct(fabric);
        CallbackFabric fab((int)fabric->rank, (int)fabric->world, fabric->all_gather, fabric->all_to_all, fabric->user, fabric->timeout_ms);
        std::vector<Prover*> provers;
        std::vector<ShardedInputs> in;
        try {
            if (!p || !main || !out || (n_prep && (!prep || !prep_chips))) throw std::invalid_argument("null argument");
            provers.push_back(p->p.get());
            in.push_back(sharded_inputs(p, main, n_main, prep_chips, prep, n_prep));
        } catch (...) {
            fab.fail();
            throw;
        }
        auto proof = std::make_unique<vgpu_proof>();
        proof->words = ShardedProof::run(fab, provers, in, log_min_sharded);
        *out = proof.release();
    })
}
// ---- the same with ROW-RANGE inputs: every sharded chip hands in only its rows [rank n / W, (rank + 1) n / W) (sharded_prover.hpp)
uint32_t vgpu_sharded_trace_is_split(uint32_t world, uint32_t log_blowup, uint32_t log_min_sharded, uint64_t height) {
    return sharded_trace_is_split(world, log_blowup, log_min_sharded, height) ? 1u : 0u;
}
int32_t vgpu_prove_sharded_rows(vgpu_prover_t* p, vgpu_comm_t* comm, const vgpu_trace_t* const* main, uint32_t n_main, const uint64_t* full_heights, const uint32_t* prep_chips,
                                const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t log_min_sharded, vgpu_proof_t** out) {
    VG_TRY({
        if (!p || !comm || !main || !full_heights || !out || (n_prep && (!prep || !prep_chips))) throw std::invalid_argument("null argument");
        if (comm->owner != p->p) throw std::invalid_argument("communicator of another prover context");
        RcclFabric fabric(comm->comm.get());
        std::vector<Prover*> provers{p->p.get()};
        std::vector<ShardedInputs> in{sharded_inputs(p, main, n_main, prep_chips, prep, n_prep, full_heights)};
        auto proof = std::make_unique<vgpu_proof>();
        proof->words = ShardedProof::run(fabric, provers, in, log_min_sharded);
        *out = proof.release();
    })
}
int32_t vgpu_prove_sharded_rows_fabric(vgpu_prover_t* p, const vgpu_fabric_t* fabric, const vgpu_trace_t* const* main, uint32_t n_main, const uint64_t* full_heights,
                                       const uint32_t* prep_chips, const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t log_min_sharded, vgpu_proof_t** out) {
    VG_TRY({
        check_fabric_struct(fabric);
        CallbackFabric fab((int)fabric->rank, (int)fabric->world, fabric->all_gather, fabric->all_to_all, fabric->user, fabric->timeout_ms);
        std::vector<Prover*> provers;
        std::vector<ShardedInputs> in;
        try {
            if (!p || !main || !full_heights || !out || (n_prep && (!prep || !prep_chips))) throw std::invalid_argument("null argument");
            provers.push_back(p->p.get());
            in.push_back(sharded_inputs(p, main, n_main, prep_chips, prep, n_prep, full_heights));
        } catch (...) {
            fab.fail();
            throw;
        }
        auto proof = std::make_unique<vgpu_proof>();
        proof->words = ShardedProof::run(fab, provers, in, log_min_sharded);
        *out = proof.release();
    })
}
int32_t vgpu_prove_sharded_rows_local(vgpu_prover_t* const* provers, uint32_t world, const vgpu_trace_t* const* main, uint32_t n_main, const uint64_t* full_heights,
                                      const uint32_t* prep_chips, const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t log_min_sharded, vgpu_proof_t** out) {
    VG_TRY({
        if (!provers || !world || !main || !full_heights || !out || (n_prep && (!prep || !prep_chips))) throw std::invalid_argument("null argument");
        LocalFabric fabric((int)world);
        std::vector<Prover*> ps;
        std::vector<ShardedInputs> in;
        for (uint32_t r = 0; r < world; r++) {
            if (!provers[r]) throw std::invalid_argument("null prover");
            for (uint32_t q = 0; q < r; q++) if (provers[q]->p == provers[r]->p) throw std::invalid_argument("the ranks need distinct prover contexts");
            ps.push_back(provers[r]->p.get());
            in.push_back(sharded_inputs(provers[r], main + (size_t)r * n_main, n_main, prep_chips, n_prep ? prep + (size_t)r * n_prep : nullptr, n_prep, full_heights));
        }
        auto proof = std::make_unique<vgpu_proof>();
        proof->words = ShardedProof::run(fabric, ps, in, log_min_sharded);
        *out = proof.release();
    })
}
int32_t vgpu_fabric_selftest(const vgpu_fabric_t* fabric, uint32_t n_words, uint32_t fail_rank) {
    VG_TRY({
        check_fabric_struct(fabric);
        CallbackFabric fab((int)fabric->rank, (int)fabric->world, fabric->all_gather, fabric->all_to_all, fabric->user, fabric->timeout_ms);
        const uint32_t W = fabric->world, me = fabric->rank;
        auto word = [](uint32_t from, uint32_t to, uint32_t k) { return 0x9E3779B9u * (from + 1) + 0x85EBCA6Bu * (to + 1) + k; };
        try {
            // 1. all_gather through the guarded entry point (status round first)
            std::vector<uint32_t> mine(n_words), all;
            for (uint32_t k = 0; k < n_words; k++) mine[k] = word(me, W, k);
            fab.all_gather({mine.data()}, n_words, all);
            for (uint32_t r = 0; r < W; r++)
                for (uint32_t k = 0; k < n_words; k++)
                    if (all[(size_t)r * n_words + k] != word(r, W, k)) throw std::runtime_error("fabric selftest: all_gather delivered a wrong word");
            // 2. a rank that fails between two exchanges, as a failing proof would
            if (me == fail_rank) throw std::invalid_argument("fabric selftest: rank " + std::to_string(me) + " fails on request");
            // 3. all_to_all of blocks whose size depends on the pair
            fab.agree();
            // the blocks live in the fabric's shared state: a callback abandoned at the deadline may still write into them
            struct Blocks { std::vector<std::vector<uint32_t>> sb, rb; };
            auto blocks = std::make_shared<Blocks>();
            blocks->sb.resize(W); blocks->rb.resize(W);
            fab.sh->keep = blocks;
            auto& sb = blocks->sb; auto& rb = blocks->rb;
            std::vector<const uint32_t*> sp(W, nullptr);
            std::vector<uint32_t*> rp(W, nullptr);
            std::vector<uint64_t> sw(W, 0), rw(W, 0);
            for (uint32_t s = 0; s < W; s++) {
                if (s == me) continue;
                sb[s].resize(n_words + 3 * me + s); rb[s].resize(n_words + 3 * s + me);
                for (size_t k = 0; k < sb[s].size(); k++) sb[s][k] = word(me, s, (uint32_t)k);
                sp[s] = sb[s].data(); sw[s] = sb[s].size(); rp[s] = rb[s].data(); rw[s] = rb[s].size();
            }
            fab.host_all_to_all(sp, sw, rp, rw);
            for (uint32_t s = 0; s < W; s++)
                for (size_t k = 0; k < rb[s].size(); k++)
                    if (rb[s][k] != word(s, me, (uint32_t)k)) throw std::runtime_error("fabric selftest: all_to_all delivered a wrong word");
            fab.agree();  // and a last round, in which a late failure of a peer would still surface
        } catch (const FabricPeerFailure&) {
            throw;
        } catch (...) {
            fab.fail();
            throw;
        }
    })
}
int32_t vgpu_prove_sharded_local(vgpu_prover_t* const* provers, uint32_t world, const vgpu_trace_t* const* main, uint32_t n_main,
                                 const uint32_t* prep_chips, const vgpu_trace_t* const* prep, uint32_t n_prep, uint32_t log_min_sharded, vgpu_proof_t** out) {
    VG_TRY({
        if (!provers || !world || !main || !out || (n_prep && (!prep || !prep_chips))) throw std::invalid_argument("null argument");
        LocalFabric fabric((int)world);
        std::vector<Prover*> ps;
        std::vector<ShardedInputs> in;
        for (uint32_t r = 0; r < world; r++) {
            if (!provers[r]) throw std::invalid_argument("null prover");
            for (uint32_t q = 0; q < r; q++) if (provers[q]->p == provers[r]->p) throw std::invalid_argument("the ranks need distinct prover contexts");
            ps.push_back(provers[r]->p.get());
            in.push_back(sharded_inputs(provers[r], main + (size_t)r * n_main, n_main, prep_chips, n_prep ? prep + (size_t)r * n_prep : nullptr, n_prep));
        }
        auto proof = std::make_unique<vgpu_proof>();
        proof->words = ShardedProof::run(fabric, ps, in, log_min_sharded);
        *out = proof.release();
    })
}

// ---- workloads
int32_t vgpu_workload_fib(uint32_t n, vgpu_workload_t** out) {
    VG_TRY({
        if (!out) throw std::invalid_argument("null out");
        auto w = std::make_unique<vgpu_workload>();
        w->vm.reset(new vwork::BasicVm(vwork::fib_program(n)));
        w->vm->run();
        w->main = w->vm->main_traces();
        w->prep_program = w->vm->program_preprocessed();
        w->prep_range = vwork::BasicVm::range_preprocessed();
        auto it = w->vm->cells.find(0x1000 + 4);
        w->result = it == w->vm->cells.end() ? 0 : it->second;
        fill_logs(*w);
        *out = w.release();
    })
}
int32_t vgpu_workload_alu(uint32_t iters, vgpu_workload_t** out) {
    VG_TRY({
        if (!out || !iters) throw std::invalid_argument("bad argument");
        auto w = std::make_unique<vgpu_workload>();
        w->vm.reset(new vwork::BasicVm(vwork::alu_program(iters)));
        w->vm->run();
        w->main = w->vm->main_traces();
        w->prep_program = w->vm->program_preprocessed();
        w->prep_range = vwork::BasicVm::range_preprocessed();
        auto it = w->vm->cells.find(0x1000 - 4);
        w->result = it == w->vm->cells.end() ? 0 : it->second;
        fill_logs(*w);
        *out = w.release();
    })
}
static void fill_logs(vgpu_workload& w) {
    const vwork::BasicVm& vm = *w.vm;
    std::vector<uint32_t> first(vm.cpu_ops.size() + 1, 0);
    for (auto& m : vm.mem_ops) first[m.clk + 1]++;
    for (size_t i = 0; i < vm.cpu_ops.size(); i++) first[i + 1] += first[i];
    w.log_cpu.resize(vm.cpu_ops.size());
    for (size_t i = 0; i < vm.cpu_ops.size(); i++) {
        const auto& r = vm.cpu_ops[i];
        vgpu_cpu_op_t& o = w.log_cpu[i];
        o.pc = r.pc; o.fp = r.fp; o.opcode = r.instr.opcode;
        for (int k = 0; k < 5; k++) o.operands[k] = r.instr.ops[k];
        o.kind = (uint32_t)r.op; o.has_imm = r.has_imm ? 1 : 0; o.imm = vwork::u32_of(r.imm); o.mem_first = first[i];
    }
    w.log_mem.resize(vm.mem_ops.size());
    for (size_t i = 0; i < vm.mem_ops.size(); i++) w.log_mem[i] = {vm.mem_ops[i].clk, vm.mem_ops[i].addr, vwork::u32_of(vm.mem_ops[i].value), vm.mem_ops[i].is_write ? 1u : 0u};
    for (auto& kv : vm.static_cells) { w.log_static.push_back(kv.first); w.log_static.push_back(vwork::u32_of(kv.second)); }
    const std::vector<vwork::AluOp>* src[4] = {&vm.add_ops, &vm.sub_ops, &vm.lt_ops, &vm.bitwise_ops};
    for (int k = 0; k < 4; k++) {
        w.log_alu[k].resize(src[k]->size());
        for (size_t i = 0; i < src[k]->size(); i++) { const auto& a = (*src[k])[i]; w.log_alu[k][i] = {a.opcode, vwork::u32_of(a.a), vwork::u32_of(a.b), vwork::u32_of(a.c)}; }
    }
    const std::vector<vwork::AluOp>* src2[4] = {&vm.mul_ops, &vm.div_ops, &vm.shift_ops, &vm.com_ops};
    for (int k = 0; k < 4; k++) {
        w.log_alu2[k].resize(src2[k]->size());
        for (size_t i = 0; i < src2[k]->size(); i++) { const auto& a = (*src2[k])[i]; w.log_alu2[k][i] = {a.opcode, vwork::u32_of(a.a), vwork::u32_of(a.b), vwork::u32_of(a.c)}; }
    }
    for (auto& v : vm.output_values) w.log_output.push_back({v.first, v.second});
}
int32_t vgpu_workload_named(const char* name, vgpu_workload_t** out) {
    VG_TRY({
        if (!out || !name) throw std::invalid_argument("bad argument");
        std::string n(name);
        std::vector<vwork::InstructionWord> prog;
        if (n == "left_imm_ops") prog = vwork::left_imm_ops_program();
        else if (n == "signed_inequality") prog = vwork::signed_inequality_program();
        else if (n == "loadfp") prog = vwork::loadfp_program();
        else if (n == "static_data") prog = vwork::static_data_program();
        else if (n.rfind("mixed_ops", 0) == 0) prog = vwork::mixed_ops_program(n.size() > 10 ? (uint32_t)std::stoul(n.substr(10)) : 40u);  // "mixed_ops" or "mixed_ops:<iters>"
        else throw std::invalid_argument("unknown program: " + n);
        auto w = std::make_unique<vgpu_workload>();
        w->vm.reset(new vwork::BasicVm(prog));
        if (n == "static_data") {  // basic/tests/test_static_data.rs:57-58
            w->vm->write_static(0x10, vwork::Word{{0, 0, 0, 0x25}});
            w->vm->write_static(0x14, vwork::Word{{0, 0, 0, 0x32}});
        }
        w->vm->run();
        w->main = w->vm->main_traces();
        w->prep_program = w->vm->program_preprocessed();
        w->prep_range = vwork::BasicVm::range_preprocessed();
        fill_logs(*w);
        *out = w.release();
    })
}
int32_t vgpu_workload_cell(const vgpu_workload_t* w, uint32_t addr, uint32_t* value) {
    VG_TRY({
        if (!w || !value) throw std::invalid_argument("bad argument");
        auto it = w->vm->cells.find(addr);
        if (it == w->vm->cells.end()) throw std::invalid_argument("memory cell never written");
        *value = it->second;
    })
}
void vgpu_workload_oplog(const vgpu_workload_t* w, vgpu_oplog_desc_t* out) {
    out->struct_size = sizeof(vgpu_oplog_desc_t);
    out->cpu = w->log_cpu.data(); out->n_cpu = w->log_cpu.size();
    out->mem = w->log_mem.data(); out->n_mem = w->log_mem.size();
    for (int k = 0; k < 4; k++) { out->alu[k] = w->log_alu[k].data(); out->n_alu[k] = w->log_alu[k].size(); }
    out->static_cells = w->log_static.data(); out->n_static = w->log_static.size() / 2;
    out->rom_len = (uint32_t)w->vm->rom.size();
    for (int k = 0; k < 4; k++) { out->alu2[k] = w->log_alu2[k].data(); out->n_alu2[k] = w->log_alu2[k].size(); }
    out->output = w->log_output.data(); out->n_output = w->log_output.size();
}
void vgpu_workload_free(vgpu_workload_t* w) { delete w; }
void vgpu_workload_stats(const vgpu_workload_t* w, uint64_t out[8]) {
    out[0] = w->vm->clock; out[1] = w->vm->cpu_ops.size(); out[2] = w->vm->mem_ops.size(); out[3] = w->vm->add_ops.size();
    out[4] = w->result; out[5] = w->vm->rom.size(); out[6] = w->main[0].height; out[7] = 0;
}
int32_t vgpu_workload_main_trace(const vgpu_workload_t* w, uint32_t chip, const uint32_t** data, uint64_t* height, uint64_t* width) {
    VG_TRY({
        if (!w || chip >= w->main.size()) throw std::invalid_argument("bad chip index");
        *data = w->main[chip].v.data(); *height = w->main[chip].height; *width = w->main[chip].width;
    })
}
int32_t vgpu_workload_preprocessed(const vgpu_workload_t* w, uint32_t k, uint32_t* chip, const uint32_t** data, uint64_t* height, uint64_t* width) {
    VG_TRY({
        if (!w || k > 1) throw std::invalid_argument("bad index");
        const vwork::RowMajor& m = k == 0 ? w->prep_program : w->prep_range;
        *chip = k == 0 ? (uint32_t)vchips::CHIP_PROGRAM : (uint32_t)vchips::CHIP_RANGE;
        *data = m.v.data(); *height = m.height; *width = m.width;
    })
}

}  // extern "C"
