// Machine::prove on one MI355X (see prover.hpp).  Phase order and transcript follow
// basic/src/lib.rs:147-675 / SURVEY.md Appendix C; PCS conventions SURVEY.md Appendix B.
#include "prover.hpp"
#include "poseidon_opt.hpp"

namespace vhost {

using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

static Ext5 ext_from_canonical(const uint32_t* w) { Ext5 e; for (int k = 0; k < 5; k++) e.c[k] = Fp::from_canonical(w[k]); return e; }
static void ext_to_canonical(const Ext5& e, uint32_t* w) { for (int k = 0; k < 5; k++) w[k] = e.c[k].canonical(); }

Prover::Prover(int device, const MachineDesc& machine, const uint32_t* poseidon_rc480, const FriParams& fri)
    : ctx_(new DeviceCtx(device)), machine_(machine), fri_(fri), perm16_(poseidon_rc480) {
    for (auto& a : machine_.airs) {
        if (a.log_quotient_degree < 1 || a.log_quotient_degree > 3) throw std::invalid_argument("chip " + a.name + ": log_quotient_degree must be 1..3");
        // native_chip: a BasicMachine ChipId (0..13: eval and interactions compiled into the kernels) or -2 (captured AIR: interpreted program + encoded
        // interactions).  Nothing else — in particular not -1, which only exists INSIDE the launchers as the template argument "no AIR constraints" that
        // program / mem / div / range map to — so that what is accepted does not depend on which quotient kernel a switch selects.
        if (a.native_chip != -2 && (a.native_chip < 0 || a.native_chip >= 14)) throw std::invalid_argument("chip " + a.name + ": native_chip must be a BasicMachine chip id (0..13) or -2 (captured AIR)");
        if (a.log_quotient_degree != 1 && a.native_chip >= 0) throw std::invalid_argument("chip " + a.name + ": the compiled chip kernels assume log_quotient_degree = 1");
        if (fri_.log_blowup < a.log_quotient_degree)
            throw std::invalid_argument("chip " + a.name + ": log_blowup must be >= log_quotient_degree (the quotient domain must lie inside the committed LDE)");
        std::vector<uint32_t> pw(a.program.instrs.size() * 2 + 2, 0);
        if (!a.program.instrs.empty()) memcpy(pw.data(), a.program.instrs.data(), a.program.instrs.size() * 8);
        prog_dev_.emplace_back(ctx_.get(), pw);
        iw_dev_.emplace_back(ctx_.get(), a.interaction_words);
    }
    // Poseidon tables for the device PoW search: round constants + circulant CosetMds coefficients
    bool sparse = false;  // sparse form of the partial rounds + convolution form of the MDS layer for the Poseidon MMCS kernels (word 1024 on)
    const std::vector<uint32_t> pos = poseidon_device_image(poseidon_rc480, perm16_, sparse);
    pow_pos_ = DBuf(ctx_.get(), pos);
    VG_HIP_CHECK(hipHostMalloc((void**)&cs_pinned_, CS_PINNED_WORDS * 4));
    ctx_->hash_kind = fri_.hash_kind;
    ctx_->poseidon_tab = pow_pos_.data;
    ctx_->poseidon_sparse = sparse;
}

// challenger.grind(bits) with the search on the device; canonical rule = smallest witness.
uint32_t Prover::grind(Challenger& ch) {
    DeviceCtx& c = *ctx_;
    const unsigned bits = fri_.pow_bits;
    if (bits == 0) { ch.check_witness(0, Fp::zero()); return 0; }
    uint32_t st[17];
    for (int i = 0; i < 16; i++) st[i] = ch.state[i].v;
    for (size_t i = 0; i < ch.in.size(); i++) st[i] = ch.in[i].v;
    st[16] = 0xffffffffu;
    const uint32_t k_pending = (uint32_t)ch.in.size();
    const uint32_t batch = 1u << (bits + 6 > 20 ? 20 : bits + 6);
    uint32_t* best_dev = pow_pos_.data + 512;
    for (uint64_t first = 0; first < vg::P; first += batch) {
        c.upload_async(pow_pos_.data + 496, st, 17 * 4);
        uint32_t count = (uint32_t)std::min<uint64_t>(batch, vg::P - first);
        vk::launch_pow_grind(c.stream, pow_pos_.data, c.poseidon_sparse, k_pending, (uint32_t)first, count, bits, best_dev);
        uint32_t best;
        c.download_small(&best, best_dev, 4);
        if (best != 0xffffffffu) {
            if (!ch.check_witness(bits, Fp::from_canonical(best))) throw std::runtime_error("pow: device witness rejected by the host challenger");
            return best;
        }
    }
    throw std::runtime_error("pow: no witness found");
}
Prover::~Prover() {
    if (cs_pinned_) (void)hipHostFree(cs_pinned_);
    if (open_pinned_) (void)hipHostFree(open_pinned_);
    prog_dev_.clear();
    iw_dev_.clear();
    prep_pd_cache_.reset();
    prep_nat_cache_.clear();
}

std::unique_ptr<DeviceTrace> Prover::upload_trace(const HostMatrix& m) {
    auto t = std::make_unique<DeviceTrace>();
    t->height = m.height; t->width = m.width;
    t->raw = DBuf(ctx_.get(), (size_t)(m.height * m.width));
    if (ctx_->proofs_running.load() > 0) {  // the caller prepares the next segment while a proof owns this context's streams
        ctx_->upload_beside_proof(t->raw.data, m.data, m.height * m.width * 4);
        return t;
    }
    VG_HIP_CHECK(hipMemcpyAsync(t->raw.data, m.data, m.height * m.width * 4, hipMemcpyHostToDevice, ctx_->stream));
    ctx_->sync();
    return t;
}

static size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }
static DBuf upload_bytes(DeviceCtx& c, const void* src, size_t bytes) {
    DBuf b(&c, (bytes + 3) / 4 + 4);
    if (bytes) VG_HIP_CHECK(hipMemcpyAsync(b.data, src, bytes, hipMemcpyHostToDevice, c.stream));
    return b;
}

std::unique_ptr<DeviceOplog> Prover::upload_oplog(const HostOplog& log) {
    if (!log.n_cpu || !log.cpu) throw std::invalid_argument("oplog: empty cpu log");
    DeviceCtx& c = *ctx_;
    auto d = std::make_unique<DeviceOplog>();
    d->n_cpu = log.n_cpu; d->n_mem = log.n_mem;
    d->cpu = upload_bytes(c, log.cpu, log.n_cpu * sizeof(vk::TgCpuOp));
    d->mem = upload_bytes(c, log.mem, log.n_mem * sizeof(vk::TgMemOp));
    for (int k = 0; k < 4; k++) { d->n_alu[k] = log.n_alu[k]; d->alu[k] = upload_bytes(c, log.alu[k], log.n_alu[k] * sizeof(vk::TgAluOp)); }
    d->rom_len = log.rom_len;
    d->n_static = log.n_static;
    d->static_cells = upload_bytes(c, log.static_cells, log.n_static * 8);
    for (int k = 0; k < 4; k++) { d->n_alu2[k] = log.n_alu2[k]; d->alu2[k] = upload_bytes(c, log.alu2[k], log.n_alu2[k] * sizeof(vk::TgAluOp)); }
    d->n_output = log.n_output;
    d->output = upload_bytes(c, log.output, log.n_output * sizeof(vk::TgOutOp));
    // first row of every window of the output tape (output/src/lib.rs:41-47: (clk_2 - clk_1) / table_len + 1 rows per window, one final
    // row): a running sum over a handful of entries, done here where the log is in host memory; the rows themselves are filled on the device
    std::vector<uint32_t> row0(log.n_output ? log.n_output : 1, 0);
    uint64_t rows = 0;
    for (uint64_t w = 0; w < log.n_output; w++) {
        row0[w] = (uint32_t)rows;
        if (w + 1 < log.n_output) {
            if (log.output[w + 1].clk < log.output[w].clk) throw std::invalid_argument("oplog: output tape entry " + std::to_string(w + 1) + " is out of clock order");
            rows += (uint64_t)((log.output[w + 1].clk - log.output[w].clk) / (uint32_t)log.n_output) + 1;
        } else rows += 1;
        if (rows > (1ull << 27)) throw std::invalid_argument("oplog: the output chip's trace would exceed 2^27 rows");
    }
    d->output_rows = rows;
    d->output_row0 = upload_bytes(c, row0.data(), row0.size() * 4);
    c.sync();
    return d;
}

bool Prover::can_generate(int chip) {
    using namespace vchips;
    // every chip of the BasicMachine: from its log (static_data from the initialised cells, range from the range-checked result words)
    return chip >= 0 && chip < NUM_CHIPS;
}

std::unique_ptr<DeviceTrace> Prover::generate_trace(const DeviceOplog& log, int chip) {
    using namespace vchips;
    if (!can_generate(chip) || (size_t)chip >= machine_.airs.size() || machine_.airs[chip].native_chip != chip)
        throw std::invalid_argument("generate_trace: no device trace generator for this chip");
    DeviceCtx& c = *ctx_;
    auto t = std::make_unique<DeviceTrace>();
    t->width = machine_.airs[chip].width;
    if (chip == CHIP_CPU) {
        t->height = next_pow2(log.n_cpu);
        t->nat = DMat(&c, t->height, t->width);
        vk::launch_tracegen_cpu(c.stream, (const vk::TgCpuOp*)log.cpu.data, log.n_cpu, (const vk::TgMemOp*)log.mem.data, log.n_mem, t->nat.view());
    } else if (chip == CHIP_MEM) {
        t->height = next_pow2(log.n_static + log.n_mem);  // static rows first; 0 rows -> one zero row
        t->nat = DMat(&c, t->height, t->width);
        const uint64_t n = log.n_mem;
        const size_t tmp_bytes = vk::tracegen_mem_sort_scratch_bytes(n ? n : 1);
        DBuf keys(&c, 2 * n + 4), idx(&c, 2 * n + 4), tmp(&c, tmp_bytes / 4 + 4);
        VG_HIP_CHECK(vk::launch_tracegen_mem(c.stream, (const vk::TgMemOp*)log.mem.data, n, log.static_cells.data, log.n_static, keys.data, idx.data, tmp.data,
                                             tmp_bytes, t->nat.view()));
        // the scratch buffers return to the pool here while the kernels may still be queued: safe, the pool only ever hands a
        // block to work enqueued LATER on the same stream
    } else if (chip == CHIP_RANGE || chip == CHIP_PROGRAM) {
        if (chip == CHIP_PROGRAM && !log.rom_len) throw std::invalid_argument("generate_trace: the operation log carries no ROM length");
        t->height = chip == CHIP_RANGE ? 256 : next_pow2(log.rom_len);
        t->nat = DMat(&c, t->height, t->width);
        DBuf counts(&c, std::max<size_t>(256, log.rom_len) + 4);
        if (chip == CHIP_RANGE)
            VG_HIP_CHECK(vk::launch_tracegen_range(c.stream, (const vk::TgCpuOp*)log.cpu.data, log.n_cpu, (const vk::TgMemOp*)log.mem.data, log.n_mem, counts.data, t->nat.view()));
        else
            VG_HIP_CHECK(vk::launch_tracegen_program(c.stream, (const vk::TgCpuOp*)log.cpu.data, log.n_cpu, next_pow2(log.n_cpu), log.rom_len, counts.data, t->nat.view()));
    } else if (chip == CHIP_MUL || chip == CHIP_DIV || chip == CHIP_SHIFT || chip == CHIP_COM) {
        const int k = chip == CHIP_MUL ? 0 : chip == CHIP_DIV ? 1 : chip == CHIP_SHIFT ? 2 : 3;
        t->height = next_pow2(log.n_alu2[k]);
        if (chip == CHIP_MUL && t->height < 1024) t->height = 1024;  // MIN_LENGTH, for the range-check counter (alu_u32/src/mul/mod.rs:39-42)
        t->nat = DMat(&c, t->height, t->width);
        vk::launch_tracegen_alu2(c.stream, chip, (const vk::TgAluOp*)log.alu2[k].data, log.n_alu2[k], t->nat.view());
    } else if (chip == CHIP_OUTPUT) {
        t->height = next_pow2(log.output_rows);
        t->nat = DMat(&c, t->height, t->width);
        vk::launch_tracegen_output(c.stream, (const vk::TgOutOp*)log.output.data, log.output_row0.data, log.n_output, log.output_rows, t->nat.view());
    } else if (chip == CHIP_STATIC_DATA) {
        t->height = next_pow2(log.n_static);
        t->nat = DMat(&c, t->height, t->width);
        vk::launch_tracegen_idle(c.stream, 2, log.static_cells.data, log.n_static, t->nat.view());
    } else {
        const int k = chip == CHIP_ADD ? 0 : chip == CHIP_SUB ? 1 : chip == CHIP_LT ? 2 : 3;
        t->height = next_pow2(log.n_alu[k]);
        t->nat = DMat(&c, t->height, t->width);
        vk::launch_tracegen_alu(c.stream, chip, (const vk::TgAluOp*)log.alu[k].data, log.n_alu[k], t->nat.view());
    }
    c.check_launch("generate_trace");
    return t;
}

void Prover::download_trace(const DeviceTrace& t, uint32_t* out) {
    DeviceCtx& c = *ctx_;
    if (t.nat.empty()) { c.download(out, t.raw.data, t.height * t.width * 4); return; }
    DBuf tmp(&c, (size_t)(t.height * t.width));
    vk::launch_export_rows(c.stream, t.nat.view(), 0, t.height, tmp.data);
    c.download(out, tmp.data, t.height * t.width * 4);
}

namespace {
struct PointKey {
    uint32_t w[5];
    bool operator<(const PointKey& o) const { return memcmp(w, o.w, 20) < 0; }
};
PointKey key_of(const Ext5& e) { PointKey k; for (int i = 0; i < 5; i++) k.w[i] = e.c[i].v; return k; }

struct GatherList {
    std::vector<uint32_t> desc;
    uint32_t next_dst = 0;
    // returns dst offset
    uint32_t add(const uint32_t* src, uint64_t stride, uint32_t count, uint32_t kind) {
        put_ptr(desc, src);
        put_u64(desc, stride);
        desc.push_back(count | (kind << 28));
        desc.push_back(next_dst);
        uint32_t at = next_dst;
        next_dst += count;
        return at;
    }
    size_t n() const { return desc.size() / 6; }
};
}  // namespace

// Constants of one chip's quotient launch: [K alpha powers][M bus alphas][max_fields betas][cumulative_sum] (QuotientArgs::consts)
static uint32_t quotient_consts(const AirDesc& air, const Ext5 rnd[3], const Ext5& alpha, const Ext5& cumulative_sum, std::vector<uint32_t>& pool) {
    const uint32_t M = (uint32_t)air.interactions.size(), K = air.program.num_asserts + M + 3;
    std::vector<Ext5> ap(K);
    Ext5 p = Ext5::one();
    for (uint32_t k = 0; k < K; k++) { ap[K - 1 - k] = p; p *= alpha; }  // constraint k is scaled by alpha^(K-1-k)
    for (auto& e : ap) put_ext(pool, e);
    size_t maxf = 0;
    for (auto& it : air.interactions) {
        put_ext(pool, (it.is_local() ? rnd[0] : rnd[1]).pow((uint64_t)it.bus_index + 1));  // generate_rlc_elements (chip.rs:291-331)
        maxf = std::max(maxf, it.fields.size());
    }
    Ext5 bp = Ext5::one();
    for (size_t j = 0; j < maxf; j++) { put_ext(pool, bp); bp *= rnd[2]; }
    put_ext(pool, cumulative_sum);
    return K;
}

DMat Prover::permutation_trace(int chip, const DMat& main_nat, const DMat* prep_nat, const Ext5 rnd[3], Ext5* cumulative_sum) {
    DeviceCtx& c = *ctx_;
    const AirDesc& air = machine_.airs.at((size_t)chip);
    const uint32_t M = (uint32_t)air.interactions.size();
    std::vector<uint32_t> pool;
    size_t maxf = 0;
    for (auto& it : air.interactions) { put_ext(pool, (it.is_local() ? rnd[0] : rnd[1]).pow((uint64_t)it.bus_index + 1)); maxf = std::max(maxf, it.fields.size()); }
    Ext5 bp = Ext5::one();
    for (size_t j = 0; j < maxf; j++) { put_ext(pool, bp); bp *= rnd[2]; }
    pool.push_back(0);
    DBuf pool_dev(&c, pool), scratch(&c, (size_t)vk::perm_scratch_words(main_nat.height));
    DMat perm(&c, main_nat.height, 5 * (M + 1));
    vk::launch_perm_trace(c.stream, main_nat.view(), prep_nat ? prep_nat->view() : vk::DMatView{nullptr, 0, 0, 0}, iw_dev_[chip].data, pool_dev.data, M, perm.view(), scratch.data, fri_.interpret_air ? -2 : machine_.airs[(size_t)chip].native_chip);
    c.check_launch("perm trace");
    if (cumulative_sum) {  // last row of the running-sum column (lib.rs:247-250)
        GatherList gl;
        const uint64_t n = main_nat.height;
        gl.add(perm.data + (uint64_t)(5 * M) * n + (n - 1), n, 5, 0);
        DBuf gd(&c, gl.desc), gout(&c, 8);
        vk::launch_gather(c.stream, gd.data, gl.n(), gout.data);
        uint32_t cs[5];
        c.download(cs, gout.data, 20);
        *cumulative_sum = ext_from_canonical(cs);
    } else c.sync();
    return perm;
}

DMat Prover::quotient_chunks(int chip, const DMat& main_lde, const DMat& perm_lde, const DMat* prep_lde, const Ext5 rnd[3], const Ext5& alpha,
                             const Ext5& cumulative_sum) {
    DeviceCtx& c = *ctx_;
    const AirDesc& air = machine_.airs.at((size_t)chip);
    const unsigned lb = fri_.log_blowup;
    const unsigned log_n = vg::log2_strict_u64(main_lde.height) - lb;
    if (main_lde.width != air.width || perm_lde.width != 5 * (air.interactions.size() + 1) || perm_lde.height != main_lde.height ||
        (air.prep_width && (!prep_lde || prep_lde->width != air.prep_width || prep_lde->height != main_lde.height)))
        throw std::invalid_argument("quotient: LDE shapes do not match chip " + air.name);
    std::vector<uint32_t> pool;
    vk::QuotientArgs a{};
    a.K = quotient_consts(air, rnd, alpha, cumulative_sum, pool);
    DBuf pool_dev(&c, pool);
    fill_quotient_args(a, chip, main_lde.view(), perm_lde.view(), prep_lde ? prep_lde->view() : vk::DMatView{nullptr, 0, 0, 0}, log_n, pool_dev.data);
    DMat q(&c, 1ull << log_n, 5ull << air.log_quotient_degree);
    a.out = q.view();
    vk::launch_quotient(c.stream, a, c.tables);
    c.check_launch("quotient");
    c.sync();  // pool_dev is released on return
    return q;
}

// Everything of QuotientArgs but `out` and K (shared by prove() and quotient_chunks)
void Prover::fill_quotient_args(vk::QuotientArgs& a, int chip, vk::DMatView main_lde, vk::DMatView perm_lde, vk::DMatView prep_lde, unsigned log_n, const uint32_t* consts_dev) {
    const AirDesc& air = machine_.airs[(size_t)chip];
    const Fp s = Fp::from_canonical(vg::GENERATOR);
    a.main_lde = main_lde; a.perm_lde = perm_lde; a.prep_lde = prep_lde;
    a.log_n = (int)log_n;
    a.prog = (const vair::Instr*)prog_dev_[chip].data;
    a.n_instrs = (uint32_t)air.program.instrs.size();
    a.n_regs = air.program.num_regs;
    a.n_air_asserts = air.program.num_asserts;
    a.native_chip = fri_.interpret_air ? vk::QuotientArgs::INTERPRET : air.native_chip;
    a.iw = iw_dev_[chip].data;
    a.consts = consts_dev;
    a.coset_shift = s.v;
    a.coset_shift_inv = s.inv().v;
    // ZerofierOnCoset::new(log_n, lqd, s) (App. B11): Z_H(s w_{Qn}^i) = s^n w_Q^(i mod Q) - 1
    a.lqd = (int)air.log_quotient_degree;
    const Fp sn = s.exp_power_of_2(log_n), wq = vg::two_adic_generator(air.log_quotient_degree);
    Fp wp = Fp::one();
    for (unsigned r = 0; r < (1u << air.log_quotient_degree); r++) { Fp z = sn * wp - Fp::one(); a.zh[r] = z.v; a.zh_inv[r] = z.inv().v; wp *= wq; }
    a.g_inv = vg::two_adic_generator(log_n).inv().v;
}


std::vector<uint32_t> Prover::prove(const std::vector<const DeviceTrace*>& main, const std::vector<std::pair<int, const DeviceTrace*>>& preprocessed,
                                    PhaseTimes* times, ProveDebugOut* dbg) {
    DeviceCtx& c = *ctx_;
    c.activate();
    // one proof at a time per context: a second caller (another thread's vgpu_prove, the worker of a second vgpu_prove_async) waits here
    // until the running proof has finished — the calls are served one after the other, none is refused
    std::unique_lock<std::mutex> one_at_a_time(c.prove_mu);
    c.activate();
    struct Running {  // lets uploads of another thread know that the context's main stream is owned (DeviceCtx::upload_beside_proof)
        std::atomic<int>& n;
        explicit Running(std::atomic<int>& a) : n(a) { n.fetch_add(1); }
        ~Running() { n.fetch_sub(1); }
    } running(c.proofs_running);
    const size_t NC = machine_.airs.size();
    if (main.size() != NC) throw std::invalid_argument("prove: need one main trace per chip");
    const Fp s = Fp::from_canonical(vg::GENERATOR);  // pcs.coset_shift()
    const unsigned lb = fri_.log_blowup;
    Challenger ch(&perm16_);
    PhaseTimes tm;
    auto t_total = Clock::now();
    auto t0 = Clock::now();

    // ---------------- ingest: row-major canonical -> column-major Montgomery (natural row order)
    std::vector<unsigned> log_deg(NC);
    std::vector<DMat> main_own(NC);
    std::vector<const DMat*> main_nat(NC);
    Section ingest_section(&c);
    for (size_t i = 0; i < NC; i++) {
        if (main[i]->width != machine_.airs[i].width) throw std::invalid_argument("prove: trace width mismatch for chip " + machine_.airs[i].name);
        uint64_t h = main[i]->height;
        if (h == 0 || (h & (h - 1))) throw std::invalid_argument("prove: trace heights must be powers of two");
        log_deg[i] = vg::log2_strict_u64(h);
        if (!main[i]->nat.empty()) { main_nat[i] = &main[i]->nat; continue; }  // generated on the device: already in working layout
        main_own[i] = DMat(&c, h, main[i]->width);
        main_nat[i] = &main_own[i];
        vk::launch_ingest(c.stream_for(i, h), main[i]->raw.data, main_own[i].view(), false);
    }
    std::vector<std::pair<int, uint64_t>> prep_key;
    for (auto& pr : preprocessed) prep_key.emplace_back(pr.first, pr.second->uid);
    const bool prep_hit = prep_cache_enabled_ && !prep_key.empty() && prep_key == prep_key_ && prep_pd_cache_;
    if (!prep_hit) { prep_key_.clear(); prep_pd_cache_.reset(); prep_nat_cache_.clear(); prep_nat_cache_.resize(preprocessed.size()); }
    std::vector<DMat>& prep_nat = prep_nat_cache_;
    // cache off (the default): nothing of the preprocessed round outlives this proof — the copies, their LDEs and the tree go back to the pool
    // when prove() leaves, normally or not (vgpu_prover_trim can then return them)
    struct DropPrep { Prover* p; ~DropPrep() { if (!p->prep_cache_enabled_) { p->prep_key_.clear(); p->prep_pd_cache_.reset(); p->prep_nat_cache_.clear(); } } } drop_prep{this};
    std::vector<int> prep_slot(NC, -1);
    for (size_t k = 0; k < preprocessed.size(); k++) {
        const DeviceTrace* t = preprocessed[k].second;
        int chip = preprocessed[k].first;
        if (chip < 0 || (size_t)chip >= NC || prep_slot[chip] >= 0) throw std::invalid_argument("prove: bad or repeated preprocessed chip index");
        if (t->width != machine_.airs[chip].prep_width || t->height != main[chip]->height) throw std::invalid_argument("prove: preprocessed trace shape mismatch");
        prep_slot[chip] = (int)k;
        if (prep_hit) continue;
        prep_nat[k] = DMat(&c, t->height, t->width);
        if (!t->nat.empty())  // a device-resident working-layout trace handed in as preprocessed: copy, never read the (absent) raw image
            VG_HIP_CHECK(hipMemcpyAsync(prep_nat[k].data, t->nat.data, t->height * t->width * 4, hipMemcpyDeviceToDevice, c.stream_for(k, t->height)));
        else
            vk::launch_ingest(c.stream_for(k, t->height), t->raw.data, prep_nat[k].view(), false);
    }
    ingest_section.join();
    c.check_launch("ingest");
    tm.ingest = ms_since(t0);  // host-side phase boundaries: the GPU work of a phase completes at the next true synchronisation (a root download)

    // ---------------- preprocessed + main commitments (lib.rs:189-225)
    t0 = Clock::now();
    // The preprocessed commitment is small (range table, program ROM: one-tile LDEs, a tree of single-workgroup launches) and independent of
    // the main one: it RIDES on the auxiliary stream beside the main round's big LDE passes and its root arrives with the main round's
    // synchronisation (pcs.hpp: CommitRider) instead of costing a commit and a synchronisation of its own in front of them.  The transcript
    // order is the reference's: preprocessed root, then main root.  (One after the other: +0.3 ms per lone proof, profiles/r04_ab_latency.json.)
    ProverData* prep_pd = nullptr;
    std::vector<CommitInput> prep_in;
    const bool ride = !prep_nat.empty() && !prep_hit;
    if (ride)
        for (auto& m : prep_nat) prep_in.push_back({&m, false, false});
    std::unique_ptr<ProverData> main_pd;
    {
        std::vector<CommitInput> in;
        for (auto m : main_nat) in.push_back({const_cast<DMat*>(m), false, false});  // consume = false: never modified
        const CommitRider rider{&prep_in, &prep_pd_cache_};
        main_pd = commit_batches(&c, in, nullptr, fri_, ride ? &rider : nullptr);
    }
    if (!prep_nat.empty()) {
        if (!prep_hit && prep_cache_enabled_) prep_key_ = prep_key;  // off: the key stays empty, the next proof recomputes
        prep_pd = prep_pd_cache_.get();
        ch.observe_digest(prep_pd->tree.root);
    }
    ch.observe_digest(main_pd->tree.root);
    tm.commit_main = ms_since(t0);

    // ---------------- permutation traces (lib.rs:227-261)
    t0 = Clock::now();
    Ext5 rnd[3];
    for (int i = 0; i < 3; i++) rnd[i] = ch.sample_ext();
    std::vector<std::vector<Ext5>> bus_alphas(NC), betas(NC);
    std::vector<DMat> perm_nat(NC);
    std::vector<Ext5> cumulative_sums(NC);
    uint32_t* cs_host = nullptr;  // 5 canonical words per chip, valid after the next stream synchronisation
    DBuf cs_keep;
    {
        std::vector<uint32_t> pool;
        std::vector<size_t> off(NC);
        for (size_t i = 0; i < NC; i++) {
            auto& its = machine_.airs[i].interactions;
            size_t maxf = 0;
            for (auto& it : its) {
                const Ext5& r = it.is_local() ? rnd[0] : rnd[1];  // generate_rlc_elements (chip.rs:291-331)
                bus_alphas[i].push_back(r.pow((uint64_t)it.bus_index + 1));
                maxf = std::max(maxf, it.fields.size());
            }
            Ext5 bp = Ext5::one();
            for (size_t j = 0; j < maxf; j++) { betas[i].push_back(bp); bp *= rnd[2]; }
            off[i] = pool.size();
            for (auto& a : bus_alphas[i]) put_ext(pool, a);
            for (auto& b : betas[i]) put_ext(pool, b);
        }
        pool.push_back(0);
        DBuf pool_dev(&c, pool);
        GatherList gl;
        std::vector<DBuf> scratch;
        Section perm_section(&c);
        for (size_t i = 0; i < NC; i++) {
            uint32_t M = (uint32_t)machine_.airs[i].interactions.size();
            uint64_t n = main_nat[i]->height;
            perm_nat[i] = DMat(&c, n, 5 * (M + 1));
            vk::DMatView pv{nullptr, 0, 0, 0};
            if (prep_slot[i] >= 0) pv = prep_nat[prep_slot[i]].view();
            scratch.emplace_back(&c, (size_t)vk::perm_scratch_words(n));
            vk::launch_perm_trace(c.stream_for(i, n), main_nat[i]->view(), pv, iw_dev_[i].data, pool_dev.data + off[i], M, perm_nat[i].view(), scratch.back().data, fri_.interpret_air ? -2 : machine_.airs[i].native_chip);
            gl.add(perm_nat[i].data + (uint64_t)(5 * M) * n + (n - 1), n, 5, 0);  // cumulative sum = last row, last column (lib.rs:247-250)
        }
        perm_section.join();
        c.check_launch("perm trace");
        DBuf gd(&c, gl.desc), gout(&c, gl.next_dst);
        vk::launch_gather(c.stream, gd.data, gl.n(), gout.data);
        // asynchronous copy into pinned memory; it has landed once the permutation commit below has synchronised on its root
        if (gl.next_dst > CS_PINNED_WORDS) throw std::invalid_argument("prove: too many chips");
        cs_host = cs_pinned_;  // a pinned area of its own: the generic staging buffer is reused by the root download below
        VG_HIP_CHECK(hipMemcpyAsync(cs_host, gout.data, gl.next_dst * 4, hipMemcpyDeviceToHost, c.stream));
        cs_keep = std::move(gout);
    }
    auto read_cumulative_sums = [&]() { for (size_t i = 0; i < NC; i++) cumulative_sums[i] = ext_from_canonical(cs_host + 5 * i); };
    if (dbg && dbg->check_constraints) { c.sync(); read_cumulative_sums(); }  // the debug check needs them before the commit
    tm.perm = ms_since(t0);
    if (dbg && dbg->keep_matrices) {
        dbg->perm_traces.resize(NC);
        for (size_t i = 0; i < NC; i++) {
            auto& pm = perm_nat[i];
            DBuf tmp(&c, (size_t)(pm.height * pm.width));
            vk::launch_export_rows(c.stream, pm.view(), 0, pm.height, tmp.data);
            dbg->perm_traces[i].resize(pm.height * pm.width);
            c.download(dbg->perm_traces[i].data(), tmp.data, dbg->perm_traces[i].size() * 4);
        }
    }
    if (dbg && dbg->check_constraints) {
        // check_constraints per chip (basic/src/lib.rs:270-372) and check_cumulative_sums (:373-375), as debug builds of the
        // reference do before committing: every AIR / permutation constraint on the trace itself.  The fold of the permutation
        // constraints uses powers of the third permutation challenge (any nonzero element serves a check).
        std::vector<uint32_t> pool;
        std::vector<size_t> off(NC);
        std::vector<uint32_t> Ks(NC);
        for (size_t i = 0; i < NC; i++) {
            auto& air = machine_.airs[i];
            uint32_t M = (uint32_t)air.interactions.size(), K = air.program.num_asserts + M + 3;
            Ks[i] = K;
            off[i] = pool.size();
            Ext5 pw = Ext5::one();
            std::vector<Ext5> ap(K);
            for (uint32_t k = 0; k < K; k++) { ap[K - 1 - k] = pw; pw *= rnd[2]; }
            for (auto& e : ap) put_ext(pool, e);
            for (auto& e : bus_alphas[i]) put_ext(pool, e);
            for (auto& e : betas[i]) put_ext(pool, e);
            put_ext(pool, cumulative_sums[i]);
        }
        DBuf pool_dev(&c, pool);
        std::vector<uint32_t> init(2 * NC, 0xffffffffu);
        DBuf bad_dev(&c, init);
        for (size_t i = 0; i < NC; i++) {
            auto& air = machine_.airs[i];
            vk::QuotientArgs a{};
            a.main_lde = main_nat[i]->view();
            a.perm_lde = perm_nat[i].view();
            a.prep_lde = prep_slot[i] >= 0 ? prep_nat[prep_slot[i]].view() : vk::DMatView{nullptr, 0, 0, 0};
            a.log_n = (int)log_deg[i];
            a.prog = (const vair::Instr*)prog_dev_[i].data;
            a.n_instrs = (uint32_t)air.program.instrs.size();
            a.n_regs = air.program.num_regs;
            a.n_air_asserts = air.program.num_asserts;
            a.iw = iw_dev_[i].data;
            a.consts = pool_dev.data + off[i];
            a.K = Ks[i];
            a.native_chip = fri_.interpret_air ? vk::QuotientArgs::INTERPRET : air.native_chip;
            vk::launch_check_constraints(c.stream, a, reinterpret_cast<unsigned long long*>(bad_dev.data) + i);
        }
        c.check_launch("check_constraints");
        std::vector<uint32_t> bad(2 * NC);
        c.download(bad.data(), bad_dev.data, bad.size() * 4);
        for (size_t i = 0; i < NC; i++) {
            const uint64_t v = ((uint64_t)bad[2 * i + 1] << 32) | bad[2 * i];
            if (v == ~0ull) continue;
            const uint64_t row = v >> 16;
            const uint32_t code = (uint32_t)(v & 0xffff);
            std::string what = code == 0xFFFE ? "a permutation (bus) constraint" : code == 0xFFFD ? "an AIR constraint" : "AIR constraint " + std::to_string(code);
            throw std::invalid_argument("check_constraints: chip " + machine_.airs[i].name + ": " + what + " is not satisfied on row " + std::to_string(row));
        }
        Ext5 total = Ext5::zero();
        for (auto& s5 : cumulative_sums) total += s5;
        if (!total.is_zero()) throw std::invalid_argument("check_cumulative_sums: the chips' cumulative sums do not cancel (a bus is unbalanced)");
    }
    t0 = Clock::now();
    std::unique_ptr<ProverData> perm_pd;
    {
        std::vector<CommitInput> in;
        for (auto& m : perm_nat) in.push_back({&m, false, false});
        perm_pd = commit_batches(&c, in, nullptr, fri_);
    }
    ch.observe_digest(perm_pd->tree.root);
    read_cumulative_sums();  // the root download synchronised the stream: the asynchronous copy has landed
    perm_nat.clear();
    main_own.clear();
    tm.commit_perm = ms_since(t0);

    // ---------------- quotients (lib.rs:263-599)
    t0 = Clock::now();
    const Ext5 alpha = ch.sample_ext();
    std::vector<DMat> quot(NC);
    std::vector<char> quot_is_natural(NC, 0);
    std::vector<Fp> quot_shifts(NC);
    {
        std::vector<uint32_t> pool;
        std::vector<size_t> off(NC);
        std::vector<uint32_t> Ks(NC);
        // the powers of alpha ONCE for all chips (this loop sits between two kernels of a lone proof with the GPU idle: the per-chip
        // recomputation cost 4x the products)
        uint32_t maxK = 0;
        for (size_t i = 0; i < NC; i++) maxK = std::max<uint32_t>(maxK, machine_.airs[i].program.num_asserts + (uint32_t)machine_.airs[i].interactions.size() + 3);
        std::vector<Ext5> apow_all(maxK);
        { Ext5 p = Ext5::one(); for (auto& e : apow_all) { e = p; p *= alpha; } }
        for (size_t i = 0; i < NC; i++) {
            auto& air = machine_.airs[i];
            uint32_t M = (uint32_t)air.interactions.size(), K = air.program.num_asserts + M + 3;
            Ks[i] = K;
            off[i] = pool.size();
            for (uint32_t k = 0; k < K; k++) put_ext(pool, apow_all[K - 1 - k]);  // constraint k is scaled by alpha^(K-1-k)
            for (auto& e : bus_alphas[i]) put_ext(pool, e);
            for (auto& e : betas[i]) put_ext(pool, e);
            put_ext(pool, cumulative_sums[i]);
        }
        DBuf pool_dev(&c, pool);
        Section quotient_section(&c);
        for (size_t i = 0; i < NC; i++) {
            auto& air = machine_.airs[i];
            vk::QuotientArgs a{};
            fill_quotient_args(a, (int)i, main_pd->ldes[i].view(), perm_pd->ldes[i].view(),
                               prep_slot[i] >= 0 ? prep_pd->ldes[prep_slot[i]].view() : vk::DMatView{nullptr, 0, 0, 0}, log_deg[i], pool_dev.data + off[i]);
            a.K = Ks[i];
            quot[i] = DMat(&c, 1ull << log_deg[i], 5ull << air.log_quotient_degree);
            a.out = quot[i].view();
            // chunk rows in natural order (staged through LDS into 64-byte runs): the quotient round then takes the fused LDE like the other
            // two; higher quotient degrees (k_quotient_general) keep the bit-reversed positions and the unfused passes.
            a.out_natural = air.log_quotient_degree == 1 ? 1 : 0;
            quot_is_natural[i] = a.out_natural != 0;
            vk::launch_quotient(c.stream_for(i, 1ull << log_deg[i]), a, c.tables);
            quot_shifts[i] = s.exp_power_of_2(air.log_quotient_degree);  // lib.rs:593-596
        }
        quotient_section.join();
        c.check_launch("quotient");
    }
    tm.quotient = ms_since(t0);
    if (dbg && dbg->keep_matrices) {
        dbg->quotient_chunks.resize(NC);
        for (size_t i = 0; i < NC; i++) {
            auto& q = quot[i];
            DMat nat(&c, q.height, q.width);
            if (q.height > 1 && !quot_is_natural[i]) vk::launch_bitrev_rows(c.stream, q.view(), nat.view());
            else VG_HIP_CHECK(hipMemcpyAsync(nat.data, q.data, q.height * q.width * 4, hipMemcpyDeviceToDevice, c.stream));
            DBuf tmp(&c, (size_t)(q.height * q.width));
            vk::launch_export_rows(c.stream, nat.view(), 0, q.height, tmp.data);
            dbg->quotient_chunks[i].resize(q.height * q.width);
            c.download(dbg->quotient_chunks[i].data(), tmp.data, dbg->quotient_chunks[i].size() * 4);
        }
    }
    t0 = Clock::now();
    std::unique_ptr<ProverData> quot_pd;
    {
        std::vector<CommitInput> in;
        for (size_t i = 0; i < quot.size(); i++) in.push_back({&quot[i], !quot_is_natural[i], !quot_is_natural[i]});
        quot_pd = commit_batches(&c, in, &quot_shifts, fri_);
    }
    ch.observe_digest(quot_pd->tree.root);
    quot.clear();
    tm.commit_quotient = ms_since(t0);

    // ---------------- opening (lib.rs:606-619): pcs.open_multi_batches over the three rounds
    const Ext5 zeta = ch.sample_ext();
    std::vector<OpenRound> rounds(3);
    rounds[0].pd = main_pd.get(); rounds[1].pd = perm_pd.get(); rounds[2].pd = quot_pd.get();
    for (size_t i = 0; i < NC; i++) {
        Fp g = vg::two_adic_generator(log_deg[i]);
        rounds[0].points.push_back({zeta, zeta * g});
        rounds[1].points.push_back({zeta, zeta * g});
        rounds[2].points.push_back({zeta.exp_power_of_2(machine_.airs[i].log_quotient_degree)});
    }
    PcsOpening opening = open_multi_batches(rounds, ch);
    tm.open_values = opening.ms_values; tm.open_reduce = opening.ms_reduce; tm.fri = opening.ms_fri;
    t0 = Clock::now();
    const auto& opened = opening.opened;

    // ---------------- assemble MachineProof (flat "VPF1" words; machine/src/proof.rs:13-44, App. B12)
    std::vector<uint32_t> pw;
    {
        size_t chip_words = 26;
        for (size_t i = 0; i < NC; i++) chip_words += 11 + 5 * (2 * machine_.airs[i].width + 10 * (machine_.airs[i].interactions.size() + 1) + 10);
        pw.reserve(chip_words + opening.proof_words.size());
    }
    pw.push_back(PROOF_MAGIC);
    pw.push_back((uint32_t)NC);
    for (int k = 0; k < 8; k++) pw.push_back(main_pd->tree.root[k]);
    for (int k = 0; k < 8; k++) pw.push_back(perm_pd->tree.root[k]);
    for (int k = 0; k < 8; k++) pw.push_back(quot_pd->tree.root[k]);
    auto put_vec = [&](const std::vector<Ext5>& v) {
        pw.push_back((uint32_t)v.size());
        for (auto& e : v) { uint32_t w5[5]; ext_to_canonical(e, w5); pw.insert(pw.end(), w5, w5 + 5); }
    };
    for (size_t i = 0; i < NC; i++) {
        pw.push_back(log_deg[i]);
        put_vec(opened[0][i][0]); put_vec(opened[0][i][1]);
        put_vec(opened[1][i][0]); put_vec(opened[1][i][1]);
        put_vec(opened[2][i][0]);
        uint32_t w5[5]; ext_to_canonical(cumulative_sums[i], w5); pw.insert(pw.end(), w5, w5 + 5);
    }
    pw.insert(pw.end(), opening.proof_words.begin(), opening.proof_words.end());
    tm.queries = opening.ms_queries + ms_since(t0);
    c.profiler.collect();
    tm.total = ms_since(t_total);
    if (times) *times = tm;
    if (dbg) {
        if (prep_pd) memcpy(dbg->prep_root, prep_pd->tree.root, 32); else memset(dbg->prep_root, 0, 32);
        for (int i = 0; i < 3; i++) ext_to_canonical(rnd[i], dbg->perm_challenges + 5 * i);
        ext_to_canonical(alpha, dbg->alpha);
        ext_to_canonical(zeta, dbg->zeta);
    }
    return pw;
}

// k_reduce_openings' per-point table: g(X) = prod_{i=1..4} (X - frob^i z) = X^4 + g3 X^3 + .. + g0 over Ext5 and m(X) = (X - z) g(X), the minimal
// polynomial of z over the base field (X^5 + m4 X^4 + .. + m0): 1/(z - x) = -g(x)/m(x) for x in the base field.  Words: [m0..m4][g0]..[g3].
void put_min_poly(std::vector<uint32_t>& w, const Ext5& z) {
    Ext5 g[5] = {Ext5::one(), Ext5::zero(), Ext5::zero(), Ext5::zero(), Ext5::zero()};  // coefficients, low first; degree grows to 4
    Ext5 conj = z;
    for (int i = 1; i <= 4; i++) {
        conj = conj.frobenius();
        for (int k = i; k >= 1; k--) g[k] = g[k - 1] - conj * g[k];  // g *= (X - conj)
        g[0] = Ext5::zero() - conj * g[0];
    }
    // m = (X - z) * g: m_k = g_{k-1} - z * g_k (g_5 = 0, g_{-1} = 0); every coefficient must land in the base field
    Ext5 m[5];
    for (int k = 0; k < 5; k++) m[k] = (k ? g[k - 1] : Ext5::zero()) - z * g[k];
    for (int k = 0; k < 5; k++) {
        for (int l = 1; l < 5; l++) if (!m[k].c[l].is_zero()) throw std::logic_error("open: minimal polynomial coefficient outside the base field");
        w.push_back(m[k].c[0].v);
    }
    for (int k = 0; k < 4; k++) put_ext(w, g[k]);
}

// pcs.open_multi_batches (basic/src/lib.rs:611-619; Plonky3 TwoAdicFriPcs): opens every matrix of every
// round at its points, reduces the openings per LDE height, runs the FRI commit phase, grinds, answers the queries.  `ch` is the
// caller's transcript (the reference passes `&mut challenger`): it leaves advanced exactly as the reference leaves it.
// Any number of rounds, matrices, columns and points per matrix: wide matrices and long point lists go through the kernels in
// chunks (k_col_dot: <= 2 points and <= 1024 / (5 points) columns per launch; k_reduce_openings: <= 4 distinct points per launch).
PcsOpening Prover::open_multi_batches(const std::vector<OpenRound>& rounds, Challenger& ch) {
    DeviceCtx& c = *ctx_;
    c.activate();
    const Fp s = Fp::from_canonical(vg::GENERATOR);
    const unsigned lb = fri_.log_blowup;
    const size_t NR = rounds.size();
    if (!NR) throw std::invalid_argument("open: no rounds");
    for (auto& r : rounds) {
        if (!r.pd || r.pd->ldes.empty()) throw std::invalid_argument("open: empty round");
        if (r.points.size() != r.pd->ldes.size()) throw std::invalid_argument("open: one point list per committed matrix is required");
    }
    PcsOpening res;
    auto t0 = Clock::now();
    // The opening's first transcript step (TwoAdicFriPcs::open samples alpha before anything else, App. B9; the opened values are not
    // observed): taken HERE, before the kernels of the opened values are even enqueued, so that everything of the reduced openings that does
    // not depend on the values themselves — the powers of alpha, the offsets alpha^num_reduced, the point slots — is computed by the host
    // WHILE the GPU evaluates the opened values instead of after their download with the GPU idle (a 0.13 ms gap of a lone proof).
    const Ext5 alpha_b = ch.sample_ext();
    struct MatEntry { const DMat* lde; std::vector<std::tuple<uint32_t, Ext5, Ext5>> pts; size_t r, i; };  // (slot, alpha^offset, Y)
    struct Group { std::vector<Ext5> zs; std::map<PointKey, uint32_t> slot; std::vector<MatEntry> mats; uint64_t num_reduced = 0; Ext5 offset = Ext5::one(); };  // offset = alpha^num_reduced
    std::map<unsigned, Group> groups;
    std::vector<Ext5> apow;
    auto prepare_groups = [&]() {  // everything but Y
        size_t max_width = 0;
        for (auto& r : rounds) for (auto& l : r.pd->ldes) max_width = std::max<size_t>(max_width, l.width);
        apow.resize(max_width);
        { Ext5 p = Ext5::one(); for (auto& a : apow) { a = p; p *= alpha_b; } }
        for (size_t r = 0; r < NR; r++)
            for (size_t i = 0; i < rounds[r].pd->ldes.size(); i++) {
                const DMat& lde = rounds[r].pd->ldes[i];
                unsigned lh = vg::log2_strict_u64(lde.height);
                Group& g = groups[lh];
                MatEntry me{&lde, {}, r, i};
                for (size_t p = 0; p < rounds[r].points[i].size(); p++) {
                    const Ext5& z = rounds[r].points[i][p];
                    auto key = key_of(z);
                    auto it = g.slot.find(key);
                    uint32_t slot;
                    if (it == g.slot.end()) { slot = (uint32_t)g.zs.size(); g.slot[key] = slot; g.zs.push_back(z); } else slot = it->second;
                    me.pts.emplace_back(slot, g.offset, Ext5::zero());
                    g.num_reduced += lde.width;
                    if (lde.width) g.offset *= apow[lde.width - 1] * alpha_b;  // alpha^width: two products instead of a square-and-multiply per (matrix, point)
                }
                g.mats.push_back(std::move(me));
            }
    };

    // ---- opened values p_j(z) by barycentric evaluation over the first n rows of each bit-reversed LDE (App. B9)
    res.opened.resize(NR);
    struct Job { size_t r, i; int p0, np; uint64_t c0, cw; size_t w[2]; size_t scale_off, out_off; };
    std::vector<Job> jobs;
    size_t out_words = 0;
    // the reduced openings' descriptor pool (built and UPLOADED before the opened-value kernels are enqueued, see below)
    struct RLaunch { unsigned lh; size_t off; bool accumulate; uint64_t total_width; int n_points; bool vec_ok; };
    std::vector<RLaunch> launches;
    DBuf reduce_pool_dev;
    size_t apw_at = 0, ydesc_at = 0, yoff_at = 0, n_y_slots = 0;
    DBuf out_dev;             // the opened values, canonical words, as k_col_dot_finish leaves them
    uint32_t* out_host = nullptr;  // their pinned landing area: read after the opening's one synchronisation (the end of the FRI commit phase)
    {
        std::map<std::pair<unsigned, PointKey>, size_t> wkey;  // distinct (log_n, point) -> weight vector
        std::map<unsigned, Fp> den_inv;                          // log_n -> 1 / (n s^(n-1))
        std::map<PointKey, std::vector<Ext5>> zpow2;             // point -> z^(2^k), k = 0 ..
        struct WEntry { unsigned log_n; Ext5 z; size_t pool_off; DBuf buf; };
        std::vector<WEntry> wlist;
        std::vector<uint32_t> pool;
        for (size_t r = 0; r < NR; r++) {
            res.opened[r].resize(rounds[r].pd->ldes.size());
            for (size_t i = 0; i < rounds[r].pd->ldes.size(); i++) {
                const DMat& lde = rounds[r].pd->ldes[i];
                const unsigned ln = vg::log2_strict_u64(lde.height) - lb;
                const uint64_t n = 1ull << ln;
                const auto& pts = rounds[r].points[i];
                res.opened[r][i].assign(pts.size(), std::vector<Ext5>(lde.width));
                for (size_t p0 = 0; p0 < pts.size(); p0 += 2) {
                    const int np = (int)std::min<size_t>(2, pts.size() - p0);
                    size_t widx[2] = {0, 0};
                    const size_t scale_off = pool.size();
                    for (int p = 0; p < np; p++) {
                        const Ext5& z = pts[p0 + p];
                        // scale = (z^n - s^n) / (n s^(n-1)): the host sits between the transcript and the first launch here with the GPU idle, so
                        // 1 / (n s^(n-1)) is kept per height and the squarings z^(2^k) per distinct point (zeta serves every chip)
                        auto dit = den_inv.find(ln);
                        if (dit == den_inv.end()) dit = den_inv.emplace(ln, (Fp::from_canonical((uint32_t)(n % vg::P)) * s.pow(n - 1)).inv()).first;
                        std::vector<Ext5>& zp = zpow2[key_of(z)];
                        if (zp.empty()) zp.push_back(z);
                        while (zp.size() <= ln) zp.push_back(zp.back().square());
                        Ext5 zer = zp[ln] - s.exp_power_of_2(ln);
                        put_ext(pool, zer * dit->second);
                    }
                    for (int p = 0; p < np; p++) {
                        const Ext5& z = pts[p0 + p];
                        auto key = std::make_pair(ln, key_of(z));
                        auto it = wkey.find(key);
                        if (it == wkey.end()) { wkey[key] = wlist.size(); widx[p] = wlist.size(); wlist.push_back(WEntry{ln, z, 0, DBuf()}); }
                        else widx[p] = it->second;
                    }
                    const uint64_t max_cols = vk::col_dot_max_columns(np);
                    for (uint64_t c0 = 0; c0 < lde.width; c0 += max_cols) {
                        const uint64_t cw = std::min<uint64_t>(max_cols, lde.width - c0);
                        jobs.push_back(Job{r, i, (int)p0, np, c0, cw, {widx[0], widx[np - 1]}, scale_off, out_words});
                        out_words += cw * np * 5;
                    }
                }
            }
        }
        for (auto& w : wlist) { w.pool_off = pool.size(); put_min_poly(pool, w.z); }
        pool.push_back(0);
        DBuf pool_dev(&c, pool);
        {   // every weight vector of the opening in one launch
            std::vector<uint32_t> jobs;
            uint32_t blocks = 0;
            double rows = 0;
            for (auto& w : wlist) {
                const uint64_t n = 1ull << w.log_n;
                w.buf = DBuf(&c, (size_t)vk::bary_buffer_words(n));
                jobs.push_back(blocks); jobs.push_back(0);
                put_u64(jobs, n);
                put_ptr(jobs, pool_dev.data + w.pool_off);
                put_ptr(jobs, w.buf.data);
                put_ptr(jobs, vk::bary_weights_has_image(n) ? w.buf.data + 5 * n : nullptr);
                blocks += vk::bary_weights_blocks(n);
                rows += (double)n;
            }
            if (!wlist.empty()) {
                DBuf jobs_dev(&c, jobs);
                vk::launch_bary_weights_batch(c.stream, jobs_dev.data, (uint32_t)wlist.size(), blocks, rows, s, c.tables);
                // jobs_dev returns to the pool while the launch is queued: safe, the pool hands a block only to work enqueued later on this stream
            }
        }
        // The descriptors of the reduced openings need nothing the kernels below compute (the one value-dependent part, Y, is formed on the device
        // by k_open_y): they are built and uploaded HERE, while the weight kernel runs and BEFORE the column-dot launches fork onto the auxiliary
        // stream — a copy enqueued behind that join waits for the other queue's signal through the runtime (0.08 ms of a lone proof with the GPU idle).
        prepare_groups();
        {
            // Y = sum_col alpha^col y_col per (matrix, point) is the one part that needs the VALUES: its slot in the descriptors stays empty here and
            // k_open_y fills it on the device (y_slots: where, and from which col_dot outputs)
            struct YSlot { size_t pool_off, r, i, p; };
            std::vector<YSlot> y_slots;
            // one descriptor per (height, chunk of <= MAX_OPEN_POINTS distinct points); later chunks accumulate into the vector
            std::vector<uint32_t> pool;  // (a pool of its own: the barycentric pool above is already on its way)
            for (auto& kv : groups) {
                Group& g = kv.second;
                for (uint32_t s0 = 0; s0 < g.zs.size() || s0 == 0; s0 += vk::MAX_OPEN_POINTS_PER_LAUNCH) {
                    const uint32_t s1 = std::min<uint32_t>((uint32_t)g.zs.size(), s0 + vk::MAX_OPEN_POINTS_PER_LAUNCH);
                    std::vector<const MatEntry*> live;
                    size_t max_w = 0;
                    uint64_t total_width = 0;
                    bool vec_ok = true;
                    for (auto& me : g.mats) {
                        bool any = false;
                        for (auto& t : me.pts) any |= std::get<0>(t) >= s0 && std::get<0>(t) < s1;
                        if (any) { live.push_back(&me); max_w = std::max<size_t>(max_w, me.lde->width); total_width += me.lde->width; vec_ok &= vk::reduce_vec_ok(me.lde->data, me.lde->height); }
                    }
                    launches.push_back({kv.first, pool.size(), s0 != 0, total_width, (int)(s1 - s0), vec_ok});
                    pool.push_back((uint32_t)live.size());
                    pool.push_back(s1 - s0);
                    pool.push_back((uint32_t)max_w);
                    for (uint32_t q = s0; q < s1; q++) put_min_poly(pool, g.zs[q]);
                    for (size_t col = 0; col < max_w; col++) put_ext(pool, apow[col]);
                    for (auto* me : live) {
                        put_ptr(pool, me->lde->data);
                        put_u64(pool, me->lde->height);
                        pool.push_back((uint32_t)me->lde->width);
                        uint32_t cnt = 0;
                        for (auto& t : me->pts) cnt += std::get<0>(t) >= s0 && std::get<0>(t) < s1;
                        pool.push_back(cnt);
                        for (size_t pi = 0; pi < me->pts.size(); pi++) {
                            auto& t = me->pts[pi];
                            if (std::get<0>(t) >= s0 && std::get<0>(t) < s1) {
                                pool.push_back(std::get<0>(t) - s0);
                                put_ext(pool, std::get<1>(t));
                                y_slots.push_back({pool.size(), me->r, me->i, pi});
                                put_ext(pool, Ext5::zero());  // Y: written by k_open_y
                            }
                        }
                    }
                    if (g.zs.size() <= s1) break;
                }
            }
            // Y of every (matrix, point), on the device: its tables ride in the same upload as the reduce descriptors (one host-to-device copy in the
            // dependent chain, not four)
            apw_at = pool.size();
            for (auto& a : apow) put_ext(pool, a);
            std::vector<uint32_t> ydesc, yoff;
            for (auto& ys : y_slots) {
                yoff.push_back((uint32_t)ydesc.size());
                ydesc.push_back((uint32_t)ys.pool_off);
                const size_t n_seg_at = ydesc.size();
                ydesc.push_back(0);
                for (auto& j : jobs)
                    if (j.r == ys.r && j.i == ys.i && (size_t)j.p0 <= ys.p && ys.p < (size_t)(j.p0 + j.np)) {
                        ydesc.push_back((uint32_t)j.out_off); ydesc.push_back((uint32_t)j.np); ydesc.push_back((uint32_t)(ys.p - (size_t)j.p0));
                        ydesc.push_back((uint32_t)j.c0); ydesc.push_back((uint32_t)j.cw);
                        ydesc[n_seg_at]++;
                    }
            }
            ydesc_at = pool.size();
            pool.insert(pool.end(), ydesc.begin(), ydesc.end());
            yoff_at = pool.size();
            pool.insert(pool.end(), yoff.begin(), yoff.end());
            pool.push_back(0);
            n_y_slots = y_slots.size();
            reduce_pool_dev = DBuf(&c, pool);
        }
        out_dev = DBuf(&c, out_words + 4);
        std::vector<DBuf> partials;
        // the finishes of all column-dot launches in ONE launch behind the join (a finish launch behind every column-dot launch: +0.15 ms per lone proof,
        // profiles/r05_ab_latency.txt; the sharded prover's single-stream openings keep that form): the job table is built and uploaded before the launches
        // fork (a copy enqueued behind the join would wait for the other queue's signal)
        std::vector<uint32_t> fin_jobs;
        uint32_t fin_blocks = 0;
        for (auto& j : jobs) {
            const DMat& lde = rounds[j.r].pd->ldes[j.i];
            const uint64_t n = lde.height >> lb;
            partials.emplace_back(&c, (size_t)(vk::col_dot_slots(n) * j.cw * j.np * 5));
            fin_blocks = vk::col_dot_finish_job(fin_jobs, fin_blocks, n, j.cw, j.np, partials.back().data, pool_dev.data + j.scale_off, out_dev.data + j.out_off);
        }
        DBuf fin_dev;
        if (!fin_jobs.empty()) fin_dev = DBuf(&c, fin_jobs);
        Section open_section(&c);
        size_t job_idx = 0;
        for (auto& j : jobs) {
            const DMat& lde = rounds[j.r].pd->ldes[j.i];
            const uint64_t n = lde.height >> lb;
            vk::DMatView sub{lde.data + j.c0 * lde.height, lde.height, j.cw, lde.height};
            vk::launch_col_dot(c.stream_for(job_idx, n), sub, n, j.np, wlist[j.w[0]].buf.data, wlist[j.w[1]].buf.data, partials[job_idx].data,
                               pool_dev.data + j.scale_off, out_dev.data + j.out_off, false);
            job_idx++;
        }
        open_section.join();
        if (!fin_jobs.empty()) vk::launch_col_dot_finish_batch(c.stream, fin_dev.data, (uint32_t)jobs.size(), fin_blocks);
        c.check_launch("opened values");
        // the values travel to the host BESIDE the reduced openings and the FRI commit phase (no synchronisation here: k_open_y below reads them
        // on the device); a pinned area of this prover's own, the generic staging buffer is reused by the downloads that follow
        {
            std::lock_guard<std::recursive_mutex> lk(c.host_mu);
            if (open_pinned_words_ < out_words + 4) {
                if (open_pinned_) { VG_HIP_CHECK(hipStreamSynchronize(c.stream)); VG_HIP_CHECK(hipHostFree(open_pinned_)); open_pinned_ = nullptr; open_pinned_words_ = 0; }
                VG_HIP_CHECK(hipHostMalloc((void**)&open_pinned_, (out_words + 4) * 4));
                open_pinned_words_ = out_words + 4;
            }
            out_host = open_pinned_;
        }
    }
    res.ms_values = ms_since(t0);

    // ---- reduced openings per LDE height (App. B9): ro[x] += alpha^offset * sum_j alpha^j (ys_j - row_j(x)) / (z - x)
    t0 = Clock::now();
    std::map<unsigned, DBuf> ro;  // log_height -> pair-layout vector
    unsigned log_max = 0;
    {
        DBuf& pool_dev = reduce_pool_dev;
        vk::launch_open_y(c.stream, out_dev.data, pool_dev.data + apw_at, pool_dev.data + ydesc_at, pool_dev.data + yoff_at, (uint32_t)n_y_slots, pool_dev.data);
        for (auto& kv : groups) { ro[kv.first] = DBuf(&c, (size_t)(5ull << kv.first)); log_max = std::max(log_max, kv.first); }
        Section reduce_section(&c);
        std::map<unsigned, hipStream_t> stream_of;  // chunks of one height must stay on one stream (they accumulate in order)
        size_t grp_idx = 0;
        // (the second tallest height on the auxiliary stream, beside the tallest one's launch instead of behind it, was measured: no gain — profiles/r05_ab_latency.txt)
        for (auto& l : launches) {
            const uint64_t L = 1ull << l.lh;
            if (!stream_of.count(l.lh)) stream_of[l.lh] = c.stream_for(grp_idx++, L);
            vk::launch_reduce_openings(stream_of[l.lh], pool_dev.data + l.off, L, s, c.tables, ro[l.lh].data, l.total_width, l.accumulate, l.n_points, l.vec_ok);
        }
        reduce_section.join();
        c.check_launch("reduce openings");
        // the opened values travel to the host from here on, behind the last kernel that reads them and beside the FRI commit phase
        if (out_words) VG_HIP_CHECK(hipMemcpyAsync(out_host, out_dev.data, out_words * 4, hipMemcpyDeviceToHost, c.stream));
    }
    res.ms_reduce = ms_since(t0);
    if (log_max < lb + 0u) throw std::invalid_argument("open: nothing to open");

    // ---- FRI commit phase (App. B10)
    t0 = Clock::now();
    std::vector<DBuf> layer_bufs;          // layer i vector (length 2^(log_max - i)), pair layout
    std::vector<DeviceTree> layer_trees;
    std::vector<std::array<uint32_t, 8>> commit_phase_commits;
    DBuf cur = std::move(ro[log_max]);
    ro.erase(log_max);
    // The transcript moves to the device for the commit phase: per layer k_fri_challenge observes the root where
    // the tree left it and samples beta, the fold reads beta from device memory — no host round trip inside the
    // chain of dependent layers.  Roots, the final values and the sponge state come back in one sync below.
    const unsigned n_layers = log_max - lb;
    std::vector<uint32_t> chw(vk::DEV_CHALLENGER_WORDS, 0);
    for (int i = 0; i < 16; i++) chw[i] = ch.state[i].v;
    for (size_t i = 0; i < ch.in.size(); i++) chw[16 + i] = ch.in[i].v;
    chw[32] = (uint32_t)ch.in.size();
    for (size_t i = 0; i < ch.out.size(); i++) chw[33 + i] = ch.out[i].v;
    chw[49] = (uint32_t)ch.out.size();
    DBuf ch_dev(&c, chw), betas_dev(&c, (size_t)(5 * n_layers + 8)), commits_dev(&c, (size_t)(8 * n_layers + 8));
    unsigned li = 0;
    for (unsigned lf = log_max; lf-- > lb; li++) {
        uint64_t L = 2ull << lf, half = L >> 1;  // current length 2^(lf+1)
        layer_trees.emplace_back();
        layer_trees.back().drop_bottom = true;  // (layers of >= 2^16 pairs)
        // the challenger step rides on the tree-top launch (its first wave, on the root it has just written): one launch and one gap less per layer
        const DeviceTree::TopChallenger step{pow_pos_.data, ch_dev.data, betas_dev.data + 5 * li, commits_dev.data + 8 * li};
        layer_trees.back().build(&c, {vk::DMatView{cur.data, half, 10, half}}, false, nullptr, &step);
        DBuf next(&c, (size_t)(5 * half));
        auto it = ro.find(lf);
        vk::launch_fri_fold(c.stream, cur.data, L, betas_dev.data + 5 * li, it != ro.end() ? it->second.data : nullptr, c.tables, next.data);
        if (it != ro.end()) ro.erase(it);  // folded in: back to the pool (the fold is enqueued; the pool orders reuse on this stream)
        layer_bufs.push_back(std::move(cur));
        cur = std::move(next);
    }
    c.check_launch("fri fold");
    // ---- the query openings' TEMPLATE, built and uploaded while the GPU runs the commit phase just enqueued: every descriptor of the gather
    // except the query index, and every word of the proof tail the host writes itself (the data-dependent ones — roots, final polynomial, witness —
    // as placeholders).  After the indices are sampled only they travel (k_gather_q).
    const size_t NQ = fri_.num_queries, NL = layer_trees.size();
    if (NQ > 256) throw std::invalid_argument("open: at most 256 queries");
    std::vector<uint32_t> templ, bottom_jobs;
    std::vector<std::pair<uint32_t, uint32_t>> fix;  // (position in the tail, value) of the words the host writes
    size_t fix_roots_at = 0, fix_final_at = 0;
    uint32_t tail_pos = 0;
    {
        size_t n_desc = 0, n_fix = 8 * NL + 16;
        for (size_t l = 0; l < NL; l++) n_desc += 1 + layer_trees[l].log_max_height;
        for (size_t r = 0; r < NR; r++) n_desc += rounds[r].pd->ldes.size() + rounds[r].pd->tree.log_max_height;
        n_desc *= NQ;
        n_fix += NQ * (2 + 2 * NL + NR * 2);
        for (size_t r = 0; r < NR; r++) n_fix += NQ * rounds[r].pd->ldes.size();
        templ.reserve(8 * n_desc + 8);
        fix.reserve(n_fix);
        uint32_t& pos = tail_pos;
        auto host_word = [&](uint32_t v) { fix.emplace_back(pos++, v); };
        auto gather = [&](const uint32_t* base, uint64_t stride, uint32_t count, uint32_t kind, size_t q, uint32_t mode, uint32_t shift, uint32_t aux) {
            const uint64_t pv = (uint64_t)base;
            const uint32_t d8[8] = {(uint32_t)pv, (uint32_t)(pv >> 32), (uint32_t)stride, (uint32_t)(stride >> 32), count | (kind << 28), pos, (uint32_t)q | (mode << 8) | (shift << 16), aux};
            templ.insert(templ.end(), d8, d8 + 8);
            pos += count;
        };
        // the path of leaf (index >> shift0) in tree t: level l's sibling is ((index >> shift0) >> l) ^ 1.  The layers a big tree did not keep
        // (DeviceTree::dropped) are recomputed from the committed rows by a job each (vk::launch_*_bottom_q) instead of gathered.
        auto gather_path = [&](const DeviceTree& t, size_t q, uint32_t shift0) {
            host_word(t.log_max_height);
            for (unsigned l = 0; l < t.log_max_height; l++) {
                if (l < t.dropped) {
                    const uint64_t pv = (uint64_t)t.leaf_ptr;
                    const uint32_t j8[8] = {(uint32_t)pv, (uint32_t)(pv >> 32), (uint32_t)t.leaf_stride, (uint32_t)(t.leaf_stride >> 32), (uint32_t)t.leaf_elems, pos, (uint32_t)q | (l << 8) | (shift0 << 16), 0};
                    bottom_jobs.insert(bottom_jobs.end(), j8, j8 + 8);
                    pos += 8;
                } else gather(t.layers[l].data, 1, 8, 1, q, 1, shift0 + l, 0);
            }
        };
        host_word((uint32_t)NL);
        fix_roots_at = fix.size();
        for (size_t l = 0; l < NL; l++) for (int w = 0; w < 8; w++) host_word(0);  // the commit-phase roots: filled in after the synchronisation
        host_word((uint32_t)NQ);
        for (size_t q = 0; q < NQ; q++) {
            host_word((uint32_t)NL);
            for (size_t l = 0; l < NL; l++) {
                // idx_i = index >> l, sibling value at pair idx_i >> 1 of the half the bit idx_i & 1 does NOT select (pair layout, 5 limbs `half` apart)
                const uint64_t half = 1ull << (log_max - 1 - l);
                gather(layer_bufs[l].data, half, 5, 0, q, 2, (uint32_t)l, (uint32_t)(5 * half));
                gather_path(layer_trees[l], q, (uint32_t)l + 1);
            }
        }
        fix_final_at = fix.size();
        for (int k = 0; k < 6; k++) host_word(0);  // final polynomial (5) and proof-of-work witness: filled in later
        host_word((uint32_t)NQ);
        for (size_t q = 0; q < NQ; q++) {
            host_word((uint32_t)NR);
            for (size_t r = 0; r < NR; r++) {
                const DeviceTree& t = rounds[r].pd->tree;
                const uint32_t shift_r = log_max - t.log_max_height;  // idx_r = index >> shift_r
                host_word((uint32_t)rounds[r].pd->ldes.size());
                for (auto& lde : rounds[r].pd->ldes) {
                    const unsigned lh = vg::log2_strict_u64(lde.height);
                    host_word((uint32_t)lde.width);
                    gather(lde.data, lde.height, (uint32_t)lde.width, 0, q, 0, shift_r + (t.log_max_height - lh), 0);
                }
                gather_path(t, q, shift_r);
            }
        }
    }
    const size_t tail_words = tail_pos;
    templ.push_back(0);
    const size_t n_bottom = bottom_jobs.size() / 8;
    bottom_jobs.push_back(0);
    DBuf bottom_dev(&c, bottom_jobs);
    DBuf templ_dev(&c, templ), gout(&c, tail_words + 4);  // the upload is enqueued behind the commit phase; nothing waits for it until the gather
    // `cur` now holds 2^lb values that must all be equal (a constant polynomial)
    std::vector<uint32_t> fin(5ull << lb), commits(8 * n_layers + 8);
    {   // sponge state, roots and final values: three copies into one pinned area, one synchronisation
        const size_t n0 = chw.size(), n1 = 8 * (size_t)n_layers, n2 = fin.size();
        uint32_t* pin = (uint32_t*)c.pinned_buffer((n0 + n1 + n2) * 4);
        VG_HIP_CHECK(hipMemcpyAsync(pin, ch_dev.data, n0 * 4, hipMemcpyDeviceToHost, c.stream));
        if (n1) VG_HIP_CHECK(hipMemcpyAsync(pin + n0, commits_dev.data, n1 * 4, hipMemcpyDeviceToHost, c.stream));
        VG_HIP_CHECK(hipMemcpyAsync(pin + n0 + n1, cur.data, n2 * 4, hipMemcpyDeviceToHost, c.stream));
        c.sync();
        memcpy(chw.data(), pin, n0 * 4);
        memcpy(commits.data(), pin + n0, n1 * 4);
        memcpy(fin.data(), pin + n0 + n1, n2 * 4);
    }
    // the opened values arrived long ago (same stream): into the result (one streaming copy out of the pinned area first)
    {
        std::vector<uint32_t> out(out_words + 4);
        if (out_words) memcpy(out.data(), out_host, out_words * 4);
        for (auto& j : jobs)
            for (uint64_t col = 0; col < j.cw; col++)
                for (int p = 0; p < j.np; p++) res.opened[j.r][j.i][j.p0 + p][j.c0 + col] = ext_from_canonical(&out[j.out_off + (col * j.np + p) * 5]);
    }
    out_dev = DBuf();
    for (int i = 0; i < 16; i++) ch.state[i] = Fp::raw(chw[i]);
    ch.in.clear();
    for (uint32_t i = 0; i < chw[32]; i++) ch.in.push_back(Fp::raw(chw[16 + i]));
    ch.out.clear();
    for (uint32_t i = 0; i < chw[49]; i++) ch.out.push_back(Fp::raw(chw[33 + i]));
    for (unsigned i = 0; i < n_layers; i++) {
        std::array<uint32_t, 8> root;
        memcpy(root.data(), commits.data() + 8 * i, 32);
        memcpy(layer_trees[i].root, root.data(), 32);
        commit_phase_commits.push_back(root);
    }
    // pair layout of a length-2^lb vector: (2^lb / 2) rows x 10 columns
    const uint64_t frows = (1ull << lb) >> 1;
    auto elem = [&](uint64_t idx) { Ext5 e; for (int k = 0; k < 5; k++) e.c[k] = Fp::raw(fin[((idx & 1) * 5 + k) * frows + (idx >> 1)]); return e; };
    Ext5 fp0 = elem(0);
    for (uint64_t q = 1; q < (1ull << lb); q++) if (elem(q) != fp0) throw std::runtime_error("fri: final polynomial is not constant");
    layer_bufs.push_back(std::move(cur));  // keep alive (not opened)
    ro.clear();
    uint32_t fpw[5];
    ext_to_canonical(fp0, fpw);
    if (fri_.observe_final_poly) ch.observe_ext(fp0);
    uint32_t pow_witness = grind(ch);
    res.ms_fri = ms_since(t0);

    // ---- queries.  The gather kernel writes every opened row, sibling value and Merkle path straight to its position in the proof
    // words (the TwoAdicFriPcsProof tail of the "VPF1" layout, App. B12); the host only patches in the length fields and the few
    // values it already holds (commit-phase roots, final polynomial, proof-of-work witness).  One flat descriptor array, sized up
    // front: building it is the only host work between the last FRI kernel and the gather.
    t0 = Clock::now();
    std::vector<uint32_t> indices(NQ + 1, 0);
    for (size_t q = 0; q < NQ; q++) indices[q] = (uint32_t)ch.sample_bits(log_max);
    for (size_t l = 0; l < NL; l++) for (int w = 0; w < 8; w++) fix[fix_roots_at + 8 * l + w].second = commit_phase_commits[l][w];
    for (int k = 0; k < 5; k++) fix[fix_final_at + k].second = fpw[k];
    fix[fix_final_at + 5].second = pow_witness;
    DBuf idx_dev(&c, indices);
    vk::launch_gather_q(c.stream, templ_dev.data, (templ.size() - 1) / 8, idx_dev.data, gout.data);
    if (c.hash_kind == 1) vk::launch_poseidon_bottom_q(c.stream, c.poseidon_tab, c.poseidon_sparse, bottom_dev.data, (uint32_t)n_bottom, idx_dev.data, gout.data);
    else vk::launch_keccak_bottom_q(c.stream, bottom_dev.data, (uint32_t)n_bottom, idx_dev.data, gout.data);
    c.check_launch("query gather");
    res.proof_words.resize(tail_words);
    c.download_small(res.proof_words.data(), gout.data, tail_words * 4);  // through pinned memory
    for (auto& f : fix) res.proof_words[f.first] = f.second;
    res.ms_queries = ms_since(t0);
    return res;
}

}  // namespace vhost
