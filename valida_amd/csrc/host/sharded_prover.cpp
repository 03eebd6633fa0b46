// ONE proof over W GPUs: Machine::prove (basic/src/lib.rs:147-675) with the large objects of the proof in row-range shards.
//
// The reference commits all chips' LDEs of a round under one Merkle tree and runs one serial transcript (basic/src/lib.rs:199,223,
// 258,599,601-619), so a sharded prover has to reproduce those very roots, opened values and FRI layers.  What makes that cheap:
//
//   Storage rows are in committed (bit-reversed) order, so the row range [r L/W, (r+1) L/W) of an LDE on s H_L is the sub-coset
//   s w_L^e H_{L/W}, e = bitrev_W(r), AGAIN in bit-reversed order: a rank's shard of every committed matrix is itself a bona fide
//   bit-reversed LDE — of the same polynomial, on a smaller coset — and the single-GPU kernels run on it unchanged with the shift
//   s w_L^e where they used s.
//
// Phases (W ranks, rank r; `big` = LDE of at least max(4 W, 2^log_min_sharded) rows, anything shorter is computed by every rank):
//   commit      columns of the big matrices are dealt round-robin (global column g -> rank g mod W); each rank extends its columns
//               (coset_lde), ALL-TO-ALL into row ranges, subtree over the range (shorter matrices with >= W rows contribute their row
//               range from the replicated LDE, the rest is injected above), ALL-GATHER of the W subtree roots, top log2 W levels on
//               every rank.  [first phase of §8(f)-4: sharded.hpp]
//   permutation every rank computes the (cheap, row-serial) permutation traces of all chips from the replicated main traces.
//   quotient    the successor x g_n of a shard's point lies in ANOTHER shard (sub-coset e + 2), at the same or the following natural
//               index: one HALO exchange hands every rank its successor shard of the three LDEs, the quotient kernel reads `next`
//               rows from it (QuotientArgs::main_nx ..), selectors and Z_H come out right from the shifted coset, and the rank's
//               quotient chunks are a row range of the chunk matrix.  ALL-TO-ALL (rows -> columns) feeds the quotient commit.
//   opening     p(z) by the barycentric formula over the WHOLE LDE domain: its sum splits over the shards (partial sums, ALL-GATHER,
//               added on the host); reduced openings are row-local; every FRI layer is folded inside the shards (the fold's 1 / x is
//               the local one times w^-e: folded into beta), its tree built like a commitment round; once a layer is shorter than the
//               threshold it is gathered and the remaining layers run on every rank.  The transcript is replicated.
//   queries     every rank lays out the same proof tail and fills in the rows / digests it holds; the tails are OR-ed (ALL-GATHER).
// Any log_blowup >= 1 (the quotient domain is the first L >> (log_blowup - 1) storage rows: the first W >> (log_blowup - 1) ranks' row ranges);
// log_quotient_degree = 1 (every chip of the reference); W a power of two.
#include "sharded_prover.hpp"
#include <algorithm>
#include <array>
#include <tuple>

namespace vhost {

namespace {

Ext5 sp_ext_from_canonical(const uint32_t* w) { Ext5 e; for (int k = 0; k < 5; k++) e.c[k] = Fp::from_canonical(w[k]); return e; }
void sp_ext_to_canonical(const Ext5& e, uint32_t* w) { for (int k = 0; k < 5; k++) w[k] = e.c[k].canonical(); }

struct SpPointKey {
    uint32_t w[5];
    bool operator<(const SpPointKey& o) const { return memcmp(w, o.w, 20) < 0; }
};
SpPointKey sp_key_of(const Ext5& e) { SpPointKey k; for (int i = 0; i < 5; i++) k.w[i] = e.c[i].v; return k; }

int owner_of(uint64_t global_col, int W) { return (int)(global_col % (uint64_t)W); }

// A Merkle tree held either whole (every rank) or as one subtree per rank + the replicated top levels
struct ShTree {
    bool sharded = false;
    DeviceTree tree;                          // sharded: over this rank's row range
    std::vector<std::vector<uint32_t>> top;   // sharded: top[t] = the W >> t digests t levels above the subtree roots (top[0] = the roots)
    unsigned log_total = 0, log_local = 0;    // log2 leaves of the whole tree / of the part `tree` covers
    uint32_t root[8] = {0};
};

struct ShMat {
    uint64_t L = 0, width = 0, col_base = 0;
    bool big = false;
    DMat shard;  // big: rows [rank L/W, (rank+1) L/W) of every column
    DMat lde;    // !big: the whole LDE
};
struct ShRound {
    std::vector<ShMat> mats;
    ShTree t;
};

struct CommitIn {
    const DMat* nat = nullptr;    // whole matrix, natural row order (left untouched)
    DMat* bitrev_full = nullptr;  // whole matrix, rows at bit-reversed positions (consumed)
    DMat* own_bitrev = nullptr;   // this rank's columns only (ascending), rows at bit-reversed positions (consumed)
    const DMat* rows_nat = nullptr;  // row-range inputs: this rank's ROWS [rank n / W, (rank + 1) n / W) of every column, natural order
    DMat* own_nat = nullptr;      // ... dealt into this rank's columns (whole, natural order) by the round's first exchange
    uint64_t height = 0, width = 0;  // of the WHOLE matrix
};

struct FriLayer {
    bool sharded = false;
    uint64_t len = 0;  // elements of the whole layer
    DBuf buf;          // pair layout: the whole layer or this rank's range
    ShTree t;
};

struct Rank {
    Prover* p = nullptr;
    DeviceCtx* c = nullptr;
    int rank = 0;
    uint32_t e = 0;  // bitrev_W(rank): this rank's row range of an LDE on s H_L is the sub-coset s w_L^e H_{L/W}
    std::unique_ptr<Challenger> ch;
    std::vector<DMat> main_own, prep_nat, perm_nat;
    std::vector<const DMat*> main_nat;
    std::vector<int> prep_slot;
    std::vector<unsigned> log_deg;
    ShRound prep_rs, main_rs, perm_rs, quot_rs;
    std::vector<Ext5> cumulative_sums;
    std::vector<std::vector<Ext5>> bus_alphas, betas;
    Ext5 rnd[3], alpha;
    // commit scratch
    std::vector<DBuf> sendbuf, recvbuf;
    std::vector<DMat> own_lde, own_nat;
    std::vector<char> split;  // per chip: the main / permutation trace of this chip is held as a row range
    // quotient
    DBuf halo_send, halo_recv;
    std::vector<DMat> quot_full, quot_shard, quot_own;
    std::vector<uint32_t> words;
    std::vector<uint32_t> perm_totals;  // canonical: per chip the last running sum of this rank's rows (row-range inputs)
};

}  // namespace

// One failing rank must fail every rank (fabric.hpp, FAILURE PROTOCOL): whatever this rank throws between two collectives — a shape
// mismatch, an allocation failure, a HIP error — is reported to its peers through the status round they run before their next collective.
std::vector<uint32_t> ShardedProof::run(Fabric& f, const std::vector<Prover*>& provers, const std::vector<ShardedInputs>& in, unsigned log_min_sharded) {
    try {
        return run_impl(f, provers, in, log_min_sharded);
    } catch (const FabricPeerFailure&) {
        throw;  // a peer failed first and everyone already knows
    } catch (...) {
        f.fail();
        throw;
    }
}

std::vector<uint32_t> ShardedProof::run_impl(Fabric& f, const std::vector<Prover*>& provers, const std::vector<ShardedInputs>& in, unsigned log_min_sharded) {
    const int W = f.world, NH = (int)f.hosted.size();
    if (W < 1 || (W & (W - 1)) || W > 1024) throw std::invalid_argument("sharded prove: the number of ranks must be a power of two (at most 1024)");
    if (NH < 1 || (int)provers.size() != NH || (int)in.size() != NH) throw std::invalid_argument("sharded prove: one prover context and one set of traces per hosted rank");
    for (auto* p : provers) if (!p) throw std::invalid_argument("sharded prove: null prover");
    const unsigned logW = vg::log2_strict_u64((uint64_t)W);
    const MachineDesc& md = provers[0]->machine_;
    const FriParams fri = provers[0]->fri_;
    const size_t NC = md.airs.size();
    if (fri.log_blowup < 1) throw std::invalid_argument("sharded prove: log_blowup must be at least 1 (= the chips' log_quotient_degree)");
    for (auto& a : md.airs) if (a.log_quotient_degree != 1) throw std::invalid_argument("sharded prove: chip " + a.name + ": log_quotient_degree must be 1");
    if (log_min_sharded > 27) throw std::invalid_argument("sharded prove: log_min_sharded out of range");
    const unsigned lb = fri.log_blowup;
    // a sharded object has at least 4 rows per rank, and more rows than the final FRI layer (2^lb values, held whole by every rank)
    const uint64_t min_big = std::max<uint64_t>(std::max<uint64_t>(4ull * (uint64_t)W, 1ull << log_min_sharded), 2ull << lb);
    const Fp s = Fp::from_canonical(vg::GENERATOR);  // pcs.coset_shift()

    std::vector<Rank> rk((size_t)NH);
    for (int k = 0; k < NH; k++) {
        Rank& R = rk[k];
        R.p = provers[k]; R.c = &provers[k]->ctx(); R.rank = f.hosted[k];
        R.e = logW ? vg::reverse_bits_len((uint32_t)R.rank, logW) : 0u;
        R.ch.reset(new Challenger(&R.p->perm16_));
        if (R.p->fri_.log_blowup != fri.log_blowup || R.p->fri_.num_queries != fri.num_queries || R.p->fri_.pow_bits != fri.pow_bits || R.p->fri_.hash_kind != fri.hash_kind ||
            R.p->fri_.observe_final_poly != fri.observe_final_poly || R.p->machine_.airs.size() != NC)
            throw std::invalid_argument("sharded prove: the ranks' prover contexts must share one configuration");
        if (in[k].main.size() != NC) throw std::invalid_argument("sharded prove: need one main trace per chip and rank");
    }
    // every hosted context for the whole proof; a context that is busy with another proof is not waited for (a sharded proof holds several
    // contexts at once: waiting in some order on each could deadlock two such calls) — the call is refused
    struct Running {
        std::vector<DeviceCtx*> held;
        void take(DeviceCtx& c) {
            if (!c.prove_mu.try_lock()) throw std::invalid_argument("sharded prove: another proof is already running on one of the prover contexts");
            c.proofs_running.fetch_add(1);
            held.push_back(&c);
        }
        ~Running() { for (auto* c : held) { c->proofs_running.fetch_sub(1); c->prove_mu.unlock(); } }
    } running;
    for (auto& R : rk) running.take(*R.c);

    // the sub-coset of rank `R` inside a domain of 2^logL points: generator power w_L^e
    auto rho_of = [&](const Rank& R, unsigned logL) { return vg::two_adic_generator(logL).pow((uint64_t)R.e); };

    // ---------------------------------------------------------------------------------------------------------------------------
    // One commitment round (pcs.commit_batches / commit_shifted_batches, lib.rs:199,223,258,599) over the ranks
    // ---------------------------------------------------------------------------------------------------------------------------
    auto commit_phase_a = [&](Rank& R, std::vector<CommitIn>& cin, const std::vector<Fp>* shifts, ShRound& rs, Fabric::A2A& plan) {
        DeviceCtx* c = R.c;
        c->activate();
        const Fp g = Fp::from_canonical(vg::GENERATOR);
        rs.mats.clear();
        rs.mats.resize(cin.size());
        rs.t = ShTree();
        uint64_t base = 0, H = 0;
        bool sharded = false;
        for (size_t i = 0; i < cin.size(); i++) {
            ShMat& m = rs.mats[i];
            if (cin[i].height == 0 || (cin[i].height & (cin[i].height - 1))) throw std::invalid_argument("sharded commit: matrix heights must be powers of two");
            m.L = cin[i].height << lb; m.width = cin[i].width; m.col_base = base; base += m.width;
            m.big = m.L >= min_big;
            sharded |= m.big;
            H = std::max(H, m.L);
        }
        rs.t.sharded = sharded;
        rs.t.log_total = vg::log2_strict_u64(H);
        rs.t.log_local = sharded ? rs.t.log_total - logW : rs.t.log_total;
        R.own_lde.clear();
        R.own_lde.resize(cin.size());
        std::vector<std::vector<uint64_t>> own(cin.size());
        for (size_t i = 0; i < cin.size(); i++) {
            ShMat& m = rs.mats[i];
            const Fp shift = shifts ? g * (*shifts)[i].inv() : g;
            if (!m.big) {  // computed whole by every rank
                if (cin[i].nat) { CommitInput ci{const_cast<DMat*>(cin[i].nat), false, false}; m.lde = coset_lde(c, c->stream, ci, lb, shift); }
                else if (cin[i].bitrev_full) { CommitInput ci{cin[i].bitrev_full, true, true}; m.lde = coset_lde(c, c->stream, ci, lb, shift); }
                else throw std::logic_error("sharded commit: a replicated matrix needs the whole input");
                continue;
            }
            for (uint64_t col = 0; col < m.width; col++) if (owner_of(m.col_base + col, W) == R.rank) own[i].push_back(col);
            if (own[i].empty()) continue;
            if (cin[i].nat) {
                const DMat& nat = *cin[i].nat;
                if (own[i].size() == nat.width) {  // every column is this rank's (W = 1): extend the matrix where it lies
                    CommitInput ci{const_cast<DMat*>(&nat), false, false};
                    R.own_lde[i] = coset_lde(c, c->stream, ci, lb, shift);
                } else {
                    // this rank's columns are every W-th one (owner_of): ONE strided copy gathers them
                    DMat mine(c, nat.height, own[i].size());
                    VG_HIP_CHECK(hipMemcpy2DAsync(mine.data, mine.height * 4, nat.data + own[i][0] * nat.height, (size_t)W * nat.height * 4, mine.height * 4, own[i].size(),
                                                  hipMemcpyDeviceToDevice, c->stream));
                    CommitInput ci{&mine, false, false};
                    R.own_lde[i] = coset_lde(c, c->stream, ci, lb, shift);
                }
            } else if (cin[i].own_bitrev) {
                if (cin[i].own_bitrev->width != own[i].size() || cin[i].own_bitrev->height != cin[i].height) throw std::logic_error("sharded commit: own-column input of the wrong shape");
                CommitInput ci{cin[i].own_bitrev, true, true};
                R.own_lde[i] = coset_lde(c, c->stream, ci, lb, shift);
            } else if (cin[i].own_nat) {
                if (cin[i].own_nat->width != own[i].size() || cin[i].own_nat->height != cin[i].height) throw std::logic_error("sharded commit: dealt columns of the wrong shape");
                CommitInput ci{cin[i].own_nat, false, false};
                R.own_lde[i] = coset_lde(c, c->stream, ci, lb, shift);
                cin[i].own_nat->reset();
            } else throw std::logic_error("sharded commit: a sharded matrix needs the whole input or this rank's columns");
        }
        if (!sharded) return;
        // The row range of every peer goes out straight from the extended columns and arrives straight in the peer's shard: per big matrix
        // one segment of `own columns` runs of L / W words (a run per owned column; in the receiver's shard that owner's columns are W columns
        // apart).  No pack / unpack copies: the exchange moves every LDE word exactly once.
        plan = Fabric::A2A(c, W);
        R.sendbuf.clear(); R.recvbuf.clear();
        for (size_t i = 0; i < cin.size(); i++) {
            ShMat& m = rs.mats[i];
            if (!m.big) continue;
            const uint64_t rows = m.L / W;
            m.shard = DMat(c, rows, m.width);
            for (int t = 0; t < W; t++) {
                if (!own[i].empty()) plan.add_send(t, R.own_lde[i].data + (uint64_t)t * rows, rows, own[i].size(), m.L);
                // columns owned by rank t: the first is (t - col_base) mod W, then every W-th
                const uint64_t first = (uint64_t)(((int64_t)t - (int64_t)(m.col_base % (uint64_t)W) + W) % W);
                if (first < m.width) plan.add_recv(t, m.shard.data + first * rows, rows, (m.width - first + W - 1) / W, (uint64_t)W * rows);
            }
        }
        c->check_launch("sharded commit: lde");
    };
    auto commit_phase_b = [&](Rank& R, ShRound& rs) {
        DeviceCtx* c = R.c;
        c->activate();
        std::vector<vk::DMatView> views;
        if (!rs.t.sharded) {  // nothing reaches the threshold: every rank holds the whole round
            for (auto& m : rs.mats) views.push_back(m.lde.view());
            rs.t.tree.build(c, views);
            memcpy(rs.t.root, rs.t.tree.root, 32);
            R.own_lde.clear();
            return;
        }
        // the shards were filled by the exchange; the extended columns may go (the peers have finished reading them: the exchange returns
        // when the data has arrived)
        R.own_lde.clear();
        for (auto& m : rs.mats) {
            if (m.big) views.push_back(m.shard.view());
            else if (m.L >= (uint64_t)W) { const uint64_t rows = m.L / W; views.push_back(vk::DMatView{m.lde.data + (uint64_t)R.rank * rows, rows, m.width, m.L}); }
        }
        rs.t.tree.build(c, views);  // reads the subtree root back (synchronises)
    };
    // top log2 W levels over the gathered subtree roots, matrices of fewer than W rows injected at their level; all levels kept on the host
    auto finish_top = [&](Rank& R, ShTree& t, const std::vector<uint32_t>& roots, const std::vector<ShMat>* mats) {
        t.top.clear();
        t.top.push_back(roots);
        if (W == 1) { memcpy(t.root, roots.data(), 32); return; }
        DeviceCtx* c = R.c;
        c->activate();
        DBuf prev(c, roots);
        vk::KeccakTopArgs top{};
        top.prev = prev.data; top.first_len = (uint64_t)W / 2; top.levels = 0;
        std::vector<DBuf> layers;
        std::vector<uint64_t> ptrs;
        std::vector<std::pair<size_t, size_t>> inj;
        for (uint64_t len = (uint64_t)W / 2; len >= 1; len /= 2) {
            const size_t first = ptrs.size();
            if (mats)
                for (auto& m : *mats)
                    if (!m.big && m.L == len) for (uint64_t col = 0; col < m.width; col++) ptrs.push_back((uint64_t)(m.lde.data + col * m.L));
            inj.push_back({first, ptrs.size() - first});
            layers.emplace_back(c, (size_t)len * 8);
            if (len == 1) break;
        }
        DBuf ptr_buf(c, ptrs.size() * 2 + 4);
        if (!ptrs.empty()) c->upload_async(ptr_buf.data, ptrs.data(), ptrs.size() * 8);
        const uint32_t* const* pd = (const uint32_t* const*)ptr_buf.data;
        if (layers.size() > (size_t)vk::KECCAK_TOP_MAX_LEVELS) throw std::invalid_argument("sharded commit: too many ranks");
        for (size_t l = 0; l < layers.size(); l++) {
            top.out[l] = layers[l].data;
            top.cols[l] = inj[l].second ? pd + inj[l].first : nullptr;
            top.n_elems[l] = (int)inj[l].second;
            top.levels++;
        }
        if (c->hash_kind == 1) vk::launch_poseidon_top(c->stream, c->poseidon_tab, c->poseidon_sparse, top); else vk::launch_keccak_top(c->stream, top);
        c->check_launch("sharded top");
        uint64_t len = (uint64_t)W / 2;
        for (size_t l = 0; l < layers.size(); l++, len /= 2) {
            std::vector<uint32_t> d((size_t)len * 8);
            c->download_small(d.data(), layers[l].data, d.size() * 4);
            t.top.push_back(std::move(d));
        }
        memcpy(t.root, t.top.back().data(), 32);
    };
    auto gather_subtree_roots = [&](std::vector<ShTree*> trees, std::vector<uint32_t>& roots) {
        std::vector<const uint32_t*> contrib;
        for (auto* t : trees) contrib.push_back(t->tree.root);
        f.all_gather(contrib, 8, roots);
    };
    // Row-range inputs (ShardedInputs::full_height): rows -> columns.  Rank r holds rows [r n / W, (r + 1) n / W) of every column of a split
    // matrix; the column's owner (global column g -> rank g mod W, as everywhere) receives the W segments and lays them end to end: its
    // columns whole, in natural order, ready for the LDE.  One all-to-all per commitment round, (W - 1) / W of the round's trace words per rank.
    auto deal_rows_to_columns = [&](std::vector<std::vector<CommitIn>>& cin) {
        bool any = false;
        for (auto& ci : cin[0]) any |= ci.rows_nat != nullptr;
        if (!any) return;
        std::vector<Fabric::A2A> plan((size_t)NH);
        auto col_base_of = [&](const std::vector<CommitIn>& v, size_t i) { uint64_t b = 0; for (size_t q = 0; q < i; q++) b += v[q].width; return b; };
        for (int k = 0; k < NH; k++) {
            Rank& R = rk[k];
            DeviceCtx& c = *R.c;
            c.activate();
            plan[k] = Fabric::A2A(&c, W);
            R.own_nat.clear();
            R.own_nat.resize(cin[k].size());
            for (size_t i = 0; i < cin[k].size(); i++) {
                if (!cin[k][i].rows_nat) continue;
                const DMat& m = *cin[k][i].rows_nat;
                const uint64_t n = cin[k][i].height, rows = n / (uint64_t)W, base = col_base_of(cin[k], i), width = cin[k][i].width;
                if (m.height != rows || m.width != width) throw std::invalid_argument("sharded commit: a row range of the wrong shape");
                auto first_of = [&](int t) { return (uint64_t)(((int64_t)t - (int64_t)(base % (uint64_t)W) + W) % W); };
                auto count_of = [&](int t) { const uint64_t f0 = first_of(t); return f0 < width ? (width - f0 + W - 1) / W : 0; };
                // to rank t: my rows of the columns it owns (every W-th column of my row range) ...
                for (int t = 0; t < W; t++) if (count_of(t)) plan[k].add_send(t, m.data + first_of(t) * rows, rows, count_of(t), (uint64_t)W * rows);
                // ... and from rank s its rows of MY columns, laid end to end: column q of mine is whole after the W segments
                const uint64_t mine = count_of(R.rank);
                if (mine) {
                    R.own_nat[i] = DMat(&c, n, mine);
                    for (int src = 0; src < W; src++) plan[k].add_recv(src, R.own_nat[i].data + (uint64_t)src * rows, rows, mine, n);
                }
            }
        }
        f.all_to_all(plan);
        for (int k = 0; k < NH; k++)
            for (size_t i = 0; i < cin[k].size(); i++) {
                if (!cin[k][i].rows_nat) continue;
                cin[k][i].rows_nat = nullptr;
                if (!rk[k].own_nat[i].empty()) cin[k][i].own_nat = &rk[k].own_nat[i];
            }
    };
    auto commit_round = [&](std::vector<std::vector<CommitIn>>& cin, const std::vector<Fp>* shifts, ShRound Rank::*which) {
        deal_rows_to_columns(cin);
        std::vector<Fabric::A2A> plan((size_t)NH);
        for (int k = 0; k < NH; k++) commit_phase_a(rk[k], cin[k], shifts, rk[k].*which, plan[k]);
        const bool sharded = (rk[0].*which).t.sharded;
        if (sharded) f.all_to_all(plan);
        for (int k = 0; k < NH; k++) commit_phase_b(rk[k], rk[k].*which);
        if (sharded) {
            std::vector<ShTree*> trees;
            for (int k = 0; k < NH; k++) trees.push_back(&(rk[k].*which).t);
            std::vector<uint32_t> roots;
            gather_subtree_roots(trees, roots);
            for (int k = 0; k < NH; k++) finish_top(rk[k], (rk[k].*which).t, roots, &(rk[k].*which).mats);
        }
    };

    // ---------------------------------------------------------------------------------------------------------------------------
    // ingest (row-major canonical -> column-major Montgomery, natural row order) on every rank
    // ---------------------------------------------------------------------------------------------------------------------------
    for (int k = 0; k < NH; k++) {
        Rank& R = rk[k];
        DeviceCtx& c = *R.c;
        c.activate();
        const auto& main = in[k].main;
        R.log_deg.resize(NC); R.main_own.resize(NC); R.main_nat.resize(NC);
        R.split.assign(NC, 0);
        const bool row_ranges = !in[k].full_height.empty();
        if (row_ranges && in[k].full_height.size() != NC) throw std::invalid_argument("sharded prove: one full height per chip");
        for (size_t i = 0; i < NC; i++) {
            if (!main[i]) throw std::invalid_argument("sharded prove: null trace");
            if (main[i]->width != md.airs[i].width) throw std::invalid_argument("sharded prove: trace width mismatch for chip " + md.airs[i].name);
            const uint64_t hl = main[i]->height;           // the rows this rank holds
            const uint64_t h = row_ranges ? in[k].full_height[i] : hl;
            if (h == 0 || (h & (h - 1))) throw std::invalid_argument("sharded prove: trace heights must be powers of two");
            if (row_ranges) {
                // the rule of sharded_trace_is_split: exactly the chips whose LDE is sharded come as row ranges
                const bool split = W > 1 && (h << lb) >= min_big;
                if (hl != (split ? h / (uint64_t)W : h))
                    throw std::invalid_argument("sharded prove: chip " + md.airs[i].name + " must hand in " + (split ? "its row range (height / ranks rows)" : "its whole trace") +
                                                ": " + std::to_string(hl) + " rows given, full height " + std::to_string(h));
                R.split[i] = split ? 1 : 0;
            }
            R.log_deg[i] = vg::log2_strict_u64(h);
            if (!main[i]->nat.empty()) { R.main_nat[i] = &main[i]->nat; continue; }
            R.main_own[i] = DMat(&c, hl, main[i]->width);
            R.main_nat[i] = &R.main_own[i];
            vk::launch_ingest(c.stream, main[i]->raw.data, R.main_own[i].view(), false);
        }
        const auto& prep = in[k].prep;
        R.prep_nat.resize(prep.size());
        R.prep_slot.assign(NC, -1);
        for (size_t q = 0; q < prep.size(); q++) {
            const DeviceTrace* t = prep[q].second;
            const int chip = prep[q].first;
            if (!t || chip < 0 || (size_t)chip >= NC || R.prep_slot[chip] >= 0) throw std::invalid_argument("sharded prove: bad or repeated preprocessed chip index");
            if (t->width != md.airs[chip].prep_width || t->height != (1ull << R.log_deg[chip])) throw std::invalid_argument("sharded prove: preprocessed trace shape mismatch (preprocessed traces are handed in whole)");
            R.prep_nat[q] = DMat(&c, t->height, t->width);
            if (!t->nat.empty()) VG_HIP_CHECK(hipMemcpyAsync(R.prep_nat[q].data, t->nat.data, t->height * t->width * 4, hipMemcpyDeviceToDevice, c.stream));
            else vk::launch_ingest(c.stream, t->raw.data, R.prep_nat[q].view(), false);
            R.prep_slot[chip] = (int)q;
        }
        c.check_launch("ingest");
        if (k > 0) {
            if (R.log_deg != rk[0].log_deg || R.prep_slot != rk[0].prep_slot || R.split != rk[0].split) throw std::invalid_argument("sharded prove: the ranks' traces must have the same shapes");
        }
    }
    const std::vector<unsigned>& log_deg = rk[0].log_deg;
    const std::vector<int>& prep_slot = rk[0].prep_slot;
    const size_t NP = in[0].prep.size();
    if (NH < W) {
        // ranks in other processes: every exchange below is sized from the shapes, so they must be THE SAME everywhere before the first one
        std::vector<uint32_t> shape;
        for (size_t i = 0; i < NC; i++) { shape.push_back(log_deg[i]); shape.push_back((uint32_t)(prep_slot[i] + 1)); shape.push_back((uint32_t)rk[0].split[i]); }
        shape.push_back(log_min_sharded); shape.push_back(fri.log_blowup); shape.push_back(fri.num_queries); shape.push_back(fri.pow_bits); shape.push_back((uint32_t)fri.hash_kind);
        shape.push_back(fri.observe_final_poly ? 1u : 0u);
        std::vector<const uint32_t*> contrib((size_t)NH, shape.data());
        std::vector<uint32_t> all;
        f.all_gather(contrib, shape.size(), all);
        for (int r = 0; r < W; r++)
            if (memcmp(all.data() + (size_t)r * shape.size(), shape.data(), shape.size() * 4) != 0)
                throw FabricPeerFailure("sharded prove: rank " + std::to_string(r) + " was called with traces of other shapes or another configuration than rank " +
                                        std::to_string(f.hosted[0]) + " (every rank sees this and gives up the proof)");
    }

    // ---------------------------------------------------------------------------------------------------------------------------
    // preprocessed + main commitments (lib.rs:189-225)
    // ---------------------------------------------------------------------------------------------------------------------------
    if (NP) {
        std::vector<std::vector<CommitIn>> cin((size_t)NH);
        for (int k = 0; k < NH; k++)
            for (auto& m : rk[k].prep_nat) { CommitIn ci; ci.nat = &m; ci.height = m.height; ci.width = m.width; cin[k].push_back(ci); }
        commit_round(cin, nullptr, &Rank::prep_rs);
        for (auto& R : rk) R.ch->observe_digest(R.prep_rs.t.root);
    }
    {
        std::vector<std::vector<CommitIn>> cin((size_t)NH);
        for (int k = 0; k < NH; k++)
            for (size_t i = 0; i < NC; i++) {
                const DMat* m = rk[k].main_nat[i];
                CommitIn ci;
                if (rk[k].split[i]) ci.rows_nat = m; else ci.nat = m;
                ci.height = 1ull << log_deg[i]; ci.width = m->width;
                cin[k].push_back(ci);
            }
        commit_round(cin, nullptr, &Rank::main_rs);
        for (auto& R : rk) R.ch->observe_digest(R.main_rs.t.root);
    }

    // ---------------------------------------------------------------------------------------------------------------------------
    // permutation traces (lib.rs:227-261): every chip, on every rank
    // ---------------------------------------------------------------------------------------------------------------------------
    for (auto& R : rk) {
        DeviceCtx& c = *R.c;
        c.activate();
        for (int i = 0; i < 3; i++) R.rnd[i] = R.ch->sample_ext();
        R.bus_alphas.assign(NC, {}); R.betas.assign(NC, {});
        std::vector<uint32_t> pool;
        std::vector<size_t> off(NC);
        for (size_t i = 0; i < NC; i++) {
            size_t maxf = 0;
            for (auto& it : md.airs[i].interactions) {
                const Ext5& r = it.is_local() ? R.rnd[0] : R.rnd[1];  // generate_rlc_elements (chip.rs:291-331)
                R.bus_alphas[i].push_back(r.pow((uint64_t)it.bus_index + 1));
                maxf = std::max(maxf, it.fields.size());
            }
            Ext5 bp = Ext5::one();
            for (size_t j = 0; j < maxf; j++) { R.betas[i].push_back(bp); bp *= R.rnd[2]; }
            off[i] = pool.size();
            for (auto& a : R.bus_alphas[i]) put_ext(pool, a);
            for (auto& b : R.betas[i]) put_ext(pool, b);
        }
        pool.push_back(0);
        DBuf pool_dev(&c, pool);
        std::vector<DBuf> scratch;
        std::vector<uint32_t> desc;
        R.perm_nat.clear();
        R.perm_nat.resize(NC);
        for (size_t i = 0; i < NC; i++) {
            const uint32_t M = (uint32_t)md.airs[i].interactions.size();
            const uint64_t n = R.main_nat[i]->height;  // the rows this rank holds: a row range of a split chip
            R.perm_nat[i] = DMat(&c, n, 5 * (M + 1));
            vk::DMatView pv{nullptr, 0, 0, 0};
            if (prep_slot[i] >= 0) {
                const DMat& pm = R.prep_nat[prep_slot[i]];  // whole on every rank: a split chip reads its row range of it
                pv = R.split[i] ? vk::DMatView{pm.data + (uint64_t)R.rank * n, n, pm.width, pm.height} : pm.view();
            }
            scratch.emplace_back(&c, (size_t)vk::perm_scratch_words(n));
            vk::launch_perm_trace(c.stream, R.main_nat[i]->view(), pv, R.p->iw_dev_[i].data, pool_dev.data + off[i], M, R.perm_nat[i].view(), scratch.back().data, R.p->fri_.interpret_air ? -2 : R.p->machine_.airs[i].native_chip);
            // cumulative sum = last row of the running-sum column (lib.rs:247-250)
            put_ptr(desc, R.perm_nat[i].data + (uint64_t)(5 * M) * n + (n - 1));
            put_u64(desc, n);
            desc.push_back(5u);
            desc.push_back((uint32_t)(5 * i));
        }
        c.check_launch("perm trace");
        DBuf gd(&c, desc), gout(&c, 5 * NC + 4);
        vk::launch_gather(c.stream, gd.data, NC, gout.data);
        std::vector<uint32_t> cs(5 * NC);
        c.download_small(cs.data(), gout.data, cs.size() * 4);  // synchronises: pool_dev / scratch may go
        R.cumulative_sums.resize(NC);
        for (size_t i = 0; i < NC; i++) R.cumulative_sums[i] = sp_ext_from_canonical(cs.data() + 5 * i);
        R.perm_totals = cs;  // a split chip's entry is the total of ITS rows only: completed below
    }
    {
        // Row-range inputs: the running sum of a split chip (chip.rs:176-205) was scanned over each rank's own rows.  ONE exchange of the
        // ranks' totals (5 words per chip and rank) gives every rank the sum of the rows before its range — added to its column — and the
        // chip's cumulative sum, the total over all ranks.
        bool any_split = false;
        for (char sp : rk[0].split) any_split |= sp != 0;
        if (any_split) {
            std::vector<const uint32_t*> contrib;
            for (auto& R : rk) contrib.push_back(R.perm_totals.data());
            std::vector<uint32_t> all;
            f.all_gather(contrib, 5 * NC, all);
            for (auto& R : rk) {
                DeviceCtx& c = *R.c;
                c.activate();
                std::vector<uint32_t> offs;
                std::vector<size_t> at(NC, 0);
                for (size_t i = 0; i < NC; i++) {
                    if (!R.split[i]) continue;
                    Ext5 before = Ext5::zero(), total = Ext5::zero();
                    for (int r = 0; r < W; r++) {
                        const Ext5 t = sp_ext_from_canonical(&all[(size_t)r * 5 * NC + 5 * i]);
                        if (r < R.rank) before += t;
                        total += t;
                    }
                    R.cumulative_sums[i] = total;
                    at[i] = offs.size();
                    put_ext(offs, before);
                }
                offs.push_back(0);
                DBuf offs_dev(&c, offs);
                for (size_t i = 0; i < NC; i++) {
                    if (!R.split[i] || R.rank == 0) continue;
                    const uint32_t M = (uint32_t)md.airs[i].interactions.size();
                    DMat& pm = R.perm_nat[i];
                    vk::launch_add_ext_const(c.stream, pm.data + (uint64_t)(5 * M) * pm.height, pm.height, pm.height, offs_dev.data + at[i]);
                }
                c.check_launch("perm trace offsets");
                c.sync();  // offs_dev goes
            }
        }
        for (auto& R : rk) R.perm_totals.clear();
    }
    {
        std::vector<std::vector<CommitIn>> cin((size_t)NH);
        for (int k = 0; k < NH; k++)
            for (size_t i = 0; i < NC; i++) {
                DMat& m = rk[k].perm_nat[i];
                CommitIn ci;
                if (rk[k].split[i]) ci.rows_nat = &m; else ci.nat = &m;
                ci.height = 1ull << log_deg[i]; ci.width = m.width;
                cin[k].push_back(ci);
            }
        commit_round(cin, nullptr, &Rank::perm_rs);
        for (auto& R : rk) {
            R.ch->observe_digest(R.perm_rs.t.root);
            // no synchronisation: blocks released here are handed out again only to work enqueued LATER on this context's stream (runtime.hpp),
            // and every exchange drains the stream before it touches memory from outside it
            R.perm_nat.clear(); R.main_own.clear(); R.prep_nat.clear();
        }
    }

    // ---------------------------------------------------------------------------------------------------------------------------
    // quotients (lib.rs:263-599)
    // ---------------------------------------------------------------------------------------------------------------------------
    for (auto& R : rk) R.alpha = R.ch->sample_ext();
    // The quotient domain s H_{2n} is the first L >> dq storage rows of an LDE of L = n 2^lb rows (dq = lb - 1; machine/src/quotient.rs:41-47
    // takes the same rows as a strided view of the natural-order LDE): the row ranges of the first Wq = max(1, W >> dq) ranks.  Those ranks
    // evaluate the quotient, the others only receive their columns of the chunks afterwards.  Inside the quotient domain rank r < Wq holds the
    // sub-coset s w_Q^eq H_{Q / Wq}, eq = bitrev_Wq(r) = e >> dq, and everything below is the blowup-2 picture with (W, e) -> (Wq, eq).
    const unsigned dq = lb - 1;
    const int Wq = std::max(1, W >> dq);
    const unsigned logWq = vg::log2_strict_u64((uint64_t)Wq);
    auto quot_rank = [&](const Rank& R) { return R.rank < Wq; };
    auto eq_of = [&](const Rank& R) { return R.e >> std::min(dq, logW); };
    // rows of chip i's LDE shard that lie in the quotient domain (ranks below Wq): the whole shard once W >= 2^dq
    auto quot_rows = [&](uint64_t L) { return (L >> dq) / (uint64_t)Wq; };
    // the rank holding the successors of rank r's points: sub-coset eq + 2 (the trace generator is w_Q^2)
    auto next_rank = [&](const Rank& R) { return logWq ? (int)vg::reverse_bits_len((eq_of(R) + 2u) % (uint32_t)Wq, logWq) : 0; };
    auto pred_rank = [&](const Rank& R) { return logWq ? (int)vg::reverse_bits_len((eq_of(R) + (uint32_t)Wq - (2u % (uint32_t)Wq)) % (uint32_t)Wq, logWq) : 0; };
    std::vector<size_t> big_chips;
    for (size_t i = 0; i < NC; i++) if (rk[0].main_rs.mats[i].big) big_chips.push_back(i);
    const bool need_halo = Wq > 2 && !big_chips.empty();
    // halo layout (the same on every rank): per big chip [preprocessed shard][main shard][permutation shard], each a contiguous column-major block
    std::vector<size_t> halo_off(NC, 0);
    size_t halo_words = 0;
    for (size_t i : big_chips) {
        const ShMat& mm = rk[0].main_rs.mats[i];
        const ShMat& pm = rk[0].perm_rs.mats[i];
        const uint64_t rows = quot_rows(mm.L);
        halo_off[i] = halo_words;
        if (prep_slot[i] >= 0) halo_words += rows * rk[0].prep_rs.mats[prep_slot[i]].width;
        halo_words += rows * mm.width + rows * pm.width;
    }
    if (need_halo) {  // Wq > 2 implies W >= 4 * 2^dq: a quotient rank's shard lies in the quotient domain with all its rows
        std::vector<Fabric::A2A> plan((size_t)NH);
        for (int k = 0; k < NH; k++) {
            Rank& R = rk[k];
            DeviceCtx& c = *R.c;
            c.activate();
            plan[k] = Fabric::A2A(&c, W);
            if (!quot_rank(R)) continue;
            R.halo_recv = DBuf(&c, halo_words + 4);
            const int to = pred_rank(R), from = next_rank(R);
            for (size_t i : big_chips) {  // the first `rows` rows of every column of the three shards (a shard may be taller than its part of the quotient domain)
                auto put = [&](const DMat& sh) { plan[k].add_send(to, sh.data, quot_rows(rk[0].main_rs.mats[i].L), sh.width, sh.height); };
                if (prep_slot[i] >= 0) put(R.prep_rs.mats[prep_slot[i]].shard);
                put(R.main_rs.mats[i].shard);
                put(R.perm_rs.mats[i].shard);
            }
            {   // the same segments, laid end to end in the halo block (the quotient kernel reads each as a contiguous column-major matrix)
                size_t pos = 0;
                for (size_t i : big_chips) {
                    const uint64_t rows = quot_rows(rk[0].main_rs.mats[i].L);
                    auto get = [&](uint64_t width) { plan[k].add_recv(from, R.halo_recv.data + pos, rows, width, rows); pos += rows * width; };
                    if (prep_slot[i] >= 0) get(R.prep_rs.mats[prep_slot[i]].width);
                    get(R.main_rs.mats[i].width);
                    get(R.perm_rs.mats[i].width);
                }
            }
        }
        f.all_to_all(plan);
    }
    for (auto& R : rk) {
        DeviceCtx& c = *R.c;
        c.activate();
        std::vector<uint32_t> pool;
        std::vector<size_t> off(NC);
        std::vector<uint32_t> Ks(NC);
        for (size_t i = 0; i < NC; i++) {
            auto& air = md.airs[i];
            const uint32_t M = (uint32_t)air.interactions.size(), K = air.program.num_asserts + M + 3;
            Ks[i] = K;
            off[i] = pool.size();
            std::vector<Ext5> ap(K);
            Ext5 pw = Ext5::one();
            for (uint32_t q = 0; q < K; q++) { ap[K - 1 - q] = pw; pw *= R.alpha; }  // constraint k is scaled by alpha^(K-1-k)
            for (auto& x : ap) put_ext(pool, x);
            for (auto& x : R.bus_alphas[i]) put_ext(pool, x);
            for (auto& x : R.betas[i]) put_ext(pool, x);
            put_ext(pool, R.cumulative_sums[i]);
        }
        DBuf pool_dev(&c, pool);
        R.quot_full.clear(); R.quot_full.resize(NC);
        R.quot_shard.clear(); R.quot_shard.resize(NC);
        const uint32_t eq = eq_of(R);
        const uint32_t d = (eq + 2u) / (uint32_t)Wq;  // the successor's natural index inside its shard: same (0) or following (1); 2 on one rank
        for (size_t i = 0; i < NC; i++) {
            const ShMat& mm = R.main_rs.mats[i];
            const ShMat& pm = R.perm_rs.mats[i];
            const ShMat* qm = prep_slot[i] >= 0 ? &R.prep_rs.mats[prep_slot[i]] : nullptr;
            vk::QuotientArgs a{};
            if (!mm.big) {
                R.p->fill_quotient_args(a, (int)i, mm.lde.view(), pm.lde.view(), qm ? qm->lde.view() : vk::DMatView{nullptr, 0, 0, 0}, log_deg[i], pool_dev.data + off[i]);
                a.K = Ks[i];
                R.quot_full[i] = DMat(&c, 1ull << log_deg[i], 10);
                a.out = R.quot_full[i].view();
                a.out_natural = 1;  // chunk rows in natural order: the quotient round takes the fused LDE like the other two (as in prover.cpp)
                vk::launch_quotient(c.stream, a, c.tables);
                continue;
            }
            if (!quot_rank(R)) continue;  // this rank's row range lies outside the quotient domain
            const uint64_t rows = quot_rows(mm.L);        // this rank's storage rows inside the quotient domain (the first rows of its shard)
            const unsigned log_rows = vg::log2_strict_u64(rows);
            auto qview = [&](const DMat& sh) { return vk::DMatView{sh.data, rows, sh.width, sh.height}; };
            R.p->fill_quotient_args(a, (int)i, qview(mm.shard), qview(pm.shard), qm ? qview(qm->shard) : vk::DMatView{nullptr, 0, 0, 0}, log_rows - 1, pool_dev.data + off[i]);
            a.K = Ks[i];
            if (Wq > 1) {
                // the rows are the LDE on s' H_rows, s' = s w_L^e = s w_Q^eq: x, 1/x, the decomposition pairs (x, -x) follow from the shift; the
                // selectors and Z_H belong to the TRACE domain H_n (n = Q / 2): x^n = s^n (-1)^(natural index), and the natural index
                // eq + Wq m of every point of this range has the parity of eq
                const unsigned logL = vg::log2_strict_u64(mm.L);
                const Fp sp = s * rho_of(R, logL);
                a.coset_shift = sp.v; a.coset_shift_inv = sp.inv().v;
                Fp sn = s.exp_power_of_2(log_deg[i]);
                if (eq & 1u) sn = -sn;
                const Fp z = sn - Fp::one();
                a.zh[0] = a.zh[1] = z.v;
                a.zh_inv[0] = a.zh_inv[1] = z.inv().v;
                a.g_inv = vg::two_adic_generator(log_deg[i]).inv().v;
                a.next_step_p1 = d + 1u;
                if (need_halo) {
                    const uint32_t* h = R.halo_recv.data + halo_off[i];
                    if (qm) { a.prep_nx = h; h += rows * qm->width; }
                    a.main_nx = h; h += rows * mm.width;
                    a.perm_nx = h;
                }
            }
            R.quot_shard[i] = DMat(&c, rows / 2, 10);
            a.out = R.quot_shard[i].view();
            a.out_natural = 1;  // natural order of THIS range (the sub-coset eq): global natural index eq + Wq m
            vk::launch_quotient(c.stream, a, c.tables);
        }
        c.check_launch("quotient");
        R.halo_recv = DBuf();  // (and pool_dev) released in stream order: no synchronisation needed
    }
    // rows -> columns: the quotient commit extends whole columns
    std::vector<Fp> quot_shifts(NC, s.exp_power_of_2(1));  // lib.rs:593-596 with log_quotient_degree = 1
    {
        std::vector<std::vector<CommitIn>> cin((size_t)NH);
        if (!big_chips.empty()) {
            // senders: the Wq ranks that evaluated the quotient (chunk rows [r rows, (r + 1) rows) of every big chip); receivers: all W ranks
            // (global chunk column g -> rank g mod W, as in every commitment round)
            auto chunk_rows = [&](size_t i) { return quot_rows(rk[0].main_rs.mats[i].L) / 2; };
            std::vector<Fabric::A2A> plan((size_t)NH);
            for (int k = 0; k < NH; k++) {
                Rank& R = rk[k];
                DeviceCtx& c = *R.c;
                c.activate();
                plan[k] = Fabric::A2A(&c, W);
                R.quot_own.clear(); R.quot_own.resize(NC);
                for (size_t i : big_chips) {
                    const uint64_t rows = chunk_rows(i), n = rows * (uint64_t)Wq, base = 10 * (uint64_t)i;
                    auto first_of = [&](int t) { return (uint64_t)(((int64_t)t - (int64_t)(base % (uint64_t)W) + W) % W); };
                    auto count_of = [&](int t) { const uint64_t f0 = first_of(t); return f0 < 10 ? (10 - f0 + W - 1) / W : 0; };
                    // a quotient rank sends every rank its chunk rows of that rank's columns (every W-th of the ten) ...
                    if (quot_rank(R)) {
                        const DMat& q = R.quot_shard[i];
                        for (int t = 0; t < W; t++) if (count_of(t)) plan[k].add_send(t, q.data + first_of(t) * q.height, q.height, count_of(t), (uint64_t)W * q.height);
                    }
                    // ... and every rank receives, from each quotient rank, that rank's rows of its own columns, laid end to end
                    const uint64_t mine = count_of(R.rank);
                    if (mine) {
                        R.quot_own[i] = DMat(&c, n, mine);
                        for (int src = 0; src < Wq; src++) plan[k].add_recv(src, R.quot_own[i].data + (uint64_t)src * rows, rows, mine, n);
                    }
                }
            }
            f.all_to_all(plan);
            for (auto& R : rk) R.quot_shard.clear();
            if (Wq > 1)  // the blocks lie end to end, block r = sub-coset bitrev_Wq(r): interleave them into the global natural order
                for (int k = 0; k < NH; k++) {
                    Rank& R = rk[k];
                    DeviceCtx& c = *R.c;
                    c.activate();
                    for (size_t i : big_chips) {
                        if (R.quot_own[i].empty()) continue;
                        DMat nat(&c, R.quot_own[i].height, R.quot_own[i].width);
                        vk::launch_interleave_blocks(c.stream, R.quot_own[i].data, nat.data, chunk_rows(i), logWq, nat.width);
                        R.quot_own[i] = std::move(nat);  // the old block returns to the pool in stream order
                    }
                    c.check_launch("quotient interleave");
                }
        }
        for (int k = 0; k < NH; k++) {
            Rank& R = rk[k];
            for (size_t i = 0; i < NC; i++) {
                CommitIn ci;
                ci.height = 1ull << log_deg[i]; ci.width = 10;
                if (R.main_rs.mats[i].big) ci.own_nat = R.quot_own[i].empty() ? nullptr : &R.quot_own[i];
                else ci.nat = &R.quot_full[i];
                cin[k].push_back(ci);
            }
        }
        commit_round(cin, &quot_shifts, &Rank::quot_rs);
        for (auto& R : rk) { R.ch->observe_digest(R.quot_rs.t.root); R.quot_full.clear(); R.quot_own.clear(); }
    }

    // ---------------------------------------------------------------------------------------------------------------------------
    // opening (lib.rs:606-619): pcs.open_multi_batches over the three rounds
    // ---------------------------------------------------------------------------------------------------------------------------
    const size_t NR = 3;
    struct Opening {
        std::vector<std::vector<std::vector<std::vector<Ext5>>>> opened;  // [round][matrix][point][column]
        std::vector<uint32_t> out;                                         // this rank's (partial) values, canonical words
        Ext5 alpha_b;
        std::map<unsigned, DBuf> ro;
        std::vector<FriLayer> layers;
        std::vector<std::array<uint32_t, 8>> commits;
        DBuf cur;
        bool cur_sharded = false;
        uint32_t fpw[5];
        uint32_t pow_witness = 0;
        std::vector<uint32_t> tail;
    };
    std::vector<Opening> op((size_t)NH);
    std::vector<std::vector<std::vector<std::vector<Ext5>>>> points((size_t)NH);  // [rank][round][matrix] -> points
    for (int k = 0; k < NH; k++) {
        Rank& R = rk[k];
        const Ext5 zeta = R.ch->sample_ext();
        points[k].resize(NR);
        for (size_t i = 0; i < NC; i++) {
            const Fp g = vg::two_adic_generator(log_deg[i]);
            points[k][0].push_back({zeta, zeta * g});
            points[k][1].push_back({zeta, zeta * g});
            points[k][2].push_back({zeta.exp_power_of_2(1)});
        }
    }
    auto round_of = [&](Rank& R, size_t r) -> ShRound& { return r == 0 ? R.main_rs : r == 1 ? R.perm_rs : R.quot_rs; };

    // ---- opened values.  Sharded matrix: p(z) = (z^L - s^L) / (L s^(L-1)) sum_{j < L} y_j r_j / (z - s r_j) over the WHOLE LDE domain, whose sum
    // splits over the row ranges; on a range r_j = w_L^e r_jl, so k_bary_weights / k_col_dot run on the shard with the shift s w_L^e and the
    // host folds w_L^e into the scale.  Replicated matrix: the usual sum over the first n rows (App. B9).
    struct Job { size_t r, i; int p0, np; uint64_t c0, cw; size_t w[2]; size_t scale_off, out_off; bool big; };
    std::vector<Job> jobs0;
    size_t out_words = 0;
    for (int k = 0; k < NH; k++) {
        Rank& R = rk[k];
        DeviceCtx& c = *R.c;
        c.activate();
        Opening& O = op[k];
        O.opened.resize(NR);
        struct WEntry { bool big; unsigned logL; uint64_t rows; Fp shift; Ext5 z; size_t pool_off; DBuf buf; };
        std::map<std::tuple<int, unsigned, SpPointKey>, size_t> wkey;
        std::vector<WEntry> wlist;
        std::vector<uint32_t> pool;
        std::vector<Job> jobs;
        size_t ow = 0;
        for (size_t r = 0; r < NR; r++) {
            ShRound& rs = round_of(R, r);
            O.opened[r].resize(rs.mats.size());
            for (size_t i = 0; i < rs.mats.size(); i++) {
                const ShMat& m = rs.mats[i];
                const unsigned logL = vg::log2_strict_u64(m.L), ln = logL - lb;
                const uint64_t n = 1ull << ln;
                const auto& pts = points[k][r][i];
                O.opened[r][i].assign(pts.size(), std::vector<Ext5>(m.width));
                const Fp rho = m.big ? rho_of(R, logL) : Fp::one();
                for (size_t p0 = 0; p0 < pts.size(); p0 += 2) {
                    const int np = (int)std::min<size_t>(2, pts.size() - p0);
                    size_t widx[2] = {0, 0};
                    const size_t scale_off = pool.size();
                    for (int p = 0; p < np; p++) {
                        const Ext5& z = pts[p0 + p];
                        if (m.big) {
                            Ext5 zer = z.exp_power_of_2(logL) - s.exp_power_of_2(logL);
                            Fp den = Fp::from_canonical((uint32_t)(m.L % vg::P)) * s.pow(m.L - 1);
                            put_ext(pool, zer * (den.inv() * rho));
                        } else {
                            Ext5 zer = z.exp_power_of_2(ln) - s.exp_power_of_2(ln);  // scale = (z^n - s^n) / (n s^(n-1))
                            Fp den = Fp::from_canonical((uint32_t)(n % vg::P)) * s.pow(n - 1);
                            put_ext(pool, zer * den.inv());
                        }
                    }
                    for (int p = 0; p < np; p++) {
                        const Ext5& z = pts[p0 + p];
                        auto key = std::make_tuple(m.big ? 1 : 0, m.big ? logL : ln, sp_key_of(z));
                        auto it = wkey.find(key);
                        if (it == wkey.end()) {
                            wkey[key] = wlist.size(); widx[p] = wlist.size();
                            wlist.push_back(WEntry{m.big, logL, m.big ? m.L / W : n, m.big ? s * rho : s, z, 0, DBuf()});
                        } else widx[p] = it->second;
                    }
                    const uint64_t max_cols = vk::col_dot_max_columns(np);
                    for (uint64_t c0 = 0; c0 < m.width; c0 += max_cols) {
                        const uint64_t cw = std::min<uint64_t>(max_cols, m.width - c0);
                        jobs.push_back(Job{r, i, (int)p0, np, c0, cw, {widx[0], widx[np - 1]}, scale_off, ow, m.big});
                        ow += cw * np * 5;
                    }
                }
            }
        }
        for (auto& w : wlist) { w.pool_off = pool.size(); put_min_poly(pool, w.z); }
        pool.push_back(0);
        DBuf pool_dev(&c, pool);
        {  // all weight vectors of this rank in ONE launch (k_bary_weights_batch, as in the single-GPU prover); a job carries its own shift
            std::vector<uint32_t> bj;
            uint32_t blocks = 0;
            double rows = 0;
            auto put64 = [&](uint64_t v) { bj.push_back((uint32_t)v); bj.push_back((uint32_t)(v >> 32)); };
            for (auto& w : wlist) {
                w.buf = DBuf(&c, (size_t)vk::bary_buffer_words(w.rows));
                bj.push_back(blocks); bj.push_back(w.shift.v);  // a Montgomery word of a non-zero field element is never 0
                put64(w.rows);
                put64((uint64_t)(pool_dev.data + w.pool_off));
                put64((uint64_t)w.buf.data);
                put64((uint64_t)(vk::bary_weights_has_image(w.rows) ? w.buf.data + 5 * w.rows : nullptr));
                blocks += vk::bary_weights_blocks(w.rows);
                rows += (double)w.rows;
            }
            if (!wlist.empty()) {
                DBuf jobs_dev(&c, bj);
                vk::launch_bary_weights_batch(c.stream, jobs_dev.data, (uint32_t)wlist.size(), blocks, rows, s, c.tables);
            }
        }
        DBuf out_dev(&c, ow + 4);
        std::vector<DBuf> partials;
        for (auto& j : jobs) {
            const ShMat& m = round_of(R, j.r).mats[j.i];
            const DMat& lde = m.big ? m.shard : m.lde;
            const uint64_t rows = m.big ? lde.height : (lde.height >> lb);
            partials.emplace_back(&c, (size_t)(vk::col_dot_slots(rows) * j.cw * j.np * 5));
            vk::DMatView sub{lde.data + j.c0 * lde.height, lde.height, j.cw, lde.height};
            vk::launch_col_dot(c.stream, sub, rows, j.np, wlist[j.w[0]].buf.data, wlist[j.w[1]].buf.data, partials.back().data, pool_dev.data + j.scale_off, out_dev.data + j.out_off);
        }
        c.check_launch("opened values");
        O.out.assign(ow, 0);
        if (ow) c.download_small(O.out.data(), out_dev.data, ow * 4); else c.sync();  // synchronises: the temporaries of this block may go
        if (k == 0) { jobs0 = jobs; out_words = ow; }
        else if (ow != out_words) throw std::logic_error("sharded open: ranks disagree on the layout of the opened values");
    }
    {
        std::vector<const uint32_t*> contrib;
        for (auto& O : op) contrib.push_back(O.out.data());
        std::vector<uint32_t> all;
        f.all_gather(contrib, out_words, all);
        for (int k = 0; k < NH; k++) {
            Opening& O = op[k];
            for (auto& j : jobs0)
                for (uint64_t col = 0; col < j.cw; col++)
                    for (int p = 0; p < j.np; p++) {
                        const size_t at = j.out_off + (col * j.np + p) * 5;
                        Ext5 v;
                        if (j.big) {
                            v = Ext5::zero();
                            for (int r = 0; r < W; r++) v += sp_ext_from_canonical(&all[(size_t)r * out_words + at]);
                        } else v = sp_ext_from_canonical(&O.out[at]);
                        O.opened[j.r][j.i][j.p0 + p][j.c0 + col] = v;
                    }
        }
    }

    // ---- reduced openings per LDE height (App. B9), row-local: a sharded height is reduced over the rank's range with the shifted coset
    unsigned log_max = 0;
    for (int k = 0; k < NH; k++) {
        Rank& R = rk[k];
        DeviceCtx& c = *R.c;
        c.activate();
        Opening& O = op[k];
        O.alpha_b = R.ch->sample_ext();
        const Ext5& alpha_b = O.alpha_b;
        struct MatEntry { const uint32_t* data; uint64_t stride; uint64_t width; std::vector<std::tuple<uint32_t, Ext5, Ext5>> pts; };  // (slot, alpha^offset, Y)
        struct Group { std::vector<Ext5> zs; std::map<SpPointKey, uint32_t> slot; std::vector<MatEntry> mats; uint64_t num_reduced = 0; Ext5 offset = Ext5::one(); };  // offset = alpha^num_reduced
        std::map<unsigned, Group> groups;
        size_t max_width = 0;
        for (size_t r = 0; r < NR; r++) for (auto& m : round_of(R, r).mats) max_width = std::max<size_t>(max_width, m.width);
        std::vector<Ext5> apow(max_width);
        { Ext5 pw = Ext5::one(); for (auto& a : apow) { a = pw; pw *= alpha_b; } }
        for (size_t r = 0; r < NR; r++) {
            ShRound& rs = round_of(R, r);
            for (size_t i = 0; i < rs.mats.size(); i++) {
                const ShMat& m = rs.mats[i];
                const unsigned lh = vg::log2_strict_u64(m.L);
                Group& g = groups[lh];
                MatEntry me{m.big ? m.shard.data : m.lde.data, m.big ? m.shard.height : m.lde.height, m.width, {}};
                for (size_t p = 0; p < points[k][r][i].size(); p++) {
                    const Ext5& z = points[k][r][i][p];
                    auto key = sp_key_of(z);
                    auto it = g.slot.find(key);
                    uint32_t slot;
                    if (it == g.slot.end()) { slot = (uint32_t)g.zs.size(); g.slot[key] = slot; g.zs.push_back(z); } else slot = it->second;
                    Ext5 Y = Ext5::zero();
                    auto& ys = O.opened[r][i][p];
                    for (size_t col = 0; col < ys.size(); col++) Y += apow[col] * ys[col];
                    me.pts.emplace_back(slot, g.offset, Y);
                    g.num_reduced += m.width;
                    if (m.width) g.offset *= apow[m.width - 1] * alpha_b;  // alpha^width (as in prover.cpp)
                }
                g.mats.push_back(std::move(me));
            }
        }
        struct Launch { unsigned lh; size_t off; bool accumulate; uint64_t total_width; int n_points; bool vec_ok; };
        std::vector<Launch> launches;
        std::vector<uint32_t> pool;
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
                    if (any) { live.push_back(&me); max_w = std::max<size_t>(max_w, me.width); total_width += me.width; vec_ok &= vk::reduce_vec_ok(me.data, me.stride); }
                }
                launches.push_back({kv.first, pool.size(), s0 != 0, total_width, (int)(s1 - s0), vec_ok});
                pool.push_back((uint32_t)live.size());
                pool.push_back(s1 - s0);
                pool.push_back((uint32_t)max_w);
                for (uint32_t q = s0; q < s1; q++) put_min_poly(pool, g.zs[q]);
                for (size_t col = 0; col < max_w; col++) put_ext(pool, apow[col]);
                for (auto* me : live) {
                    put_ptr(pool, me->data);
                    put_u64(pool, me->stride);
                    pool.push_back((uint32_t)me->width);
                    uint32_t cnt = 0;
                    for (auto& t : me->pts) cnt += std::get<0>(t) >= s0 && std::get<0>(t) < s1;
                    pool.push_back(cnt);
                    for (auto& t : me->pts)
                        if (std::get<0>(t) >= s0 && std::get<0>(t) < s1) { pool.push_back(std::get<0>(t) - s0); put_ext(pool, std::get<1>(t)); put_ext(pool, std::get<2>(t)); }
                }
                if (g.zs.size() <= s1) break;
            }
        }
        DBuf pool_dev(&c, pool);
        for (auto& kv : groups) {
            const uint64_t L = 1ull << kv.first, rows = L >= min_big ? L / W : L;
            O.ro[kv.first] = DBuf(&c, (size_t)(5 * rows));
            log_max = std::max(log_max, kv.first);
        }
        for (auto& l : launches) {
            const uint64_t L = 1ull << l.lh;
            const bool big = L >= min_big;
            vk::launch_reduce_openings(c.stream, pool_dev.data + l.off, big ? L / W : L, big ? s * rho_of(R, l.lh) : s, c.tables, O.ro[l.lh].data, l.total_width, l.accumulate, l.n_points, l.vec_ok);
        }
        c.check_launch("reduce openings");  // pool_dev is released in stream order
    }
    if (log_max < lb) throw std::invalid_argument("sharded open: nothing to open");

    // ---- FRI commit phase (App. B10): a layer of at least min_big elements lives in row ranges
    // A layer that lives in row ranges needs the ranks' subtree roots on the host (an exchange per layer); from the first layer every
    // rank holds WHOLE — with one rank: from the first layer — the transcript moves to the device as in the single-GPU prover
    // (k_fri_challenge observes each root where the tree kernel left it and samples beta for the fold): the dependent chain of the
    // remaining layers is enqueued without a host round trip, and roots, sponge state and final values come back in one go.
    const bool multi = W > 1;
    for (auto& O : op) {
        O.cur = std::move(O.ro[log_max]);
        O.ro.erase(log_max);
        O.cur_sharded = multi && (1ull << log_max) >= min_big;
    }
    struct DevFri { bool on = false; unsigned count = 0; size_t first_layer = 0; DBuf ch_dev, betas_dev, commits_dev; };
    std::vector<DevFri> dev((size_t)NH);
    for (unsigned lf = log_max; lf-- > lb;) {
        const uint64_t len = 2ull << lf, half = len >> 1;  // current length 2^(lf+1)
        const bool sharded = multi && len >= min_big, next_sharded = multi && half >= min_big;
        if (!sharded) {
            for (int k = 0; k < NH; k++) {
                Rank& R = rk[k];
                DeviceCtx& c = *R.c;
                c.activate();
                Opening& O = op[k];
                DevFri& D = dev[(size_t)k];
                if (O.cur_sharded) throw std::logic_error("sharded fri: layer state out of step");
                if (!D.on) {
                    const unsigned remaining = lf - lb + 1;
                    std::vector<uint32_t> chw(vk::DEV_CHALLENGER_WORDS, 0);
                    const Challenger& ch = *R.ch;
                    for (int i = 0; i < 16; i++) chw[i] = ch.state[i].v;
                    for (size_t i = 0; i < ch.in.size(); i++) chw[16 + i] = ch.in[i].v;
                    chw[32] = (uint32_t)ch.in.size();
                    for (size_t i = 0; i < ch.out.size(); i++) chw[33 + i] = ch.out[i].v;
                    chw[49] = (uint32_t)ch.out.size();
                    D.ch_dev = DBuf(&c, chw);
                    D.betas_dev = DBuf(&c, (size_t)(5 * remaining + 8));
                    D.commits_dev = DBuf(&c, (size_t)(8 * remaining + 8));
                    D.first_layer = O.layers.size();
                    D.on = true;
                }
                O.layers.emplace_back();
                FriLayer& ly = O.layers.back();
                ly.sharded = false; ly.len = len;
                ly.t.sharded = false;
                ly.t.log_total = ly.t.log_local = vg::log2_strict_u64(half);
                // the challenger step rides on the tree-top launch, as in the single-GPU prover (prover.cpp, FRI commit phase)
                const DeviceTree::TopChallenger step{R.p->pow_pos_.data, D.ch_dev.data, D.betas_dev.data + 5 * D.count, D.commits_dev.data + 8 * D.count};
                ly.t.tree.build(&c, {vk::DMatView{O.cur.data, half, 10, half}}, false, nullptr, &step);
                DBuf next(&c, (size_t)(5 * half));
                auto it = O.ro.find(lf);
                vk::launch_fri_fold(c.stream, O.cur.data, len, D.betas_dev.data + 5 * D.count, it != O.ro.end() ? it->second.data : nullptr, c.tables, next.data);
                c.check_launch("fri fold");
                D.count++;
                ly.buf = std::move(O.cur);
                O.cur = std::move(next);
                O.cur_sharded = false;
            }
            continue;
        }
        // the layer's tree
        for (int k = 0; k < NH; k++) {
            Rank& R = rk[k];
            DeviceCtx& c = *R.c;
            c.activate();
            Opening& O = op[k];
            if (O.cur_sharded != sharded) throw std::logic_error("sharded fri: layer state out of step");
            O.layers.emplace_back();
            FriLayer& ly = O.layers.back();
            ly.sharded = sharded; ly.len = len;
            const uint64_t prow = sharded ? half / W : half;  // rows of the pair matrix this rank holds
            ly.t.sharded = sharded;
            ly.t.log_total = vg::log2_strict_u64(half);
            ly.t.log_local = vg::log2_strict_u64(prow);
            ly.t.tree.build(&c, {vk::DMatView{O.cur.data, prow, 10, prow}}, true);
            if (!sharded) memcpy(ly.t.root, ly.t.tree.root, 32);
        }
        if (sharded) {
            std::vector<ShTree*> trees;
            for (auto& O : op) trees.push_back(&O.layers.back().t);
            std::vector<uint32_t> roots;
            gather_subtree_roots(trees, roots);
            for (int k = 0; k < NH; k++) finish_top(rk[k], op[k].layers.back().t, roots, nullptr);
        }
        // observe the root, sample beta, fold
        std::vector<std::vector<uint32_t>> next_host((size_t)NH);
        for (int k = 0; k < NH; k++) {
            Rank& R = rk[k];
            DeviceCtx& c = *R.c;
            c.activate();
            Opening& O = op[k];
            FriLayer& ly = O.layers.back();
            R.ch->observe_digest(ly.t.root);
            std::array<uint32_t, 8> root;
            memcpy(root.data(), ly.t.root, 32);
            O.commits.push_back(root);
            Ext5 beta = R.ch->sample_ext();
            // a row range of the layer sits on w_len^e H: its 1 / x is the local one times w_len^-e, folded into beta
            if (sharded) beta = beta * rho_of(R, lf + 1).inv();
            std::vector<uint32_t> bw;
            put_ext(bw, beta);
            bw.resize(8, 0);
            DBuf beta_dev(&c, bw);
            const uint64_t my_len = sharded ? len / W : len;
            DBuf next(&c, (size_t)(5 * (my_len / 2)));
            auto it = O.ro.find(lf);
            const uint32_t* add = (it != O.ro.end() && sharded == next_sharded) ? it->second.data : nullptr;
            vk::launch_fri_fold(c.stream, O.cur.data, my_len, beta_dev.data, add, c.tables, next.data);
            c.check_launch("fri fold");
            if (sharded && !next_sharded) {  // the next layer is short: every rank takes all of it
                next_host[k].resize((size_t)(5 * (my_len / 2)));
                c.download(next_host[k].data(), next.data, next_host[k].size() * 4);
            } else c.sync();  // beta_dev goes
            ly.buf = std::move(O.cur);
            O.cur = std::move(next);
            O.cur_sharded = next_sharded;
        }
        if (sharded && !next_sharded) {
            const uint64_t my_half = half / W, qloc = my_half / 2, q = half / 2;  // elements / pair rows of a range, pair rows of the whole next layer
            std::vector<const uint32_t*> contrib;
            for (auto& v : next_host) contrib.push_back(v.data());
            std::vector<uint32_t> all;
            f.all_gather(contrib, (size_t)(10 * qloc), all);
            for (int k = 0; k < NH; k++) {
                Rank& R = rk[k];
                DeviceCtx& c = *R.c;
                c.activate();
                Opening& O = op[k];
                std::vector<uint32_t> full((size_t)(10 * q));
                for (int r = 0; r < W; r++)
                    for (uint64_t col = 0; col < 10; col++)
                        memcpy(&full[(size_t)(col * q + (uint64_t)r * qloc)], &all[(size_t)r * 10 * qloc + (size_t)(col * qloc)], (size_t)qloc * 4);
                auto it = O.ro.find(lf);
                if (it != O.ro.end()) {  // the reduced opening of this (replicated) height joins here
                    std::vector<uint32_t> add((size_t)(10 * q));
                    c.download(add.data(), it->second.data, add.size() * 4);
                    for (size_t x = 0; x < full.size(); x++) full[x] = (Fp::raw(full[x]) + Fp::raw(add[x])).v;
                }
                O.cur = DBuf(&c, full);
            }
        }
    }
    for (int k = 0; k < NH; k++) {
        DevFri& D = dev[(size_t)k];
        if (!D.on) continue;
        Rank& R = rk[k];
        DeviceCtx& c = *R.c;
        c.activate();
        Opening& O = op[k];
        std::vector<uint32_t> chw(vk::DEV_CHALLENGER_WORDS), commits((size_t)(8 * D.count + 8));
        {   // sponge state and roots: two copies into one pinned area, one synchronisation
            const size_t n0 = chw.size(), n1 = 8 * (size_t)D.count;
            std::lock_guard<std::recursive_mutex> lk(c.host_mu);
            uint32_t* pin = (uint32_t*)c.pinned_buffer((n0 + n1 + 8) * 4);
            VG_HIP_CHECK(hipMemcpyAsync(pin, D.ch_dev.data, n0 * 4, hipMemcpyDeviceToHost, c.stream));
            if (n1) VG_HIP_CHECK(hipMemcpyAsync(pin + n0, D.commits_dev.data, n1 * 4, hipMemcpyDeviceToHost, c.stream));
            c.sync();
            memcpy(chw.data(), pin, n0 * 4);
            memcpy(commits.data(), pin + n0, n1 * 4);
        }
        Challenger& ch = *R.ch;
        for (int i = 0; i < 16; i++) ch.state[i] = Fp::raw(chw[i]);
        ch.in.clear();
        for (uint32_t i = 0; i < chw[32]; i++) ch.in.push_back(Fp::raw(chw[16 + i]));
        ch.out.clear();
        for (uint32_t i = 0; i < chw[49]; i++) ch.out.push_back(Fp::raw(chw[33 + i]));
        for (unsigned i = 0; i < D.count; i++) {
            std::array<uint32_t, 8> root;
            memcpy(root.data(), commits.data() + 8 * i, 32);
            FriLayer& ly = O.layers[D.first_layer + i];
            memcpy(ly.t.root, root.data(), 32);
            memcpy(ly.t.tree.root, root.data(), 32);
            O.commits.push_back(root);
        }
        D = DevFri();
    }
    // `cur` now holds 2^lb values that must all be equal (a constant polynomial)
    for (int k = 0; k < NH; k++) {
        Rank& R = rk[k];
        DeviceCtx& c = *R.c;
        c.activate();
        Opening& O = op[k];
        if (O.cur_sharded) throw std::logic_error("sharded fri: the final layer cannot be sharded");
        std::vector<uint32_t> fin(5ull << lb);
        c.download(fin.data(), O.cur.data, fin.size() * 4);
        const uint64_t frows = (1ull << lb) >> 1;
        auto elem = [&](uint64_t idx) { Ext5 e; for (int q = 0; q < 5; q++) e.c[q] = Fp::raw(fin[((idx & 1) * 5 + q) * frows + (idx >> 1)]); return e; };
        const Ext5 fp0 = elem(0);
        for (uint64_t q = 1; q < (1ull << lb); q++) if (elem(q) != fp0) throw std::runtime_error("sharded fri: final polynomial is not constant");
        sp_ext_to_canonical(fp0, O.fpw);
        if (fri.observe_final_poly) R.ch->observe_ext(fp0);
        O.pow_witness = R.p->grind(*R.ch);
        O.ro.clear();
    }

    // ---- queries: every rank lays out the same tail and fills in what it holds; the tails are OR-ed
    size_t tail_words = 0;
    for (int k = 0; k < NH; k++) {
        Rank& R = rk[k];
        DeviceCtx& c = *R.c;
        c.activate();
        Opening& O = op[k];
        std::vector<uint64_t> indices(fri.num_queries);
        for (auto& ix : indices) ix = R.ch->sample_bits(log_max);
        const size_t NQ = indices.size(), NL = O.layers.size();
        std::vector<uint32_t> desc;
        std::vector<std::pair<uint32_t, uint32_t>> fix;
        uint32_t pos = 0;
        auto host_word = [&](uint32_t v) { fix.emplace_back(pos++, v); };
        auto gather = [&](const uint32_t* src, uint64_t stride, uint32_t count, uint32_t kind) {
            const uint64_t pv = (uint64_t)src;
            const uint32_t d6[6] = {(uint32_t)pv, (uint32_t)(pv >> 32), (uint32_t)stride, (uint32_t)(stride >> 32), count | (kind << 28), pos};
            desc.insert(desc.end(), d6, d6 + 6);
            pos += count;
        };
        auto skip = [&](uint32_t count) { pos += count; };  // held by another rank: stays zero here
        auto gather_path = [&](const ShTree& t, uint64_t idx) {
            host_word(t.log_total);
            if (!t.sharded) {
                for (unsigned l = 0; l < t.log_total; l++) gather(t.tree.layers[l].data + 8 * ((idx >> l) ^ 1), 1, 8, 1);
                return;
            }
            const uint64_t owner = idx >> t.log_local, loc = idx & ((1ull << t.log_local) - 1);
            for (unsigned l = 0; l < t.log_local; l++) {
                if ((int)owner == R.rank) gather(t.tree.layers[l].data + 8 * ((loc >> l) ^ 1), 1, 8, 1); else skip(8);
            }
            for (unsigned l = t.log_local; l < t.log_total; l++) {
                const uint32_t* dg = &t.top[l - t.log_local][8 * ((idx >> l) ^ 1)];
                for (int q = 0; q < 8; q++) host_word(dg[q]);
            }
        };
        host_word((uint32_t)O.commits.size());
        for (auto& r : O.commits) for (uint32_t w : r) host_word(w);
        host_word((uint32_t)NQ);
        for (size_t q = 0; q < NQ; q++) {
            const uint64_t index = indices[q];
            host_word((uint32_t)NL);
            for (size_t l = 0; l < NL; l++) {
                const FriLayer& ly = O.layers[l];
                const uint64_t idx_i = index >> l, sib = idx_i ^ 1, pair = idx_i >> 1, half = ly.len >> 1;
                if (!ly.sharded) gather(ly.buf.data + (5 * (sib & 1)) * half + pair, half, 5, 0);
                else {
                    const uint64_t prow = half / W, owner = pair / prow, loc = pair % prow;
                    if ((int)owner == R.rank) gather(ly.buf.data + (5 * (sib & 1)) * prow + loc, prow, 5, 0); else skip(5);
                }
                gather_path(ly.t, pair);
            }
        }
        for (int q = 0; q < 5; q++) host_word(O.fpw[q]);
        host_word(O.pow_witness);
        host_word((uint32_t)NQ);
        for (size_t q = 0; q < NQ; q++) {
            const uint64_t index = indices[q];
            host_word((uint32_t)NR);
            for (size_t r = 0; r < NR; r++) {
                ShRound& rs = round_of(R, r);
                const uint64_t idx_r = index >> (log_max - rs.t.log_total);
                host_word((uint32_t)rs.mats.size());
                for (auto& m : rs.mats) {
                    const unsigned lh = vg::log2_strict_u64(m.L);
                    const uint64_t row = idx_r >> (rs.t.log_total - lh);
                    host_word((uint32_t)m.width);
                    if (!m.big) gather(m.lde.data + row, m.L, (uint32_t)m.width, 0);
                    else {
                        const uint64_t rows = m.L / W, owner = row / rows, loc = row % rows;
                        if ((int)owner == R.rank) gather(m.shard.data + loc, rows, (uint32_t)m.width, 0); else skip((uint32_t)m.width);
                    }
                }
                gather_path(rs.t, idx_r);
            }
        }
        const size_t tw = pos;
        if (k == 0) tail_words = tw; else if (tw != tail_words) throw std::logic_error("sharded open: ranks disagree on the layout of the proof");
        DBuf gd(&c, desc), gout(&c, tw + 4);
        VG_HIP_CHECK(hipMemsetAsync(gout.data, 0, (tw + 4) * 4, c.stream));
        vk::launch_gather(c.stream, gd.data, desc.size() / 6, gout.data);
        c.check_launch("query gather");
        O.tail.assign(tw, 0);
        c.download(O.tail.data(), gout.data, tw * 4);
        for (auto& fx : fix) O.tail[fx.first] = fx.second;
    }
    {
        std::vector<const uint32_t*> contrib;
        for (auto& O : op) contrib.push_back(O.tail.data());
        std::vector<uint32_t> all;
        f.all_gather(contrib, tail_words, all);
        for (auto& O : op) {
            for (size_t x = 0; x < tail_words; x++) {
                uint32_t v = 0;
                for (int r = 0; r < W; r++) v |= all[(size_t)r * tail_words + x];
                O.tail[x] = v;
            }
        }
    }

    // ---------------------------------------------------------------------------------------------------------------------------
    // assemble MachineProof (flat "VPF1" words; machine/src/proof.rs:13-44, App. B12) on every rank
    // ---------------------------------------------------------------------------------------------------------------------------
    for (int k = 0; k < NH; k++) {
        Rank& R = rk[k];
        Opening& O = op[k];
        std::vector<uint32_t>& pw = R.words;
        pw.push_back(PROOF_MAGIC);
        pw.push_back((uint32_t)NC);
        for (int q = 0; q < 8; q++) pw.push_back(R.main_rs.t.root[q]);
        for (int q = 0; q < 8; q++) pw.push_back(R.perm_rs.t.root[q]);
        for (int q = 0; q < 8; q++) pw.push_back(R.quot_rs.t.root[q]);
        auto put_vec = [&](const std::vector<Ext5>& v) {
            pw.push_back((uint32_t)v.size());
            for (auto& x : v) { uint32_t w5[5]; sp_ext_to_canonical(x, w5); pw.insert(pw.end(), w5, w5 + 5); }
        };
        for (size_t i = 0; i < NC; i++) {
            pw.push_back(log_deg[i]);
            put_vec(O.opened[0][i][0]); put_vec(O.opened[0][i][1]);
            put_vec(O.opened[1][i][0]); put_vec(O.opened[1][i][1]);
            put_vec(O.opened[2][i][0]);
            uint32_t w5[5]; sp_ext_to_canonical(R.cumulative_sums[i], w5); pw.insert(pw.end(), w5, w5 + 5);
        }
        pw.insert(pw.end(), O.tail.begin(), O.tail.end());
        R.c->activate();
        R.c->sync();
        R.c->profiler.collect();
    }
    for (int k = 1; k < NH; k++) if (rk[k].words != rk[0].words) throw std::runtime_error("sharded prove: the ranks assembled different proofs");
    return rk[0].words;
}

}  // namespace vhost
