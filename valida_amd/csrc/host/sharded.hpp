// One commitment round of ONE proof sharded over W GPUs (intra-proof sharding, for traces beyond a single
// GPU's appetite).  The reference commits all chips' LDEs under one Merkle tree per round (basic/src/lib.rs:199,223,258,599), so
// a sharded pcs.commit_batches has to produce that very root:
//   1. COLUMN shards.  Global column g (columns of all matrices numbered in commit order) belongs to rank g mod W; every NTT is
//      per column, so each rank extends its own columns with no exchange (coset_lde, pcs.hpp).
//   2. ALL-TO-ALL.  A leaf digest absorbs a whole row, so the tree wants ROW-RANGE shards: rank s receives rows
//      [s L_i / W, (s+1) L_i / W) of every column of every matrix with L_i >= W rows — 4 b E (W-1) / W^2 bytes per rank over xGMI.
//   3. SUBTREES.  Storage rows are in committed (bit-reversed) order and the tree pairs adjacent storage rows, so an aligned row
//      range is a proper subtree: rank s hashes its rows (the kernels take a column POINTER table, the received segments are used
//      where they landed) and builds the tree over them, shorter matrices injected at their layer as usual.
//   4. ROOTS.  All-gather of the W subtree roots (32 W bytes); every rank finishes the top log2 W levels, injecting the matrices
//      with fewer than W rows (their tiny LDEs are computed redundantly by every rank).
// The phases are written once; the exchange is either RCCL (one process per GPU: Comm, comm.hpp) or, on a box with one GPU,
// device-to-device copies between W prover contexts standing in for the ranks (commit_sharded_local) — same phases, same root.
#pragma once
#include "comm.hpp"
#include "pcs.hpp"

namespace vhost {

struct ShardRank {
    DeviceCtx* c = nullptr;
    int rank = 0, W = 1;
    struct Mat {
        uint64_t L = 0, width = 0;        // LDE height, total width
        bool big = false;                 // L >= W: column-sharded and exchanged; else replicated
        std::vector<uint64_t> own;        // columns this rank extends (all of them when !big)
        DMat lde;                         // L x own.size(), committed row order
        uint64_t col_base = 0;            // global index of column 0
    };
    std::vector<Mat> mats;
    std::vector<DBuf> sendbuf, recvbuf;   // per peer; (matrix, owned column) segments in ascending order, L_i / W words each
    std::vector<size_t> send_words, recv_words;
    DeviceTree subtree;
    uint32_t subtree_root[8] = {0};

    static int owner(uint64_t global_col, int W) { return (int)(global_col % (uint64_t)W); }

    // phase 1: LDE of this rank's columns.  nat[i]: natural-order evaluations (column-major working layout) of matrix i
    void extend_own_columns(const std::vector<const DMat*>& nat, const std::vector<Fp>* coset_shifts, const FriParams& fri) {
        const Fp g = Fp::from_canonical(vg::GENERATOR);
        uint64_t base = 0;
        mats.resize(nat.size());
        for (size_t i = 0; i < nat.size(); i++) {
            Mat& m = mats[i];
            m.L = nat[i]->height << fri.log_blowup; m.width = nat[i]->width; m.col_base = base; m.big = m.L >= (uint64_t)W;
            base += m.width;
            for (uint64_t col = 0; col < m.width; col++) if (!m.big || owner(m.col_base + col, W) == rank) m.own.push_back(col);
            if (m.own.empty()) continue;
            DMat mine(c, nat[i]->height, m.own.size());
            for (size_t k = 0; k < m.own.size(); k++)
                VG_HIP_CHECK(hipMemcpyAsync(mine.data + k * mine.height, nat[i]->data + m.own[k] * nat[i]->height, mine.height * 4, hipMemcpyDeviceToDevice, c->stream));
            CommitInput in{&mine, false, false};
            m.lde = coset_lde(c, c->stream, in, fri.log_blowup, coset_shifts ? g * (*coset_shifts)[i].inv() : g);
        }
    }
    // how many words rank `from` sends to rank `to` (both sides compute it from the shapes alone)
    size_t words_between(int from, int to) const {
        (void)to;
        size_t n = 0;
        for (auto& m : mats)
            if (m.big) for (uint64_t col = 0; col < m.width; col++) if (owner(m.col_base + col, W) == from) n += m.L / W;
        return n;
    }
    // phase 2a: pack the row range of every peer
    void pack() {
        sendbuf.clear(); send_words.assign(W, 0);
        recvbuf.clear(); recv_words.assign(W, 0);
        for (int s = 0; s < W; s++) {
            send_words[s] = words_between(rank, s);
            recv_words[s] = words_between(s, rank);
            sendbuf.emplace_back(c, send_words[s] + 4);
            recvbuf.emplace_back(c, recv_words[s] + 4);
            size_t pos = 0;
            for (auto& m : mats) {
                if (!m.big) continue;
                const uint64_t rows = m.L / W;
                for (size_t k = 0; k < m.own.size(); k++, pos += rows)
                    VG_HIP_CHECK(hipMemcpyAsync(sendbuf[s].data + pos, m.lde.data + k * m.L + (uint64_t)s * rows, rows * 4, hipMemcpyDeviceToDevice, c->stream));
            }
        }
    }
    // phase 3: the subtree over this rank's row range, columns read where the exchange left them
    void build_subtree() {
        std::vector<ColMat> cms;
        std::vector<size_t> pos(W, 0);
        for (auto& m : mats) {
            if (!m.big) continue;
            const uint64_t rows = m.L / W;
            ColMat cm;
            cm.height = rows;
            for (uint64_t col = 0; col < m.width; col++) {
                const int src = owner(m.col_base + col, W);
                cm.cols.push_back(recvbuf[src].data + pos[src]);
                pos[src] += rows;
            }
            cms.push_back(std::move(cm));
        }
        if (cms.empty()) throw std::invalid_argument("sharded commit: every matrix is shorter than the number of ranks");
        subtree.build_cols(c, cms);
        memcpy(subtree_root, subtree.root, 32);
    }
    // phase 4: the top log2 W levels over the gathered subtree roots (rank-major, 8 words each), small matrices injected
    void finish_top(const std::vector<uint32_t>& roots, uint32_t root[8]) {
        if (W == 1) { memcpy(root, roots.data(), 32); return; }
        DBuf prev(c, roots);
        vk::KeccakTopArgs top{};
        top.prev = prev.data; top.first_len = (uint64_t)W / 2; top.levels = 0;
        std::vector<DBuf> layers;
        std::vector<uint64_t> ptrs;
        std::vector<std::pair<size_t, size_t>> inj;  // per level: (first, count) in ptrs
        for (uint64_t len = (uint64_t)W / 2; len >= 1; len /= 2) {
            size_t first = ptrs.size();
            for (auto& m : mats)
                if (!m.big && m.L == len) for (uint64_t col = 0; col < m.width; col++) ptrs.push_back((uint64_t)(m.lde.data + col * m.L));
            inj.push_back({first, ptrs.size() - first});
            layers.emplace_back(c, (size_t)len * 8);
            if (len == 1) break;
        }
        DBuf ptr_buf(c, ptrs.size() * 2 + 4);
        if (!ptrs.empty()) c->upload(ptr_buf.data, ptrs.data(), ptrs.size() * 8);
        const uint32_t* const* pd = (const uint32_t* const*)ptr_buf.data;
        for (size_t l = 0; l < layers.size(); l++) {
            top.out[l] = layers[l].data;
            top.cols[l] = inj[l].second ? pd + inj[l].first : nullptr;
            top.n_elems[l] = (int)inj[l].second;
            top.levels++;
        }
        if (top.levels > vk::KECCAK_TOP_MAX_LEVELS) throw std::invalid_argument("sharded commit: too many ranks");
        if (c->hash_kind == 1) vk::launch_poseidon_top(c->stream, c->poseidon_tab, c->poseidon_sparse, top); else vk::launch_keccak_top(c->stream, top);
        c->check_launch("sharded top");
        c->download_small(root, layers.back().data, 32);
    }
};

// One rank's share over RCCL (one process per GPU).
inline void commit_sharded_rccl(Comm& comm, const std::vector<const DMat*>& nat, const std::vector<Fp>* shifts, const FriParams& fri, uint32_t root[8]) {
    ShardRank r;
    r.c = comm.ctx; r.rank = comm.rank; r.W = comm.world;
    if (r.W & (r.W - 1)) throw std::invalid_argument("sharded commit: the number of ranks must be a power of two");
    r.c->activate();
    r.extend_own_columns(nat, shifts, fri);
    r.pack();
    std::vector<const uint32_t*> sp; std::vector<uint32_t*> rp;
    for (int s = 0; s < r.W; s++) { sp.push_back(r.sendbuf[s].data); rp.push_back(r.recvbuf[s].data); }
    comm.all_to_all_words(sp, r.send_words, rp, r.recv_words);
    r.build_subtree();
    std::vector<uint32_t> roots(8 * (size_t)r.W);
    comm.all_gather_words(r.subtree_root, 8, roots.data());
    r.finish_top(roots, root);
}

// The same phases with W prover contexts of ONE process standing in for the ranks (a box with one GPU, or several GPUs of one
// process with peer access): the exchange is a device-to-device copy per (source, destination) pair.
inline void commit_sharded_local(const std::vector<DeviceCtx*>& ctxs, const std::vector<std::vector<const DMat*>>& nat, const std::vector<Fp>* shifts,
                                 const FriParams& fri, uint32_t root[8]) {
    const int W = (int)ctxs.size();
    if (W < 1 || (W & (W - 1))) throw std::invalid_argument("sharded commit: the number of ranks must be a power of two");
    std::vector<ShardRank> rk(W);
    for (int r = 0; r < W; r++) {
        rk[r].c = ctxs[r]; rk[r].rank = r; rk[r].W = W;
        ctxs[r]->activate();
        rk[r].extend_own_columns(nat[r], shifts, fri);
        rk[r].pack();
        ctxs[r]->sync();
    }
    for (int r = 0; r < W; r++)       // the all-to-all
        for (int s = 0; s < W; s++)
            if (rk[r].send_words[s]) VG_HIP_CHECK(hipMemcpy(rk[s].recvbuf[r].data, rk[r].sendbuf[s].data, rk[r].send_words[s] * 4, hipMemcpyDeviceToDevice));
    VG_HIP_CHECK(hipDeviceSynchronize());  // device-to-device copies may return before they complete
    std::vector<uint32_t> roots(8 * (size_t)W);
    for (int r = 0; r < W; r++) {
        ctxs[r]->activate();
        rk[r].build_subtree();
        memcpy(&roots[8 * r], rk[r].subtree_root, 32);  // the all-gather
    }
    uint32_t first[8];
    for (int r = 0; r < W; r++) {
        ctxs[r]->activate();
        uint32_t got[8];
        rk[r].finish_top(roots, got);
        if (r == 0) memcpy(first, got, 32);
        else if (memcmp(first, got, 32)) throw std::runtime_error("sharded commit: ranks disagree on the root");
    }
    memcpy(root, first, 32);
}

}  // namespace vhost
