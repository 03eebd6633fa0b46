// Device PCS: the GPU counterpart of `TwoAdicFriPcs` as Valida drives it through
// UnivariatePcsWithLde (machine/src/config.rs:17-22):
//   commit_batches / commit_shifted_batches   basic/src/lib.rs:199,223,258,599
//   get_ldes                                  basic/src/lib.rs:201,225,261   (zero-copy: the LDEs stay in HBM)
//   coset_shift / log_blowup                  basic/src/lib.rs:594, machine/src/quotient.rs:41,98
// Conventions: SURVEY.md App. B3-B5.  Opening (open_multi_batches) lives in prover.cpp because it is
// interleaved with the host transcript.
#pragma once
#include <algorithm>
#include <functional>
#include <memory>
#include "runtime.hpp"

namespace vhost {

struct FriParams {
    unsigned log_blowup = 1, num_queries = 40, pow_bits = 8;
    bool observe_final_poly = false;
    bool interpret_air = false;  // quotient: force the register-program interpreter even for the in-tree chips
    int hash_kind = 0;           // MMCS hash: 0 Keccak-256 (reference), 1 Poseidon-16 sponge / truncated permutation (north-star variant)
};

// A committed matrix given by one device pointer per column (height words each): what the sharded commit hashes — its columns
// arrive from different ranks and are used where they landed.
struct ColMat {
    uint64_t height = 0;
    std::vector<const uint32_t*> cols;
};

// One Merkle tree over column-major device matrices of mixed heights (FieldMerkleTreeMmcs, App. B5).
struct DeviceTree {
    static constexpr uint64_t TOP_FIRST_LEN = 256;
    DeviceCtx* ctx = nullptr;
    std::vector<DBuf> layers;          // layers[i]: (max_height >> i) digests of 8 words; back() = root
    std::vector<uint64_t> layer_len;
    unsigned log_max_height = 0;
    uint32_t root[8] = {0};            // canonical
    // drop_bottom (set before build): a tree of at least DROP_MIN_LEAVES leaves built on the context's main stream returns its leaf layer — and the layer
    // above it, unless that one injects rows — to the pool as soon as the layers above are enqueued (half / three quarters of the tree's digests; the pool
    // hands a block only to work enqueued later on the same stream).  `dropped` layers have no buffer; the query phase recomputes their 40 sibling digests
    // from the committed rows (vk::launch_*_bottom_q), which `leaf_ptr / leaf_stride / leaf_elems` describe: a column-pointer table (stride 0, kept alive
    // in ptr_table_) or the one strided matrix of a FRI layer.  Only Prover::open_multi_batches reads such trees; the sharded prover keeps every layer.
    static constexpr uint64_t DROP_MIN_LEAVES = 1ull << 16;
    bool drop_bottom = false;
    unsigned dropped = 0;
    const void* leaf_ptr = nullptr;
    uint64_t leaf_stride = 0;
    int leaf_elems = 0;

    // mats: views in commit order.  Enqueues all kernels; root is read back (sync) at the end.
    // fetch_root = false leaves the root on the device only (layers.back()): the FRI commit phase consumes it there.
    // before_injection: called once, just before the first kernel that reads a matrix SHORTER than the tallest ones is enqueued (a layer
    // that injects rows) — commit_batches joins the stream that extends those matrices there, so the leaves and the first layers of the
    // tree (Keccak: integer-VALU work) run beside the remaining LDEs (half memory phases) instead of after them.
    // challenger: the FRI commit phase's DuplexChallenger step as an epilogue of the tree-top launch (KeccakTopArgs::ch_*), or null
    struct TopChallenger { const uint32_t* pos; uint32_t* state; uint32_t* beta5; uint32_t* commit8; };
    // on: the stream the tree's kernels are enqueued on (null = the context's main stream).  A tree built on the auxiliary stream keeps its root
    // on the device (fetch_root = false) and its pointer table alive in the tree.
    void build(DeviceCtx* c, const std::vector<vk::DMatView>& mats, bool fetch_root = true, const std::function<void()>* before_injection = nullptr,
               const TopChallenger* challenger = nullptr, hipStream_t on = nullptr) {
        // reset on every exit (a throwing build_impl included): the pointers refer to the caller's stack
        struct Reset { DeviceTree* t; ~Reset() { t->before_injection_ = nullptr; t->challenger_ = nullptr; t->on_ = nullptr; } } reset{this};
        before_injection_ = before_injection;
        challenger_ = challenger;
        on_ = on;
        if (on && on != c->stream && fetch_root) throw std::logic_error("mmcs: a tree on another stream leaves its root on the device");
        std::vector<ColMat> cms(mats.size());
        for (size_t i = 0; i < mats.size(); i++) {
            cms[i].height = mats[i].height;
            for (uint64_t col = 0; col < mats[i].width; col++) cms[i].cols.push_back(mats[i].data + col * mats[i].stride);
        }
        build_impl(c, cms, mats.size() == 1 ? &mats[0] : nullptr, fetch_root);
    }
    void build_cols(DeviceCtx* c, const std::vector<ColMat>& mats) { before_injection_ = nullptr; challenger_ = nullptr; on_ = nullptr; build_impl(c, mats, nullptr, true); }

  private:
    const std::function<void()>* before_injection_ = nullptr;
    const TopChallenger* challenger_ = nullptr;
    hipStream_t on_ = nullptr;
    DBuf ptr_table_;  // the column-pointer table of a tree whose root stays on the device (its kernels may still be queued when build returns)
    // single_view: the tree is over ONE strided matrix (every FRI layer tree): no pointer table, columns are base + k * stride
    void build_impl(DeviceCtx* c, const std::vector<ColMat>& cms, const vk::DMatView* single_view, bool fetch_root) {
        ctx = c;
        std::vector<size_t> order(cms.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cms[a].height > cms[b].height; });
        // column-pointer lists per height group, uploaded in one transfer
        struct Group { uint64_t height; size_t first, count; };
        std::vector<Group> groups;
        std::vector<uint64_t> ptrs;
        for (size_t pos = 0; pos < order.size();) {
            uint64_t h = cms[order[pos]].height;
            Group g{h, ptrs.size(), 0};
            while (pos < order.size() && cms[order[pos]].height == h)
                for (const uint32_t* col : cms[order[pos++]].cols) ptrs.push_back((uint64_t)col);
            g.count = ptrs.size() - g.first;
            groups.push_back(g);
        }
        const bool single = single_view != nullptr;

        const hipStream_t st = on_ ? on_ : c->stream;
        DBuf ptr_buf;
        if (!single) {
            ptr_buf = DBuf(c, ptrs.size() * 2);
            c->upload_async(ptr_buf.data, ptrs.data(), ptrs.size() * 8);
            if (st != c->stream) {  // the upload travels on the main stream: the kernels on `st` wait for it
                VG_HIP_CHECK(hipEventRecord(c->rider_ev, c->stream));
                VG_HIP_CHECK(hipStreamWaitEvent(st, c->rider_ev, 0));
            }
        }
        const uint32_t* const* pd = (const uint32_t* const*)ptr_buf.data;

        uint64_t maxh = groups[0].height;
        log_max_height = vg::log2_strict_u64(maxh);
        layers.clear(); layer_len.clear();
        layers.emplace_back(c, (size_t)maxh * 8);
        layer_len.push_back(maxh);
        const bool pos = c->hash_kind == 1;
        const uint32_t* tab = c->poseidon_tab;
        bool leaves_in_top = false, leaves_in_mid = false;
        if (pos) {
            if (single) vk::launch_poseidon_leaves_strided(st, tab, c->poseidon_sparse, single_view->data, single_view->stride, (int)single_view->width, maxh, layers[0].data);
            else vk::launch_poseidon_leaves(st, tab, c->poseidon_sparse, pd + groups[0].first, (int)groups[0].count, maxh, layers[0].data);
        } else if (single && vk::keccak_top_takes_leaves(maxh)) leaves_in_top = true;  // the small FRI layers: leaves, levels and challenger step in ONE launch
        else if (single && maxh / 2 > TOP_FIRST_LEN && vk::keccak_levels_take_leaves(maxh)) leaves_in_mid = true;  // the middle FRI layers: leaves + the layers down to 512 parents
        else if (single) vk::launch_keccak_leaves_strided(st, single_view->data, single_view->stride, (int)single_view->width, maxh, layers[0].data);
        else vk::launch_keccak_leaves(st, pd + groups[0].first, (int)groups[0].count, maxh, layers[0].data);
        size_t gi = 1;
        // mid: the consecutive layers of 256 < parents <= 32768 (at most seven) as ONE launch of 64-parent workgroups; top: the rest in one workgroup
        vk::KeccakTopArgs top{}, mid{}, pgrp{};  // pgrp: the Poseidon tree's current group of row-parallel layers (launch_poseidon_levels)
        int pgroup = -1;
        auto flush_pos = [&](bool ends_at_root) {
            if (!pgrp.levels) return;
            if (ends_at_root && challenger_) { pgrp.ch_pos = challenger_->pos; pgrp.ch_state = challenger_->state; pgrp.ch_beta5 = challenger_->beta5; pgrp.ch_commit8 = challenger_->commit8; }
            vk::launch_poseidon_levels(st, tab, c->poseidon_sparse, pgrp);
            pgrp.levels = 0;
        };
        auto add_level = [&](vk::KeccakTopArgs& g, uint64_t len, const Group* inj) {
            if (g.levels == 0) { g.prev = layers[layers.size() - 2].data; g.first_len = len; }
            g.out[g.levels] = layers.back().data;
            g.cols[g.levels] = inj ? pd + inj->first : nullptr;
            g.n_elems[g.levels] = inj ? (int)inj->count : 0;
            g.levels++;
        };
        auto flush_mid = [&] {
            if (!mid.levels) return;
            mid.block_len = vk::KECCAK_LEVELS_BLOCK_LEN;
            if (leaves_in_mid) {
                if (mid.prev != layers[0].data) throw std::logic_error("mmcs: leaf prologue without a launch over the leaf layer");
                mid.leaf_base = single_view->data; mid.leaf_stride = single_view->stride; mid.leaf_elems = (int)single_view->width; mid.leaf_rows = maxh;
            }
            vk::launch_keccak_levels(st, mid);
            mid.levels = 0;
        };
        for (uint64_t len = maxh / 2; len >= 1; len /= 2) {
            layers.emplace_back(c, (size_t)len * 8);
            layer_len.push_back(len);
            const Group* inj = (gi < groups.size() && groups[gi].height == len) ? &groups[gi] : nullptr;
            if (inj && before_injection_) { (*before_injection_)(); before_injection_ = nullptr; }
            // Layers of more than TOP_FIRST_LEN parents are spread over the whole GPU (the big ones a launch each); the rest of the tree is one
            // single-workgroup launch.  256, not the 1024 a workgroup could take: inside one workgroup a 1024-parent layer puts four
            // waves on each SIMD of ONE CU and costs 27 us (512 parents: 15 us), as a launch of its own across the CUs 9 + 2 us.
            if (pos) {  // big layers a thread per node, the rest in groups of five layers by 16-row workgroups (kernels/poseidon_mmcs.hip)
                if (!vk::poseidon_levels_take(len)) vk::launch_poseidon_compress(st, tab, c->poseidon_sparse, layers[layers.size() - 2].data, inj ? pd + inj->first : nullptr, inj ? (int)inj->count : 0, len, layers.back().data);
                else {
                    if (pgrp.levels && vk::poseidon_levels_group(len) != pgroup) flush_pos(false);
                    pgroup = vk::poseidon_levels_group(len);
                    add_level(pgrp, len, inj);
                }
            } else if (len > TOP_FIRST_LEN) {
                if (vk::keccak_levels_fused(len)) add_level(mid, len, inj);
                else vk::launch_keccak_compress(st, layers[layers.size() - 2].data, inj ? pd + inj->first : nullptr, inj ? (int)inj->count : 0, len,
                                                layers.back().data);
            } else {  // the last <= 9 layers go into one launch
                flush_mid();
                add_level(top, len, inj);
            }
            if (inj) gi++;
            if (len == 1) break;
        }
        flush_mid();
        if (leaves_in_mid && layers.size() < 2) throw std::logic_error("mmcs: leaf prologue on a tree without parents");
        if (leaves_in_top) {
            if (!top.levels || top.prev != layers[0].data) throw std::logic_error("mmcs: leaf prologue without a top launch over the leaf layer");
            top.leaf_base = single_view->data; top.leaf_stride = single_view->stride; top.leaf_elems = (int)single_view->width; top.leaf_rows = maxh;
        }
        flush_pos(true);
        if (challenger_ && !pos) {
            if (!top.levels) throw std::logic_error("mmcs: the challenger epilogue needs a tree with at least one parent layer");
            top.ch_pos = challenger_->pos; top.ch_state = challenger_->state; top.ch_beta5 = challenger_->beta5; top.ch_commit8 = challenger_->commit8;
        }
        if (top.levels) vk::launch_keccak_top(st, top);
        if (gi != groups.size()) throw std::runtime_error("mmcs: matrix heights must be powers of two >= 1 and <= max height");
        c->check_launch("mmcs build");
        dropped = 0;
        if (drop_bottom && st == c->stream && !c->in_section && maxh >= DROP_MIN_LEAVES && layers.size() >= 4) {
            leaf_elems = (int)groups[0].count;
            if (single) { leaf_ptr = single_view->data; leaf_stride = single_view->stride; }
            else { leaf_ptr = pd + groups[0].first; leaf_stride = 0; }
            layers[0] = DBuf();
            dropped = 1;
            if (!(groups.size() > 1 && groups[1].height == maxh / 2)) { layers[1] = DBuf(); dropped = 2; }
        }
        if (fetch_root) c->download_small(root, layers.back().data, 32);  // sync: also keeps ptr_buf alive until the kernels finished
        if (!fetch_root || dropped) ptr_table_ = std::move(ptr_buf);  // the kernels may still be queued / the query phase reads the table: it lives as long as the tree
    }

};

// ProverData of one commitment round: the committed (bit-reversed) LDEs + their tree.
struct ProverData {
    std::vector<DMat> ldes;  // commit order
    DeviceTree tree;
};

// One matrix handed to commit: column-major evaluations over H_n with rows in natural order, or already
// at bit-reversed positions (`rows_bitrev`).  `consume`: the buffer may be used as the coefficient
// scratch and released (only meaningful with rows_bitrev); otherwise it is left untouched.
struct CommitInput {
    DMat* mat;
    bool rows_bitrev;
    bool consume;
};

// Returns the bit-reversed LDE on lde_shift * H_{n << log_blowup}.
inline DMat coset_lde(DeviceCtx* c, hipStream_t st, const CommitInput& in, unsigned log_blowup, Fp lde_shift) {
    const uint64_t n = in.mat->height, b = 1ull << log_blowup;
    if (n == 0 || (n & (n - 1))) throw std::invalid_argument("commit: matrix heights must be powers of two");
    if (vg::log2_strict_u64(n) + log_blowup > 27)  // BabyBear's two-adicity: no larger multiplicative subgroup exists
        throw std::invalid_argument("commit: LDE height 2^" + std::to_string(vg::log2_strict_u64(n) + log_blowup) + " exceeds the field's two-adicity (2^27)");
    const unsigned k = vg::log2_strict_u64(n);
    // natural-order input: the fused three-pass pipeline (kernels/ntt.hip, k_lde_a / k_lde_mid / k_lde_c) — no bit-reversal copy, the
    // coefficients stay in LDS.  VGPU_LDE_FUSED=0 keeps the unfused passes (A/B runs).
    static const bool fused = [] { const char* e = getenv("VGPU_LDE_FUSED"); return !(e && e[0] == '0'); }();
    if (!in.rows_bitrev && fused) {
        DMat lde(c, n * b, in.mat->width);
        const vk::LdeTables lt = c->lde_tables((int)k, (int)log_blowup, lde_shift);
        // scratch of passes A -> MID (S1: n rows) and MID -> C (S2: b n rows).  (Column groups sized for the Infinity Cache with the scratch reused were
        // measured in round 5 and sped up no pass at any size: profiles/r05_ab_session1_standin_groups_quotient.txt.)
        DMat s1, s2;
        if (k > 12) { s1 = DMat(c, n, in.mat->width); s2 = DMat(c, n * b, in.mat->width); }
        vk::launch_lde_natural(st, in.mat->view(), lde.view(), (int)log_blowup, c->tables, lt, s1.empty() ? vk::DMatView{nullptr, 0, 0, 0} : s1.view(),
                               s2.empty() ? vk::DMatView{nullptr, 0, 0, 0} : s2.view());
        c->check_launch("coset_lde");
        return lde;  // s1 / s2 return to the pool while the kernels may still be queued: safe for the same reason as `coeffs` below
    }
    DMat coeffs;
    if (in.rows_bitrev && in.consume) coeffs = std::move(*in.mat);
    else {
        coeffs = DMat(c, n, in.mat->width);
        if (in.rows_bitrev) VG_HIP_CHECK(hipMemcpyAsync(coeffs.data, in.mat->data, n * in.mat->width * 4, hipMemcpyDeviceToDevice, st));
        else vk::launch_bitrev_rows(st, in.mat->view(), coeffs.view());
    }
    vk::launch_intt(st, coeffs.view(), c->tables);
    DMat lde(c, n * b, coeffs.width);
    Fp w = vg::two_adic_generator(k + log_blowup), wt = Fp::one();
    for (uint64_t t = 0; t < b; t++) {
        uint64_t block = vg::reverse_bits_len((uint32_t)t, log_blowup);
        vk::launch_coset_ntt(st, coeffs.view(), lde.view(), block * n, lde_shift * wt, c->tables);
        wt *= w;
    }
    c->check_launch("coset_lde");
    // `coeffs` returns to the pool when this function exits while kernels reading it may still be
    // queued: safe because the pool only ever hands a block to work enqueued LATER on the same stream.
    return lde;
}

// pcs.commit_shifted_batches (App. B4): lde_i on (31 / coset_shift_i) * H, bit-reversed rows, one MMCS.
// A second, independent and SMALL commitment (the preprocessed traces of a proof: range table and program ROM) issued on the auxiliary
// stream inside another round's commit: its one-tile LDEs and its tree — a chain of single-workgroup, latency-bound launches — run beside
// that round's big LDE passes instead of in front of them, and its root arrives with that round's own synchronisation.
struct CommitRider {
    const std::vector<CommitInput>* mats;
    std::unique_ptr<ProverData>* out;  // receives the rider's ProverData; its tree.root is valid when commit_batches returns
};
inline std::unique_ptr<ProverData> commit_batches(DeviceCtx* c, const std::vector<CommitInput>& mats, const std::vector<Fp>* coset_shifts,
                                                  const FriParams& fri, const CommitRider* rider = nullptr) {
    auto pd = std::make_unique<ProverData>();
    Fp g = Fp::from_canonical(vg::GENERATOR);
    // Per-matrix LDE pipelines are independent: the tallest matrices are extended first, on the main stream, the small ones on the auxiliary
    // stream, joined before the tree.  (Building the tree's first layers beside the remaining LDEs was measured in round 3 and bought nothing —
    // both sides are issue-bound, profiles/r03_ab_commit_overlap.json — and is gone.)
    uint64_t maxh = 0;
    for (auto& m : mats) maxh = std::max<uint64_t>(maxh, m.mat->height);
    Section lde_section(c);
    if (rider) {
        auto rp = std::make_unique<ProverData>();
        hipStream_t ax = c->aux[0];
        for (auto& m : *rider->mats) rp->ldes.push_back(coset_lde(c, ax, m, fri.log_blowup, g));
        std::vector<vk::DMatView> rviews;
        for (auto& l : rp->ldes) rviews.push_back(l.view());
        rp->tree.build(c, rviews, false, nullptr, nullptr, ax);
        VG_HIP_CHECK(hipMemcpyAsync(c->rider_root_pin, rp->tree.layers.back().data, 32, hipMemcpyDeviceToHost, ax));  // lands before the join below completes
        *rider->out = std::move(rp);
    }
    pd->ldes.resize(mats.size());
    for (int pass = 0; pass < 2; pass++)
        for (size_t i = 0; i < mats.size(); i++) {
            const bool tallest = mats[i].mat->height == maxh;
            if (tallest != (pass == 0)) continue;
            Fp shift = coset_shifts ? g * (*coset_shifts)[i].inv() : g;
            // (the big matrices below the tallest height on the auxiliary stream too — their LDE passes beside the tallest ones' — was measured in round 6:
            // a lone proof 20.14 vs 20.22 ms, three in flight 15.86 vs 15.70: profiles/r06_ab_lde_split.txt)
            hipStream_t st = c->stream_for(i, mats[i].mat->height);
            pd->ldes[i] = coset_lde(c, st, mats[i], fri.log_blowup, shift);
        }
    std::vector<vk::DMatView> views;
    for (auto& l : pd->ldes) views.push_back(l.view());
    lde_section.join();
    pd->tree.drop_bottom = true;  // (big trees only: DeviceTree::DROP_MIN_LEAVES)
    pd->tree.build(c, views);
    // The rider's root: the build above synchronised on the main stream AFTER it had waited for the auxiliary stream at the join; the auxiliary
    // stream is still synchronised here in its own right (it has long drained: microseconds).
    if (rider) {
        VG_HIP_CHECK(hipStreamSynchronize(c->aux[0]));
        memcpy((*rider->out)->tree.root, c->rider_root_pin, 32);
    }
    return pd;
}

}  // namespace vhost
