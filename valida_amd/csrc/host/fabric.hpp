// The exchanges of a proof sharded over W GPUs, behind one interface so that the sharded prover
// (sharded_prover.cpp) is written once: the driver walks the ranks THIS process hosts through the phases and calls the fabric
// between them.
//   * LocalFabric — all W ranks live in this process as W prover contexts (a box with one GPU: the 1-GPU realisation the parity
//     tests run; or several GPUs of one process): an exchange is a device-to-device copy per (source, destination) pair.
//   * RcclFabric  — one rank per process, one process per GPU (RCCL's own model): all-gather / grouped send-recv over xGMI through
//     the library's communicator (comm.hpp).
//   * CallbackFabric — one rank per process with the HOST'S OWN transport (vgpu_fabric_t of the C ABI: two function pointers over host
//     buffers): what a Rust host with its MPI / TCP / shared-memory layer plugs in, and what lets the sharded prover run one rank per
//     process on any box (tests: torch.distributed gloo between processes sharing one GPU).
//
// FAILURE PROTOCOL (multi-process fabrics).  A rank that throws alone would leave its peers blocked in the next collective for ever.
// Every collective therefore runs in FOUR steps:
//     stage      everything of the collective that can fail on this rank ALONE — sizing and (re)allocating the staging buffers, the
//                device-to-host copies of what it sends, the stream synchronisation.  Nothing collective has happened yet: a throw
//                here is an ordinary failure between two collectives.
//     agree      a one-word STATUS all-gather.  A rank that failed (here or anywhere since the last collective) contributes a non-zero
//                status instead of entering the collective (`fail`, called once by ShardedProof::run's catch-all); its peers read it and
//                throw FabricPeerFailure — every rank returns an error, none hangs.
//     transport  the exchange itself.  Only the transport can fail now (a callback's non-zero status, an RCCL error, the DEADLINE): that
//                is fatal for the fabric — it is marked `poisoned`, no further collective (not even the status word of `fail`) is
//                attempted on it, because the peers are inside THIS collective and a different one would be mismatched.  The peers
//                leave through the transport's own error or through the deadline.
//     finish     host-to-device copies of what arrived.  A throw here is again a failure between two collectives.
// DEADLINE.  `timeout_ms` (vgpu_fabric_t::timeout_ms, vgpu_comm_set_timeout_ms; 0 = none) bounds every transport step: a callback that has
// not returned in time is abandoned on its helper thread (with the buffers it may still write to), an RCCL operation that has not
// completed is aborted with ncclCommAbort; the rank throws FabricTransportFailure and the fabric is poisoned.  A peer that died —
// segfault, OOM kill — can therefore cost the survivors at most timeout_ms, not a hang.
// LocalFabric drives all ranks from one thread: nothing to agree on, nothing to time out.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>
#include "comm.hpp"

namespace vhost {

struct FabricPeerFailure : std::runtime_error {
    using std::runtime_error::runtime_error;
};
// the transport itself failed or ran into the deadline in the middle of a collective: the fabric is unusable from here on
struct FabricTransportFailure : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Test-only fault injection (documented in DESIGN.md §7): VGPU_FAILPOINT="<name>@<rank>" makes the named step throw on that rank, once
// per process — honoured only together with VGPU_TESTING=1 (a second, explicit opt-in: a stray variable in a production environment must
// not be able to fail proofs).  Names: fabric_stage (inside the all-to-all's staging, before the status round), fabric_finish (after the exchange);
// local_stage_copies@0 does not throw: it sends the first LocalFabric's exchanges through the host-staged path (tests/test_sharded_prove_gpu.py).
inline bool fabric_failpoint(const char* name, int rank) {
    static const std::string spec = [] {
        const char* t = getenv("VGPU_TESTING");
        const char* e = getenv("VGPU_FAILPOINT");
        return std::string(e && t && t[0] == '1' ? e : "");
    }();
    if (spec.empty()) return false;
    static bool fired = false;
    if (fired || spec != std::string(name) + "@" + std::to_string(rank)) return false;
    fired = true;
    return true;
}

struct Fabric {
    int world = 1;
    std::vector<int> hosted;  // the ranks living in this process, ascending
    bool peer_failed = false;
    bool poisoned = false;    // a transport step failed: no collective may follow on this fabric
    uint32_t timeout_ms = 0;  // deadline of one transport step (0: none)
    virtual ~Fabric() {}
    struct Seg;
    struct A2A;
    // contrib[k]: n words of hosted rank hosted[k] (host memory).  out: world * n words, rank-major — what every rank receives.
    void all_gather(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) {
        check_usable();
        ag_stage(n);
        agree();
        transport([&] { ag_transport(contrib, n, out); });
    }
    void all_to_all(std::vector<A2A>& plan) {
        check_usable();
        a2a_stage(plan);
        agree();
        transport([&] { a2a_transport(plan); });
        a2a_finish(plan);
    }
    // Status round before a collective: throws FabricPeerFailure on EVERY rank if any rank reported a failure.
    void agree() {
        if (!needs_agreement()) return;
        check_usable();
        const uint32_t ok = 0;
        std::vector<uint32_t> all;
        transport([&] { ag_transport({&ok}, 1, all); });
        for (int r = 0; r < world; r++)
            if (all[(size_t)r]) { peer_failed = true; throw FabricPeerFailure("sharded prove: rank " + std::to_string(r) + " failed (status " + std::to_string(all[(size_t)r]) + "); every rank gives up this proof"); }
    }
    // Called by a rank that cannot go on (never throws): its peers learn it in their next agree().  Not after a FabricPeerFailure — then the
    // peers have already left — and not on a poisoned fabric: the peers are inside the collective that broke, not in a status round.
    void fail(uint32_t status = 1) noexcept {
        if (!needs_agreement() || peer_failed || poisoned) return;
        try { std::vector<uint32_t> all; const uint32_t st = status ? status : 1u; transport([&] { ag_transport({&st}, 1, all); }); peer_failed = true; } catch (...) {}
    }
    virtual bool needs_agreement() const { return (int)hosted.size() < world; }

  protected:
    void check_usable() const {
        if (poisoned) throw FabricTransportFailure("fabric: an earlier exchange failed or timed out inside the transport; this fabric cannot be used again");
    }
    template <class F>
    void transport(F&& f) {
        try { f(); }
        catch (const FabricPeerFailure&) { throw; }
        catch (const std::exception& e) { poisoned = true; throw FabricTransportFailure(e.what()); }
        catch (...) { poisoned = true; throw; }
    }
    // stage: sizes / allocates whatever ag_transport needs for n words per rank (and one status word); may throw freely
    virtual void ag_stage(size_t n) { (void)n; }
    virtual void ag_transport(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) = 0;
    virtual void a2a_stage(std::vector<A2A>& plan) { (void)plan; }
    virtual void a2a_transport(std::vector<A2A>& plan) = 0;
    virtual void a2a_finish(std::vector<A2A>& plan) { (void)plan; }

  public:
    // Device all-to-all.  plan[k] belongs to hosted rank hosted[k]: the segments send[s] go to rank s, the segments recv[s] arrive from rank s,
    // in order.  A segment is `n_runs` runs of `run` words, `pitch` words apart (a run per column of a column-major matrix: the row range a
    // peer needs of every column this rank owns) — the exchanges move the data ONCE, from where a kernel left it to where the next kernel
    // reads it; there is no pack / unpack copy around them (the RCCL fabric packs inside, the callback fabric stages through host memory
    // anyway).  A sender's k-th segment to s and the receiver's k-th segment from it have the same shape.  Returns when the data has arrived
    // (the contexts' streams are drained).
    struct Seg {
        uint32_t* p = nullptr;
        size_t run = 0, n_runs = 0, pitch = 0;
        size_t words() const { return run * n_runs; }
    };
    struct A2A {
        DeviceCtx* c = nullptr;
        std::vector<std::vector<Seg>> send, recv;
        explicit A2A(DeviceCtx* ctx = nullptr, int world = 0) : c(ctx), send(world), recv(world) {}
        void add_send(int to, const uint32_t* p, size_t run, size_t n_runs = 1, size_t pitch = 0) { if (run && n_runs) send[(size_t)to].push_back({const_cast<uint32_t*>(p), run, n_runs, n_runs > 1 ? pitch : run}); }
        void add_recv(int from, uint32_t* p, size_t run, size_t n_runs = 1, size_t pitch = 0) { if (run && n_runs) recv[(size_t)from].push_back({p, run, n_runs, n_runs > 1 ? pitch : run}); }
        size_t send_words(int to) const { size_t n = 0; for (auto& g : send[(size_t)to]) n += g.words(); return n; }
        size_t recv_words(int from) const { size_t n = 0; for (auto& g : recv[(size_t)from]) n += g.words(); return n; }
    };
    // one segment <-> a contiguous block, or segment -> segment of the same shape (device to device)
    static void copy_seg(void* dst, size_t dst_pitch_words, const void* src, size_t src_pitch_words, const Seg& g, hipMemcpyKind kind, hipStream_t st) {
        if (g.n_runs == 1) VG_HIP_CHECK(hipMemcpyAsync(dst, src, g.run * 4, kind, st));
        else VG_HIP_CHECK(hipMemcpy2DAsync(dst, dst_pitch_words * 4, src, src_pitch_words * 4, g.run * 4, g.n_runs, kind, st));
    }
    static void copy_matching(const std::vector<Seg>& from, const std::vector<Seg>& to, hipStream_t st) {
        if (from.size() != to.size()) throw std::logic_error("fabric: send / receive segment lists disagree");
        for (size_t k = 0; k < from.size(); k++) {
            if (from[k].run != to[k].run || from[k].n_runs != to[k].n_runs) throw std::logic_error("fabric: send / receive segments of different shapes");
            copy_seg(to[k].p, to[k].pitch, from[k].p, from[k].pitch, from[k], hipMemcpyDeviceToDevice, st);
        }
    }
};

struct LocalFabric : Fabric {
    explicit LocalFabric(int w) { world = w; for (int r = 0; r < w; r++) hosted.push_back(r); }
    ~LocalFabric() override { if (stage_host) (void)hipHostFree(stage_host); }
    // Contexts on DIFFERENT devices (vgpu_prove_sharded_local over several GPUs of one process): the strided device-to-device copies below need
    // peer access between the two devices.  Decided ONCE per fabric (first all-to-all), outside the copy loops: every ordered pair of distinct
    // devices is either connected (hipDeviceEnablePeerAccess; process-wide, so "already enabled" is fine) or marked STAGED — its copies go
    // through page-locked host memory, with one warning on stderr (a slow path that says so; one process per GPU, vgpu_prove_sharded, avoids it).
  private:
    bool peers_ready = false;
    bool stage_all = false;  // test hook (VGPU_TESTING=1 VGPU_FAILPOINT=local_stage_copies@0): EVERY exchange of this fabric through the staged path — the one way a 1-GPU box can run it
    std::vector<std::pair<int, int>> staged_pairs;  // (reader device, owner device) without a peer path
    uint32_t* stage_host = nullptr;
    size_t stage_words = 0;
    void prepare_peers(const std::vector<A2A>& plan) {
        if (peers_ready) return;
        stage_all = fabric_failpoint("local_stage_copies", 0);
        if (stage_all) fprintf(stderr, "vgpu: fabric: test hook local_stage_copies: every exchange of this fabric is staged through host memory\n");
        std::vector<int> devs;
        for (auto& p : plan) if (std::find(devs.begin(), devs.end(), p.c->device) == devs.end()) devs.push_back(p.c->device);
        for (int dev : devs)
            for (int peer : devs) {
                if (dev == peer) continue;
                int can = 0;
                VG_HIP_CHECK(hipDeviceCanAccessPeer(&can, dev, peer));
                if (can) {
                    VG_HIP_CHECK(hipSetDevice(dev));
                    const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) throw std::runtime_error(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                    (void)hipGetLastError();
                } else {
                    staged_pairs.emplace_back(dev, peer);
                    fprintf(stderr, "vgpu: fabric: device %d has no peer path to device %d: exchanges between them are staged through host memory (slow); "
                                    "one process per GPU (vgpu_prove_sharded) uses RCCL instead\n", dev, peer);
                }
            }
        peers_ready = true;
    }
    bool staged(int reader, int owner) const {
        return stage_all || (reader != owner && std::find(staged_pairs.begin(), staged_pairs.end(), std::make_pair(reader, owner)) != staged_pairs.end());
    }
    void copy_staged(const std::vector<Seg>& from, int from_dev, const std::vector<Seg>& to, int to_dev) {
        if (from.size() != to.size()) throw std::logic_error("fabric: send / receive segment lists disagree");
        for (size_t k = 0; k < from.size(); k++) {
            if (from[k].run != to[k].run || from[k].n_runs != to[k].n_runs) throw std::logic_error("fabric: send / receive segments of different shapes");
            if (from[k].words() > stage_words) {
                if (stage_host) VG_HIP_CHECK(hipHostFree(stage_host));
                stage_host = nullptr;
                stage_words = from[k].words();
                VG_HIP_CHECK(hipHostMalloc((void**)&stage_host, stage_words * 4, hipHostMallocPortable));
            }
            VG_HIP_CHECK(hipSetDevice(from_dev));
            copy_seg(stage_host, from[k].run, from[k].p, from[k].pitch, from[k], hipMemcpyDeviceToHost, nullptr);
            VG_HIP_CHECK(hipStreamSynchronize(nullptr));
            VG_HIP_CHECK(hipSetDevice(to_dev));
            copy_seg(to[k].p, to[k].pitch, stage_host, from[k].run, from[k], hipMemcpyHostToDevice, nullptr);
            VG_HIP_CHECK(hipStreamSynchronize(nullptr));
        }
    }
  protected:
    void ag_transport(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) override {
        out.resize((size_t)world * n);
        for (int r = 0; r < world; r++) if (n) memcpy(out.data() + (size_t)r * n, contrib[r], n * 4);
    }
    void a2a_transport(std::vector<A2A>& plan) override {
        prepare_peers(plan);
        // every context's queued work first: the blocks to be sent must be complete, and a receive buffer fresh from a context's pool
        // may still be read by kernels that context enqueued before the block was recycled (the pool orders reuse on the context's OWN
        // stream only; the copies below run outside it)
        for (auto& p : plan) { p.c->activate(); p.c->sync(); }
        for (int r = 0; r < world; r++)
            for (int s = 0; s < world; s++) {
                if (staged(plan[s].c->device, plan[r].c->device)) { copy_staged(plan[r].send[(size_t)s], plan[r].c->device, plan[s].recv[(size_t)r], plan[s].c->device); continue; }
                VG_HIP_CHECK(hipSetDevice(plan[s].c->device));
                copy_matching(plan[r].send[(size_t)s], plan[s].recv[(size_t)r], plan[s].c->stream);
            }
        for (auto& p : plan) { VG_HIP_CHECK(hipSetDevice(p.c->device)); VG_HIP_CHECK(hipDeviceSynchronize()); }  // device-to-device copies may return early
    }
};

struct RcclFabric : Fabric {
    Comm* comm;
    // The staging of the all-gathers lives in the COMMUNICATOR (Comm::fab_*: device [send: cap][recv: cap * world], pinned host of the same shape) and is
    // grown in ag_stage, before the status round: a fabric is built per vgpu_prove_sharded call and allocates nothing (hipFree synchronises the whole
    // device: a per-proof allocation stalled every other proof in flight on it).
    explicit RcclFabric(Comm* c) : comm(c) { world = c->world; hosted.push_back(c->rank); timeout_ms = c->timeout_ms; c->grow_fabric_staging(64); }
    RcclFabric(const RcclFabric&) = delete;

  protected:
    void grow(size_t n) { comm->grow_fabric_staging(n); }
    void ag_stage(size_t n) override { grow(n ? n : 1); }
    void ag_transport(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) override {
        out.resize((size_t)world * n);
        if (!n) return;
        comm->require_alive();
        DeviceCtx* c = comm->ctx;
        c->activate();
        uint32_t* const dev = comm->fab_dev;
        uint32_t* const host = comm->fab_host;
        const size_t cap = comm->fab_cap;
        memcpy(host, contrib[0], n * 4);
        VG_HIP_CHECK(hipMemcpyAsync(dev, host, n * 4, hipMemcpyHostToDevice, c->stream));
        comm->mark_start(c->stream);  // the deadline runs from here, not from whatever the stream still had queued
        VG_NCCL_CHECK(RcclApi::get().AllGather(dev, dev + cap, n, ncclUint32, comm->comm, c->stream));
        VG_HIP_CHECK(hipMemcpyAsync(host + cap, dev + cap, (size_t)world * n * 4, hipMemcpyDeviceToHost, c->stream));
        comm->wait(c->stream, timeout_ms);
        memcpy(out.data(), host + cap, (size_t)world * n * 4);
    }
    // packed staging of the all-to-all's segments (one contiguous block per peer: one ncclSend / ncclRecv per peer and direction)
    std::vector<DBuf> pack_send, pack_recv;
    void a2a_stage(std::vector<A2A>& plan) override {
        A2A& p = plan.at(0);
        if ((int)p.send.size() != world || (int)p.recv.size() != world) throw std::logic_error("fabric: plan of the wrong world size");
        DeviceCtx* c = p.c;
        c->activate();
        pack_send.clear(); pack_recv.clear();
        pack_send.resize((size_t)world); pack_recv.resize((size_t)world);
        for (int s = 0; s < world; s++) {
            if (s == comm->rank) continue;
            const size_t sw = p.send_words(s), rw = p.recv_words(s);
            if (sw && !(p.send[(size_t)s].size() == 1 && p.send[(size_t)s][0].n_runs == 1)) {
                pack_send[(size_t)s] = DBuf(c, sw + 4);
                size_t pos = 0;
                for (auto& g : p.send[(size_t)s]) { copy_seg(pack_send[(size_t)s].data + pos, g.run, g.p, g.pitch, g, hipMemcpyDeviceToDevice, c->stream); pos += g.words(); }
            }
            if (rw && !(p.recv[(size_t)s].size() == 1 && p.recv[(size_t)s][0].n_runs == 1)) pack_recv[(size_t)s] = DBuf(c, rw + 4);
        }
        c->check_launch("fabric: pack");
    }
    void a2a_transport(std::vector<A2A>& plan) override {
        A2A& p = plan.at(0);
        std::vector<const uint32_t*> sp((size_t)world, nullptr);
        std::vector<uint32_t*> rp((size_t)world, nullptr);
        std::vector<size_t> sw((size_t)world, 0), rw((size_t)world, 0);
        for (int s = 0; s < world; s++) {
            if (s == comm->rank) continue;
            sw[(size_t)s] = p.send_words(s); rw[(size_t)s] = p.recv_words(s);
            if (sw[(size_t)s]) sp[(size_t)s] = pack_send[(size_t)s].data ? pack_send[(size_t)s].data : p.send[(size_t)s][0].p;
            if (rw[(size_t)s]) rp[(size_t)s] = pack_recv[(size_t)s].data ? pack_recv[(size_t)s].data : p.recv[(size_t)s][0].p;
        }
        comm->mark_start(p.c->stream);
        comm->all_to_all_words(sp, sw, rp, rw);
        comm->wait(p.c->stream, timeout_ms);
    }
    void a2a_finish(std::vector<A2A>& plan) override {
        A2A& p = plan.at(0);
        DeviceCtx* c = p.c;
        copy_matching(p.send[(size_t)comm->rank], p.recv[(size_t)comm->rank], c->stream);  // the block this rank keeps
        for (int s = 0; s < world; s++) {
            if (s == comm->rank || !pack_recv[(size_t)s].data) continue;
            size_t pos = 0;
            for (auto& g : p.recv[(size_t)s]) { copy_seg(g.p, g.pitch, pack_recv[(size_t)s].data + pos, g.run, g, hipMemcpyDeviceToDevice, c->stream); pos += g.words(); }
        }
        c->sync();
        pack_send.clear(); pack_recv.clear();
    }
};

// The caller's transport (include/vgpu.h: vgpu_fabric_t).  Both callbacks work on HOST buffers and return 0 on success; device blocks of
// the all-to-all are staged through page-locked memory (one D2H of everything this rank sends, one H2D of everything it receives; the
// block a rank keeps for itself is copied on the device).
struct CallbackFabric : Fabric {
    using AllGatherFn = int32_t (*)(void* user, const uint32_t* words, uint64_t n_words, uint32_t* out);
    using AllToAllFn = int32_t (*)(void* user, const uint32_t* const* send, const uint64_t* send_words, uint32_t* const* recv, const uint64_t* recv_words);
    // What a callback may still touch after the deadline abandoned it lives HERE, shared with the helper thread: the staging buffers, the
    // pointer tables handed to the callback, and the job slot.  The fabric can be destroyed while an abandoned callback is still blocked
    // in the host's transport; the state dies with its last owner.
    struct Shared {
        std::mutex mu;
        std::condition_variable cv;
        std::function<int32_t()> job;
        bool has_job = false, done = false, quit = false;
        int32_t rc = 0;
        uint32_t* host_send = nullptr;
        uint32_t* host_recv = nullptr;
        size_t host_send_words = 0, host_recv_words = 0;
        std::vector<uint32_t> ag_in, ag_out;
        std::vector<const uint32_t*> hs;
        std::vector<uint32_t*> hr;
        std::vector<uint64_t> sw, rw;
        std::shared_ptr<void> keep;  // caller-owned blocks an abandoned callback may still touch (vgpu_fabric_selftest)
        ~Shared() {
            if (host_send) (void)hipHostFree(host_send);
            if (host_recv) (void)hipHostFree(host_recv);
        }
    };
    void* user;
    int rank;
    AllGatherFn ag;
    AllToAllFn a2a;
    std::shared_ptr<Shared> sh = std::make_shared<Shared>();
    std::thread helper;
    CallbackFabric(int rank_, int world_, AllGatherFn ag_, AllToAllFn a2a_, void* user_, uint32_t timeout_ms_ = 0) : user(user_), rank(rank_), ag(ag_), a2a(a2a_) {
        if (world_ < 1 || rank_ < 0 || rank_ >= world_ || !ag_ || !a2a_) throw std::invalid_argument("fabric: bad rank / world or a null callback");
        world = world_;
        timeout_ms = timeout_ms_;
        hosted.push_back(rank_);
    }
    ~CallbackFabric() override {
        if (helper.joinable()) {
            { std::lock_guard<std::mutex> lk(sh->mu); sh->quit = true; }
            sh->cv.notify_all();
            helper.join();
        }
    }
    CallbackFabric(const CallbackFabric&) = delete;
    bool needs_agreement() const override { return world > 1; }
    // the host-buffer exchange alone (also what vgpu_fabric_selftest drives without a device)
    void host_all_to_all(const std::vector<const uint32_t*>& send, const std::vector<uint64_t>& send_words, const std::vector<uint32_t*>& recv,
                         const std::vector<uint64_t>& recv_words) {
        check_usable();
        sh->hs = send; sh->sw = send_words; sh->hr = recv; sh->rw = recv_words;
        transport([&] { run_a2a(); });
    }

  protected:
    // Runs one callback under the deadline.  timeout_ms == 0: on the calling thread.  Otherwise on the helper thread; if it has not
    // returned in time the thread is detached with the shared state it may still write to, and this rank leaves with an error.
    int32_t call(std::function<int32_t()> job, const char* what) {
        if (!timeout_ms) return job();
        if (!helper.joinable()) {
            std::shared_ptr<Shared> s = sh;
            helper = std::thread([s] {
                for (;;) {
                    std::function<int32_t()> j;
                    {
                        std::unique_lock<std::mutex> lk(s->mu);
                        s->cv.wait(lk, [&] { return s->has_job || s->quit; });
                        if (s->quit) return;
                        j = std::move(s->job);
                        s->has_job = false;
                    }
                    const int32_t rc = j();
                    { std::lock_guard<std::mutex> lk(s->mu); s->rc = rc; s->done = true; }
                    s->cv.notify_all();
                }
            });
        }
        std::unique_lock<std::mutex> lk(sh->mu);
        sh->job = std::move(job); sh->has_job = true; sh->done = false;
        sh->cv.notify_all();
        if (!sh->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return sh->done; })) {
            sh->quit = true;  // should the callback ever return, the thread ends
            lk.unlock();
            helper.detach();
            throw std::runtime_error(std::string("fabric: the host's ") + what + " callback did not return within " + std::to_string(timeout_ms) +
                                     " ms (a peer that died or never entered the exchange?); the callback is abandoned");
        }
        return sh->rc;
    }
    void run_a2a() {
        std::shared_ptr<Shared> s = sh;
        AllToAllFn fn = a2a; void* u = user;
        const int32_t rc = call([s, fn, u] { return fn(u, s->hs.data(), s->sw.data(), s->hr.data(), s->rw.data()); }, "all_to_all");
        if (rc != 0) throw std::runtime_error("fabric: the host's all_to_all callback failed with status " + std::to_string(rc));
    }
    void ag_stage(size_t n) override {
        sh->ag_in.reserve(n ? n : 1);
        sh->ag_out.reserve((size_t)world * (n ? n : 1));
    }
    void ag_transport(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) override {
        out.assign((size_t)world * n, 0u);
        if (!n) return;
        sh->ag_in.assign(contrib.at(0), contrib.at(0) + n);
        sh->ag_out.assign((size_t)world * n, 0u);
        std::shared_ptr<Shared> s = sh;
        AllGatherFn fn = ag; void* u = user;
        const int32_t rc = call([s, fn, u, n] { return fn(u, s->ag_in.data(), (uint64_t)n, s->ag_out.data()); }, "all_gather");
        if (rc != 0) throw std::runtime_error("fabric: the host's all_gather callback failed with status " + std::to_string(rc));
        out = sh->ag_out;
    }
    static void grow(uint32_t*& buf, size_t& have, size_t need) {
        if (need <= have) return;
        if (buf) VG_HIP_CHECK(hipHostFree(buf));
        buf = nullptr; have = 0;
        VG_HIP_CHECK(hipHostMalloc((void**)&buf, need * 4));
        have = need;
    }
    // everything that can fail on this rank alone: staging buffers, the device-to-host copies of the blocks it sends, its own block on
    // the device, the synchronisation — BEFORE the status round (fabric.hpp, FAILURE PROTOCOL)
    void a2a_stage(std::vector<A2A>& plan) override {
        A2A& p = plan.at(0);
        DeviceCtx* c = p.c;
        c->activate();
        if ((int)p.send.size() != world || (int)p.recv.size() != world) throw std::logic_error("fabric: plan of the wrong world size");
        size_t ns = 0, nr = 0;
        for (int s = 0; s < world; s++) if (s != rank) { ns += p.send_words(s); nr += p.recv_words(s); }
        Shared& S = *sh;
        grow(S.host_send, S.host_send_words, ns ? ns : 1);
        grow(S.host_recv, S.host_recv_words, nr ? nr : 1);
        S.hs.assign((size_t)world, nullptr); S.hr.assign((size_t)world, nullptr);
        S.sw.assign((size_t)world, 0); S.rw.assign((size_t)world, 0);
        size_t so = 0, ro = 0;
        for (int s = 0; s < world; s++) {
            if (s == rank) continue;
            const size_t sw = p.send_words(s), rw = p.recv_words(s);
            if (sw) {
                size_t pos = so;
                for (auto& g : p.send[(size_t)s]) { copy_seg(S.host_send + pos, g.run, g.p, g.pitch, g, hipMemcpyDeviceToHost, c->stream); pos += g.words(); }
                S.hs[(size_t)s] = S.host_send + so; S.sw[(size_t)s] = sw; so += sw;
            }
            if (rw) { S.hr[(size_t)s] = S.host_recv + ro; S.rw[(size_t)s] = rw; ro += rw; }
        }
        copy_matching(p.send[(size_t)rank], p.recv[(size_t)rank], c->stream);  // the block this rank keeps for itself: on the device
        if (fabric_failpoint("fabric_stage", rank)) throw std::runtime_error("fabric: failpoint fabric_stage");
        c->sync();  // the blocks to send are on the host; a receive block fresh from the pool is no longer read by queued kernels
    }
    void a2a_transport(std::vector<A2A>&) override { run_a2a(); }
    void a2a_finish(std::vector<A2A>& plan) override {
        A2A& p = plan.at(0);
        DeviceCtx* c = p.c;
        if (fabric_failpoint("fabric_finish", rank)) throw std::runtime_error("fabric: failpoint fabric_finish");
        for (int s = 0; s < world; s++) {
            if (s == rank) continue;
            size_t pos = 0;
            for (auto& g : p.recv[(size_t)s]) { copy_seg(g.p, g.pitch, sh->hr[(size_t)s] + pos, g.run, g, hipMemcpyHostToDevice, c->stream); pos += g.words(); }
        }
        c->sync();
    }
};

}  // namespace vhost
