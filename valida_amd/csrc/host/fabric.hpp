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
// FAILURE PROTOCOL (multi-process fabrics).  A rank validates its inputs and allocates between collectives; one that throws alone
// would leave its peers blocked in the next collective for ever.  Hence every collective is preceded by a one-word STATUS all-gather
// (`agree`): a rank that failed contributes a non-zero status instead of entering the collective (`fail`, called once by
// ShardedProof::run's catch-all), its peers read it in their next `agree` and throw as well — every rank returns an error, none hangs.
// Cost: one tiny all-gather per collective (a few dozen per proof).  LocalFabric drives all ranks from one thread: nothing to agree on.
#pragma once
#include "comm.hpp"

namespace vhost {

struct FabricPeerFailure : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Fabric {
    int world = 1;
    std::vector<int> hosted;  // the ranks living in this process, ascending
    bool peer_failed = false;
    virtual ~Fabric() {}
    // contrib[k]: n words of hosted rank hosted[k] (host memory).  out: world * n words, rank-major — what every rank receives.
    void all_gather(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) { agree(); do_all_gather(contrib, n, out); }
    struct A2A;
    void all_to_all(std::vector<A2A>& plan) { agree(); do_all_to_all(plan); }
    // Status round before a collective: throws FabricPeerFailure on EVERY rank if any rank reported a failure.
    void agree() {
        if (!needs_agreement()) return;
        const uint32_t ok = 0;
        std::vector<uint32_t> all;
        do_all_gather({&ok}, 1, all);
        for (int r = 0; r < world; r++)
            if (all[(size_t)r]) { peer_failed = true; throw FabricPeerFailure("sharded prove: rank " + std::to_string(r) + " failed (status " + std::to_string(all[(size_t)r]) + "); every rank gives up this proof"); }
    }
    // Called by a rank that cannot go on (never throws): its peers learn it in their next agree().  Not after a FabricPeerFailure — then the
    // peers have already left.
    void fail(uint32_t status = 1) noexcept {
        if (!needs_agreement() || peer_failed) return;
        try { std::vector<uint32_t> all; const uint32_t st = status ? status : 1u; do_all_gather({&st}, 1, all); peer_failed = true; } catch (...) {}
    }
    virtual bool needs_agreement() const { return (int)hosted.size() < world; }

  protected:
    virtual void do_all_gather(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) = 0;
    virtual void do_all_to_all(std::vector<A2A>& plan) = 0;

  public:
    // Device all-to-all.  plan[k] belongs to hosted rank hosted[k]: block send[s] (send_words[s] words) goes to rank s, recv[s]
    // (recv_words[s] words) arrives from rank s.  Returns when the data has arrived (the contexts' streams are drained).
    struct A2A {
        DeviceCtx* c = nullptr;
        std::vector<const uint32_t*> send;
        std::vector<size_t> send_words;
        std::vector<uint32_t*> recv;
        std::vector<size_t> recv_words;
        explicit A2A(DeviceCtx* ctx = nullptr, int world = 0) : c(ctx), send(world, nullptr), send_words(world, 0), recv(world, nullptr), recv_words(world, 0) {}
    };
};

struct LocalFabric : Fabric {
    explicit LocalFabric(int w) { world = w; for (int r = 0; r < w; r++) hosted.push_back(r); }
  protected:
    void do_all_gather(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) override {
        out.resize((size_t)world * n);
        for (int r = 0; r < world; r++) if (n) memcpy(out.data() + (size_t)r * n, contrib[r], n * 4);
    }
    void do_all_to_all(std::vector<A2A>& plan) override {
        // every context's queued work first: the blocks to be sent must be complete, and a receive buffer fresh from a context's pool
        // may still be read by kernels that context enqueued before the block was recycled (the pool orders reuse on the context's OWN
        // stream only; the copies below run outside it)
        for (auto& p : plan) { p.c->activate(); p.c->sync(); }
        for (int r = 0; r < world; r++)
            for (int s = 0; s < world; s++) {
                if (plan[r].send_words[s] != plan[s].recv_words[r]) throw std::logic_error("fabric: send / receive sizes disagree");
                if (plan[r].send_words[s]) VG_HIP_CHECK(hipMemcpy(plan[s].recv[r], plan[r].send[s], plan[r].send_words[s] * 4, hipMemcpyDeviceToDevice));
            }
        for (auto& p : plan) { VG_HIP_CHECK(hipSetDevice(p.c->device)); VG_HIP_CHECK(hipDeviceSynchronize()); }  // device-to-device copies may return early
    }
};

struct RcclFabric : Fabric {
    Comm* comm;
    explicit RcclFabric(Comm* c) : comm(c) { world = c->world; hosted.push_back(c->rank); }

  protected:
    void do_all_gather(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) override {
        out.resize((size_t)world * n);
        if (!n) return;
        DeviceCtx* c = comm->ctx;
        c->activate();
        DBuf send(c, n + 4), recv(c, (size_t)world * n + 4);
        VG_HIP_CHECK(hipMemcpyAsync(send.data, contrib[0], n * 4, hipMemcpyHostToDevice, c->stream));
        VG_NCCL_CHECK(RcclApi::get().AllGather(send.data, recv.data, n, ncclUint32, comm->comm, c->stream));
        VG_HIP_CHECK(hipMemcpyAsync(out.data(), recv.data, (size_t)world * n * 4, hipMemcpyDeviceToHost, c->stream));
        c->sync();
    }
    void do_all_to_all(std::vector<A2A>& plan) override {
        A2A& p = plan.at(0);
        comm->all_to_all_words(p.send, p.send_words, p.recv, p.recv_words);
        p.c->sync();
    }
};

// The caller's transport (include/vgpu.h: vgpu_fabric_t).  Both callbacks work on HOST buffers and return 0 on success; device blocks of
// the all-to-all are staged through page-locked memory (one D2H of everything this rank sends, one H2D of everything it receives; the
// block a rank keeps for itself is copied on the device).
struct CallbackFabric : Fabric {
    using AllGatherFn = int32_t (*)(void* user, const uint32_t* words, uint64_t n_words, uint32_t* out);
    using AllToAllFn = int32_t (*)(void* user, const uint32_t* const* send, const uint64_t* send_words, uint32_t* const* recv, const uint64_t* recv_words);
    void* user;
    int rank;
    AllGatherFn ag;
    AllToAllFn a2a;
    uint32_t* host_send = nullptr;
    uint32_t* host_recv = nullptr;
    size_t host_send_words = 0, host_recv_words = 0;
    CallbackFabric(int rank_, int world_, AllGatherFn ag_, AllToAllFn a2a_, void* user_) : user(user_), rank(rank_), ag(ag_), a2a(a2a_) {
        if (world_ < 1 || rank_ < 0 || rank_ >= world_ || !ag_ || !a2a_) throw std::invalid_argument("fabric: bad rank / world or a null callback");
        world = world_;
        hosted.push_back(rank_);
    }
    ~CallbackFabric() override {
        if (host_send) (void)hipHostFree(host_send);
        if (host_recv) (void)hipHostFree(host_recv);
    }
    CallbackFabric(const CallbackFabric&) = delete;
    bool needs_agreement() const override { return world > 1; }
    // the host-buffer exchange alone (also what vgpu_fabric_selftest drives without a device)
    void host_all_to_all(const std::vector<const uint32_t*>& send, const std::vector<uint64_t>& send_words, const std::vector<uint32_t*>& recv,
                         const std::vector<uint64_t>& recv_words) {
        const int32_t rc = a2a(user, send.data(), send_words.data(), recv.data(), recv_words.data());
        if (rc != 0) throw std::runtime_error("fabric: the host's all_to_all callback failed with status " + std::to_string(rc));
    }

  protected:
    void do_all_gather(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) override {
        out.assign((size_t)world * n, 0u);
        if (!n) return;
        const int32_t rc = ag(user, contrib.at(0), (uint64_t)n, out.data());
        if (rc != 0) throw std::runtime_error("fabric: the host's all_gather callback failed with status " + std::to_string(rc));
    }
    static void grow(uint32_t*& buf, size_t& have, size_t need) {
        if (need <= have) return;
        if (buf) VG_HIP_CHECK(hipHostFree(buf));
        buf = nullptr; have = 0;
        VG_HIP_CHECK(hipHostMalloc((void**)&buf, need * 4));
        have = need;
    }
    void do_all_to_all(std::vector<A2A>& plan) override {
        A2A& p = plan.at(0);
        DeviceCtx* c = p.c;
        c->activate();
        size_t ns = 0, nr = 0;
        for (int s = 0; s < world; s++) if (s != rank) { ns += p.send_words[s]; nr += p.recv_words[s]; }
        grow(host_send, host_send_words, ns ? ns : 1);
        grow(host_recv, host_recv_words, nr ? nr : 1);
        std::vector<const uint32_t*> hs((size_t)world, nullptr);
        std::vector<uint32_t*> hr((size_t)world, nullptr);
        std::vector<uint64_t> sw((size_t)world, 0), rw((size_t)world, 0);
        size_t so = 0, ro = 0;
        for (int s = 0; s < world; s++) {
            if (s == rank) continue;
            if (p.send_words[s]) {
                VG_HIP_CHECK(hipMemcpyAsync(host_send + so, p.send[s], p.send_words[s] * 4, hipMemcpyDeviceToHost, c->stream));
                hs[(size_t)s] = host_send + so; sw[(size_t)s] = p.send_words[s]; so += p.send_words[s];
            }
            if (p.recv_words[s]) { hr[(size_t)s] = host_recv + ro; rw[(size_t)s] = p.recv_words[s]; ro += p.recv_words[s]; }
        }
        if (p.send_words[rank]) {
            if (p.send_words[rank] != p.recv_words[rank]) throw std::logic_error("fabric: send / receive sizes disagree");
            VG_HIP_CHECK(hipMemcpyAsync(p.recv[rank], p.send[rank], p.send_words[rank] * 4, hipMemcpyDeviceToDevice, c->stream));
        }
        c->sync();  // the blocks to send are on the host; a receive block fresh from the pool is no longer read by queued kernels
        host_all_to_all(hs, sw, hr, rw);
        for (int s = 0; s < world; s++)
            if (s != rank && p.recv_words[s]) VG_HIP_CHECK(hipMemcpyAsync(p.recv[s], hr[(size_t)s], p.recv_words[s] * 4, hipMemcpyHostToDevice, c->stream));
        c->sync();
    }
};

}  // namespace vhost
