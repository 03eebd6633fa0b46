// The exchanges of a proof sharded over W GPUs, behind one interface so that the sharded prover
// (sharded_prover.cpp) is written once: the driver walks the ranks THIS process hosts through the phases and calls the fabric
// between them.
//   * LocalFabric — all W ranks live in this process as W prover contexts (a box with one GPU: the 1-GPU realisation the parity
//     tests run; or several GPUs of one process): an exchange is a device-to-device copy per (source, destination) pair.
//   * RcclFabric  — one rank per process, one process per GPU (RCCL's own model): all-gather / grouped send-recv over xGMI through
//     the library's communicator (comm.hpp).
#pragma once
#include "comm.hpp"

namespace vhost {

struct Fabric {
    int world = 1;
    std::vector<int> hosted;  // the ranks living in this process, ascending
    virtual ~Fabric() {}
    // contrib[k]: n words of hosted rank hosted[k] (host memory).  out: world * n words, rank-major — what every rank receives.
    virtual void all_gather(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) = 0;
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
    virtual void all_to_all(std::vector<A2A>& plan) = 0;
};

struct LocalFabric : Fabric {
    explicit LocalFabric(int w) { world = w; for (int r = 0; r < w; r++) hosted.push_back(r); }
    void all_gather(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) override {
        out.resize((size_t)world * n);
        for (int r = 0; r < world; r++) if (n) memcpy(out.data() + (size_t)r * n, contrib[r], n * 4);
    }
    void all_to_all(std::vector<A2A>& plan) override {
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
    void all_gather(const std::vector<const uint32_t*>& contrib, size_t n, std::vector<uint32_t>& out) override {
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
    void all_to_all(std::vector<A2A>& plan) override {
        A2A& p = plan.at(0);
        comm->all_to_all_words(p.send, p.send_words, p.recv, p.recv_words);
        p.c->sync();
    }
};

}  // namespace vhost
