// RCCL inside the library: the path's one collective (an all-gather of each segment's Merkle roots over xGMI)
// and the exchanges of the sharded commit (§8(f)-4) issued by the library itself, so that a non-Python host — the reference's
// Rust — owns them.  librccl.so is loaded on first use (dlopen): single-GPU users never pay for it and libvgpu.so carries no
// link-time dependency on it.  One communicator per prover context, one process per GPU (RCCL's own model); collectives are
// enqueued on the context's stream.
#pragma once
#include <dlfcn.h>
#include <chrono>
#include <thread>
#include <rccl/rccl.h>
#include "runtime.hpp"

namespace vhost {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    static RcclApi& get() {
        static RcclApi api = load();
        return api;
    }
    static RcclApi load() {
        RcclApi a;
        // a copy the process already loaded (a host that also uses RCCL through another runtime — bench.py's torch — must not end up
        // with two RCCL instances) before a fresh one
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (a.lib) break;
        }
        if (!a.lib)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (a.lib) break;
            }
        if (!a.lib) throw std::runtime_error(std::string("hip: cannot load librccl.so: ") + dlerror());
        auto sym = [&](const char* n) { void* p = dlsym(a.lib, n); if (!p) throw std::runtime_error(std::string("hip: librccl.so lacks ") + n); return p; };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.CommAbort = (decltype(a.CommAbort))sym("ncclCommAbort");
        a.CommGetAsyncError = (decltype(a.CommGetAsyncError))sym("ncclCommGetAsyncError");
        a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
        a.Send = (decltype(a.Send))sym("ncclSend");
        a.Recv = (decltype(a.Recv))sym("ncclRecv");
        a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        return a;
    }
};

#define VG_NCCL_CHECK(expr)                                                                                                  \
    do {                                                                                                                     \
        ncclResult_t _r = (expr);                                                                                            \
        if (_r != ncclSuccess) throw std::runtime_error(std::string("hip: rccl: " #expr ": ") + RcclApi::get().GetErrorString(_r)); \
    } while (0)

struct Comm {
    DeviceCtx* ctx;
    ncclComm_t comm = nullptr;
    int rank, world;
    // The roots all-gather has its own stream and buffers: it is issued by the host thread that collects finished proofs while
    // other proofs of this context may be in flight on the context's streams (and pool) — it must neither wait for them nor
    // touch the pool.
    hipStream_t coll_stream = nullptr;
    uint32_t* coll_buf = nullptr;       // device: [send: COLL_WORDS][recv: COLL_WORDS * world]
    uint32_t* coll_host = nullptr;      // pinned staging of the same shape
    static constexpr size_t COLL_WORDS = 256;
    // Deadline of one collective (vgpu_comm_set_timeout_ms; 0 = wait for ever, the default; VGPU_COMM_TIMEOUT_MS presets it): a collective a
    // peer never enters — it died, or it failed alone — is aborted with ncclCommAbort instead of blocking this rank for ever.  After an
    // abort the communicator is gone (`comm` null): every later call on it is refused.
    uint32_t timeout_ms = [] { const char* e = getenv("VGPU_COMM_TIMEOUT_MS"); return e ? (uint32_t)strtoul(e, nullptr, 10) : 0u; }();
    hipEvent_t wait_ev = nullptr;
    // mark_start(s): an event recorded on `s` just BEFORE a collective is enqueued.  wait() first waits for it without a deadline — what the stream
    // still had queued ahead of the collective (LDE passes, other kernels: local work that finishes whatever the peers do) is not charged to the
    // collective — and starts the clock there.  What the deadline still covers by design: the time this rank waits for SLOWER peers to
    // enter the collective.  timeout_ms must therefore exceed the worst compute skew between ranks (first-proof pool sizing included): seconds, not
    // milliseconds; bench.py uses 120 s.
    hipEvent_t start_ev = nullptr;
    bool start_marked = false;
    hipStream_t start_stream = nullptr;  // the stream the mark was recorded on: a mark left behind by an enqueue that threw is never applied to another stream's wait
    void mark_start(hipStream_t s) {
        if (!start_ev) VG_HIP_CHECK(hipEventCreateWithFlags(&start_ev, hipEventDisableTiming));
        start_marked = false;
        VG_HIP_CHECK(hipEventRecord(start_ev, s));
        start_marked = true;
        start_stream = s;
    }
    // staging of the RCCL fabric's all-gathers (fabric.hpp: RcclFabric), owned here so that it outlives the per-proof fabric objects
    uint32_t* fab_dev = nullptr;
    uint32_t* fab_host = nullptr;
    size_t fab_cap = 0;
    void grow_fabric_staging(size_t n) {
        if (n <= fab_cap) return;
        VG_HIP_CHECK(hipSetDevice(ctx->device));
        VG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (fab_dev) { VG_HIP_CHECK(hipFree(fab_dev)); fab_dev = nullptr; }
        if (fab_host) { VG_HIP_CHECK(hipHostFree(fab_host)); fab_host = nullptr; }
        fab_cap = 0;
        VG_HIP_CHECK(hipMalloc((void**)&fab_dev, n * (size_t)(world + 1) * 4));
        VG_HIP_CHECK(hipHostMalloc((void**)&fab_host, n * (size_t)(world + 1) * 4));
        fab_cap = n;
    }
    Comm(DeviceCtx* c, const ncclUniqueId& id, int rank_, int world_) : ctx(c), rank(rank_), world(world_) {
        if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("comm: bad rank / world");
        VG_HIP_CHECK(hipSetDevice(c->device));
        VG_NCCL_CHECK(RcclApi::get().CommInitRank(&comm, world, id, rank));
        VG_HIP_CHECK(hipStreamCreateWithFlags(&coll_stream, hipStreamNonBlocking));
        VG_HIP_CHECK(hipMalloc((void**)&coll_buf, COLL_WORDS * (size_t)(world + 1) * 4));
        VG_HIP_CHECK(hipHostMalloc((void**)&coll_host, COLL_WORDS * (size_t)(world + 1) * 4));
    }
    ~Comm() {
        (void)hipSetDevice(ctx->device);
        if (coll_stream) { (void)hipStreamSynchronize(coll_stream); (void)hipStreamDestroy(coll_stream); }
        if (coll_buf) (void)hipFree(coll_buf);
        if (coll_host) (void)hipHostFree(coll_host);
        if (wait_ev) (void)hipEventDestroy(wait_ev);
        if (start_ev) (void)hipEventDestroy(start_ev);
        if (fab_dev) (void)hipFree(fab_dev);
        if (fab_host) (void)hipHostFree(fab_host);
        if (comm) (void)RcclApi::get().CommDestroy(comm);
    }
    void require_alive() const {
        if (!comm) throw std::runtime_error("hip: rccl: this communicator was aborted after a collective timed out or failed; create a new one");
    }
    // Waits for everything enqueued on `s` (the collective just issued included).  Without a deadline: a plain stream synchronisation.  With
    // one: polls an event and RCCL's asynchronous error state; on an error or at the deadline the communicator is ABORTED (its kernels leave
    // the device, the stream drains) and the call throws.
    void wait(hipStream_t s, uint32_t deadline_ms) {
        if (!deadline_ms) { start_marked = false; VG_HIP_CHECK(hipStreamSynchronize(s)); return; }
        if (!wait_ev) VG_HIP_CHECK(hipEventCreateWithFlags(&wait_ev, hipEventDisableTiming));
        VG_HIP_CHECK(hipEventRecord(wait_ev, s));
        // local work queued ahead of the collective is not the collective's time.  Only a mark of THIS stream counts (one recorded earlier on the same
        // stream by an enqueue that then threw lies before everything waited for here: harmless); any other stale mark is dropped.
        const bool mine = start_marked && start_stream == s;
        start_marked = false;
        if (mine) VG_HIP_CHECK(hipEventSynchronize(start_ev));
        const auto t0 = std::chrono::steady_clock::now();
        auto& api = RcclApi::get();
        for (unsigned spin = 0;; spin++) {
            const hipError_t e = hipEventQuery(wait_ev);
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) throw std::runtime_error(std::string("hipEventQuery: ") + hipGetErrorString(e));
            ncclResult_t async = ncclSuccess;
            const bool bad = (spin & 63) == 63 && comm && api.CommGetAsyncError(comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress;
            const bool late = std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(deadline_ms);
            if (bad || late) {
                ncclComm_t dead = comm;
                comm = nullptr;
                (void)api.CommAbort(dead);
                (void)hipStreamSynchronize(s);
                throw std::runtime_error(bad ? std::string("hip: rccl: asynchronous error: ") + api.GetErrorString(async) + "; communicator aborted"
                                             : "hip: rccl: a collective did not complete within " + std::to_string(deadline_ms) + " ms (a peer that died or never entered it?); communicator aborted");
            }
            if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50)); else std::this_thread::yield();
        }
    }
    Comm(const Comm&) = delete;
    // every rank's `n_words` words to every rank: out[r * n_words + k] = rank r's word k.  Host buffers; staged through HBM.
    void all_gather_words(const uint32_t* words, size_t n_words, uint32_t* out) {
        if (n_words > COLL_WORDS) throw std::invalid_argument("comm: at most 256 words per rank in one roots all-gather");
        require_alive();
        VG_HIP_CHECK(hipSetDevice(ctx->device));
        memcpy(coll_host, words, n_words * 4);
        uint32_t* recv = coll_buf + COLL_WORDS;
        VG_HIP_CHECK(hipMemcpyAsync(coll_buf, coll_host, n_words * 4, hipMemcpyHostToDevice, coll_stream));
        mark_start(coll_stream);
        VG_NCCL_CHECK(RcclApi::get().AllGather(coll_buf, recv, n_words, ncclUint32, comm, coll_stream));
        VG_HIP_CHECK(hipMemcpyAsync(coll_host + COLL_WORDS, recv, n_words * (size_t)world * 4, hipMemcpyDeviceToHost, coll_stream));
        wait(coll_stream, timeout_ms);
        memcpy(out, coll_host + COLL_WORDS, n_words * (size_t)world * 4);
    }
    // device-to-device exchange: block `send[s]` (count words) goes to rank s, `recv[s]` arrives from rank s (the all-to-all of
    // the sharded commit: column shards -> row-range shards)
    void all_to_all_words(const std::vector<const uint32_t*>& send, const std::vector<size_t>& send_words, const std::vector<uint32_t*>& recv,
                          const std::vector<size_t>& recv_words) {
        require_alive();
        ctx->activate();
        auto& api = RcclApi::get();
        VG_NCCL_CHECK(api.GroupStart());
        for (int s = 0; s < world; s++) {
            if (s == rank) continue;
            if (send_words[s]) VG_NCCL_CHECK(api.Send(send[s], send_words[s], ncclUint32, s, comm, ctx->stream));
            if (recv_words[s]) VG_NCCL_CHECK(api.Recv(recv[s], recv_words[s], ncclUint32, s, comm, ctx->stream));
        }
        VG_NCCL_CHECK(api.GroupEnd());
        if (send_words[rank]) VG_HIP_CHECK(hipMemcpyAsync(recv[rank], send[rank], send_words[rank] * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
};

}  // namespace vhost
