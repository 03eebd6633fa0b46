// Device runtime for one GPU: stream, constant tables, a caching HBM allocator (a proof allocates the
// same few dozen buffers every time — after the first proof nothing in the timed path calls hipMalloc),
// pinned staging memory, and the column-major device matrix type.
#pragma once
#include <new>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>
#include "../kernels/launch.hpp"

namespace vhost {
using vg::Ext5;
using vg::Fp;

struct DeviceCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    vk::DeviceTables tables{};
    std::multimap<size_t, void*> free_blocks;
    std::map<void*, size_t> live_blocks;
    size_t bytes_allocated = 0, peak_live = 0, live = 0;
    uint32_t* table_mem = nullptr;
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // Pinned staging ring for the small host -> device uploads of a proof (constant pools, pointer tables, gather descriptors):
    // the bytes are copied here and an ASYNCHRONOUS copy is enqueued on the main stream — no host synchronisation per upload
    // (each one used to cost a stream round trip of tens of microseconds during which the GPU idled).
    uint8_t* stage = nullptr;
    static constexpr size_t STAGE_BYTES = 16u << 20;
    size_t stage_pos = 0;
    vk::Profiler profiler;
    // MMCS hash of this context (vgpu_config.hash_kind): 0 = Keccak-256 (the reference's configuration), 1 = Poseidon-16 sponge /
    // truncated permutation; poseidon_tab = [480 round constants][16 MDS coefficients] on the device (owned by the Prover)
    int hash_kind = 0;
    const uint32_t* poseidon_tab = nullptr;
    bool poseidon_sparse = false;  // poseidon_tab + 1024 holds valid sparse-partial-round tables (host/poseidon_opt.hpp)
    // Auxiliary streams for fork/join sections: independent per-chip pipelines (tiny matrices are
    // latency-bound single-block launches) overlap with the big chips' kernels on the main stream.
    // One auxiliary stream.  The device exposes a handful of hardware queues (4 by default) that streams take in creation
    // order; measured on MI355X with several prover contexts per GPU: 1 aux stream per context beats 0, 2 and 3 (with 3,
    // the main streams of two contexts land on the same queue and the proofs meant to overlap serialise), and raising
    // GPU_MAX_HW_QUEUES beyond the number of streams hurts.  DESIGN.md "Measurement".
    static constexpr int NUM_AUX = 1;
    hipStream_t aux[NUM_AUX] = {};
    hipEvent_t fork_ev = nullptr, join_ev[NUM_AUX] = {};
    // a commit that rides on the auxiliary stream beside another round's LDEs (pcs.hpp: CommitRider): the event that orders its pointer-table
    // upload (main stream) before its kernels (aux), and the page-locked landing area of its root
    hipEvent_t rider_ev = nullptr;
    uint32_t* rider_root_pin = nullptr;
    bool in_section = false;
    std::vector<void*> deferred;  // blocks released inside a section return to the pool at the join
    // The pool is shared by the thread that drives a proof (vgpu_prove_async's worker) and by whichever host thread frees a
    // handle meanwhile (vgpu_trace_free / vgpu_oplog_free / a garbage collector): every pool operation takes this lock.
    std::mutex pool_mu;
    // Uploads handed in by a caller thread WHILE a proof owns the main stream (a host that prepares segment i+1 during proof i):
    // a stream of their own, created on first use, so the copy neither queues behind the proof's kernels nor shares its waits.
    std::atomic<int> proofs_running{0};
    // One proof at a time per context (its streams, pool sections and pinned buffers are not shared): callers QUEUE on this mutex — the
    // reference's `Machine: Sync` (machine/src/machine.rs:13) lets several threads call prove on one machine, and so may a host here.
    std::mutex prove_mu;
    std::mutex upload_mu;
    hipStream_t upload_stream = nullptr;
    hipEvent_t upload_ev = nullptr;

    explicit DeviceCtx(int dev) : device(dev) {
        VG_HIP_CHECK(hipSetDevice(dev));
        // stream priorities and stream-to-queue arrangements were measured in round 5 (profiles/r04_stream_priorities.txt, r05_queue_arrangements.txt): the runtime's defaults are the optimum
        VG_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (int i = 0; i < NUM_AUX; i++) {
            VG_HIP_CHECK(hipStreamCreateWithFlags(&aux[i], hipStreamNonBlocking));
            VG_HIP_CHECK(hipEventCreateWithFlags(&join_ev[i], hipEventDisableTiming));
        }
        VG_HIP_CHECK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
        VG_HIP_CHECK(hipEventCreateWithFlags(&rider_ev, hipEventDisableTiming));
        VG_HIP_CHECK(hipHostMalloc((void**)&rider_root_pin, 64));
        init_tables();
    }
    // Every C-ABI entry that launches work calls this first: binds the calling thread to this context's
    // device and points the launchers' profiler hook at this context (never at a destroyed one).
    void activate() {
        VG_HIP_CHECK(hipSetDevice(device));
        vk::g_profiler = &profiler;
    }
    ~DeviceCtx() {
        if (vk::g_profiler == &profiler) vk::g_profiler = nullptr;
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(stream);
        for (int i = 0; i < NUM_AUX; i++) { (void)hipStreamSynchronize(aux[i]); (void)hipStreamDestroy(aux[i]); (void)hipEventDestroy(join_ev[i]); }
        (void)hipEventDestroy(fork_ev);
        if (rider_ev) (void)hipEventDestroy(rider_ev);
        if (rider_root_pin) (void)hipHostFree(rider_root_pin);
        if (sync_ev) (void)hipEventDestroy(sync_ev);
        for (auto& kv : free_blocks) (void)hipFree(kv.second);
        for (auto& kv : live_blocks) (void)hipFree(kv.first);
        if (table_mem) (void)hipFree(table_mem);
        for (auto& kv : lde_tabs) (void)hipFree(kv.second);
        if (upload_stream) { (void)hipStreamSynchronize(upload_stream); (void)hipStreamDestroy(upload_stream); (void)hipEventDestroy(upload_ev); }
        if (pinned) (void)hipHostFree(pinned);
        if (stage) (void)hipHostFree(stage);
        (void)hipStreamDestroy(stream);
    }
    DeviceCtx(const DeviceCtx&) = delete;

    void* alloc(size_t bytes) {
        std::lock_guard<std::mutex> lk(pool_mu);
        if (bytes == 0) bytes = 4;
        bytes = (bytes + 255) & ~(size_t)255;
        auto it = free_blocks.find(bytes);
        void* p;
        if (it != free_blocks.end()) { p = it->second; free_blocks.erase(it); }
        else {
            hipError_t e = hipMalloc(&p, bytes);
            if (e == hipErrorOutOfMemory) {  // cached blocks of other sizes may be holding the memory: give them back, retry once
                (void)hipGetLastError();
                trim_locked();
                e = hipMalloc(&p, bytes);
            }
            if (e != hipSuccess) throw std::bad_alloc();
            bytes_allocated += bytes;
        }
        live_blocks[p] = bytes;
        live += bytes;
        if (live > peak_live) peak_live = live;
        return p;
    }
    // Return every cached (free) block to the driver; live blocks are untouched.  Synchronises the device first: a cached
    // block may still be read by work in flight.
    size_t trim() {
        std::lock_guard<std::mutex> lk(pool_mu);
        return trim_locked();
    }
    size_t trim_locked() {
        (void)hipDeviceSynchronize();
        size_t freed = 0;
        for (auto& kv : free_blocks) { (void)hipFree(kv.second); freed += kv.first; }
        free_blocks.clear();
        bytes_allocated -= freed;
        return freed;
    }
    // Fork: aux streams wait for everything enqueued so far on the main stream.
    void fork() {
        if (in_section) throw std::runtime_error("nested fork");
        VG_HIP_CHECK(hipEventRecord(fork_ev, stream));
        for (int i = 0; i < NUM_AUX; i++) VG_HIP_CHECK(hipStreamWaitEvent(aux[i], fork_ev, 0));
        std::lock_guard<std::mutex> lk(pool_mu);
        in_section = true;
    }
    // Join: the main stream waits for all aux work; blocks released meanwhile become reusable.
    void join() {
        for (int i = 0; i < NUM_AUX; i++) {
            VG_HIP_CHECK(hipEventRecord(join_ev[i], aux[i]));
            VG_HIP_CHECK(hipStreamWaitEvent(stream, join_ev[i], 0));
        }
        std::vector<void*> d;
        {
            std::lock_guard<std::mutex> lk(pool_mu);
            in_section = false;
            d.swap(deferred);
        }
        for (void* p : d) release(p);
    }
    // Stream for independent work item `i` of estimated size `rows`: big items stay on the main stream.
    hipStream_t stream_for(size_t i, uint64_t rows) const {
        if (!in_section || rows >= (1ull << 16)) return stream;
        return aux[i % NUM_AUX];
    }
    void release(void* p) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(pool_mu);
        if (in_section) { deferred.push_back(p); return; }
        auto it = live_blocks.find(p);
        if (it == live_blocks.end()) {  // called from destructors: never throw (a logic error, reported and survived)
            fprintf(stderr, "vgpu: release of unknown device block %p ignored\n", p);
            return;
        }
        free_blocks.insert({it->second, p});
        live -= it->second;
        live_blocks.erase(it);
    }
    uint32_t* alloc_words(size_t n) { return (uint32_t*)alloc(n * 4); }
    void* pinned_buffer(size_t bytes) {
        if (bytes > pinned_bytes) {
            if (pinned) VG_HIP_CHECK(hipHostFree(pinned));
            pinned_bytes = bytes < (1u << 20) ? (1u << 20) : bytes;
            VG_HIP_CHECK(hipHostMalloc(&pinned, pinned_bytes));
        }
        return pinned;
    }
    // Waiting for the main stream.  hipStreamSynchronize parks the thread, and its wake-up costs 50-150 us — with a dozen true
    // synchronisation points per proof (roots, opened values, final values) that is ~1 ms of an idle GPU per LONE proof.  POLLING an
    // event avoids it but burns a core per waiting proof thread, which is only right when cores are plentiful: VGPU_SPIN_WAIT=1
    // selects it (bench.py does for a single rank; with one rank per GPU and three proof threads each, a node's ranks would fight
    // over the host cores), the default parks.
    hipEvent_t sync_ev = nullptr;
    static bool spin_wait() {
        static const bool on = [] { const char* e = getenv("VGPU_SPIN_WAIT"); return e && e[0] == '1'; }();
        return on;
    }
    // Host-side staging state (the pinned ring of upload_async, the lazily created polling event, the pinned download buffer) is shared by
    // every ABI entry point that works on this context; the one-proof-at-a-time guard covers prove only, so a caller thread that commits /
    // opens / builds a permutation trace beside an outstanding ticket meets the proof thread here: one lock around all of it.
    std::recursive_mutex host_mu;
    void sync() {
        if (!spin_wait()) { VG_HIP_CHECK(hipStreamSynchronize(stream)); return; }
        hipEvent_t ev;
        {
            // one polling event per waiting thread would be cleaner; waits of one context are rare enough to share it under the lock
            std::lock_guard<std::recursive_mutex> lk(host_mu);
            if (!sync_ev) VG_HIP_CHECK(hipEventCreateWithFlags(&sync_ev, hipEventDisableTiming));
            ev = sync_ev;
            VG_HIP_CHECK(hipEventRecord(ev, stream));
        }
        for (;;) {
            hipError_t e = hipEventQuery(ev);
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) throw std::runtime_error(std::string("hipEventQuery: ") + hipGetErrorString(e));
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    void check_launch(const char* what) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    }
    // small synchronous transfers (roots, challenges, descriptors)
    void upload(void* dst, const void* src, size_t bytes) {
        VG_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        sync();
    }
    // The same from a second host thread while a proof is running on this context.  `dst` comes from the pool: a block on the free list
    // may still be read by kernels the proving thread enqueued before it released the block, and every such kernel is ordered before an
    // event recorded on the main stream now (releases inside a fork/join section are deferred to the join) — the copy waits for that event.
    void upload_beside_proof(void* dst, const void* src, size_t bytes) {
        std::lock_guard<std::mutex> lk(upload_mu);
        if (!upload_stream) {
            VG_HIP_CHECK(hipStreamCreateWithFlags(&upload_stream, hipStreamNonBlocking));
            VG_HIP_CHECK(hipEventCreateWithFlags(&upload_ev, hipEventDisableTiming));
        }
        VG_HIP_CHECK(hipEventRecord(upload_ev, stream));
        VG_HIP_CHECK(hipStreamWaitEvent(upload_stream, upload_ev, 0));
        VG_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, upload_stream));
        VG_HIP_CHECK(hipStreamSynchronize(upload_stream));
    }
    void download(void* dst, const void* src, size_t bytes) {
        VG_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
        sync();
    }
    // small D2H through pinned memory (a pageable destination makes hipMemcpyAsync take the slow staged path)
    void download_small(void* dst, const void* src, size_t bytes) {
        std::lock_guard<std::recursive_mutex> lk(host_mu);  // the pinned buffer is shared (and may be re-allocated)
        void* pin = pinned_buffer(bytes);
        VG_HIP_CHECK(hipMemcpyAsync(pin, src, bytes, hipMemcpyDeviceToHost, stream));
        sync();
        memcpy(dst, pin, bytes);
    }
    // Host -> device without a synchronisation: valid for consumers enqueued later on the main stream (or on the aux streams after
    // a fork).  The source may be reused at once.
    void upload_async(void* dst, const void* src, size_t bytes) {
        if (!bytes) return;
        if (bytes > STAGE_BYTES / 2) { upload(dst, src, bytes); return; }
        std::lock_guard<std::recursive_mutex> lk(host_mu);  // ring position + the wrap's drain
        if (!stage) VG_HIP_CHECK(hipHostMalloc((void**)&stage, STAGE_BYTES));
        size_t pos = (stage_pos + 63) & ~(size_t)63;
        if (pos + bytes > STAGE_BYTES) {  // wrap: everything staged so far must have been read by the copy engine
            VG_HIP_CHECK(hipStreamSynchronize(stream));
            pos = 0;
        }
        memcpy(stage + pos, src, bytes);
        VG_HIP_CHECK(hipMemcpyAsync(dst, stage + pos, bytes, hipMemcpyHostToDevice, stream));
        stage_pos = pos + bytes;
    }
    // Tables of the fused LDE per (log height, log blowup, coset shift): built on the host the first time a shape is extended in this
    // context (a few thousand field products), copied synchronously — every later launch on any stream of the context sees them — and
    // kept for the context's lifetime (a proof uses a handful of shapes; words per entry: b (n_lo + n_hi) <= 2^16).
    std::map<std::tuple<int, int, uint32_t>, uint32_t*> lde_tabs;
    std::mutex lde_tabs_mu;
    vk::LdeTables lde_tables(int k, int log_blowup, Fp shift) {
        std::lock_guard<std::mutex> lk(lde_tabs_mu);
        const auto key = std::make_tuple(k, log_blowup, shift.v);
        auto it = lde_tabs.find(key);
        if (it == lde_tabs.end()) {
            std::vector<uint32_t> h(vk::lde_tables_words(k, log_blowup));
            vk::build_lde_tables(k, log_blowup, shift, h.data());
            uint32_t* d = nullptr;
            VG_HIP_CHECK(hipMalloc((void**)&d, h.size() * 4));
            VG_HIP_CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            it = lde_tabs.emplace(key, d).first;
        }
        const size_t n_lo = (size_t)1 << vk::lde_k_lo(k);
        return vk::LdeTables{it->second, it->second + ((size_t)1 << log_blowup) * n_lo};
    }
    uint32_t* upload_words(const std::vector<uint32_t>& w) {
        uint32_t* d = alloc_words(w.size());
        upload_async(d, w.data(), w.size() * 4);
        return d;
    }

  private:
    void init_tables() {
        std::vector<uint32_t> t(vk::device_tables_words(), 0);
        vk::build_device_tables(tables, t.data());
        VG_HIP_CHECK(hipMalloc((void**)&table_mem, t.size() * 4));
        VG_HIP_CHECK(hipMemcpy(table_mem, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        vk::bind_device_tables(tables, table_mem);
    }
};

// RAII fork/join: joins on scope exit (also on exceptions) unless join() was called explicitly.
struct Section {
    DeviceCtx* c;
    bool open;
    explicit Section(DeviceCtx* ctx) : c(ctx), open(true) { c->fork(); }
    void join() { if (open) { open = false; c->join(); } }
    ~Section() { if (open) { try { c->join(); } catch (...) { std::lock_guard<std::mutex> lk(c->pool_mu); c->in_section = false; } } }
    Section(const Section&) = delete;
    Section& operator=(const Section&) = delete;
};

// Column-major Montgomery matrix in HBM.
struct DMat {
    DeviceCtx* ctx = nullptr;
    uint32_t* data = nullptr;
    uint64_t height = 0, width = 0;
    DMat() {}
    DMat(DeviceCtx* c, uint64_t h, uint64_t w) : ctx(c), data(c->alloc_words(h * w)), height(h), width(w) {}
    DMat(DMat&& o) noexcept { *this = std::move(o); }
    DMat& operator=(DMat&& o) noexcept {
        if (this != &o) { reset(); ctx = o.ctx; data = o.data; height = o.height; width = o.width; o.data = nullptr; o.ctx = nullptr; }
        return *this;
    }
    DMat(const DMat&) = delete;
    DMat& operator=(const DMat&) = delete;
    ~DMat() { reset(); }
    void reset() { if (data && ctx) ctx->release(data); data = nullptr; }
    vk::DMatView view() const { return vk::DMatView{data, height, width, height}; }
    bool empty() const { return data == nullptr; }
};

// RAII device word buffer
struct DBuf {
    DeviceCtx* ctx = nullptr;
    uint32_t* data = nullptr;
    size_t words = 0;
    DBuf() {}
    DBuf(DeviceCtx* c, size_t n) : ctx(c), data(c->alloc_words(n)), words(n) {}
    DBuf(DeviceCtx* c, const std::vector<uint32_t>& w) : ctx(c), data(c->upload_words(w)), words(w.size()) {}
    DBuf(DBuf&& o) noexcept { *this = std::move(o); }
    DBuf& operator=(DBuf&& o) noexcept {
        if (this != &o) { reset(); ctx = o.ctx; data = o.data; words = o.words; o.data = nullptr; o.ctx = nullptr; }
        return *this;
    }
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf() { reset(); }
    void reset() { if (data && ctx) ctx->release(data); data = nullptr; }
};

inline void put_ext(std::vector<uint32_t>& w, const Ext5& e) { for (int k = 0; k < 5; k++) w.push_back(e.c[k].v); }
inline void put_ptr(std::vector<uint32_t>& w, const void* p) { uint64_t v = (uint64_t)p; w.push_back((uint32_t)v); w.push_back((uint32_t)(v >> 32)); }
inline void put_u64(std::vector<uint32_t>& w, uint64_t v) { w.push_back((uint32_t)v); w.push_back((uint32_t)(v >> 32)); }

}  // namespace vhost
