// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg):
// every launcher opens a ProfScope naming the kernel and the ALGORITHMIC bytes that launch must move
// (compulsory reads + writes, each array once; DESIGN.md "Kernels").  Disabled by default: zero cost
// beyond one thread_local pointer test per launch.
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <string>
#include <vector>

namespace vk {

struct KernelStat { uint64_t launches = 0; double ms = 0; double bytes = 0; double valu_ops = 0; };

struct Profiler {
    bool enabled = false;
    struct Rec { const char* name; hipEvent_t start, stop; double bytes, valu_ops; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    std::map<std::string, KernelStat> stats;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
    // Call after the stream has been synchronised.
    void collect() {
        for (auto& r : recs) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
                KernelStat& s = stats[r.name];
                s.launches++; s.ms += ms; s.bytes += r.bytes; s.valu_ops += r.valu_ops;
            }
            pool.push_back(r.start);
            pool.push_back(r.stop);
        }
        recs.clear();
    }
    void reset() { collect(); stats.clear(); }
    ~Profiler() { collect(); for (auto e : pool) (void)hipEventDestroy(e); }
};

extern thread_local Profiler* g_profiler;

struct ProfScope {
    Profiler* p;
    hipStream_t st;
    size_t idx;
    // valu_ops: algorithmic wave64 VALU instructions of the launch, for the kernels whose roofline is the integer
    // issue rate rather than HBM (the Keccak kernels); 0 = not modelled
    ProfScope(const char* name, hipStream_t s, double bytes, double valu_ops = 0) : p(g_profiler), st(s), idx(0) {
        if (!p || !p->enabled) { p = nullptr; return; }
        Profiler::Rec r{name, p->get(), p->get(), bytes, valu_ops};
        (void)hipEventRecord(r.start, st);
        idx = p->recs.size();
        p->recs.push_back(r);
    }
    ~ProfScope() { if (p) (void)hipEventRecord(p->recs[idx].stop, st); }
};

}  // namespace vk
