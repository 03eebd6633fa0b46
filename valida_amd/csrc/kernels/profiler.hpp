// Optional per-kernel timing with HIP events stamped by the launches themselves (bench.py's roofline leg):
// every launcher opens a ProfScope naming the kernel and the ALGORITHMIC bytes that launch must move
// (compulsory reads + writes, each array once; DESIGN.md "Kernels").  Disabled by default: zero cost
// beyond one thread_local pointer test per launch.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <map>
#include <string>
#include <vector>

namespace vk {

struct KernelStat { uint64_t launches = 0; double ms = 0; double bytes = 0; double valu_ops = 0; };

struct Profiler {
    bool enabled = false;
    std::string only;  // when not empty: only launches of this kernel name carry events (the bench's timed region times its dominant kernel alone)
    struct Rec { const char* name; hipEvent_t start, stop; double bytes, valu_ops; bool counts_as_launch; };
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
                s.launches += r.counts_as_launch ? 1 : 0; s.ms += ms; s.bytes += r.bytes; s.valu_ops += r.valu_ops;
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

// A launcher opens a ProfScope naming the kernel(s) it is about to launch; every VK_LAUNCH inside the scope asks it for a
// (start, stop) event pair and hands the pair to hipExtLaunchKernelGGL, which stamps them with the kernel's OWN begin / end
// timestamps.  No hipEventRecord, i.e. no marker packets between the kernels and no extra host call per event (two records per
// launch used to cost the host ~3.5 us each and 5 % of the bench's throughput).
struct ProfScope;
extern thread_local ProfScope* g_scope;

struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };

struct ProfScope {
    Profiler* p;
    ProfScope* outer;
    const char* name;
    double bytes, valu_ops;
    bool first = true;
    // valu_ops: algorithmic wave64 VALU instructions of the launch, for the kernels whose roofline is the integer
    // issue rate rather than HBM (the Keccak kernels); 0 = not modelled
    ProfScope(const char* name_, hipStream_t, double bytes_, double valu_ops_ = 0) : p(g_profiler), outer(g_scope), name(name_), bytes(bytes_), valu_ops(valu_ops_) {
        if (!p || !p->enabled || (!p->only.empty() && p->only != name_)) p = nullptr;
        g_scope = this;
    }
    ~ProfScope() { g_scope = outer; }
    ProfScope(const ProfScope&) = delete;
    // events for the next kernel launched inside this scope (null when profiling is off); the first launch of a scope carries the
    // scope's algorithmic bytes and counts as its "launch", further kernels of the same scope only add their time
    LaunchEvents next() {
        LaunchEvents e;
        if (!p) return e;
        Profiler::Rec r{name, p->get(), p->get(), first ? bytes : 0.0, first ? valu_ops : 0.0, first};
        first = false;
        p->recs.push_back(r);
        e.start = r.start; e.stop = r.stop;
        return e;
    }
};

inline LaunchEvents launch_events() { return g_scope ? g_scope->next() : LaunchEvents{}; }

}  // namespace vk

// Kernel launch of this library: plain when profiling is off, with the launch's own timestamp events when it is on.
#define VK_LAUNCH(kernel, grid, block, lds, stream, ...)                                                                  \
    do {                                                                                                                  \
        vk::LaunchEvents _ev = vk::launch_events();                                                                       \
        if (_ev.start) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, _ev.start, _ev.stop, 0, __VA_ARGS__);       \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                            \
    } while (0)
