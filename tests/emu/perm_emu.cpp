// The permutation-trace kernels of valida_amd/csrc/kernels/perm.hip (k_perm_recip and the three-phase block scan) compiled for the HOST under
// tools/hipemu and checked against the oracle's generate_permutation_trace (tests/test_perm_emu_cpu.py).  Test infrastructure; nothing in the
// product links it.
#define HIPEMU_CHECKS 1
#define HIPEMU_STATIC_SHARED 1
#include <hip/hip_runtime.h>  // tools/hipemu/hip/hip_runtime.h (first on the include path)

#include "../../valida_amd/csrc/kernels/perm.hip"

namespace vk {
thread_local Profiler* g_profiler = nullptr;
thread_local ProfScope* g_scope = nullptr;
}  // namespace vk

using vg::Fp;
using vg::Ext5;

extern "C" {
// perm (n x 5 (M + 1), row-major canonical) = the flattened permutation trace of a chip whose interactions are given in the DEVICE encoding
// `iw` (kernels/interactions.hpp), on the row-major canonical main trace (n x w), for the three Ext5 challenges `rnd15`; is_global / bus
// per interaction say which challenge and power make its alpha (generate_rlc_elements, machine/src/chip.rs:291-331), as prover.cpp does.
// native_chip: -2 = the encoded walk (k_perm_recip); a vchips::ChipId = that chip's compiled-in interactions with the batched inversion (k_perm_recip_native)
int emu_perm_trace(const uint32_t* main, uint64_t n, uint64_t w, const uint32_t* iw, const uint32_t* rnd15, const uint32_t* is_global, const uint32_t* bus, uint32_t* perm, int native_chip) {
    const uint32_t M = iw[0], maxf = iw[1];
    Ext5 rnd[3];
    for (int i = 0; i < 3; i++) for (int k = 0; k < 5; k++) rnd[i].c[k] = Fp::from_canonical(rnd15[5 * i + k]);
    std::vector<uint32_t> chal;
    for (uint32_t m = 0; m < M; m++) {
        const Ext5 a = (is_global[m] ? rnd[1] : rnd[0]).pow((uint64_t)bus[m] + 1);
        for (int k = 0; k < 5; k++) chal.push_back(a.c[k].v);
    }
    Ext5 bp = Ext5::one();
    for (uint32_t j = 0; j < maxf; j++) { for (int k = 0; k < 5; k++) chal.push_back(bp.c[k].v); bp *= rnd[2]; }
    chal.push_back(0);
    std::vector<uint32_t> cols((size_t)n * w), out((size_t)n * 5 * (M + 1)), scratch((size_t)vk::perm_scratch_words(n) + 1);
    for (uint64_t r = 0; r < n; r++) for (uint64_t j = 0; j < w; j++) cols[(size_t)j * n + r] = Fp::from_canonical(main[r * w + j]).v;
    vk::launch_perm_trace(nullptr, vk::DMatView{cols.data(), n, w, n}, vk::DMatView{nullptr, 0, 0, 0}, iw, chal.data(), M, vk::DMatView{out.data(), n, 5 * (uint64_t)(M + 1), n}, scratch.data(), native_chip);
    const uint64_t pw = 5 * (uint64_t)(M + 1);
    for (uint64_t r = 0; r < n; r++) for (uint64_t j = 0; j < pw; j++) perm[r * pw + j] = Fp::raw(out[(size_t)j * n + r]).canonical();
    return 0;
}
}
