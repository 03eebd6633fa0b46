// The NTT / LDE kernels of valida_amd/csrc/kernels/ntt.hip compiled for the HOST under tools/hipemu (fibers per workgroup thread,
// __syncthreads() as a yield): the very kernel source, its index arithmetic and barrier structure, checked on the CPU against the oracle's
// committed LDE before any GPU minute is spent (tests/test_ntt_emu_cpu.py).  Test infrastructure; nothing in the product links it.
#define HIPEMU_CHECKS 1
#include <hip/hip_runtime.h>  // tools/hipemu/hip/hip_runtime.h (first on the include path)

#include "../../valida_amd/csrc/kernels/ntt.hip"

namespace vk {
uint32_t lds[48 * 1024];  // the workgroup's dynamic LDS (`extern __shared__ uint32_t lds[]` of every kernel resolves to this)
thread_local Profiler* g_profiler = nullptr;
thread_local ProfScope* g_scope = nullptr;
}  // namespace vk

namespace {
struct Tables {
    std::vector<uint32_t> image;
    vk::DeviceTables tb{};
    Tables() : image(vk::device_tables_words()) {
        vk::build_device_tables(tb, image.data());
        vk::bind_device_tables(tb, image.data());
    }
};
Tables& tables() { static Tables t; return t; }

using vg::Fp;
// row-major canonical (n x w) -> column-major Montgomery
std::vector<uint32_t> to_cols(const uint32_t* m, uint64_t n, uint64_t w) {
    std::vector<uint32_t> c(n * w);
    for (uint64_t r = 0; r < n; r++) for (uint64_t j = 0; j < w; j++) c[j * n + r] = Fp::from_canonical(m[r * w + j]).v;
    return c;
}
void from_cols(const std::vector<uint32_t>& c, uint64_t n, uint64_t w, uint32_t* m) {
    for (uint64_t r = 0; r < n; r++) for (uint64_t j = 0; j < w; j++) m[r * w + j] = Fp::raw(c[j * n + r]).canonical();
}
}  // namespace

extern "C" {
// out (row-major canonical, (n << log_blowup) x w) = the committed (bit-reversed) LDE of m on shift * H, through the FUSED pipeline
int emu_lde_natural(const uint32_t* m, uint64_t n, uint64_t w, int log_blowup, uint32_t shift, uint32_t* out) {
    try {
        const int k = (int)vg::log2_strict_u64(n);
        const uint64_t L = n << log_blowup;
        std::vector<uint32_t> nat = to_cols(m, n, w), lde(L * w, 0xDEADBEEFu), s1(k > 12 ? n * w : 1, 0xDEADBEEFu), s2(k > 12 ? L * w : 1, 0xDEADBEEFu);
        std::vector<uint32_t> lt(vk::lde_tables_words(k, log_blowup));
        vk::build_lde_tables(k, log_blowup, Fp::from_canonical(shift), lt.data());
        const size_t n_lo = (size_t)1 << vk::lde_k_lo(k);
        vk::LdeTables ltv{lt.data(), lt.data() + ((size_t)1 << log_blowup) * n_lo};
        vk::launch_lde_natural(nullptr, vk::DMatView{nat.data(), n, w, n}, vk::DMatView{lde.data(), L, w, L}, log_blowup, tables().tb, ltv,
                               vk::DMatView{s1.data(), n, w, n}, vk::DMatView{s2.data(), L, w, L});
        from_cols(lde, L, w, out);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "emu_lde_natural: %s\n", e.what());
        return 1;
    }
}
// the same through the UNFUSED passes the product already runs on the device (host-side row bit-reversal standing in for k_bitrev_rows):
// validates the emulator itself on kernels whose device results are known to be right
int emu_lde_unfused(const uint32_t* m, uint64_t n, uint64_t w, int log_blowup, uint32_t shift, uint32_t* out) {
    try {
        const int k = (int)vg::log2_strict_u64(n);
        const uint64_t L = n << log_blowup;
        std::vector<uint32_t> nat = to_cols(m, n, w), coeffs(n * w), lde(L * w, 0xDEADBEEFu);
        for (uint64_t j = 0; j < w; j++) for (uint64_t r = 0; r < n; r++) coeffs[j * n + vg::reverse_bits_len((uint32_t)r, (unsigned)k)] = nat[j * n + r];
        vk::DMatView cv{coeffs.data(), n, w, n};
        vk::launch_intt(nullptr, cv, tables().tb);
        Fp wgen = vg::two_adic_generator((unsigned)(k + log_blowup)), wt = Fp::one(), s = Fp::from_canonical(shift);
        for (uint64_t t = 0; t < ((uint64_t)1 << log_blowup); t++) {
            const uint64_t block = vg::reverse_bits_len((uint32_t)t, (unsigned)log_blowup);
            vk::launch_coset_ntt(nullptr, cv, vk::DMatView{lde.data(), L, w, L}, block * n, s * wt, tables().tb);
            wt *= wgen;
        }
        from_cols(lde, L, w, out);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "emu_lde_unfused: %s\n", e.what());
        return 1;
    }
}
unsigned long emu_barriers() { return hipemu::sched().barriers; }
}
