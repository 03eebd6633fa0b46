// The Keccak-256 MMCS kernels of valida_amd/csrc/kernels/merkle.hip compiled for the HOST under tools/hipemu: the thread-per-leaf / per-parent
// kernels and the single-workgroup top (VGPU_KECCAK_PAIRS=0 selects them for every layer: the lane-pair variants exchange halves with DPP,
// which fibers cannot emulate), with v_bitop3_b32 / v_alignbit_b32 as the plain functions they are.  Built with -DVK_ALIGNBIT_NOP=0: the
// no-op rides on an inline-asm statement that only the device compiler understands; the rotation itself is the same builtin.
// Checked against the oracle's Keccak MMCS (tests/test_keccak_emu_cpu.py).  Test infrastructure; nothing in the product links it.
#define HIPEMU_CHECKS 1
#include <hip/hip_runtime.h>  // tools/hipemu/hip/hip_runtime.h (first on the include path)

#include "../../valida_amd/csrc/kernels/merkle.hip"

namespace vk {
uint32_t lds[16];
thread_local Profiler* g_profiler = nullptr;
thread_local ProfScope* g_scope = nullptr;
}  // namespace vk

using vg::Fp;

extern "C" {
// Root (8 canonical words) of the mixed-height tree over two row-major canonical matrices: `tall` (h0 x w0) and `low` (h1 x w1, h1 a power of
// two <= h0, or h1 = 0 for none), injected where the layer length equals h1 — built as host/pcs.hpp's DeviceTree does: leaves, one
// launch per parent layer above `top_first_len` parents, the rest in the single-workgroup top launch.
int emu_keccak_root(const uint32_t* tall, uint64_t h0, int w0, const uint32_t* low, uint64_t h1, int w1, uint64_t top_first_len, uint32_t* root) {
    std::vector<uint32_t> c0((size_t)h0 * w0), c1((size_t)h1 * w1 + 1);
    for (uint64_t r = 0; r < h0; r++) for (int j = 0; j < w0; j++) c0[(size_t)j * h0 + r] = Fp::from_canonical(tall[r * w0 + j]).v;
    for (uint64_t r = 0; r < h1; r++) for (int j = 0; j < w1; j++) c1[(size_t)j * h1 + r] = Fp::from_canonical(low[r * w1 + j]).v;
    std::vector<const uint32_t*> p0, p1;
    for (int j = 0; j < w0; j++) p0.push_back(c0.data() + (size_t)j * h0);
    for (int j = 0; j < w1; j++) p1.push_back(c1.data() + (size_t)j * h1);
    std::vector<std::vector<uint32_t>> layers;
    layers.emplace_back((size_t)h0 * 8);
    vk::launch_keccak_leaves(nullptr, p0.data(), w0, h0, layers[0].data());
    vk::KeccakTopArgs top{};
    for (uint64_t len = h0 / 2; len >= 1; len /= 2) {
        layers.emplace_back((size_t)len * 8);
        const bool inj = h1 && len == h1;
        if (len > top_first_len) {
            vk::launch_keccak_compress(nullptr, layers[layers.size() - 2].data(), inj ? p1.data() : nullptr, inj ? w1 : 0, len, layers.back().data());
        } else {
            if (top.levels == 0) { top.prev = layers[layers.size() - 2].data(); top.first_len = len; }
            top.out[top.levels] = layers.back().data();
            top.cols[top.levels] = inj ? p1.data() : nullptr;
            top.n_elems[top.levels] = inj ? w1 : 0;
            top.levels++;
        }
        if (len == 1) break;
    }
    if (top.levels) vk::launch_keccak_top(nullptr, top);
    memcpy(root, layers.back().data(), 32);
    return 0;
}
}
