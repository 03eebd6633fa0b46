// The Poseidon-16 MMCS kernels of valida_amd/csrc/kernels/poseidon_mmcs.hip compiled for the HOST under tools/hipemu: the very device
// functions (the sparse partial rounds, the MDS layer as a cyclic convolution through the butterfly network of butterfly.hpp) and the
// leaf / parent kernels, checked on the CPU against the oracle's Poseidon before any GPU minute is spent (tests/test_poseidon_emu_cpu.py).
// Test infrastructure; nothing in the product links it.
#define HIPEMU_CHECKS 1
#include <hip/hip_runtime.h>  // tools/hipemu/hip/hip_runtime.h (first on the include path)

#include "../../valida_amd/csrc/kernels/poseidon_mmcs.hip"
#include "../../valida_amd/csrc/host/poseidon_opt.hpp"

namespace vk {
uint32_t lds[16];
thread_local Profiler* g_profiler = nullptr;
thread_local ProfScope* g_scope = nullptr;
}  // namespace vk

using vg::Fp;

extern "C" {
// state (16 canonical words) <- Perm16(state) through the DEVICE function.  form 0: plain rounds (dense MDS everywhere), 1: the kernels'
// schedule (sparse partial rounds, convolution MDS).  Returns 1 when form 1's tables were valid (else the plain form ran), 0 otherwise.
int emu_poseidon16_permute(const uint32_t* rc480, uint32_t* state, int form) {
    vhost::Poseidon16 p(rc480);
    bool sparse = false;
    const std::vector<uint32_t> pos = vhost::poseidon_device_image(rc480, p, sparse);
    const vk::PoseidonTab tab = vk::tab_of(pos.data(), form && sparse);
    Fp st[16];
    for (int i = 0; i < 16; i++) st[i] = Fp::from_canonical(state[i]);
    vk::poseidon16_permute(st, tab);
    for (int i = 0; i < 16; i++) state[i] = st[i].canonical();
    return form && sparse;
}
// digests (n_rows x 8 canonical words) of the rows of a column-major Montgomery-free matrix given row-major canonical (n_rows x width):
// k_poseidon_leaves launched through the emulator, then ONE parent layer over them with k_poseidon_compress (n_rows even)
int emu_poseidon_leaves_and_parents(const uint32_t* rc480, const uint32_t* m, uint64_t n_rows, int width, uint32_t* leaves, uint32_t* parents) {
    vhost::Poseidon16 p(rc480);
    bool sparse = false;
    const std::vector<uint32_t> pos = vhost::poseidon_device_image(rc480, p, sparse);
    std::vector<uint32_t> cols((size_t)n_rows * width);
    for (uint64_t r = 0; r < n_rows; r++) for (int j = 0; j < width; j++) cols[(size_t)j * n_rows + r] = Fp::from_canonical(m[r * width + j]).v;
    vk::launch_poseidon_leaves_strided(nullptr, pos.data(), sparse, cols.data(), n_rows, width, n_rows, leaves);
    if (n_rows >= 2) vk::launch_poseidon_compress(nullptr, pos.data(), sparse, leaves, nullptr, 0, n_rows / 2, parents);
    return sparse;
}
}
