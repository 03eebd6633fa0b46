// Probe of v_mfma_i32_16x16x64_i8 on gfx950: with the loading convention "lane l holds row (A) / column (B) l % 16 and the 16 consecutive
// k = 16 * (l / 16) .. + 15 as the bytes of its four operand VGPRs", is D[i][j] = sum_k A[i][k] B[k][j] with D at col = l & 15, row = 4 (l >> 4) + reg?
// (The contraction only needs the SAME k-map for A and B; this checks that and the C/D map, with asymmetric random operands.)
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_i8_probe.hip -o build/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int8_t* A /*16x64 row-major*/, const int8_t* B /*64x16 row-major: B[k][j]*/, int* D /*16x16*/) {
    const int l = threadIdx.x, ij = l & 15, g = l >> 4;
    v4i a, b, c = {0, 0, 0, 0};
    for (int q = 0; q < 4; q++) {
        uint32_t wa = 0, wb = 0;
        for (int t = 0; t < 4; t++) {
            const int kk = 16 * g + 4 * q + t;
            wa |= (uint32_t)(uint8_t)A[ij * 64 + kk] << (8 * t);
            wb |= (uint32_t)(uint8_t)B[kk * 16 + ij] << (8 * t);
        }
        a[q] = (int)wa; b[q] = (int)wb;
    }
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * g + r) * 16 + ij] = c[r];
}
int main() {
    int8_t hA[16 * 64], hB[64 * 16];
    srand(7);
    for (auto& x : hA) x = (int8_t)(rand() % 256 - 128);
    for (auto& x : hB) x = (int8_t)(rand() % 128);
    int8_t *dA, *dB; int* dD; int hD[256];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    if (hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
    int bad = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
        int s = 0;
        for (int kk = 0; kk < 64; kk++) s += (int)hA[i * 64 + kk] * (int)hB[kk * 16 + j];
        bad += s != hD[i * 16 + j];
    }
    printf("mfma_i32_16x16x64_i8 with the consecutive-k convention: %d of 256 outputs differ from the reference\n", bad);
    return bad != 0;
}
