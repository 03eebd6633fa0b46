/* A plain C99 host of libvgpu.so: what a non-Python, non-C++ caller (the reference's Rust through `extern "C"`) binds.
 * Host-only entry points run anywhere; with a device the program goes on to prove prove_fibonacci and prints the proof's size.
 *   gcc -std=c99 -Iinclude examples/c_host.c -o c_host -Lvalida_amd -lvgpu -Wl,-rpath,$PWD/valida_amd && ./c_host
 * (tests/test_host_cpu.py::test_c99_host_links_and_runs builds and runs it on the CPU box: the prover must refuse with VGPU_ERR_HIP there.) */
#include <stdio.h>
#include <string.h>
#include "vgpu.h"

int main(void) {
    vgpu_machine_t* machine = NULL;
    vgpu_challenger_t* ch = NULL;
    vgpu_prover_t* prover = NULL;
    vgpu_config_t cfg;
    uint32_t info[8], sample[5], state[16];
    uint64_t x = 0x56414C494441ull; /* any 480 values below p serve as Poseidon round constants (configuration input) */
    int32_t code;
    int i;

    printf("%s\n", vgpu_version());
    if (vgpu_machine_basic(&machine) != VGPU_OK || vgpu_machine_num_chips(machine) != 14) { fprintf(stderr, "machine: %s\n", vgpu_last_error()); return 1; }
    if (vgpu_machine_chip_info(machine, 0, info) != VGPU_OK) return 2;
    printf("chip 0: width %u\n", (unsigned)info[0]);

    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.log_blowup = 1; cfg.num_queries = 40; cfg.pow_bits = 8; cfg.hash_kind = VGPU_HASH_KECCAK256;
    for (i = 0; i < 480; i++) { x = x * 6364136223846793005ull + 1442695040888963407ull; cfg.poseidon_rc[i] = (uint32_t)((x >> 33) % 2013265921u); }

    /* the transcript is host code: DuplexChallenger over Poseidon-16 */
    if (vgpu_challenger_new(cfg.poseidon_rc, &ch) != VGPU_OK) return 3;
    for (i = 0; i < 16; i++) state[i] = (uint32_t)i;
    vgpu_challenger_observe(ch, state, 16);
    vgpu_challenger_sample(ch, sample, 5);
    vgpu_poseidon16_permute(cfg.poseidon_rc, state);
    if (sample[4] != state[11] || sample[0] != state[15]) { fprintf(stderr, "challenger: samples are popped from the END of the permuted state\n"); return 4; }
    vgpu_challenger_free(ch);

    code = vgpu_prover_create(&cfg, machine, &prover);
    if (code != VGPU_OK) { /* no usable device: the product has no CPU fallback */
        printf("no device: code %d (%s)\n", (int)code, vgpu_last_error());
        vgpu_machine_free(machine);
        return code == VGPU_ERR_HIP ? 0 : 5;
    }
    {
        vgpu_workload_t* w = NULL;
        vgpu_proof_t* proof = NULL;
        vgpu_trace_t* main_t[14];
        vgpu_trace_t* prep_t[2];
        uint32_t prep_chips[2];
        uint64_t stats[8];
        if (vgpu_workload_fib(25, &w) != VGPU_OK) return 6;
        vgpu_workload_stats(w, stats);
        for (i = 0; i < 14; i++) {
            uint64_t h = 0, wd = 0;
            const uint32_t* data = NULL;
            if (vgpu_workload_main_trace(w, (uint32_t)i, &data, &h, &wd) != VGPU_OK) return 7;
            if (vgpu_trace_upload(prover, data, h, wd, &main_t[i]) != VGPU_OK) { fprintf(stderr, "upload: %s\n", vgpu_last_error()); return 7; }
        }
        for (i = 0; i < 2; i++) {
            uint64_t h = 0, wd = 0;
            const uint32_t* data = NULL;
            if (vgpu_workload_preprocessed(w, (uint32_t)i, &prep_chips[i], &data, &h, &wd) != VGPU_OK) return 8;
            if (vgpu_trace_upload(prover, data, h, wd, &prep_t[i]) != VGPU_OK) return 8;
        }
        if (vgpu_prove(prover, (const vgpu_trace_t* const*)main_t, 14, prep_chips, (const vgpu_trace_t* const*)prep_t, 2, 0, &proof) != VGPU_OK) {
            fprintf(stderr, "prove: %s\n", vgpu_last_error());
            return 9;
        }
        printf("prove_fibonacci: %llu proof words\n", (unsigned long long)vgpu_proof_len(proof));
        vgpu_proof_free(proof);
        for (i = 0; i < 14; i++) vgpu_trace_free(main_t[i]);
        for (i = 0; i < 2; i++) vgpu_trace_free(prep_t[i]);
        vgpu_workload_free(w);
    }
    vgpu_prover_destroy(prover);
    vgpu_machine_free(machine);
    return 0;
}
