"""Leaf-hash time against the shape of the tallest matrix: commit one random matrix of h rows x w columns (LDE 2h rows) for several (h, w),
under rocprofv3 --kernel-trace (tools/gpu_r4_leaves.sh prints the k_keccak_leaves / k_keccak_compress durations per shape)."""
import numpy as np
import torch
import valida_amd as va

P = 2013265921
rc = va.poseidon_round_constants()
prover = va.Prover(va.Machine.basic(), rc, log_blowup=1, device=0)
rng = np.random.default_rng(5)
for lh, w in ((22, 14), (21, 14), (22, 10), (21, 10), (20, 10), (22, 5), (22, 20), (23, 10)):
    m = rng.integers(0, P, size=(1 << lh, w), dtype=np.uint32)
    d = prover.upload(m)
    for _ in range(3):
        pd = prover.commit_batches([d])
        torch.cuda.synchronize()
    print("shape", lh, w, flush=True)
    del pd, d
