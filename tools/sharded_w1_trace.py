"""One C2 proof by the sharded prover over W contexts of one GPU (default W = 1), a few times — to be run under rocprofv3 --kernel-trace
(tools/gpu_r4_w1.sh) for the launch sequence of the sharded path next to the single-GPU prover's."""
import sys, time
import numpy as np
import torch
import valida_amd as va
import bench

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rc = va.poseidon_round_constants()
machine = va.Machine.basic()
ps = [va.Prover(machine, rc, log_blowup=1, device=0) for _ in range(W)]
wl = va.Workload.fib(bench.segment_loop_bound(20, 0))
mt, prep = wl.main_traces(), wl.preprocessed()
up = va.upload_replicated(ps, mt, prep)
pr = va.prove_sharded_local(ps, mt, prep, uploaded=up)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    va.prove_sharded_local(ps, mt, prep, uploaded=up)
torch.cuda.synchronize()
print("W", W, "ms per proof", (time.perf_counter() - t0) / 4 * 1e3)
