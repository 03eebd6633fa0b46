#!/bin/bash
# Round 4, session 4: sharded prover after the segment-list exchanges (no pack / unpack copies) and row-range inputs: its tests, then what ONE proof
# over W prover contexts of this GPU costs (replicated traces and row ranges), A/B against the library of the previous commit.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04s4; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_prove_gpu.py tests/test_zz_sharded_multiprocess_gpu.py -m gpu -q -x > "$OUT/pytest_sharded.log" 2>&1; tail -3 "$OUT/pytest_sharded.log"
python - "$OUT" <<'P'
import sys, time, json, os
import numpy as np
sys.path.insert(0, os.getcwd())
import valida_amd as va
import torch
out = sys.argv[1]
rc = va.poseidon_round_constants()
m = va.Machine.basic()
w = va.Workload.fib(149794)
mt, prep = w.main_traces(), w.preprocessed()
res = {}
p0 = va.Prover(m, rc)
d = [p0.upload(x) for x in mt]; dp = [(c, p0.upload(x)) for c, x in prep]
ref = p0.prove(d, dp)
for _ in range(3): p0.prove(d, dp)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): p0.prove(d, dp)
torch.cuda.synchronize(); res["single_gpu_prover_ms"] = (time.perf_counter() - t0) / 5 * 1e3
provers = [p0] + [va.Prover(m, rc) for _ in range(7)]
for W in (1, 2, 4, 8):
    ps = provers[:W]
    up = va.upload_replicated(ps, mt, prep)
    pr = va.prove_sharded_local(ps, mt, prep, uploaded=up)
    same = bool(np.array_equal(pr.words, ref.words))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): va.prove_sharded_local(ps, mt, prep, uploaded=up)
    torch.cuda.synchronize()
    res["replicated_W%d" % W] = {"ms_per_proof": (time.perf_counter() - t0) / 3 * 1e3, "same_words": same}
    del up
    if W > 1:
        pr = va.prove_sharded_rows_local(ps, mt, prep)
        same = bool(np.array_equal(pr.words, ref.words))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): va.prove_sharded_rows_local(ps, mt, prep)
        torch.cuda.synchronize()
        res["row_ranges_W%d_incl_upload" % W] = {"ms_per_proof": (time.perf_counter() - t0) / 3 * 1e3, "same_words": same}
print(json.dumps(res, indent=1))
open(out + "/sharded_cost_%s.json" % os.environ.get("LABEL", "new"), "w").write(json.dumps(res, indent=1))
P
