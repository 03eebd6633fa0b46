#!/bin/bash
# Round 4, session 2 (ON the MI355X box): the issue-rate microbenchmarks with the Poseidon-16 in-register ceiling, and the BATCH-CEILING
# experiment of VERDICT r03 item 3 — a proof 4x taller has, kernel for kernel, the grids of a batch of four segments in one launch each
# (LDE passes, leaves, compress layers, quotient, reduced openings all scale with the rows) and one transcript's worth of latency-bound
# chains: its time / 4 bounds what `vgpu_prove_batch(K = 4)` in one context could reach per segment, and the same with 2 / 3 contexts in
# flight what several batches in flight could.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04s2; mkdir -p "$OUT"; export TMPDIR=/tmp
MICROBENCH_PREBUILT=1 python tools/microbench.py "$OUT/microbench.txt" > /dev/null 2> "$OUT/microbench.err"
B="python bench.py --no-cpu-baseline --no-extra-legs"
for rep in 1 2; do
  $B --inflight 3 > "$OUT/rows20_inflight3_$rep.json" 2> /dev/null
  $B --inflight 1 > "$OUT/rows20_inflight1_$rep.json" 2> /dev/null
  for m in 1 2 3; do $B --log-rows 22 --inflight $m --steps 6 --warmup 3 > "$OUT/rows22_inflight${m}_$rep.json" 2> "$OUT/rows22_inflight${m}_$rep.err"; done
done
$B --log-rows 18 --inflight 3 --steps 48 --warmup 12 > "$OUT/rows18_inflight3.json" 2> /dev/null
$B --log-rows 18 --inflight 6 --steps 48 --warmup 12 > "$OUT/rows18_inflight6.json" 2> /dev/null
python - "$OUT" <<'P'
import json, sys, glob, os
out = sys.argv[1]
for f in sorted(glob.glob(out + "/rows*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s ms_per_step %8.3f  proofs/s %7.2f  kernel_ms_total %s" % (os.path.basename(f), d["ms_per_step"], d["value"], d.get("kernel_ms_total_per_step")))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
P
