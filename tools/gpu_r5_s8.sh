#!/bin/bash
# Round 5, session 8: lone-proof latency A/B — column-dot finishes in one launch (default) against a finish launch each (VGPU_DOT_FINISH_BATCH=0), and the reduced
# openings of the second tallest height beside the tallest one's (VGPU_REDUCE_SPLIT=1); parity first
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5_s8; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sharded_prove_gpu.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "rc=$?"; tail -1 "$OUT/pytest.log"
VGPU_REDUCE_SPLIT=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or c2 or full" > "$OUT/pytest_split.log" 2>&1; echo "split rc=$?"; tail -1 "$OUT/pytest_split.log"
for rep in 1 2 3 4 5 6; do
  for cfg in "A:" "B:VGPU_DOT_FINISH_BATCH=0" "C:VGPU_REDUCE_SPLIT=1"; do
    lab=${cfg%%:*}; e=${cfg#*:}
    env $e python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 16 --warmup 4 --no-kernel-events > "$OUT/${lab}_rep${rep}.single.json" 2>>"$OUT/err.txt"
    env $e python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6 > "$OUT/${lab}_rep${rep}.three.json" 2>>"$OUT/err.txt"
  done
done
python - "$OUT" <<'P'
import glob, json, sys, statistics
for kind in ("single", "three"):
    for lab in "ABC":
        v = [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob("%s/%s_rep*.%s.json" % (sys.argv[1], lab, kind)))]
        print(kind, lab, "mean %.3f median %.3f" % (statistics.mean(v), statistics.median(v)), [round(x, 2) for x in v])
P
