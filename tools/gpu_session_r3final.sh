#!/bin/bash
# Round 3, final session: the whole -m gpu suite, a proofs-in-flight sweep, then the profile session (tools/profile_round.sh r03).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r03_pytest_gpu_final.log 2>&1
grep -E "passed|failed" gpurun_out/r03_pytest_gpu_final.log | tail -2; grep -E "^(FAILED|ERROR)" gpurun_out/r03_pytest_gpu_final.log | head
for m in 2 3 4; do python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6 --inflight $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight $m', round(d['value'],2), round(d['ms_per_step'],2))"; done | tee gpurun_out/r03_inflight_sweep.txt
tools/profile_round.sh r03 2>&1 | tail -2
