#!/bin/bash
# Round 3, final session: the whole -m gpu suite, a proofs-in-flight sweep, a proofs-in-flight sweep earlier in the round (profiles/r03_inflight_sweep.txt), then the profile session (tools/profile_round.sh r03).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r03_pytest_gpu_final.log 2>&1
grep -E "passed|failed" gpurun_out/r03_pytest_gpu_final.log | tail -2; grep -E "^(FAILED|ERROR)" gpurun_out/r03_pytest_gpu_final.log | head
true
tools/profile_round.sh r03 2>&1 | tail -2
