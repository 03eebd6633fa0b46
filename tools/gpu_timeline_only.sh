#!/bin/bash
# Only the lone-proof kernel trace of tools/profile_round.sh (one rocprofv3 run) and its timeline summary: tools/gpu_timeline_only.sh <tag>
set -u
TAG=${1:-tl}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats1" -o run -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 10 --warmup 1 > "$OUT/stats1.json" 2> "$OUT/stats1.err"
cd "$ROOT" && python tools/summarize_prof.py "$TAG" "$OUT/summary" > /dev/null 2>&1; rm -rf "$OUT/stats1"
grep -E "span_us|phases_us|fri_layer_us" "$OUT/summary/${TAG}_timeline_inflight1.txt"
