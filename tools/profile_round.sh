#!/bin/bash
# Run ON the MI355X box (through gpurun) from the repo root: collects the rocprofv3 evidence bench.py's roofline
# figures are checked against.  Counters are collected in their own passes (never together with API traces).
#   tools/profile_round.sh r01        ->  gpurun_out/prof_r01/{stats1,stats2,pmc_valu,pmc_fetch,pmc_write}/...
# Then, back in the dev container:  python tools/summarize_prof.py r01   (writes profiles/r01_*.{csv,json})
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline"
# bench line alone first (no profiler attached)
$B > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
$B --inflight 1 > "$OUT/bench_inflight1.json" 2> "$OUT/bench_inflight1.err"
# 1) kernel trace + stats of the default command and of the one-proof-in-flight command
rocprofv3 --kernel-trace --stats -d "$OUT/stats2" -o run -- $B > "$OUT/stats2.json" 2> "$OUT/stats2.err"
rocprofv3 --kernel-trace --stats -d "$OUT/stats1" -o run -- $B --inflight 1 --steps 5 --warmup 2 > "$OUT/stats1.json" 2> "$OUT/stats1.err"
# 2) counters, separate passes, one proof in flight, few steps
P="$B --inflight 1 --steps 2 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc_valu" -o run -- $P > "$OUT/pmc_valu.json" 2> "$OUT/pmc_valu.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o run -- $P > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o run -- $P > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
ls -R "$OUT" | head -60
