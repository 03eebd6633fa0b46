#!/bin/bash
# Run ON the MI355X box (through gpurun) from the repo root: collects the rocprofv3 evidence bench.py's roofline
# figures are checked against.  Counters are collected in their own passes (never together with API traces).
#   tools/profile_round.sh r01        ->  gpurun_out/prof_r01/{stats1,stats2,pmc_valu,pmc_fetch,pmc_write}/...
# The summaries land in gpurun_out/prof_<tag>/summary/ (copied back by gpurun); commit them under profiles/.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline"
Q="--sustained-seconds 0"  # the profiled passes run the contract region only (a 6 s sustained region under rocprofv3 would be all trace)
# bench line alone first (no profiler attached)
$B > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
$B --inflight 1 > "$OUT/bench_inflight1.json" 2> "$OUT/bench_inflight1.err"
# 1) kernel trace + stats of the default command and of the one-proof-in-flight command
rocprofv3 --kernel-trace --stats -d "$OUT/stats2" -o run -- $B $Q --no-extra-legs --steps 24 --warmup 3 > "$OUT/stats2.json" 2> "$OUT/stats2.err"
rocprofv3 --kernel-trace --stats -d "$OUT/stats1" -o run -- $B $Q --no-extra-legs --inflight 1 --steps 10 --warmup 1 > "$OUT/stats1.json" 2> "$OUT/stats1.err"
# 2) counters, separate passes, one proof in flight, few steps
P="env VGPU_BENCH_SHARDED=0 $B $Q --inflight 1 --steps 2 --warmup 1"  # without the sharded-prover leg: its shard-size launches would mix into the per-launch averages
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/pmc_valu" -o run -- $P > "$OUT/pmc_valu.json" 2> "$OUT/pmc_valu.err"
# VALU issue utilisation per SIMD: busy cycles of the CUs and of the whole GPU next to the VALU-active wave cycles
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_busy" -o run -- $P > "$OUT/pmc_busy.json" 2> "$OUT/pmc_busy.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o run -- $P > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o run -- $P > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
# full default bench line (with the CPU baseline leg) and the other single-GPU configs
python $ROOT/bench.py > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
for w in c3 c4; do $B --workload $w --steps 6 --warmup 2 > "$OUT/bench_$w.json" 2> /dev/null; done
$B --mmcs poseidon --steps 4 --warmup 1 > "$OUT/bench_poseidon.json" 2> "$OUT/bench_poseidon.err"
# BASELINE.json configs[3] to the letter: the ALU / range-check heavy program with the Poseidon Merkle tree
$B --workload c4 --mmcs poseidon --steps 4 --warmup 1 > "$OUT/bench_c4_poseidon.json" 2> /dev/null
# summarise here: the databases are too big to be copied back, the summaries are not
cd "$ROOT" && python tools/summarize_prof.py "$TAG" "$OUT/summary" && rm -rf "$OUT"/stats1 "$OUT"/stats2 "$OUT"/pmc_valu "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_busy
ls "$OUT/summary"
