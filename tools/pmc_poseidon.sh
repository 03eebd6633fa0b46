#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pos_pmc; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
P="env VGPU_BENCH_SHARDED=0 python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --sustained-seconds 0 --mmcs poseidon --inflight 1 --steps 2 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$OUT/pmc1" -o run -- $P > "$OUT/p1.json" 2> "$OUT/p1.err"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS -d "$OUT/pmc2" -o run -- $P > "$OUT/p2.json" 2> "$OUT/p2.err"
cd $ROOT
for d in pmc1 pmc2; do python tools/pmc_query.py "$OUT/$d" > "$OUT/$d.txt" 2>&1; rm -rf "$OUT/$d"; done
grep -i "poseidon" "$OUT/pmc1.txt" | head; grep -i poseidon "$OUT/pmc2.txt" | head
