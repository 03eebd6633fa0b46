#!/bin/bash
# per-launch table of a lone proof's kernels (tools/launch_table.py)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r02t}; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace -d "$OUT/trace" -o run -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 3 --warmup 1 > "$OUT/trace.json" 2> "$OUT/trace.err"
cd "$ROOT" && python tools/launch_table.py "$OUT/trace" k_ntt k_intt k_quotient k_bitrev k_reduce k_perm > "$OUT/launch_table.txt" 2>&1
rm -rf "$OUT/trace"
head -70 "$OUT/launch_table.txt"
