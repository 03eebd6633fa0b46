#!/bin/bash
# launch sequence of a lone proof (rocprofv3 --kernel-trace, one pass): tools/gpu_r4_seq.sh
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_seq; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats1" -o run -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 10 --warmup 1 > "$OUT/stats1.json" 2> "$OUT/stats1.err"
cd $ROOT && python - <<'P'
import glob, sys
sys.path.insert(0, "tools")
import summarize_prof as sp
db = glob.glob("gpurun_out/prof_seq/stats1/**/*.db", recursive=True)
print(db)
sp.timeline(db[0], "gpurun_out/prof_seq/seq_timeline_inflight1.txt", "lone proof")
P
rm -rf "$OUT/stats1"; ls "$OUT"
