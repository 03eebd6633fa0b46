#!/bin/bash
# Round 5, session 5: repeated, alternating runs of the reduced-openings variants (the step differences are at the noise level of single runs)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5_s5; mkdir -p "$OUT"; export TMPDIR=/tmp
for rep in 1 2 3 4 5; do
  for R in 1 2 4 0; do
    VGPU_REDUCE_ROWS=$R python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6 > "$OUT/rows${R}_rep${rep}.three.json" 2>>"$OUT/err.txt"
    VGPU_REDUCE_ROWS=$R python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 > "$OUT/rows${R}_rep${rep}.single.json" 2>>"$OUT/err.txt"
  done
done
python - "$OUT" <<'P'
import glob, json, sys, statistics
for kind in ("three", "single"):
    for R in (1, 2, 4, 0):
        v = [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob("%s/rows%d_rep*.%s.json" % (sys.argv[1], R, kind)))]
        print(kind, "rows", R, "n", len(v), "mean %.3f" % statistics.mean(v), "min %.3f max %.3f" % (min(v), max(v)), [round(x, 2) for x in v])
P
