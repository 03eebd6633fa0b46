#!/bin/bash
# three-in-flight A/B by environment switch, N alternations in one session: tools/gpu_r4_ab3.sh <switch> <outdir> [N]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${2:-r4ab3}; mkdir -p "$OUT"; export TMPDIR=/tmp
SW=$1; N=${3:-6}
for i in $(seq 1 $N); do
  python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6 > "$OUT/a3_$i.json" 2>/dev/null
  env "$SW" python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6 > "$OUT/b3_$i.json" 2>/dev/null
done
python - "$OUT" $N <<'P'
import json, sys
out, n = sys.argv[1], int(sys.argv[2])
for lab in ("a3", "b3"):
    v = [json.loads(open("%s/%s_%d.json" % (out, lab, i)).read().strip().splitlines()[-1])["ms_per_step"] for i in range(1, n + 1)]
    print(lab, " ".join("%.3f" % x for x in v), "median %.3f" % sorted(v)[len(v) // 2])
P
