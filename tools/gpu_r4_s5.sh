#!/bin/bash
# Round-4 session 5: parity of the multi-layer Keccak launches (k_keccak_levels_pair), then an A/B by environment switch in one session:
# A = default (new), B = "$1" (e.g. VGPU_KECCAK_LEVELS=0 = the round-3 launches).  tools/gpu_r4_s5.sh <switch> <outdir> [reps] [pytest -k]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${2:-r4s5}; mkdir -p "$OUT"; export TMPDIR=/tmp
SW=$1; N=${3:-5}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ${4:+-k "$4"} > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"
for i in $(seq 1 $N); do
  python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 12 > "$OUT/a1_$i.json" 2>/dev/null
  env "$SW" python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 12 > "$OUT/b1_$i.json" 2>/dev/null
done
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-extra-legs > "$OUT/a3_$i.json" 2>/dev/null
  env "$SW" python bench.py --no-cpu-baseline --no-extra-legs > "$OUT/b3_$i.json" 2>/dev/null
done
python - "$OUT" $N <<'P'
import json, sys
out, n = sys.argv[1], int(sys.argv[2])
for lab, m in (("a1", n), ("b1", n), ("a3", 3), ("b3", 3)):
    v = [json.loads(open("%s/%s_%d.json" % (out, lab, i)).read().strip().splitlines()[-1])["ms_per_step"] for i in range(1, m + 1)]
    print(lab, " ".join("%.3f" % x for x in v), "median %.3f" % sorted(v)[len(v) // 2])
P
