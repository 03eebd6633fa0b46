#!/bin/bash
# A/B of the LONE-proof latency of two library builds in one session: tools/gpu_ab_lone.sh <outdir> <path of library B> [repetitions]
# alternates `bench.py --inflight 1` (12 steps) between valida_amd/libvgpu.so (A) and B.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; mkdir -p "$OUT"; export TMPDIR=/tmp
for i in $(seq 1 ${3:-5}); do
  python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps ${STEPS:-12} > "$OUT/a_$i.json" 2>/dev/null
  VGPU_LIB_PATH=$ROOT/$2 python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps ${STEPS:-12} > "$OUT/b_$i.json" 2>/dev/null
done
python - "$OUT" ${3:-5} <<'P'
import json, sys
out, n = sys.argv[1], int(sys.argv[2])
for lab in ("a", "b"):
    v = [json.loads(open("%s/%s_%d.json" % (out, lab, i)).read().strip().splitlines()[-1])["ms_per_step"] for i in range(1, n + 1)]
    print(lab, " ".join("%.3f" % x for x in v), "median %.3f" % sorted(v)[len(v) // 2])
P
