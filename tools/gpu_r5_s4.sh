#!/bin/bash
# Round 5, session 4: reduced openings with 2 / 4 rows per thread (A/B by VGPU_REDUCE_ROWS), parity first.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5_s4; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -rs > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -4 "$OUT/pytest_gpu.log"
for R in 2 4; do VGPU_REDUCE_ROWS=$R timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$OUT/pytest_rows$R.log" 2>&1; echo "rows=$R rc=$?"; tail -1 "$OUT/pytest_rows$R.log"; done
timeout 900 bash tools/gpu_ab_libs.sh r5_s4/ab rows1=ENV:VGPU_REDUCE_ROWS=1 rows2=ENV:VGPU_REDUCE_ROWS=2 rows4=ENV:VGPU_REDUCE_ROWS=4 > "$OUT/ab_summary.txt" 2>&1
python - "$OUT" <<'P'
import json, sys
for lab in ["base0", "rows1", "rows2", "rows4", "base1"]:
    for kind in ("single", "three"):
        d = json.loads(open("%s/ab/%s.%s.json" % (sys.argv[1], lab, kind)).read().strip().splitlines()[-1])
        k = d["kernel_ms_per_step"]
        print(lab, kind, "%.2f ms/step" % d["ms_per_step"], {n: round(v, 3) for n, v in k.items() if "reduce" in n or "col_dot" in n or "fri" in n})
P
