#!/bin/bash
# Round 3, session j: Keccak round variants against the default (VK_KECCAK_PIN=1: a scheduling barrier behind every lane): build/variants/keccak_rows = -DVK_KECCAK_PIN=2
# (row by row); four alternating rounds of the default bench + the lone-proof leg; then the in-register microbenchmark of the variants.
set -u
export TMPDIR=/tmp
O=gpurun_out/ab_keccak_rows; mkdir -p $O
VGPU_LIB_PATH=$PWD/build/variants/keccak_rows/libvgpu.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "fib25 or commit or mmcs or root" > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1
B="python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6"
for r in 0 1 2 3; do
  $B > $O/def_$r.json 2>$O/err.log
  VGPU_LIB_PATH=$PWD/build/variants/keccak_rows/libvgpu.so $B > $O/rows_$r.json 2>>$O/err.log
done
for r in 0 1; do
  $B --inflight 1 --steps 12 > $O/def1_$r.json 2>>$O/err.log
  VGPU_LIB_PATH=$PWD/build/variants/keccak_rows/libvgpu.so $B --inflight 1 --steps 12 > $O/rows1_$r.json 2>>$O/err.log
done
python - $O <<'PY'
import json, sys, glob
for lab in ("def", "rows", "def1", "rows1"):
    v = [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob(sys.argv[1] + "/" + lab + "_?.json"))]
    print(lab, " ".join("%.2f" % x for x in v))
PY
MICROBENCH_PREBUILT=1 timeout 400 python tools/microbench.py gpurun_out/r03_microbench.txt > /dev/null 2>&1; grep -E "keccak_f1600" gpurun_out/r03_microbench.txt | cut -c1-330
