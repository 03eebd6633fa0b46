#!/bin/bash
# Round 3, session k: lone-proof latency — the challenger step as an epilogue of the tree-top launch + the preprocessed commitment cached across
# proofs, against the library before both (build/variants/lat_old); parity tests first.
set -u
export TMPDIR=/tmp
O=gpurun_out/ab_latency; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sharded_prove_gpu.py -q -m gpu -x -p no:cacheprovider -k "not full_size and not c3" > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1
B="python bench.py --no-cpu-baseline --no-extra-legs --warmup 6"
for r in 0 1 2; do
  $B --inflight 1 --steps 12 > $O/new1_$r.json 2>$O/err.log
  VGPU_LIB_PATH=$PWD/build/variants/lat_old/libvgpu.so $B --inflight 1 --steps 12 > $O/old1_$r.json 2>>$O/err.log
done
for r in 0 1; do
  $B --steps 24 > $O/new3_$r.json 2>>$O/err.log
  VGPU_LIB_PATH=$PWD/build/variants/lat_old/libvgpu.so $B --steps 24 > $O/old3_$r.json 2>>$O/err.log
done
python - $O <<'PY'
import json, sys, glob
for lab in ("new1", "old1", "new3", "old3"):
    v = [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob(sys.argv[1] + "/" + lab + "_?.json"))]
    print(lab, " ".join("%.2f" % x for x in v))
PY
