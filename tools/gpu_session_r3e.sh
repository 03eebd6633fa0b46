#!/bin/bash
# Round 3, session e: the Poseidon-16 MDS layer as a cyclic convolution (butterfly.hpp) + the proof-of-work search on the MMCS kernels'
# permutation, against the library before the change (build/variants/pos_old): parity tests, then the Poseidon legs and the default bench.
set -u
export TMPDIR=/tmp
O=gpurun_out/ab_poseidon_fft; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "poseidon or pow or grind or fib25 or c1" > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1
B="python bench.py --no-cpu-baseline --no-extra-legs"
for r in 0 1; do
  $B --mmcs poseidon --steps 4 --warmup 1 > $O/new$r.pos.json 2>$O/err.log
  VGPU_LIB_PATH=$PWD/build/variants/pos_old/libvgpu.so $B --mmcs poseidon --steps 4 --warmup 1 > $O/old$r.pos.json 2>>$O/err.log
done
$B --workload c4 --mmcs poseidon --steps 4 --warmup 1 > $O/new.c4pos.json 2>>$O/err.log
$B --inflight 1 > $O/new.single.json 2>>$O/err.log
VGPU_LIB_PATH=$PWD/build/variants/pos_old/libvgpu.so $B --inflight 1 > $O/old.single.json 2>>$O/err.log
$B > $O/new.three.json 2>>$O/err.log
python - $O <<'P'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernel_ms_per_step"]
        print(os.path.basename(f), "%.2f p/s %.2f ms" % (d["value"], d["ms_per_step"]), {n: round(v, 2) for n, v in k.items() if "poseidon" in n or "pow" in n})
    except Exception as e:
        print(f, "ERR", e)
P
