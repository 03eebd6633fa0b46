#!/bin/bash
# Round 3, session f: Poseidon MDS layer as CRT blocks (default build) against the transforms (build/variants/pos_fft, -DVGPU_POSEIDON_MDS=1)
set -u
export TMPDIR=/tmp
O=gpurun_out/ab_poseidon_blocks; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "poseidon or pow or grind" > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1
B="python bench.py --no-cpu-baseline --no-extra-legs"
for r in 0 1; do
  $B --mmcs poseidon --steps 4 --warmup 1 > $O/blocks$r.pos.json 2>$O/err.log
  VGPU_LIB_PATH=$PWD/build/variants/pos_fft/libvgpu.so $B --mmcs poseidon --steps 4 --warmup 1 > $O/fft$r.pos.json 2>>$O/err.log
done
python - $O <<'P'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["kernel_ms_per_step"]
        print(os.path.basename(f), "%.2f p/s %.2f ms" % (d["value"], d["ms_per_step"]), {n: round(v, 2) for n, v in k.items() if "poseidon" in n or "pow" in n})
    except Exception as e:
        print(f, "ERR", e)
P
