#!/bin/bash
# Round 4, session 3 (ON the MI355X box): the Poseidon-16 permutation with deferred sparse rounds and signed-Montgomery S-boxes — in-register
# ceiling, parity tests, and the Poseidon-MMCS bench leg A/B against the round-3 form (libraries alternating inside this session).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04s3; mkdir -p "$OUT"; export TMPDIR=/tmp
MICROBENCH_PREBUILT=1 python tools/microbench.py "$OUT/microbench.txt" > /dev/null 2> "$OUT/microbench.err"
grep "poseidon16" "$OUT/microbench.txt"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "poseidon or Poseidon or pow or grind" > "$OUT/pytest_poseidon.log" 2>&1; tail -3 "$OUT/pytest_poseidon.log"
B="python bench.py --no-cpu-baseline --no-extra-legs --mmcs poseidon --steps 9 --warmup 3"
for rep in 1 2; do
  for v in ${VARIANTS:-new pos_r3 pos_defer_only pos_sbox_only}; do
    if [ $v = new ]; then L=; else L="VGPU_LIB_PATH=$ROOT/build/variants/$v/libvgpu.so"; fi
    env $L $B > "$OUT/${v}_three_$rep.json" 2> /dev/null
    env $L $B --inflight 1 > "$OUT/${v}_single_$rep.json" 2> /dev/null
  done
done
python - "$OUT" <<'P'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*_[ts]*_[12].json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d["kernel_ms_per_step"]
        print("%-30s ms_per_step %8.3f  compress %.2f leaves %.2f top %.2f  roofline frac %.3f" % (os.path.basename(f), d["ms_per_step"], k.get("k_poseidon_compress", 0), k.get("k_poseidon_leaves", 0), k.get("k_poseidon_top", 0), d["roofline"]["frac"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
P
