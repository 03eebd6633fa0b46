#!/bin/bash
# Round 5, session 3: the GPU suite on the tree (compiled interactions for all 14 chips, Poseidon row layers), the VALU-sensitivity experiment
# (Keccak with 2 / 4 of its middle rounds left out: wrong proofs, timing only), and the Poseidon-MMCS leg by row-layer threshold.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5_s3; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -rs > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -4 "$OUT/pytest_gpu.log"
timeout 600 bash tools/gpu_ab_libs.sh r5_s3/ab kskip2=build/variants/kskip2/libvgpu.so kskip4=build/variants/kskip4/libvgpu.so > "$OUT/ab_summary.txt" 2>&1
for rm in 0 1024 4096 16384 0 4096; do
  for fl in 1 3; do
    VGPU_POSEIDON_ROW_MAX=$rm python bench.py --no-cpu-baseline --no-extra-legs --mmcs poseidon --inflight $fl --steps 6 --warmup 2 > "$OUT/pos_rm${rm}_f${fl}_$RANDOM.json" 2>>"$OUT/pos.err"
  done
done
python - "$OUT" <<'P'
import glob, json, sys
for lab in ["base0", "kskip2", "kskip4", "base1"]:
    for kind in ("single", "three"):
        d = json.loads(open("%s/ab/%s.%s.json" % (sys.argv[1], lab, kind)).read().strip().splitlines()[-1])
        k = d["kernel_ms_per_step"]
        print(lab, kind, "%.2f ms/step" % d["ms_per_step"], {n: round(v, 2) for n, v in k.items() if "keccak" in n})
for f in sorted(glob.glob(sys.argv[1] + "/pos_rm*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d["kernel_ms_per_step"]
    print(f.split("/")[-1], "%.2f ms/step" % d["ms_per_step"], {n: round(v, 2) for n, v in k.items() if "poseidon" in n})
P
