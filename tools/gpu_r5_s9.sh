#!/bin/bash
# Round 5, session 9: passes A and C of the fused LDE with compile-time tile I/O for the two hot shapes (uniform base + one per-thread offset) against the generic loops
# (-DVGPU_STRIDED_IO=0), alternating repetitions; parity first (LDE element for element up to 2^24 rows, whole proofs, sharded proofs)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5_s9; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -rs > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
for rep in 1 2 3 4 5; do
  for cfg in "new:" "old:VGPU_LIB_PATH=$ROOT/build/variants/io0/libvgpu.so"; do
    lab=${cfg%%:*}; e=${cfg#*:}
    env $e python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6 > "$OUT/${lab}_rep${rep}.three.json" 2>>"$OUT/err.txt"
    env $e python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 16 --warmup 4 > "$OUT/${lab}_rep${rep}.single.json" 2>>"$OUT/err.txt"
  done
done
for cfg in "new:" "old:VGPU_LIB_PATH=$ROOT/build/variants/io0/libvgpu.so"; do
  lab=${cfg%%:*}; e=${cfg#*:}
  env $e python bench.py --no-cpu-baseline --no-extra-legs --workload c3 --steps 6 --warmup 2 > "$OUT/${lab}_c3.three.json" 2>>"$OUT/err.txt"
done
python - "$OUT" <<'P'
import glob, json, sys, statistics
for kind in ("three", "single"):
    for lab in ("new", "old"):
        ds = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob("%s/%s_rep*.%s.json" % (sys.argv[1], lab, kind)))]
        v = [d["ms_per_step"] for d in ds]
        ka = statistics.mean(d["kernel_ms_per_step"]["k_lde_a"] for d in ds); kc = statistics.mean(d["kernel_ms_per_step"]["k_lde_c"] for d in ds)
        print(kind, lab, "mean %.3f median %.3f" % (statistics.mean(v), statistics.median(v)), [round(x, 2) for x in v], "k_lde_a %.3f k_lde_c %.3f ms" % (ka, kc))
for lab in ("new", "old"):
    d = json.loads(open("%s/%s_c3.three.json" % (sys.argv[1], lab)).read().strip().splitlines()[-1])
    print("c3", lab, "%.2f ms/step" % d["ms_per_step"], {n: round(v, 2) for n, v in d["kernel_ms_per_step"].items() if "lde" in n})
P
