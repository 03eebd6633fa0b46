#!/bin/bash
# Round 5, session 6: GPU suite on the native permutation-trace kernel; A/B VGPU_PERM_NATIVE (repeated, alternating); C3 after the k_lde_mid14 change
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5_s6; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -rs > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
for rep in 1 2 3 4; do
  for PN in 0 1; do
    VGPU_PERM_NATIVE=$PN python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6 > "$OUT/pn${PN}_rep${rep}.three.json" 2>>"$OUT/err.txt"
    VGPU_PERM_NATIVE=$PN python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 > "$OUT/pn${PN}_rep${rep}.single.json" 2>>"$OUT/err.txt"
  done
done
python bench.py --no-cpu-baseline --no-extra-legs --workload c3 --steps 6 --warmup 2 > "$OUT/c3.three.json" 2>>"$OUT/err.txt"
python bench.py --no-cpu-baseline --no-extra-legs --workload c3 --steps 4 --warmup 1 --inflight 1 > "$OUT/c3.single.json" 2>>"$OUT/err.txt"
python - "$OUT" <<'P'
import glob, json, sys, statistics
for kind in ("three", "single"):
    for PN in (0, 1):
        ds = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob("%s/pn%d_rep*.%s.json" % (sys.argv[1], PN, kind)))]
        v = [d["ms_per_step"] for d in ds]
        k = [d["kernel_ms_per_step"].get("k_perm_recip", 0) for d in ds]
        print(kind, "perm_native", PN, "mean %.3f" % statistics.mean(v), [round(x, 2) for x in v], "k_perm_recip ms", [round(x, 3) for x in k])
for kind in ("three", "single"):
    d = json.loads(open("%s/c3.%s.json" % (sys.argv[1], kind)).read().strip().splitlines()[-1])
    print("c3", kind, "%.2f ms/step" % d["ms_per_step"], {n: round(v, 2) for n, v in list(d["kernel_ms_per_step"].items())[:8]})
P
