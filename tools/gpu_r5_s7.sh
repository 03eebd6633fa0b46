#!/bin/bash
# Round 5, session 7: the final tree — build() as the driver calls it, smoke(), the whole GPU suite (with the A/B-switch parity test), the default bench line,
# and power / clock readings beside a long three-in-flight run (VERDICT r04 Weak 6: is it the power limit that three proofs in flight run into?)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5_s7; mkdir -p "$OUT"; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 1500 python -m pytest tests -m gpu -x -q -rs > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
python bench.py > "$OUT/bench_default_final.json" 2> "$OUT/bench_default_final.err"
( python bench.py --no-cpu-baseline --no-extra-legs --steps 400 --warmup 6 > "$OUT/bench_long.json" 2>/dev/null ) &
BP=$!
sleep 6
for i in 1 2 3 4 5; do (rocm-smi --showpower --showclocks --showtemp --showperflevel 2>&1 | grep -v "^$" | head -40) >> "$OUT/rocm_smi_under_load.txt"; echo "---- sample $i" >> "$OUT/rocm_smi_under_load.txt"; sleep 1; done
wait $BP
(rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -40) > "$OUT/rocm_smi_idle.txt"
python - "$OUT" <<'P'
import json, sys
for f in ("bench_default_final.json", "bench_long.json"):
    d = json.loads(open(sys.argv[1] + "/" + f).read().strip().splitlines()[-1])
    print(f, "%.2f proofs/s %.2f ms/step lone %s same_proof %s clock %s" % (d["value"], d["ms_per_step"], d.get("prover_ms_single_proof_in_flight"), d.get("same_proof_as_cpu_baseline"), (d.get("shader_clock") or {}).get("GHz_mean")))
P
