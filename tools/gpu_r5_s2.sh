#!/bin/bash
# Round 5, session 2: the per-point quotient kernel with the chips' interactions compiled in (A/B against the descriptor walk, per-chip launch times),
# the quotient parity tests on the new code, and the RCCL world-of-two test with its skip reason.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5_s2; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$OUT/pytest_parity.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_parity.log"; tail -3 "$OUT/pytest_parity.log"
timeout 400 python -m pytest tests/test_zz_sharded_multiprocess_gpu.py -m gpu -q -rs -k "rccl_fabric_world_of_two" > "$OUT/pytest_rccl2.log" 2>&1; tail -5 "$OUT/pytest_rccl2.log"
export VGPU_PROF_QUOTIENT_BY_CHIP=1
timeout 900 bash tools/gpu_ab_libs.sh r5_s2/ab q_pn0=build/variants/q_pn0/libvgpu.so q_pn1w0=build/variants/q_pn1w0/libvgpu.so q_pn1w8=build/variants/q_pn1w8/libvgpu.so > "$OUT/ab_summary.txt" 2>&1
python - "$OUT/ab" <<'P'
import json, sys
for lab in ["base0", "q_pn0", "q_pn1w0", "q_pn1w8", "base1"]:
    for kind in ("single", "three"):
        d = json.loads(open("%s/%s.%s.json" % (sys.argv[1], lab, kind)).read().strip().splitlines()[-1])
        k = d["kernel_ms_per_step"]
        q = {n.replace("k_quotient.", ""): round(v, 3) for n, v in k.items() if n.startswith("k_quotient")}
        print(lab, kind, "%.2f ms/step" % d["ms_per_step"], "quotient total %.3f" % sum(q.values()), q)
P
