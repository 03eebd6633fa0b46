#!/bin/bash
# Round-3 session b ON the MI355X box: parity of the fused LDE (the whole -m gpu suite), then A/B in one session: A = fused (default), B = VGPU_LDE_FUSED=0
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r03b}; mkdir -p "$OUT"; export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider -x --deselect tests/test_gpu_parity.py::test_full_size_c2_poseidon_mmcs ) > "$OUT/pytest_gpu.log" 2>&1
grep -E "passed|failed|error" "$OUT/pytest_gpu.log" | tail -3
grep -E "^(FAILED|ERROR)" "$OUT/pytest_gpu.log" | head -20
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-extra-legs > "$OUT/a_$i.json" 2>"$OUT/a_$i.err"
  VGPU_LDE_FUSED=0 python bench.py --no-cpu-baseline --no-extra-legs > "$OUT/b_$i.json" 2>/dev/null
done
python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 > "$OUT/a_single.json" 2>/dev/null
VGPU_LDE_FUSED=0 python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 > "$OUT/b_single.json" 2>/dev/null
python bench.py --no-cpu-baseline --no-extra-legs --workload c3 --steps 4 --warmup 2 > "$OUT/a_c3.json" 2>/dev/null
VGPU_LDE_FUSED=0 python bench.py --no-cpu-baseline --no-extra-legs --workload c3 --steps 4 --warmup 2 > "$OUT/b_c3.json" 2>/dev/null
python - <<P
import json
for f in ('a_1','b_1','a_2','b_2','a_single','b_single','a_c3','b_c3'):
    try:
        d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
        print(f, round(d['value'],2), round(d['ms_per_step'],2), {n: round(v,2) for n,v in k.items() if 'ntt' in n or 'lde' in n or 'bitrev' in n}, 'pool', d.get('hbm_pool_peak_bytes'))
    except Exception as e: print(f, 'ERR', e)
P
