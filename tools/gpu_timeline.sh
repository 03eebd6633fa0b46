#!/bin/bash
# kernel timeline of a lone proof: where the GPU idles (tools/summarize_prof.py timeline)
set -u
ROOT=$(pwd); TAG=${1:-r02t}; OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats1" -o run -- python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 4 --warmup 1 > "$OUT/stats1.json" 2> "$OUT/stats1.err"
python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 6 --warmup 2 > "$OUT/bench_inflight1.json" 2> /dev/null
python $ROOT/bench.py --no-cpu-baseline > "$OUT/bench_default.json" 2> /dev/null
cd "$ROOT" && python tools/summarize_prof.py "$TAG" "$OUT/summary" && rm -rf "$OUT"/stats1
head -60 "$OUT/summary/${TAG}_timeline_inflight1.txt"
python -c "
import json
for f in ('bench_inflight1','bench_default'):
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['value'],2), round(d['ms_per_step'],2), d.get('prover_ms_single_proof_in_flight'), {k:round(v,2) for k,v in d['phase_ms'].items()})
"
