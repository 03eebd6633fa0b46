#!/bin/bash
# Grid cap of the throughput-bound Keccak kernels (VGPU_KECCAK_BLOCKS_PER_CU=k: 256 x k workgroups, grid-stride) on the round-4 kernels, one session
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r4cap}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fib25_proof or mixed_height or lane_pair or golden_fixture" 2>&1 | tail -1
echo "# blocks/CU proofs/s ms/step (three in flight) | lone ms" > "$OUT/cap.txt"
for rep in 1 2; do
for k in 0 2 3 4 5 6 8; do
  a=$(VGPU_KECCAK_BLOCKS_PER_CU=$k python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))")
  b=$(VGPU_KECCAK_BLOCKS_PER_CU=$k python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 --steps 12 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))")
  echo "$k $a | $b" >> "$OUT/cap.txt"
done
done
cat "$OUT/cap.txt"
