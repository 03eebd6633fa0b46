#!/bin/bash
# A/B of the sharded prover on one GPU: library A = valida_amd/libvgpu.so, B = "$1"; W = 1, 2, 4, 8 alternating, two passes (tools/sharded_w1_trace.py)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${2:-r4shab}; mkdir -p "$OUT"; export TMPDIR=/tmp PYTHONPATH=$ROOT
: > "$OUT/ab.txt"
for rep in 1 2; do for w in 1 2 4 8; do
  echo "A $(python tools/sharded_w1_trace.py $w)" >> "$OUT/ab.txt"
  echo "B $(VGPU_LIB_PATH=$ROOT/$1 python tools/sharded_w1_trace.py $w)" >> "$OUT/ab.txt"
done; done
cat "$OUT/ab.txt"
