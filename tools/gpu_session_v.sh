#!/bin/bash
# NTT change: LDE / commit / proof parity, then A/B against the previous build in one session
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r02v}; mkdir -p "$OUT"; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider -k "lde or commit or fib25_proof or full_size_c4 or blowup or sharded_commit" ) > "$OUT/parity.log" 2>&1
tail -5 "$OUT/parity.log"
bash tools/gpu_ab.sh ${2:-build/ab/libvgpu_prev.so} ${1:-r02v}_ab
