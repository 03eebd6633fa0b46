#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r02g}; mkdir -p "$OUT"; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 ) > "$OUT/pytest.log" 2>&1
tail -9 "$OUT/pytest.log"
bash tools/gpu_timeline.sh ${1:-r02g}t 2>&1 | tail -45
