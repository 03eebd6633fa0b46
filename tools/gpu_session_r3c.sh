#!/bin/bash
# Round 3, session c: A/B of the additive Montgomery reduction (-DVG_MONTY_ADD=1, build/variants/monty_add) with parity tests under it.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/ab_monty
VGPU_LIB_PATH=$PWD/build/variants/monty_nomad/libvgpu.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "not full_size" > gpurun_out/ab_monty/pytest_variant.log 2>&1
grep -E "passed|failed" gpurun_out/ab_monty/pytest_variant.log | tail -1
tools/gpu_ab_libs.sh ab_monty nomad0=build/variants/monty_nomad/libvgpu.so nomad1=build/variants/monty_nomad/libvgpu.so 2>&1 | tee gpurun_out/ab_monty/table.txt
