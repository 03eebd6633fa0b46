#!/bin/bash
# Round 5, session 1 (run through gpurun from the repo root): the GPU suite on the changed quotient / LDE code, then ONE-session A/B records:
#   quotient per-point kernel (buffer loads x waves per SIMD), the fused-leaf-hash STAND-IN (the gate of VERDICT r04 item 1), LDE column groups.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5_s1; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"; tail -3 "$OUT/pytest_gpu.log"
(cd /tmp && timeout 60 rocprofv3 --list-avail > "$OUT/rocprof_list_avail.txt" 2>&1)
timeout 1500 bash tools/gpu_ab_libs.sh r5_s1/ab \
  q_old=build/variants/q_old/libvgpu.so q_s1w0=build/variants/q_s1w0/libvgpu.so q_s1w8=build/variants/q_s1w8/libvgpu.so q_s0w6=build/variants/q_s0w6/libvgpu.so \
  standin=build/variants/standin/libvgpu.so grp64=ENV:VGPU_LDE_GROUP_MB=64 grp128=ENV:VGPU_LDE_GROUP_MB=128 grp256=ENV:VGPU_LDE_GROUP_MB=256 > "$OUT/ab_summary.txt" 2>&1
tail -20 "$OUT/ab_summary.txt"
