#!/bin/bash
# Round 3, session g: s_nop 0 behind every v_alignbit_b32 of the thread-per-permutation Keccak kernels (build/variants/mulhi_nop, -DVK_ALIGNBIT_NOP=1)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/ab_mulhi_nop
VGPU_LIB_PATH=$PWD/build/variants/mulhi_nop/libvgpu.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "not full_size" > gpurun_out/ab_mulhi_nop/pytest.log 2>&1
grep -E "passed|failed" gpurun_out/ab_mulhi_nop/pytest.log | tail -1
tools/gpu_ab_libs.sh ab_mulhi_nop mh0=build/variants/mulhi_nop/libvgpu.so mh1=build/variants/mulhi_nop/libvgpu.so > gpurun_out/ab_mulhi_nop/table.txt 2>&1
python - <<'P'
import json
for lab in ['base0','mh0','mh1','base1']:
    row=[lab]
    for kind in ['single','three']:
        d=json.loads(open(f'gpurun_out/ab_mulhi_nop/{lab}.{kind}.json').read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
        row.append("%s %.2f p/s %.2f ms" % (kind, d['value'], d['ms_per_step'])); row.append({n:round(v,2) for n,v in k.items() if any(t in n for t in ('lde','quotient','reduce','perm_recip'))})
    print(*row)
P
