#!/bin/bash
# Round 3, session h: three proofs in flight, four alternating rounds of: no no-ops (build/variants/keccak_nonop), the default (s_nop behind v_alignbit_b32),
# and additionally s_nop behind the v_mul_hi_u32 of every Montgomery reduction (build/variants/mulhi_nop).  24 steps each.
set -u
export TMPDIR=/tmp
O=gpurun_out/ab_nops3; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6"
for r in 0 1 2 3; do
  VGPU_LIB_PATH=$PWD/build/variants/keccak_nonop/libvgpu.so $B > $O/nonop$r.json 2>$O/err.log
  $B > $O/alignnop$r.json 2>>$O/err.log
  VGPU_LIB_PATH=$PWD/build/variants/mulhi_nop/libvgpu.so $B > $O/mulhinop$r.json 2>>$O/err.log
done
python - $O <<'P'
import json, sys, glob, os
for lab in ("nonop", "alignnop", "mulhinop"):
    v = []
    for f in sorted(glob.glob(sys.argv[1] + "/" + lab + "?.json")):
        d = json.loads(open(f).read().strip().splitlines()[-1]); v.append(d["ms_per_step"])
    print(lab, " ".join("%.2f" % x for x in v), "median %.2f" % sorted(v)[len(v) // 2 - (0 if len(v) % 2 else 1)] if v else "")
P
