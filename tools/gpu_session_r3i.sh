#!/bin/bash
# Round 3, session i: s_nop behind the half-rate instructions of a Montgomery reduction, with the Keccak no-ops in place everywhere:
# mh0 = none (build/variants/mulhi_nonop), default = behind v_mul_hi_u32, mh2 = also behind v_mul_lo_u32 (build/variants/mullo_nop).
# Four alternating rounds of the default bench (24 steps), then the Poseidon-MMCS leg (H-heavy) for the three.
set -u
export TMPDIR=/tmp
O=gpurun_out/ab_nops4; mkdir -p $O
VGPU_LIB_PATH=$PWD/build/variants/mullo_nop/libvgpu.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "not full_size" > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1
B="python bench.py --no-cpu-baseline --no-extra-legs --steps 24 --warmup 6"
for r in 0 1 2 3; do
  VGPU_LIB_PATH=$PWD/build/variants/mulhi_nonop/libvgpu.so $B > $O/mh0_$r.json 2>$O/err.log
  $B > $O/mh1_$r.json 2>>$O/err.log
  VGPU_LIB_PATH=$PWD/build/variants/mullo_nop/libvgpu.so $B > $O/mh2_$r.json 2>>$O/err.log
done
P="python bench.py --no-cpu-baseline --no-extra-legs --mmcs poseidon --steps 4 --warmup 1"
for r in 0 1; do
  VGPU_LIB_PATH=$PWD/build/variants/mulhi_nonop/libvgpu.so $P > $O/pos_mh0_$r.json 2>>$O/err.log
  $P > $O/pos_mh1_$r.json 2>>$O/err.log
  VGPU_LIB_PATH=$PWD/build/variants/mullo_nop/libvgpu.so $P > $O/pos_mh2_$r.json 2>>$O/err.log
done
python - $O <<'PY'
import json, sys, glob
for lab in ("mh0", "mh1", "mh2", "pos_mh0", "pos_mh1", "pos_mh2"):
    v = [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob(sys.argv[1] + "/" + lab + "_?.json"))]
    print(lab, " ".join("%.2f" % x for x in v))
PY
