#!/bin/bash
# Round-2 GPU session A (run through gpurun from the repo root): parity suite incl. the full-size word-for-word cases,
# instruction-rate microbenchmarks + counter calibration, full-size oracle fixtures.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r02a
mkdir -p "$OUT"
export TMPDIR=/tmp
nproc > "$OUT/nproc.txt"; free -g > "$OUT/mem.txt"; lscpu | head -20 > "$OUT/lscpu.txt"
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > "$OUT/pytest.log" 2>&1
tail -5 "$OUT/pytest.log"
python tools/microbench.py "$OUT/microbench.txt" > /dev/null 2> "$OUT/microbench.err"
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $pass -d "$OUT/mb_$pass" -o run -- $ROOT/build/microbench copies > "$OUT/mb_$pass.log" 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d "$OUT/mb_valu" -o run -- $ROOT/build/microbench rates > "$OUT/mb_valu.log" 2>&1
cd "$ROOT"
python - <<'PY' > "$OUT/schema.txt" 2>&1
import glob, sqlite3
for p in glob.glob("gpurun_out/r02a/mb_valu/**/*.db", recursive=True):
    db = sqlite3.connect(p)
    for n, t, s in db.execute("select name, type, sql from sqlite_master"):
        print(t, n, (s or "")[:300].replace("\n", " "))
    break
PY
( time python tests/golden/make_golden.py --full "$OUT/golden" c2 c4 ) > "$OUT/golden.log" 2>&1
tail -3 "$OUT/golden.log"
du -sh "$OUT"
