#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_leaves; mkdir -p "$OUT"; export TMPDIR=/tmp PYTHONPATH=$ROOT; cd /tmp
rocprofv3 --kernel-trace -d "$OUT/t" -o run -- python $ROOT/tools/leaves_scaling.py > "$OUT/run.log" 2> "$OUT/run.err"
cd $ROOT && python - <<'P'
import sqlite3, glob, re
db = sqlite3.connect(glob.glob("gpurun_out/prof_leaves/t/**/*.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
out = []
for n, s, e in rows:
    m = re.search(r"(k_keccak_leaves|k_keccak_compress)\b", n)
    if m and (e - s) > 20e3: out.append("%s %.1f" % (m.group(1), (e - s) / 1e3))
open("gpurun_out/prof_leaves/leaves.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
P
rm -rf "$OUT/t"; tail -3 "$OUT/run.err"
