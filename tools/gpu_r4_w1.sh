#!/bin/bash
# launch sequence of the sharded prover at W = 1 on one GPU (rocprofv3 --kernel-trace): tools/gpu_r4_w1.sh [W]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_w1; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
PYTHONPATH=$ROOT python $ROOT/tools/sharded_w1_trace.py ${1:-1}
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run -- python $ROOT/tools/sharded_w1_trace.py ${1:-1} > "$OUT/run.log" 2> "$OUT/run.err"
cd $ROOT && python - <<'P'
import sqlite3, glob, re
db = sqlite3.connect(glob.glob("gpurun_out/prof_w1/stats/**/*.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# last proof: from the last k_ingest group... the sharded prover ingests nothing per proof (uploaded traces): delimit by k_pow_grind
ends = [i for i, r in enumerate(rows) if "k_pow_grind" in r[0]]
a, b = ends[-2], ends[-1]
seg = rows[a + 1:b + 1]
t0 = seg[0][1]
def short(n):
    m = re.search(r"(k_\w+|__amd_rocclr_\w+)", n); return m.group(1) if m else n[:30]
per = {}
for n, s, e in seg:
    k = per.setdefault(short(n), [0, 0]); k[0] += 1; k[1] += e - s
with open("gpurun_out/prof_w1/w1_kernels.txt", "w") as f:
    f.write("span_us %.1f launches %d\n" % ((seg[-1][2] - t0) / 1e3, len(seg)))
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        f.write("%-28s %5d %10.1f\n" % (k, n, t / 1e3))
    f.write("# sequence: start_us dur_us kernel (>= 30 us)\n")
    for n, s, e in seg:
        if e - s >= 30e3: f.write("%9.1f %8.1f %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, short(n)))
print(open("gpurun_out/prof_w1/w1_kernels.txt").read()[:3000])
P
rm -rf "$OUT/stats"
