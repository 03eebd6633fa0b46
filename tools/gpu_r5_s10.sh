set -u
OUT=gpurun_out/r5_s10; mkdir -p $OUT
for rep in 1 2 3; do for f in 2 3 4; do python bench.py --no-cpu-baseline --no-extra-legs --inflight $f --steps 24 --warmup 6 > $OUT/f${f}_rep${rep}.json 2>>$OUT/err.txt; done; done
python - <<'P'
import glob, json, statistics
for f in (2, 3, 4):
    v = [json.loads(open(p).read().strip().splitlines()[-1])["ms_per_step"] for p in sorted(glob.glob("gpurun_out/r5_s10/f%d_rep*.json" % f))]
    print("inflight", f, "mean %.3f" % statistics.mean(v), [round(x, 2) for x in v])
P
