set -u
OUT=gpurun_out/r5_s11; mkdir -p $OUT
export VGPU_PROF_QUOTIENT_BY_CHIP=1
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 > $OUT/single_rep$rep.json 2>>$OUT/err.txt
  python bench.py --no-cpu-baseline --no-extra-legs > $OUT/three_rep$rep.json 2>>$OUT/err.txt
done
python - <<'P'
import glob, json, statistics
def avg(kind, chip):
    return statistics.mean(json.loads(open(p).read().strip().splitlines()[-1])["kernel_ms_per_step"].get("k_quotient." + chip, 0) for p in sorted(glob.glob("gpurun_out/r5_s11/%s_rep*.json" % kind)))
for chip in ("cpu", "mem", "add"):
    a, b = avg("single", chip), avg("three", chip)
    print("k_quotient_pt<%s>: alone %.0f us, three proofs in flight %.0f us per launch (x%.2f)" % (chip, a * 1e3, b * 1e3, b / a))
for k in ("k_keccak_compress", "k_lde_c", "k_lde_mid12", "k_lde_a", "k_reduce_openings"):
    a = statistics.mean(json.loads(open(p).read().strip().splitlines()[-1])["kernel_ms_per_step"][k] for p in sorted(glob.glob("gpurun_out/r5_s11/single_rep*.json")))
    b = statistics.mean(json.loads(open(p).read().strip().splitlines()[-1])["kernel_ms_per_step"][k] for p in sorted(glob.glob("gpurun_out/r5_s11/three_rep*.json")))
    print("%s: alone %.2f ms per proof, three in flight %.2f (x%.2f)" % (k, a, b, b / a))
P
