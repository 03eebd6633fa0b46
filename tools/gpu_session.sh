#!/bin/bash
# ONE parametrised GPU session (run ON the MI355X box through gpurun, from the repo root).  It replaces the per-session scripts of rounds 4 and 5
# (tools/gpu_r4_*.sh, gpu_r5_s*.sh, gpu_ab*.sh: in the history up to commit 8544c76; what each of them measured is recorded in profiles/HISTORY.md and in
# the header of the profiles/r0N_ab_*.txt file it produced).
#
#   tools/gpu_session.sh <tag> [options] [-- candidate ...]
#
# options
#   --tests "<pytest -k expression>" | --tests all | --tests none    parity first (default: a quick subset); under EVERY candidate when --tests-per-candidate
#   --reps N            interleaved repetitions of every candidate (default 3; boxes differ by a few per cent, so A and B always share a session)
#   --legs "three lone" which bench legs to run per candidate: three (default run, three proofs in flight), lone (--inflight 1 --no-kernel-events),
#                       lonek (one proof in flight WITH per-launch events: exclusive per-kernel times), full (with the extra legs and the CPU baseline, once per candidate),
#                       c3, c4, poseidon, poseidonlone, poseidonlonek, c4poseidon
#   --bench "<args>"    extra bench.py arguments for the three / lone legs (e.g. "--steps 24 --warmup 6")
#   --profile           tools/profile_round.sh <tag> afterwards (rocprofv3 kernel stats + PMC passes + summaries)
#   --timeline          kernel timeline of a lone proof (rocprofv3 --kernel-trace; tools/summarize_prof.py timeline)
#   --microbench        tools/microbench.py (instruction rates, copy ceiling) into <out>/microbench.txt
# candidates (default: the in-tree library alone)
#   label                       the in-tree library, no switch (the baseline; always put one first)
#   label=ENV:VAR=v[,VAR2=v2]   the in-tree library under environment switches
#   label=LIB:path/libvgpu.so   a library built by tools/build_variant.py (through VGPU_LIB_PATH), optionally followed by ,VAR=v
# Output: gpurun_out/<tag>/*.json + summary.txt (copied back by gpurun); commit what is to be judged under profiles/.
set -u
TAG=${1:?usage: tools/gpu_session.sh <tag> [options] [-- candidates]}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
TESTS="fib25_proof or mixed_height or golden_fixture or full_size_c2 or poseidon_mmcs_commit"; PER_CAND=0; REPS=3; LEGS="three lone"; BENCH=""; PROFILE=0; TIMELINE=0; MICRO=0
while [ $# -gt 0 ]; do
  case "$1" in
    --tests) TESTS=$2; shift 2;;
    --tests-per-candidate) PER_CAND=1; shift;;
    --reps) REPS=$2; shift 2;;
    --legs) LEGS=$2; shift 2;;
    --bench) BENCH=$2; shift 2;;
    --profile) PROFILE=1; shift;;
    --timeline) TIMELINE=1; shift;;
    --microbench) MICRO=1; shift;;
    --) shift; break;;
    *) echo "unknown option $1" >&2; exit 2;;
  esac
done
CANDS=("$@"); [ ${#CANDS[@]} -eq 0 ] && CANDS=(base)

cand_env() {  # prints the env assignments of a candidate spec, one per line
  local spec=$1 val
  [[ "$spec" == *=* ]] || return 0
  val=${spec#*=}
  if [[ "$val" == LIB:* ]]; then
    val=${val#LIB:}; echo "VGPU_LIB_PATH=$ROOT/${val%%,*}"
    [[ "$val" == *,* ]] && tr ',' '\n' <<< "${val#*,}"
  elif [[ "$val" == ENV:* ]]; then tr ',' '\n' <<< "${val#ENV:}"
  else echo "bad candidate $spec" >&2; exit 2; fi
}
run_tests() {  # label, env...
  local lab=$1; shift
  [ "$TESTS" = none ] && return 0
  local k=(); [ "$TESTS" != all ] && k=(-k "$TESTS")
  env "$@" timeout 1500 python -m pytest tests -m gpu -x -q "${k[@]}" > "$OUT/pytest_$lab.log" 2>&1
  echo "pytest[$lab] rc=$? $(tail -1 "$OUT/pytest_$lab.log")" | tee -a "$OUT/summary.txt"
}
B="python bench.py --no-cpu-baseline"
run_leg() {  # label rep leg env...
  local lab=$1 rep=$2 leg=$3; shift 3
  local f="$OUT/${lab}_rep${rep}.${leg}.json"
  case "$leg" in
    three)      env "$@" $B --no-extra-legs $BENCH > "$f" 2>>"$OUT/$lab.err";;
    lone)       env "$@" $B --no-extra-legs --inflight 1 --no-kernel-events --sustained-seconds 0 $BENCH > "$f" 2>>"$OUT/$lab.err";;
    lonek)      env "$@" VGPU_PROF_QUOTIENT_BY_CHIP=1 $B --no-extra-legs --inflight 1 --sustained-seconds 0 $BENCH > "$f" 2>>"$OUT/$lab.err";;  # one proof in flight WITH per-launch events: exclusive per-kernel times (kernel_ms_per_step), quotient launches named by chip
    full)       [ "$rep" = 1 ] && env "$@" python bench.py > "$f" 2>>"$OUT/$lab.err";;
    c3)         env "$@" $B --no-extra-legs --workload c3 --steps 6 --warmup 2 --sustained-seconds 0 > "$f" 2>>"$OUT/$lab.err";;
    c4)         env "$@" $B --no-extra-legs --workload c4 --sustained-seconds 0 > "$f" 2>>"$OUT/$lab.err";;
    poseidon)   env "$@" $B --no-extra-legs --mmcs poseidon --steps 6 --warmup 2 --sustained-seconds 0 > "$f" 2>>"$OUT/$lab.err";;
    poseidonlone) env "$@" $B --no-extra-legs --mmcs poseidon --inflight 1 --steps 6 --warmup 2 --no-kernel-events --sustained-seconds 0 > "$f" 2>>"$OUT/$lab.err";;
    poseidonlonek) env "$@" $B --no-extra-legs --mmcs poseidon --inflight 1 --steps 6 --warmup 2 --sustained-seconds 0 > "$f" 2>>"$OUT/$lab.err";;
    c4poseidon) env "$@" $B --no-extra-legs --workload c4 --mmcs poseidon --steps 4 --warmup 1 --sustained-seconds 0 > "$f" 2>>"$OUT/$lab.err";;
    *) echo "unknown leg $leg" >&2;;
  esac
}
first=1
for spec in "${CANDS[@]}"; do
  lab=${spec%%=*}; mapfile -t E < <(cand_env "$spec")
  if [ $first = 1 ] || [ $PER_CAND = 1 ]; then run_tests "$lab" "${E[@]}"; fi
  first=0
done
for rep in $(seq 1 "$REPS"); do
  for spec in "${CANDS[@]}"; do
    lab=${spec%%=*}; mapfile -t E < <(cand_env "$spec")
    for leg in $LEGS; do run_leg "$lab" "$rep" "$leg" "${E[@]}"; done
  done
done
python - "$OUT" "$LEGS" "${CANDS[@]%%=*}" <<'P' | tee -a "$OUT/summary.txt"
import glob, json, statistics, sys
out, legs, labs = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
for leg in legs:
    for lab in labs:
        v, s, extra = [], [], ""
        for f in sorted(glob.glob("%s/%s_rep*.%s.json" % (out, lab, leg))):
            try:
                d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
                v.append(d["ms_per_step"])
                if d.get("sustained"):
                    s.append(d["sustained"]["ms_per_step"])
                if d.get("hbm_pool_peak_bytes"):
                    extra = " pool_peak %.1f GB" % (d["hbm_pool_peak_bytes"] / 1e9)
            except Exception as e:  # noqa: BLE001
                extra += " [%s: %s]" % (f.rsplit("/", 1)[-1], e)
        if leg in ("lonek", "poseidonlonek"):
            ks = {}
            for f in sorted(glob.glob("%s/%s_rep*.%s.json" % (out, lab, leg))):
                try:
                    d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
                    for k, x in d["kernel_ms_per_step"].items():
                        ks.setdefault(k, []).append(x)
                except Exception:  # noqa: BLE001
                    pass
            print("%-12s %-14s kernel ms per lone proof (median over reps): %s" % (leg, lab, ", ".join("%s %.3f" % (k, statistics.median(x)) for k, x in sorted(ks.items(), key=lambda kv: -statistics.median(kv[1]))[:40])))
        if v:
            print("%-12s %-14s ms/step median %.3f mean %.3f %s%s%s" % (leg, lab, statistics.median(v), statistics.mean(v), [round(x, 2) for x in v],
                  (" | sustained median %.3f %s" % (statistics.median(s), [round(x, 2) for x in s])) if s else "", extra))
        else:
            print("%-12s %-14s no result%s" % (leg, lab, extra))
P
[ $MICRO = 1 ] && python tools/microbench.py > "$OUT/microbench.txt" 2>"$OUT/microbench.err"
if [ $TIMELINE = 1 ]; then
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/stats1" -o run -- python "$ROOT/bench.py" --no-cpu-baseline --no-extra-legs --inflight 1 --steps 4 --warmup 1 --sustained-seconds 0 > "$OUT/stats1.json" 2> "$OUT/stats1.err" )
  python tools/summarize_prof.py "$TAG" "$OUT/summary" && rm -rf "$OUT/stats1"
fi
[ $PROFILE = 1 ] && bash tools/profile_round.sh "$TAG"
exit 0
