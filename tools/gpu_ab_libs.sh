#!/bin/bash
# A/B of several library builds in ONE session (boxes differ by a few per cent): tools/gpu_ab_libs.sh <outdir> <label=path-or-ENV:VAR=val> ...
#   label=build/variants/x/libvgpu.so   a library built by tools/build_variant.py (VGPU_LIB_PATH)
#   label=ENV:VGPU_FOO=0                the default library under an environment switch
# Each candidate: one lone-proof run (--inflight 1) and one default run (three in flight); the default library first and last.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; shift; mkdir -p "$OUT"; export TMPDIR=/tmp
run() {  # label kind value
  local envs=()
  if [ "$2" = lib ]; then envs=(VGPU_LIB_PATH="$ROOT/$3"); elif [ "$2" = env ]; then envs=("$3"); fi
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-extra-legs --inflight 1 > "$OUT/$1.single.json" 2>"$OUT/$1.err"
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-extra-legs > "$OUT/$1.three.json" 2>>"$OUT/$1.err"
}
run base0 none x
for spec in "$@"; do
  label=${spec%%=*}; val=${spec#*=}
  if [[ "$val" == ENV:* ]]; then run "$label" env "${val#ENV:}"; else run "$label" lib "$val"; fi
done
run base1 none x
python - "$OUT" base0 "${@%%=*}" base1 <<'P'
import json, sys
out = sys.argv[1]
for lab in sys.argv[2:]:
    row = [lab]
    for kind in ("single", "three"):
        try:
            d = json.loads(open("%s/%s.%s.json" % (out, lab, kind)).read().strip().splitlines()[-1])
            k = d["kernel_ms_per_step"]
            row.append("%s %.2f p/s %.2f ms" % (kind, d["value"], d["ms_per_step"]))
            row.append({n: round(v, 2) for n, v in k.items() if any(s in n for s in ("ntt", "lde", "bitrev", "quotient"))})
        except Exception as e:
            row.append("%s ERR %s" % (kind, e))
    print(*row)
P
