#!/bin/bash
# Generic same-box A/B of bench.py under environment switches: each argument is "label:VAR=val,VAR2=val2" (or "label:" for the
# default); every variant runs ROUNDS times, interleaved.   gpurun -- bash scripts/ab.sh out "base:" "x:EVC_X=1" ...
set -u
OUT=gpurun_out/$1.txt; shift
mkdir -p gpurun_out; : > $OUT
for round in $(seq 1 ${ROUNDS:-2}); do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    line=$(env $(echo $envs | tr ',' ' ') X_=1 timeout 300 python bench.py --no_secondary --no_cpu_baseline --steps ${STEPS:-20} --warmup 3 ${BENCH_ARGS:-} 2>/dev/null | tail -1)
    python - "$label" "$line" >> $OUT <<'PY'
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    r = json.loads(line)
    rl = r.get("rooflines", {})
    print("%-40s %7.3f ms/step  fwd %.1f us  bwd %.1f us  dx %.3f  wgrad %.3f" % (label, r["ms_per_step"], rl["fwd_step"]["avg_launch_ms"] * 1e3,
          rl["bwd_step"]["avg_launch_ms"] * 1e3, rl["dx_nt"]["frac"], rl["wgrad_tn"]["frac"]))
except Exception as e:
    print("%-40s FAILED %s %s" % (label, e, line[:200]))
PY
  done
done
cat $OUT
