#!/bin/bash
# Schedule A/B on one box: the headline step under the schedule switches of DESIGN.md 5, each variant its own process,
# baseline first and last.   gpurun -- bash scripts/sched_ab.sh [out]   -> gpurun_out/<out>.txt
set -u
OUT=gpurun_out/${1:-sched_ab}.txt
mkdir -p gpurun_out
: > $OUT
run() {   # label, env...
  local label=$1; shift
  local line
  line=$(env "$@" timeout 300 python bench.py --no_secondary --no_cpu_baseline --steps 20 --warmup 3 ${BENCH_ARGS:-} 2>/dev/null | tail -1)
  python - "$label" "$line" >> $OUT <<'PY'
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    r = json.loads(line)
    rl = r.get("rooflines", {})
    print("%-44s %7.3f ms/step  fwd %.1f us  bwd %.1f us  dx %.3f  wgrad %.3f" % (label, r["ms_per_step"], rl["fwd_step"]["avg_launch_ms"] * 1e3,
          rl["bwd_step"]["avg_launch_ms"] * 1e3, rl["dx_nt"]["frac"], rl["wgrad_tn"]["frac"]))
except Exception as e:
    print("%-44s FAILED %s %s" % (label, e, line[:200]))
PY
}
run "baseline" X=1
run "student_early" EVC_STUDENT_EARLY=1
run "defer" EVC_DEFER_UPDATES=1
run "defer+student_early" EVC_DEFER_UPDATES=1 EVC_STUDENT_EARLY=1
run "opt on 16 CUs" EVC_OPT_CU_MASK=2
run "opt on 32 CUs" EVC_OPT_CU_MASK=4
run "opt on 64 CUs" EVC_OPT_CU_MASK=8
run "opt on 64 CUs + defer" EVC_OPT_CU_MASK=8 EVC_DEFER_UPDATES=1
run "opt on 256 CUs (own stream, no mask effect)" EVC_OPT_CU_MASK=32
run "baseline again" X=1
cat $OUT
