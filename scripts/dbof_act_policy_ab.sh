#!/bin/bash
# cfg 4's cluster kernel: cache policy of the bf16 tape store (268 MB per launch) - plain / sc1 write-through / nt (shipped since round 6): the kernel alone
# and the cfg-4 training step.  Variants 0 and 1: EVC_OUT=build_ab/libevc_dbof_act_policy_$v.so EVC_OBJ_DIR=build_ab/obj_act$v csrc/build.sh -DEVC_DBOF_ACT_POLICY=$v
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/dbof_act_policy_ab.txt}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for round in 1 2 3; do
  for v in 0 1 2; do
    lib=$PWD/efficientvideoclassification_youtube8m_amd/libevc_hip.so; [ $v != 2 ] && lib=$PWD/build_ab/libevc_dbof_act_policy_$v.so
    k=$(EVC_LIB=$lib timeout 200 python3 scripts/dbof_bench.py fwd 2>/dev/null | head -1 | cut -c1-70)
    s=$(EVC_LIB=$lib timeout 300 python3 bench.py --config dbof --no_cpu_baseline --steps 30 --warmup 6 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step %.3f ms (median %.3f), cluster kernel in the step %.1f us' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'] * 1e3))")
    echo "policy $v (round $round): $k | $s" >> "$OUT"
  done
done
cat "$OUT"
