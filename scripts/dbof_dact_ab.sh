#!/bin/bash
# cfg 4's dact kernel (in-place backward of max-pool + relu6 + cluster BN over the 268 MB bf16 tape): rows loaded ahead of the first store and nt loads of the tape.
# Variants: EVC_OUT=build_ab/libevc_dact_<rows>_<nt>.so EVC_OBJ_DIR=... csrc/build.sh -DEVC_DBOF_DACT_ROWS=<rows> -DEVC_DBOF_DACT_NT=<0|1>; shipped = 32 rows (the whole video), nt.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/dbof_dact_ab.txt}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
timeout 600 python -m pytest tests/test_gpu_dbof_logistic.py tests/test_gpu_kernels.py -x -q -k "dbof" 2>&1 | tail -1 >> "$OUT"
for round in 1 2; do
  for v in shipped 1_0 8_1 16_1; do
    lib=$PWD/build_ab/libevc_dact_$v.so; [ $v = shipped ] && lib=$PWD/efficientvideoclassification_youtube8m_amd/libevc_hip.so
    k=$(EVC_LIB=$lib timeout 200 python3 scripts/dbof_bench.py dact 2>/dev/null | head -1 | cut -c1-70)
    s=$(EVC_LIB=$lib timeout 300 python3 bench.py --config dbof --no_cpu_baseline --steps 30 --warmup 6 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step %.3f ms (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']))")
    echo "rows_nt $v (round $round): $k | $s" >> "$OUT"
  done
done
cat "$OUT"
