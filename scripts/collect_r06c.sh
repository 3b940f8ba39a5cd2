#!/bin/bash
# Round 6, third call: new tests (long horizon, exchange routes), the cost of the student's full "high" layout, cfg 5 with the fused MoE update at
# 1024 rows, the 12-draw long-horizon robustness study, the deterministic draw with every layout + "split" for the ladder.
set -u
O=gpurun_out/r06c
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_step.py -x -q -k "long_training" -s > $O/pytest_long.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_dp.py -x -q > $O/pytest_dp.txt 2>&1
for i in 1 2; do
  timeout 300 python bench.py --precision high --no_cpu_baseline --no_secondary --steps 20 > $O/high_light_$i.json 2> /dev/null
  EVC_HIGH_STUDENT_LIGHT=0 timeout 300 python bench.py --precision high --no_cpu_baseline --no_secondary --steps 20 > $O/high_full_$i.json 2> /dev/null
  timeout 300 python bench.py --mode student --every_n 30 --batch 1024 --no_cpu_baseline --no_secondary --steps 20 > $O/cfg5_mat_$i.json 2> /dev/null
  EVC_MOE_FUSE_MAX_ROWS=1024 timeout 300 python bench.py --mode student --every_n 30 --batch 1024 --no_cpu_baseline --no_secondary --steps 20 > $O/cfg5_fused_$i.json 2> /dev/null
done
timeout 600 python scripts/long_horizon.py train $O/long 16 1e-3 16,128,512 > $O/long_train.txt 2>&1
timeout 900 python scripts/long_horizon.py eval $O/long "bf16;high;high:full;high:nodither,full;high:full@256;split" > $O/long_eval.txt 2>&1
rm -rf $O/long
bash scripts/precision_robustness_long.sh 12 $O/precision_robustness_long.txt 512 "high:full;high:full@256;high:nodither,full;high:light" > /dev/null 2>&1
tail -4 $O/pytest_long.txt; tail -4 $O/pytest_dp.txt
cat $O/long_eval.txt | cut -c1-250
cat $O/precision_robustness_long.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06c/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["ms_per_step_median"])
    except Exception as e:
        print(f, "failed", e)
PY
