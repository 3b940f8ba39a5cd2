#!/bin/bash
# Round 6, second call: new kernel tests, long-horizon layouts (one deterministic training run, the checkpoints evaluated under four "high" layouts),
# cfg 5 A/B of the L2-level hoist threshold.
set -u
O=gpurun_out/r06b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "dynamic_range or f16_fp8 or l2norm or fp8" > $O/pytest_kernels.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dp.py -x -q -k "cfg5 or multi_rank_path or student" > $O/pytest_cfg.txt 2>&1
timeout 600 python scripts/long_horizon.py train $O/long 16 1e-3 16,128,512 > $O/long_train.txt 2>&1
timeout 600 python scripts/long_horizon.py eval $O/long bf16,high > $O/long_eval_default.txt 2>&1
EVC_HIGH_DYNAMIC_RANGE=0 timeout 600 python scripts/long_horizon.py eval $O/long high > $O/long_eval_fixed_range.txt 2>&1
EVC_HIGH_STUDENT_LIGHT=0 timeout 600 python scripts/long_horizon.py eval $O/long high > $O/long_eval_student_full.txt 2>&1
EVC_HIGH_DITHER_LAYERS= timeout 600 python scripts/long_horizon.py eval $O/long high > $O/long_eval_no_dither.txt 2>&1
EVC_HIGH_DITHER_LAYERS= EVC_HIGH_STUDENT_LIGHT=0 timeout 600 python scripts/long_horizon.py eval $O/long high > $O/long_eval_no_dither_student_full.txt 2>&1
rm -f $O/long/*.pt
for i in 1 2; do
  timeout 300 python bench.py --mode student --every_n 30 --batch 1024 --no_cpu_baseline --no_secondary --steps 20 > $O/cfg5_default_$i.json 2> /dev/null
  EVC_HOIST_BELOW=1025 timeout 300 python bench.py --mode student --every_n 30 --batch 1024 --no_cpu_baseline --no_secondary --steps 20 > $O/cfg5_hoist_$i.json 2> /dev/null
done
tail -3 $O/pytest_kernels.txt $O/pytest_cfg.txt
grep -h "high\|steps" $O/long_eval_*.txt | cut -c1-260
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06b/cfg5_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["ms_per_step_median"])
    except Exception as e:
        print(f, "failed", e)
PY
