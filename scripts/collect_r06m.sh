#!/bin/bash
# Round 6: the fused-norm gradient products with per-wave partials (kernel test, cfg 5 tests + A/B); 16 non-deterministic draws at the 16-step transient
# with the split-bf16 mode next to "high" (a draw on which split is off too is an ill-conditioned recurrence, not a layout deficiency).
set -u
O=gpurun_out/r06m
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "clip_norm_from_the_same_pass" > $O/pytest_kernels.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_step.py -x -q -k "cfg5 or small_step or three_iterations or deterministic_mode" > $O/pytest_cfg.txt 2>&1
for i in 1 2 3; do
  for v in 1 0; do
    EVC_FUSED_GRAD_NORM=$v timeout 300 python bench.py --mode student --every_n 30 --batch 1024 --no_cpu_baseline --no_secondary --steps 20 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 EVC_FUSED_GRAD_NORM=$v: %.3f ms/step (median %.3f)' % (d['ms_per_step'], d['ms_per_step_median']))" >> $O/cfg5_fused_norm_ab.txt
  done
done
bash scripts/precision_robustness_long.sh 16 $O/precision_robustness_16_split.txt 16 "high:u8;split" > /dev/null 2>&1
tail -3 $O/pytest_kernels.txt; tail -3 $O/pytest_cfg.txt
cat $O/cfg5_fused_norm_ab.txt
cat $O/precision_robustness_16_split.txt
grep "^steps\|^   " $O/precision_robustness_16_split.txt.raw | cut -c1-200 | head -60
