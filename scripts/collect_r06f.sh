#!/bin/bash
# Round 6, sixth call: the candidate layouts' budget again (emulator now saturates e4m3 like the kernels) on the three bad deterministic draws.
set -u
O=gpurun_out/r06g
mkdir -p $O
D=/tmp/evc_draws; mkdir -p $D
for seed in 5 7 8; do
  rm -rf $D/s; mkdir -p $D/s
  EVC_LONG_SEED=$seed timeout 600 python scripts/long_horizon.py train $D/s 16 1e-3 512 > /dev/null 2>&1
  echo "== init seed $seed" >> $O/budget_plans.txt
  timeout 1200 python scripts/precision_budget.py --load_sd $D/s/step512.pt --only "R6 " 2>&1 | grep -v amdgpu.ids | cut -c1-230 >> $O/budget_plans.txt
done
rm -rf $D
cat $O/budget_plans.txt
