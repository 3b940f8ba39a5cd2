#!/bin/bash
# Round 6: the "high" mode's L1 level as two-tile launches (evc_lstm_level2_fwd_high): kernel test, step tests in the mode, same-box A/B against one launch per layer and step.
set -u
O=gpurun_out/r06r
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "level2_fwd" > $O/pytest_kernels.txt 2>&1
tail -3 $O/pytest_kernels.txt
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_configs.py -x -q -k "high or trained_magnitude or dither" > $O/pytest_step.txt 2>&1
tail -3 $O/pytest_step.txt
for i in 1 2 3; do
  for v in 1 0; do
    for inp in uint8 f32; do
      EVC_FWD_WALK2_HIGH=$v timeout 300 python bench.py --precision high --input $inp --no_cpu_baseline --no_secondary --steps 20 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('high $inp EVC_FWD_WALK2_HIGH=$v: %.3f ms/step (median %.3f) fwd_step %s' % (d['ms_per_step'], d['ms_per_step_median'], json.dumps(d.get('rooflines', {}).get('fwd_step', {}))[:160]))" >> $O/walk2_high_ab.txt
    done
  done
done
cat $O/walk2_high_ab.txt
