#!/bin/bash
# Round 6, seventh call: the h_lo forms (activations' low-order halves corrected) - kernel tests, step tests, the six deterministic 512-step draws, cost.
set -u
O=gpurun_out/r06h
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "fp8 or f16 or adam or dither" > $O/pytest_kernels.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_step.py -x -q -k "high or precision or dither or images or long_training" > $O/pytest_step.txt 2>&1
D=/tmp/evc_draws; mkdir -p $D
: > $O/long_draws.txt
for seed in 5 7 8 3 4 6; do
  rm -rf $D/s; mkdir -p $D/s
  EVC_LONG_SEED=$seed timeout 600 python scripts/long_horizon.py train $D/s 16 1e-3 512 > /dev/null 2>&1
  echo "== init seed $seed" >> $O/long_draws.txt
  timeout 600 python scripts/long_horizon.py eval $D/s "high;high:nodither" 2>&1 | grep "^steps\|^   " | cut -c1-260 >> $O/long_draws.txt
  EVC_HIGH_ACT_LO=0 timeout 600 python scripts/long_horizon.py eval $D/s "high" 2>&1 | grep "^   " | sed 's/high /high(act_lo=0) /' | cut -c1-260 >> $O/long_draws.txt
done
rm -rf $D
for i in 1 2; do
  timeout 300 python bench.py --precision high --no_cpu_baseline --no_secondary --steps 20 > $O/high_actlo_$i.json 2> /dev/null
  EVC_HIGH_ACT_LO=0 timeout 300 python bench.py --precision high --no_cpu_baseline --no_secondary --steps 20 > $O/high_noactlo_$i.json 2> /dev/null
  timeout 300 python bench.py --no_cpu_baseline --no_secondary --steps 20 > $O/bf16_$i.json 2> /dev/null
done
tail -4 $O/pytest_kernels.txt; tail -4 $O/pytest_step.txt
cat $O/long_draws.txt | cut -c1-230
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06h/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["ms_per_step_median"])
    except Exception as e:
        print(f, "failed", e)
PY
