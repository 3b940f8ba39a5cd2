#!/bin/bash
# Round 6, eighth call: the integer-frame layer 0 (uint8 input) - kernel tests, step tests, the six deterministic draws on uint8 frames, cost on uint8 input.
set -u
O=gpurun_out/r06i
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "integer or l2norm or fp8 or dither" > $O/pytest_kernels.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_step.py -x -q -k "high or precision or dither or images" > $O/pytest_step.txt 2>&1
D=/tmp/evc_draws; mkdir -p $D
: > $O/long_draws.txt
for seed in 5 7 8 3 4 6; do
  rm -rf $D/s; mkdir -p $D/s
  EVC_LONG_SEED=$seed timeout 600 python scripts/long_horizon.py train $D/s 16 1e-3 512 > /dev/null 2>&1
  echo "== init seed $seed" >> $O/long_draws.txt
  timeout 600 python scripts/long_horizon.py eval $D/s "high:u8;high;high:u8@256;high:u8,nodither" 2>&1 | grep "^steps\|^   " | cut -c1-260 >> $O/long_draws.txt
done
rm -rf $D
for i in 1 2; do
  timeout 300 python bench.py --input uint8 --precision high --no_cpu_baseline --no_secondary --steps 20 > $O/u8_high_int_$i.json 2> /dev/null
  EVC_HIGH_X_INT=0 timeout 300 python bench.py --input uint8 --precision high --no_cpu_baseline --no_secondary --steps 20 > $O/u8_high_noint_$i.json 2> /dev/null
  timeout 300 python bench.py --input uint8 --no_cpu_baseline --no_secondary --steps 20 > $O/u8_bf16_$i.json 2> /dev/null
  timeout 300 python bench.py --precision high --no_cpu_baseline --no_secondary --steps 20 > $O/f32_high_$i.json 2> /dev/null
done
tail -4 $O/pytest_kernels.txt; tail -4 $O/pytest_step.txt
cat $O/long_draws.txt | cut -c1-230
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06i/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["ms_per_step_median"], d["roofline"]["avg_launch_ms"])
    except Exception as e:
        print(f, "failed", e)
PY
