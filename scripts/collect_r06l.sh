#!/bin/bash
# Round 6: the 16-step transient - twelve deterministic draws (init seeds 3..14) in high / split / bf16, the worst one's budget; the fused-norm MoE gradient
# products (kernel test, cfg 5 tests, cfg 5 A/B).
set -u
O=gpurun_out/r06l
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "clip_norm_from_the_same_pass or gemm_nt" > $O/pytest_kernels.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_step.py -x -q -k "cfg5 or small_step or three_iterations or fused_moe" > $O/pytest_cfg.txt 2>&1
for i in 1 2 3; do
  for v in 1 0; do
    EVC_FUSED_GRAD_NORM=$v timeout 300 python bench.py --mode student --every_n 30 --batch 1024 --no_cpu_baseline --no_secondary --steps 20 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 EVC_FUSED_GRAD_NORM=$v: %.3f ms/step (median %.3f) losses %s' % (d['ms_per_step'], d['ms_per_step_median'], d['losses']))" >> $O/cfg5_fused_norm_ab.txt
  done
done
D=/tmp/evc_draws; mkdir -p $D
: > $O/draws16.txt
for seed in 3 4 5 6 7 8 9 10 11 12 13 14; do
  rm -rf $D/s$seed; mkdir -p $D/s$seed
  EVC_LONG_SEED=$seed timeout 600 python scripts/long_horizon.py train $D/s$seed 16 1e-3 16 > /dev/null 2>&1
  echo "== init seed $seed" >> $O/draws16.txt
  timeout 600 python scripts/long_horizon.py eval $D/s$seed "high:u8;split;bf16" 2>&1 | grep "^steps\|^   " | cut -c1-260 >> $O/draws16.txt
done
worst_seed=$(python3 - $O/draws16.txt <<'PY'
import re, sys
cur, err = None, {}
for l in open(sys.argv[1]):
    if l.startswith("== init seed"):
        cur = l.split()[-1]
    m = re.match(r'\s+high:u8\s+.*t_gate (\S+) t_expert (\S+)', l)
    if m and cur:
        err[cur] = max(float(m.group(1)), float(m.group(2)))
print(max(err, key=err.get) if err else 3)
PY
)
echo "worst deterministic 16-step draw: init seed $worst_seed" | tee -a $O/draws16.txt
ONLY="R6 shipped|R6 D |R6 E |all f16|moe only f16|MOE fine: f16 + fp8|L1 f16, rest x3|L2 f16, rest x3|L1c0 f16: x only|L1c0 f16: h only|L1c0 f16: Wx only|L1c0 f16: Wh only|I1 only|I2 only|L2 fine f16:|FS all LSTM"
timeout 1500 python scripts/precision_budget.py --load_sd $D/s$worst_seed/step16.pt --only "$ONLY" > $O/budget_worst16.txt 2>&1
rm -rf $D
tail -3 $O/pytest_kernels.txt $O/pytest_cfg.txt
cat $O/cfg5_fused_norm_ab.txt
cat $O/draws16.txt | cut -c1-230
cat $O/budget_worst16.txt | cut -c1-210
