#!/bin/bash
# Round 6, fourth call: (1) six DETERMINISTIC draws of the 512-step towers (EVC_LONG_SEED = 3..8) evaluated in "high"; the worst one's error budget per
# product family (scripts/precision_budget.py --load_sd); (2) the data-parallel step as rank 0 of a simulated world of 8 (scripts/dp_sim_world.sh);
# (3) cfg 4's cluster kernel ablations (scripts/dbof_ablation.sh); (4) the two-rank DBoF reduce-scatter test.
set -u
O=gpurun_out/r06d
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dp.py -x -q -k "dbof" > $O/pytest_dbof_dp.txt 2>&1
D=/tmp/evc_draws; mkdir -p $D
: > $O/long_draws.txt
for seed in 3 4 5 6 7 8; do
  rm -rf $D/s$seed; mkdir -p $D/s$seed
  EVC_LONG_SEED=$seed timeout 600 python scripts/long_horizon.py train $D/s$seed 16 1e-3 512 > /dev/null 2>&1
  echo "== init seed $seed" >> $O/long_draws.txt
  timeout 600 python scripts/long_horizon.py eval $D/s$seed "high:full;high:nodither,full;split" 2>&1 | grep "^steps\|^   " | cut -c1-260 >> $O/long_draws.txt
done
worst_seed=$(python3 - $O/long_draws.txt <<'PY'
import re, sys
cur, err = None, {}
for l in open(sys.argv[1]):
    if l.startswith("== init seed"):
        cur = l.split()[-1]
    m = re.match(r'\s+high:full\s+.*t_gate (\S+) t_expert (\S+)', l)
    if m and cur:
        err[cur] = max(float(m.group(1)), float(m.group(2)))
print(max(err, key=err.get) if err else 3)
PY
)
echo "worst deterministic draw: init seed $worst_seed" | tee -a $O/long_draws.txt
ONLY="all f16|moe only f16|MOE fine: f16 + fp8|L1 f16, rest x3|L2 f16, rest x3|L1c0 f16: Wx only|L1c0 f16: x only|L1c0 f16: Wh only|L1c0 f16: h only|I1 only|I2 only|I3 only|I4 only|L2 fine f16:|W16+8 only|X16+8 only|FZ9|FZ8X|FS all LSTM|FY only"
timeout 1500 python scripts/precision_budget.py --load_sd $D/s$worst_seed/step512.pt --only "$ONLY" > $O/budget_worst.txt 2>&1
rm -rf $D
bash scripts/dp_sim_world.sh $O/dp_sim_world.txt > /dev/null 2>&1
bash scripts/dbof_ablation.sh $O/dbof_ablation.txt > /dev/null 2>&1
tail -3 $O/pytest_dbof_dp.txt
cat $O/long_draws.txt | cut -c1-230
cat $O/budget_worst.txt | cut -c1-200
cat $O/dp_sim_world.txt
cat $O/dbof_ablation.txt
