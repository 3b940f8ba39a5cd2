#!/bin/bash
# Round 6, first contact: (1) the long-horizon precision experiment (scripts/long_horizon.py), (2) rocprofv3 digests / window statistics of
# BASELINE cfg 5 (student only, every_n = 30, B = 1024) and cfg 2 (teacher only, B = 256), multi-stream and single-stream.
set -u
O=gpurun_out/r06
mkdir -p $O
timeout 600 python scripts/long_horizon.py train $O/long 16 1e-3 16,128,512 > $O/long_train.txt 2>&1
timeout 900 python scripts/long_horizon.py eval $O/long bf16,high > $O/long_eval.txt 2>&1
prof() {   # name, bench args / env through the caller
  local name=$1; shift
  bash scripts/prof.sh r06/prof_$name --no_cpu_baseline --no_secondary "$@" > $O/digest_$name.txt 2>&1
  python scripts/window_stats.py $O/prof_$name/run_kernel_trace.csv $O/window_stats_$name.csv >> $O/digest_$name.txt 2>&1
  grep -h '^{' $O/prof_$name/stdout.log > $O/bench_$name.json
}
prof cfg5 --mode student --every_n 30 --batch 1024
EVC_SINGLE_STREAM=1 prof cfg5_solo --mode student --every_n 30 --batch 1024
prof cfg2 --mode teacher
EVC_SINGLE_STREAM=1 prof cfg2_solo --mode teacher
find $O -name "run_kernel_trace.csv" -delete
rm -f $O/long/*.pt
cat $O/long_train.txt $O/long_eval.txt
head -30 $O/digest_cfg5.txt
