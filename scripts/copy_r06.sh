#!/bin/bash
# gpurun_out/r06f/ (scripts/collect_r06.sh) -> profiles/r06_* (the names profiles/README.md indexes).
set -u
S=gpurun_out/r06f; D=profiles
for n in bench_final.json bench_dbof.json bench_uint8_bf16.json bench_uint8_high.json pmc_traffic.json pmc_kernels.json pmc_kernels_cfg5.json rccl_one_rank.txt; do
  [ -s $S/$n ] && cp $S/$n $D/r06_$n
done
for n in default solo high high_solo cfg5 cfg5_solo cfg2 cfg2_solo dbof; do
  [ -s $S/digest_$n.txt ] && cp $S/digest_$n.txt $D/r06_digest_$n.txt
  [ -s $S/window_stats_$n.csv ] && cp $S/window_stats_$n.csv $D/r06_window_stats_$n.csv
  [ -s $S/prof_$n/run_kernel_stats.csv ] && cp $S/prof_$n/run_kernel_stats.csv $D/r06_bench_kernel_stats_$n.csv
done
ls -la $D | grep r06 | wc -l
