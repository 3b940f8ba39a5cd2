#!/bin/bash
# gpurun_out/r05/ (scripts/collect_r05.sh) -> profiles/r05_* (the names profiles/README.md indexes).
set -u
S=gpurun_out/r05; D=profiles
cp $S/bench_final.json $D/r05_bench_final.json
cp $S/bench_dbof.json $D/r05_bench_dbof.json
cp $S/bench_dbof_high.json $D/r05_bench_dbof_high.json
for n in default high dbof; do cp $S/prof_$n/run_kernel_stats.csv $D/r05_bench$( [ $n = dbof ] && echo _dbof )_kernel_stats$( [ $n = dbof ] || echo _$n ).csv; done
cp $S/prof_solo/run_kernel_stats.csv $D/r05_bench_kernel_stats_single_stream.csv
cp $S/prof_high_solo/run_kernel_stats.csv $D/r05_bench_kernel_stats_high_single_stream.csv
for n in default solo high high_solo dbof; do cp $S/window_stats_$n.csv $D/r05_window_stats_$n.csv; cp $S/digest_$n.txt $D/r05_digest_$n.txt; done
for n in pmc_traffic.json pmc_kernels.json pmc_kernels_high.json pmc_kernels_dbof.json pmc_kernels_dbof_one_tile_per_workgroup.json blaslt_ref.txt fwd_high_bench.txt l2_bwd_bench.txt rccl_one_rank.txt dbof_lds_conflicts.txt; do
  [ -s $S/$n ] && cp $S/$n $D/r05_$n
done
ls -la $D/r05_bench_final.json
