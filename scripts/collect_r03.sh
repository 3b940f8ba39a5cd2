#!/bin/bash
# Round-3 measurement artefacts on the GPU box (gpurun -- bash scripts/collect_r03.sh): bench lines, rocprofv3 kernel stats of
# the same commands, PMC passes -> gpurun_out/r03/ (copied into profiles/r03_* afterwards, see profiles/README.md).
set -u
O=gpurun_out/r03
mkdir -p $O
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err
timeout 300 python bench.py --config dbof --no_cpu_baseline > $O/bench_dbof.json 2> /dev/null
bash scripts/prof.sh r03/prof_default --no_cpu_baseline --no_secondary > $O/digest_default.txt 2>&1
bash scripts/prof.sh r03/prof_no_overlap --no_cpu_baseline --no_secondary --no_overlap > $O/digest_no_overlap.txt 2>&1
bash scripts/prof.sh r03/prof_high_no_overlap --no_cpu_baseline --no_secondary --no_overlap --precision high > $O/digest_high_no_overlap.txt 2>&1
bash scripts/prof.sh r03/prof_high --no_cpu_baseline --no_secondary --precision high > $O/digest_high.txt 2>&1
bash scripts/prof.sh r03/prof_dbof --config dbof --no_cpu_baseline > $O/digest_dbof.txt 2>&1
bash scripts/pmc_collect.sh $O/pmc > /dev/null 2>&1
python scripts/pmc_summarize.py $O/pmc $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
python scripts/pmc_kernels.py $O/pmc $O/pmc_kernels.json > /dev/null 2>&1
bash scripts/pmc_collect.sh $O/pmc_high --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --precision high > /dev/null 2>&1
python scripts/pmc_kernels.py $O/pmc_high $O/pmc_kernels_high.json "lstm_fwd_step f16 + e4m3 stages (teacher L1, high mode)=false, true, true>(GemmOperands, LstmFwdParams" \
  "lstm_fwd_pair f16 + e4m3 stages (L2 wavefront, high mode)=lstm_fwd_pair_kernel<" \
  "gemm_nt f16 + e4m3 stages (MoE head, high mode)=gemm_nt_kernel<TileCfg3<256, 1, 64, 2, 4, 4>, true, true>" > /dev/null 2>&1
bash scripts/pmc_collect.sh $O/pmc_dbof --config dbof --steps 3 --warmup 2 --no_cpu_baseline > /dev/null 2>&1
python scripts/pmc_kernels.py $O/pmc_dbof $O/pmc_kernels_dbof.json > /dev/null 2>&1
bash scripts/rccl_one_rank.sh $O/rccl_one_rank.txt > /dev/null 2>&1
# keep the merged output small: the raw traces and counter dumps stay on the box
rm -rf $O/pmc/*/ $O/pmc_dbof/*/ $O/pmc_high/*/ 2>/dev/null
find $O -name "run_kernel_trace.csv" -delete
ls -la $O
head -c 600 $O/bench_final.json
