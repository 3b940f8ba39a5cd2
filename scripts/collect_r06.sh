#!/bin/bash
# Round-6 measurement artefacts on the GPU box (gpurun -- bash scripts/collect_r06.sh): the default bench line, rocprofv3 kernel statistics / digests /
# steady-state window statistics of the headline (four streams and one stream), the "high" mode, BASELINE cfg 5 and cfg 2 (both stream modes), cfg 4;
# PMC passes for the headline's hot kernels and for cfg 5's three largest kernels; the one-rank RCCL run  -> gpurun_out/r06f/ (scripts/copy_r06.sh -> profiles/r06_*).
set -u
O=gpurun_out/r06f
mkdir -p $O
timeout 1200 python bench.py > $O/bench_final.json 2> $O/bench_final.err
timeout 300 python bench.py --config dbof --no_cpu_baseline > $O/bench_dbof.json 2> /dev/null
timeout 300 python bench.py --input uint8 --no_cpu_baseline --no_secondary --steps 20 > $O/bench_uint8_bf16.json 2> /dev/null
timeout 300 python bench.py --input uint8 --precision high --no_cpu_baseline --no_secondary --steps 20 > $O/bench_uint8_high.json 2> /dev/null
prof() {   # name, bench args / env through the caller
  local name=$1; shift
  bash scripts/prof.sh r06f/prof_$name --no_cpu_baseline --no_secondary "$@" > $O/digest_$name.txt 2>&1
  python scripts/window_stats.py $O/prof_$name/run_kernel_trace.csv $O/window_stats_$name.csv >> $O/digest_$name.txt 2>&1
}
prof default
EVC_SINGLE_STREAM=1 prof solo
prof high --precision high
EVC_SINGLE_STREAM=1 prof high_solo --precision high
C5="--mode student --every_n 30 --batch 1024"
prof cfg5 $C5
EVC_SINGLE_STREAM=1 prof cfg5_solo $C5
prof cfg2 --mode teacher
EVC_SINGLE_STREAM=1 prof cfg2_solo --mode teacher
bash scripts/prof.sh r06f/prof_dbof --config dbof --no_cpu_baseline > $O/digest_dbof.txt 2>&1
python scripts/window_stats.py $O/prof_dbof/run_kernel_trace.csv $O/window_stats_dbof.csv >> $O/digest_dbof.txt 2>&1
bash scripts/pmc_collect.sh $O/pmc > /dev/null 2>&1
python scripts/pmc_summarize.py $O/pmc $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
python scripts/pmc_kernels.py $O/pmc $O/pmc_kernels.json > /dev/null 2>&1
bash scripts/pmc_collect.sh $O/pmc_cfg5 --steps 3 --warmup 2 --no_cpu_baseline --no_secondary $C5 > /dev/null 2>&1
python scripts/pmc_kernels.py $O/pmc_cfg5 $O/pmc_kernels_cfg5.json "gemm_tn 256x256 (cfg 5: weight gradients of the student's L1 / L2 levels)=gemm_tn_kernel<TileCfg2<256" \
  "clip_adam (cfg 5: plain clip + TF-Adam of the materialised MoE gradient at 1024 rows)=clip_adam_kernel" \
  "gemm_nt 256x256 ring 5 (cfg 5: MoE head forward products at 1024 rows)=gemm_nt_kernel<TileCfg2<256, 1, 256, 2, 4, 5, true>" \
  "gemm_nt 320x256 (cfg 5: materialised MoE gradient product)=gemm_nt_kernel<TileCfg2<320" \
  "lstm_adam_fused (cfg 5)=lstm_adam_fused_kernel" "grad_sqnorm (cfg 5)=grad_sqnorm_kernel" > /dev/null 2>&1
STEPS=10 bash scripts/rccl_one_rank.sh $O/rccl_one_rank.txt > /dev/null 2>&1
rm -rf $O/pmc/*/ $O/pmc_cfg5/*/ 2>/dev/null
find $O -name "run_kernel_trace.csv" -delete
ls -la $O
head -c 900 $O/bench_final.json
