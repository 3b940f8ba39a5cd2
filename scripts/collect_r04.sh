#!/bin/bash
# Round-4 measurement artefacts on the GPU box (gpurun -- bash scripts/collect_r04.sh): bench lines, rocprofv3 kernel stats of the
# same commands (rocprofv3's own --stats file + the same statistics restricted to steady-state steps), PMC passes, per-stream
# timeline, one-rank RCCL runs -> gpurun_out/r04/ (copied into profiles/r04_* afterwards, see profiles/README.md).
set -u
O=gpurun_out/r04
mkdir -p $O
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err
timeout 300 python bench.py --config dbof --no_cpu_baseline > $O/bench_dbof.json 2> /dev/null
prof() {   # name, bench args / env through the caller
  local name=$1; shift
  bash scripts/prof.sh r04/prof_$name --no_cpu_baseline --no_secondary "$@" > $O/digest_$name.txt 2>&1
  python scripts/window_stats.py $O/prof_$name/run_kernel_trace.csv $O/window_stats_$name.csv >> $O/digest_$name.txt 2>&1
}
prof default
EVC_SINGLE_STREAM=1 prof solo
prof high --precision high
EVC_SINGLE_STREAM=1 prof high_solo --precision high
bash scripts/prof.sh r04/prof_dbof --config dbof --no_cpu_baseline > $O/digest_dbof.txt 2>&1
timeout 200 python scripts/step_timeline.py > $O/timeline_default.txt 2>&1
timeout 200 python scripts/step_timeline.py --precision high > $O/timeline_high.txt 2>&1
bash scripts/pmc_collect.sh $O/pmc > /dev/null 2>&1
python scripts/pmc_summarize.py $O/pmc $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
python scripts/pmc_kernels.py $O/pmc $O/pmc_kernels.json > /dev/null 2>&1
bash scripts/pmc_collect.sh $O/pmc_high --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --precision high > /dev/null 2>&1
python scripts/pmc_kernels.py $O/pmc_high $O/pmc_kernels_high.json "lstm_fwd_step f16 + e4m3 stages (teacher L1, high mode)=false, true, true>(GemmOperands, LstmFwdParams" \
  "lstm_fwd_pair f16 + e4m3 stages (L2 wavefront, high mode)=lstm_fwd_pair_kernel<" \
  "gemm_nt f16 + e4m3 stages (MoE head, high mode)=gemm_nt_kernel<TileCfg3<256, 1, 64, 2, 4, 4>, true, true>" > /dev/null 2>&1
bash scripts/pmc_collect.sh $O/pmc_dbof --config dbof --steps 3 --warmup 2 --no_cpu_baseline > /dev/null 2>&1
python scripts/pmc_kernels.py $O/pmc_dbof $O/pmc_kernels_dbof.json > /dev/null 2>&1
STEPS=10 bash scripts/rccl_one_rank.sh $O/rccl_one_rank.txt > /dev/null 2>&1
for v in 0 1; do EVC_DETERMINISTIC=$v python bench.py --no_secondary --no_cpu_baseline --steps 20 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EVC_DETERMINISTIC=$v bf16 %.3f ms/step' % r['ms_per_step'], r['losses'])"; done > $O/deterministic_cost.txt 2>&1
# keep the merged output small: the raw traces and counter dumps stay on the box
rm -rf $O/pmc/*/ $O/pmc_dbof/*/ $O/pmc_high/*/ 2>/dev/null
find $O -name "run_kernel_trace.csv" -delete
ls -la $O
head -c 700 $O/bench_final.json
