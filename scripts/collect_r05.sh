#!/bin/bash
# Round-5 measurement artefacts on the GPU box (gpurun -- bash scripts/collect_r05.sh): bench lines, rocprofv3 kernel stats of the same
# commands (+ the same statistics over steady-state steps), PMC passes, the library reference points, the DBoF kernel's LDS-conflict split
# -> gpurun_out/r05/ (copied into profiles/r05_* afterwards, see profiles/README.md).
set -u
O=gpurun_out/r05
mkdir -p $O
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err
timeout 300 python bench.py --config dbof --no_cpu_baseline > $O/bench_dbof.json 2> /dev/null
timeout 300 python bench.py --config dbof --no_cpu_baseline --precision high > $O/bench_dbof_high.json 2> /dev/null
prof() {   # name, bench args / env through the caller
  local name=$1; shift
  bash scripts/prof.sh r05/prof_$name --no_cpu_baseline --no_secondary "$@" > $O/digest_$name.txt 2>&1
  python scripts/window_stats.py $O/prof_$name/run_kernel_trace.csv $O/window_stats_$name.csv >> $O/digest_$name.txt 2>&1
}
prof default
EVC_SINGLE_STREAM=1 prof solo
prof high --precision high
EVC_SINGLE_STREAM=1 prof high_solo --precision high
bash scripts/prof.sh r05/prof_dbof --config dbof --no_cpu_baseline > $O/digest_dbof.txt 2>&1
python scripts/window_stats.py $O/prof_dbof/run_kernel_trace.csv $O/window_stats_dbof.csv >> $O/digest_dbof.txt 2>&1
bash scripts/pmc_collect.sh $O/pmc > /dev/null 2>&1
python scripts/pmc_summarize.py $O/pmc $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
python scripts/pmc_kernels.py $O/pmc $O/pmc_kernels.json > /dev/null 2>&1
bash scripts/pmc_collect.sh $O/pmc_high --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --precision high > /dev/null 2>&1
python scripts/pmc_kernels.py $O/pmc_high $O/pmc_kernels_high.json "lstm_fwd_step f16 + e4m3 stages (teacher L1, high mode)=false, true, true>(GemmOperands, LstmFwdParams" \
  "lstm_fwd_pair f16 + e4m3 stages (L2 wavefront, high mode)=lstm_fwd_pair_kernel<" \
  "gemm_nt f16 + e4m3 stages (MoE head, high mode)=gemm_nt_kernel<TileCfg3<256, 1, 64, 2, 4, 4>, true, true>" \
  "moe_update pass 2 with the f16 + e4m3 images (high mode)=moe_update_kernel<TileCfg2<128, 1, 128, 2, 4, 5, true>, 2>" > /dev/null 2>&1
bash scripts/pmc_collect.sh $O/pmc_dbof --config dbof --steps 3 --warmup 2 --no_cpu_baseline > /dev/null 2>&1
python scripts/pmc_kernels.py $O/pmc_dbof $O/pmc_kernels_dbof.json > /dev/null 2>&1
EVC_DBOF_WALK=0 bash scripts/pmc_collect.sh $O/pmc_dbof_nowalk --config dbof --steps 3 --warmup 2 --no_cpu_baseline > /dev/null 2>&1
python scripts/pmc_kernels.py $O/pmc_dbof_nowalk $O/pmc_kernels_dbof_one_tile_per_workgroup.json > /dev/null 2>&1
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_dbof_lds -o run -- python3 scripts/dbof_bench.py fwd > $O/dbof_bench_fwd.txt 2>&1 )
python scripts/dbof_lds_conflicts.py $O/pmc_dbof_lds > $O/dbof_lds_conflicts.txt 2>&1
timeout 300 python scripts/blaslt_ref.py > $O/blaslt_ref.txt 2>&1
timeout 200 python scripts/fwd_high_bench.py > $O/fwd_high_bench.txt 2>&1
timeout 200 python scripts/l2_bwd_bench.py > $O/l2_bwd_bench.txt 2>&1
STEPS=10 bash scripts/rccl_one_rank.sh $O/rccl_one_rank.txt > /dev/null 2>&1
# keep the merged output small: the raw traces and counter dumps stay on the box
rm -rf $O/pmc/*/ $O/pmc_dbof/*/ $O/pmc_dbof_nowalk/*/ $O/pmc_high/*/ $O/pmc_dbof_lds 2>/dev/null
find $O -name "run_kernel_trace.csv" -delete
ls -la $O
head -c 700 $O/bench_final.json
