#!/bin/bash
# Round 6, closing call: the default bench line of the final tree, the uint8-input lines, cfg 5's PMC summary again (with its clip + Adam kernel), the
# robustness study at the 128- and 16-step horizons, the whole GPU suite.
set -u
O=gpurun_out/r06k
mkdir -p $O
timeout 1200 python bench.py > $O/bench_final.json 2> $O/bench_final.err
timeout 300 python bench.py --input uint8 --no_cpu_baseline --no_secondary --steps 20 > $O/bench_uint8_bf16.json 2> /dev/null
timeout 300 python bench.py --input uint8 --precision high --no_cpu_baseline --no_secondary --steps 20 > $O/bench_uint8_high.json 2> /dev/null
C5="--mode student --every_n 30 --batch 1024"
bash scripts/pmc_collect.sh $O/pmc_cfg5 --steps 3 --warmup 2 --no_cpu_baseline --no_secondary $C5 > /dev/null 2>&1
python scripts/pmc_kernels.py $O/pmc_cfg5 $O/pmc_kernels_cfg5.json "gemm_tn 256x256 (cfg 5: weight gradients of the student's L1 / L2 levels)=gemm_tn_kernel<TileCfg2<256" \
  "clip_adam (cfg 5: plain clip + TF-Adam of the materialised MoE gradient at 1024 rows)=clip_adam_kernel" \
  "gemm_nt 256x256 ring 5 (cfg 5: MoE head forward products at 1024 rows)=gemm_nt_kernel<TileCfg2<256, 1, 256, 2, 4, 5, true>" \
  "gemm_nt 320x256 (cfg 5: materialised MoE gradient product, hoisted L2 projection)=gemm_nt_kernel<TileCfg2<320" \
  "lstm_adam_fused (cfg 5)=lstm_adam_fused_kernel" "grad_sqnorm (cfg 5)=grad_sqnorm_kernel" > /dev/null 2>&1
rm -rf $O/pmc_cfg5/*/
bash scripts/precision_robustness_long.sh 12 $O/precision_robustness_128.txt 128 "high:u8;high" > /dev/null 2>&1
bash scripts/precision_robustness_long.sh 12 $O/precision_robustness_16.txt 16 "high:u8;high" > /dev/null 2>&1
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
cat $O/precision_robustness_128.txt $O/precision_robustness_16.txt
head -c 600 $O/bench_final.json
