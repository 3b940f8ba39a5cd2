#!/bin/bash
# HBM traffic + MFMA busy counters of the bench step, one rocprofv3 --pmc pass per counter group, each with
# --kernel-trace only (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots").  Run on the GPU box from the repo root:
#   bash scripts/pmc_collect.sh gpurun_out/pmc_r2      then      python scripts/pmc_summarize.py gpurun_out/pmc_r2
set -u
OUT=${1:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD="python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o run -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o run -- $CMD > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -o run -- $CMD > "$OUT/mfma.log" 2>&1
ls "$OUT"/*/
