#!/bin/bash
# HBM traffic, MFMA busy and wave-state counters of a bench.py command, one rocprofv3 --pmc pass per counter group, each
# with --kernel-trace only (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots").  Run on the GPU box from the repo root:
#   bash scripts/pmc_collect.sh gpurun_out/pmc_r2 [bench.py args...]
#   python scripts/pmc_summarize.py gpurun_out/pmc_r2 profiles/r02_pmc_traffic.json      (forward step, bench.py reads it)
#   python scripts/pmc_kernels.py gpurun_out/pmc_r2 profiles/r02_pmc_kernels.json         (every hot kernel)
set -u
OUT=${1:-gpurun_out/pmc}
shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
ARGS="${*:---steps 2 --warmup 1 --no_cpu_baseline --no_secondary}"
echo "python3 bench.py $ARGS" > "$OUT/command.txt"
pass() {   # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o run -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass waves SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
ls "$OUT"/*/ | head -20
