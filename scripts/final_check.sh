#!/bin/bash
# Round-end check on the GPU box (run from the repo root: `gpurun -- bash scripts/final_check.sh`): the GPU test suite, the
# smoke entry, the default bench line, the DBoF bench line, kernel-stat and PMC profiles -> gpurun_out/ (copy what is to be
# kept into profiles/).
set -u
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/final_gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/final_gputest.log 2>&1
timeout 900 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
timeout 300 python bench.py --config dbof --no_cpu_baseline > gpurun_out/final_bench_dbof.json 2> /dev/null
if [ "${PROFILES:-1}" = "1" ]; then
  bash scripts/prof.sh final_prof_default --no_cpu_baseline --no_secondary > /dev/null 2>&1
  bash scripts/prof.sh final_prof_nooverlap --no_cpu_baseline --no_secondary --no_overlap > /dev/null 2>&1
  bash scripts/prof.sh final_prof_dbof --config dbof --no_cpu_baseline > /dev/null 2>&1
  bash scripts/pmc_collect.sh gpurun_out/final_pmc > /dev/null 2>&1
fi
cat gpurun_out/final_gputest.log
