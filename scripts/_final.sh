timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r2_gputest_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r2_gputest_final.log 2>&1
timeout 900 python bench.py > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err
bash scripts/prof.sh prof_r2f_default --no_cpu_baseline --no_secondary > /dev/null 2>&1
bash scripts/prof.sh prof_r2f_nooverlap --no_cpu_baseline --no_secondary --no_overlap > /dev/null 2>&1
bash scripts/pmc_collect.sh gpurun_out/pmc_r2f > /dev/null 2>&1
