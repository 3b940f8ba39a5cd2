timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_configs.py tests/test_gpu_dbof_logistic.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
echo "== v3 $(timeout 300 python bench.py --no_secondary --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])")"
echo "== big-NT v2 $(EVC_NT_BIG_V2=1 timeout 300 python bench.py --no_secondary --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])")"
done
