timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_configs.py -x -q 2>&1 | tail -2
for i in 1 2; do
echo "== step $(timeout 300 python bench.py --no_secondary --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])")"
done
echo "== cfg5 $(timeout 300 python bench.py --mode student --every_n 30 --batch 1024 --no_cpu_baseline --no_secondary --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")"
