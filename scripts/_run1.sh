for i in 1 2; do for v in base tn1 tn4 tn5; do
if [ $v = base ]; then L=""; else L="EVC_LIB=/root/repo/scripts/libevc_$v.so"; fi
echo "=== $v"
env $L timeout 300 python scripts/tn_bench.py 2>&1 | grep "^TN" | head -6 | cut -c1-60
done; done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -x -q 2>&1 | tail -2
echo "== step $(timeout 300 python bench.py --no_secondary --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])")"
