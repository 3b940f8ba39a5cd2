#!/bin/bash
# Round 6: the BPTT step's row-major tail with its first 8 row slots' tape / state loads issued BEFORE the main loop (EVC_BWD_TAIL_PRE): tests, then the same-box A/B
# against the loads-in-the-tail build (build_ab/libevc_bwd_pre0.so = csrc/build.sh -DEVC_BWD_TAIL_PRE=0).
set -u
O=gpurun_out/r06u
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "bwd or backward or bptt" > $O/pytest_kernels.txt 2>&1
tail -3 $O/pytest_kernels.txt
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_configs.py -x -q -k "not long_training and not long_horizon" > $O/pytest_step.txt 2>&1
tail -3 $O/pytest_step.txt
bash scripts/lib_ab.sh r06u/bwd_tail_pre_ab libevc_hip.so ../build_ab/libevc_bwd_pre0.so > /dev/null 2>&1
cat $O/bwd_tail_pre_ab.txt
