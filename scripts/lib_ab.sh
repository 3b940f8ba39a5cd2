#!/bin/bash
# Same-box A/B of alternative builds of libevc_hip.so (selected through EVC_LIB): forward / BPTT step per layer and the training step.
#   gpurun -- bash scripts/lib_ab.sh <out> libevc_hip.so libevc_x.so ...
set -u
OUT=gpurun_out/$1.txt; shift
: > $OUT
P=$PWD/efficientvideoclassification_youtube8m_amd
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $round)" >> $OUT
    EVC_LIB=$P/$lib timeout 300 python scripts/lstm_layer_bench.py 2>/dev/null | grep "fwd" | head -2 >> $OUT
    EVC_LIB=$P/$lib timeout 300 python bench.py --no_secondary --no_cpu_baseline --steps 20 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); rl=r['rooflines']
print('bench %.3f ms/step  fwd %.4f  dx %.4f  wgrad %.4f bwd %.1f us' % (r['ms_per_step'], rl['fwd_step']['frac'], rl['dx_nt']['frac'], rl['wgrad_tn']['frac'], rl['bwd_step']['avg_launch_ms']*1e3))" >> $OUT
  done
done
cat $OUT
