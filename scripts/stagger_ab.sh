#!/bin/bash
# Same-box A/B of the role-staggered loop schedules (-DEVC_STAGGER_LEAD / -DEVC_TN_STAGGER builds of the library, selected through EVC_LIB):
# forward / BPTT step per layer, TN weight-gradient shapes, the training step.   gpurun -- bash scripts/stagger_ab.sh
set -u
OUT=gpurun_out/${1:-stagger_ab}.txt
: > $OUT
P=$PWD/efficientvideoclassification_youtube8m_amd
for round in 1 2; do
  for lib in libevc_hip.so libevc_stag4.so libevc_stag8.so libevc_stag12.so; do
    echo "== $lib (round $round)" >> $OUT
    EVC_LIB=$P/$lib timeout 300 python scripts/lstm_layer_bench.py 2>/dev/null | grep -i "us\|step" | head -6 >> $OUT
    EVC_LIB=$P/$lib timeout 300 python scripts/tn_bench.py 2>/dev/null | grep "K=56640\|K=5120" | head -6 >> $OUT
    EVC_LIB=$P/$lib timeout 300 python bench.py --no_secondary --no_cpu_baseline --steps 20 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); rl=r['rooflines']
print('bench %.3f ms/step  fwd %.4f  dx %.4f  wgrad %.4f' % (r['ms_per_step'], rl['fwd_step']['frac'], rl['dx_nt']['frac'], rl['wgrad_tn']['frac']))" >> $OUT
  done
done
cat $OUT
