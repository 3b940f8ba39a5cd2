#!/bin/bash
# What do collectives in flight cost the compute streams?  One GPU, one-rank RCCL communicator (EVC_DP_FORCE=1: every collective
# of the data-parallel step is issued, no byte moves) + EVC_DP_SIM: after each collective a stand-in kernel with an RCCL-like
# footprint (B workgroups x 256 threads x L KB of LDS) stays resident for the time the collective's bytes would spend on the wire
# of an 8-GPU node at the given bus bandwidth.  It holds CUs, not HBM bandwidth or links: a model of the
# SCHEDULE's sensitivity to collectives, not of the fabric.
#   bash scripts/dp_occupancy_sim.sh [out-file]
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${1:-gpurun_out/dp_occupancy_sim.txt}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
PORT=29700
run() {   # label, env...
  local label=$1; shift
  PORT=$((PORT + 1))
  env "$@" python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus 1 --steps ${STEPS:-10} --warmup 3 --no_cpu_baseline --no_secondary 2>/dev/null | python3 -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); dp = d.get('dp', {})
        print('%-64s %7.2f ms/step  wire %6.0f MB  collective events %6.2f ms  fwd step %5.1f us' % ('$label', d['ms_per_step'],
              dp.get('wire_mb_per_rank_per_step', 0), dp.get('collective_event_ms_per_step', 0), d['roofline']['avg_launch_ms'] * 1e3))" | tee -a "$OUT"
}
python3 bench.py --no_cpu_baseline --no_secondary --steps ${STEPS:-10} --warmup 3 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); print('%-64s %7.2f ms/step' % ('no process group (plain single-GPU step)', d['ms_per_step']))" | tee -a "$OUT"
run "one-rank RCCL, no stand-in" EVC_DP_FORCE=1
for bw in 300 150; do
  for blocks in ${BLOCKS:-8 16 32 64}; do
    run "stand-in busbw $bw GB/s, $blocks WGs x 32 KB LDS (default placement)" EVC_DP_FORCE=1 EVC_DP_SIM=$bw:$blocks:32
  done
done
run "stand-in busbw 300 GB/s, 32 WGs x 0 KB LDS" EVC_DP_FORCE=1 EVC_DP_SIM=300:32:0
run "stand-in busbw 300 GB/s, 32 WGs x 32 KB, ONE communicator, readiness issue order (round 4)" EVC_DP_FORCE=1 EVC_DP_SERIAL_COMM=1 EVC_DP_SIM=300:32:32
run "stand-in busbw 300 GB/s, 32 WGs x 32 KB, ONE communicator, sequential issue order (round 3 serial)" EVC_DP_FORCE=1 EVC_DP_SERIAL_COMM=1 EVC_ISSUE_ORDER=sequential EVC_DP_SIM=300:32:32
run "stand-in busbw 150 GB/s, 32 WGs x 32 KB, ONE communicator, readiness issue order" EVC_DP_FORCE=1 EVC_DP_SERIAL_COMM=1 EVC_DP_SIM=150:32:32
run "stand-in busbw 300 GB/s, 32 WGs x 32 KB, default placement, readiness issue order" EVC_DP_FORCE=1 EVC_ISSUE_ORDER=interleaved EVC_DP_SIM=300:32:32
run "stand-in busbw 300 GB/s, 32 WGs x 32 KB, bf16 gradients" EVC_DP_FORCE=1 EVC_DP_GRAD_DTYPE=bf16 EVC_DP_SIM=300:32:32
