#!/bin/bash
# The data-parallel step of ONE rank of an 8-GPU node, on one GPU (round 6): one-rank RCCL communicator (EVC_DP_FORCE=1: every collective of the
# step is issued) + EVC_DP_SIM_WORLD=8 (this process does rank 0's per-rank work: MoE row slabs of 1/8 of the rows, factor gathers replicated to
# 8 x batch rows; distill.GradReducer) + optionally EVC_DP_SIM=<busbw>:<blocks>:<LDS KB>:8 (a stand-in kernel holds CUs for the time the
# collective's bytes would spend on the wire).  Splits the zero-byte overhead of the one-rank run (profiles/r05_rccl_one_rank.txt: +0.7 ms)
# into what only a ONE-rank world pays (all MoE rows updated through the two-phase slab path, whole matrices cloned for the slab gather) and
# what the schedule costs a rank of 8.  BASELINE cfg 3 (headline) and cfg 5 (student only, B = 1024: both MoE gradient routes).
#   bash scripts/dp_sim_world.sh [out-file]
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${1:-gpurun_out/dp_sim_world.txt}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
PORT=29720
show='
import json, sys
label = sys.argv[1]
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); dp = d.get("dp", {})
        kinds = " ".join("%s %.0f" % (k.replace("all_", "").replace("_grad", ""), v["wire_mb_per_rank_per_step"]) for k, v in sorted(dp.get("collectives", {}).items()) if v["wire_mb_per_rank_per_step"] >= 1)
        print("%-86s %7.2f ms/step (median %.2f)  wire %6.0f MB [%s]  fwd step %5.1f us" % (label, d["ms_per_step"], d["ms_per_step_median"],
              dp.get("wire_mb_per_rank_per_step", 0), kinds, d["roofline"]["avg_launch_ms"] * 1e3))'
plain() {   # label, bench args...
  local label=$1; shift
  python3 bench.py --no_cpu_baseline --no_secondary --steps ${STEPS:-12} --warmup 3 "$@" 2>/dev/null | python3 -c "$show" "$label" | tee -a "$OUT"
}
run() {   # label, "ENV=..." assignments ..., -- , bench args...
  local label=$1; shift
  local envs=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  PORT=$((PORT + 1))
  env EVC_DP_FORCE=1 "${envs[@]}" python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus 1 --steps ${STEPS:-12} --warmup 3 --no_cpu_baseline --no_secondary "$@" 2>/dev/null | python3 -c "$show" "$label" | tee -a "$OUT"
}
echo "== BASELINE cfg 3 (teacher + student, B = 256 per rank)" | tee -a "$OUT"
plain "no process group (plain single-GPU step)"
run "one-rank RCCL (a world of ONE: all MoE rows through the slab path, whole-matrix slab clones)"
run "one-rank RCCL as rank 0 of 8 (EVC_DP_SIM_WORLD=8), zero time on the wire" EVC_DP_SIM_WORLD=8
run "  + stand-in 300 GB/s busbw, 32 WGs x 32 KB" EVC_DP_SIM_WORLD=8 EVC_DP_SIM=300:32:32:8
run "  + stand-in 150 GB/s busbw, 32 WGs x 32 KB" EVC_DP_SIM_WORLD=8 EVC_DP_SIM=150:32:32:8
run "  + stand-in 300 GB/s, bf16 LSTM gradient payload" EVC_DP_SIM_WORLD=8 EVC_DP_SIM=300:32:32:8 EVC_DP_GRAD_DTYPE=bf16
echo "== BASELINE cfg 5 (student only, every_n = 30, B = 1024 per rank)" | tee -a "$OUT"
C5="--mode student --every_n 30 --batch 1024"
plain "no process group (plain single-GPU step)" $C5
run "rank 0 of 8, MoE gradient by bf16 reduce-scatter (the route its shape picks), zero wire time" EVC_DP_SIM_WORLD=8 -- $C5
run "  + stand-in 300 GB/s" EVC_DP_SIM_WORLD=8 EVC_DP_SIM=300:32:32:8 -- $C5
run "  + stand-in 300 GB/s, bf16 LSTM gradient payload" EVC_DP_SIM_WORLD=8 EVC_DP_SIM=300:32:32:8 EVC_DP_GRAD_DTYPE=bf16 -- $C5
run "rank 0 of 8, MoE gradient by factor all-gather (round 5), zero wire time" EVC_DP_SIM_WORLD=8 EVC_DP_MOE_ROUTE=factors -- $C5
run "  + stand-in 300 GB/s" EVC_DP_SIM_WORLD=8 EVC_DP_SIM=300:32:32:8 EVC_DP_MOE_ROUTE=factors -- $C5
