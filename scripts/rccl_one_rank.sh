#!/bin/bash
# The data-parallel step on ONE GPU with a one-rank RCCL communicator (EVC_DP_FORCE=1): every collective of the
# step (per-group gradient all-reduces on the side streams, the MoE factor all-gathers, the student's own
# communicator, the loss all-reduce) goes through RCCL; the result must match the plain single-GPU run.
# Runs both placements of the collectives (distill.GradReducer): in stream order on two communicators (default) and
# EVC_DP_SERIAL_COMM=1 (one communicator, one collective at a time), then the plain single-GPU step.
#   bash scripts/rccl_one_rank.sh [out-file]      (default gpurun_out/rccl_one_rank.txt; copy to profiles/ to keep)
set -e
set -o pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
export EVC_DETERMINISTIC=1      # no floating-point atomics: the three runs differ only in the code path of the MoE clip norm (Gram matrices / row-slab phases)
OUT=${1:-gpurun_out/rccl_one_rank.txt}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
run() {   # label, env assignments..., then the rest
  local label=$1; shift
  echo "== $label" | tee -a "$OUT"
  env "$@" | python3 -c "
import json, sys
seen = 0
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        seen += 1
        print('ms_per_step %.3f  frames/s %.0f  losses %s' % (d['ms_per_step'], d['value'], d['losses']))
        print('LOSSES ' + json.dumps(d['losses'], sort_keys=True))
sys.exit(0 if seen == 1 else 3)" | tee -a "$OUT"      # a crashed run prints no JSON line: fail here (pipefail), not silently
}
ARGS="bench.py --gpus 1 --steps ${STEPS:-10} --warmup 3 --no_cpu_baseline --no_secondary"
run "one-rank RCCL, collectives in stream order, student on its own communicator (default)" EVC_DP_FORCE=1 \
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 $ARGS
run "one-rank RCCL, EVC_DP_SERIAL_COMM=1 (one communicator, one collective at a time)" EVC_DP_FORCE=1 EVC_DP_SERIAL_COMM=1 \
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 $ARGS
run "no process group (plain single-GPU step)" python3 $ARGS
# the three runs train on the same synthetic batches from the same initialisation, under EVC_DETERMINISTIC=1 (round 3: default mode, whose
# atomics moved the trajectory by a few percent after ~20 Adam iterations on noise - compared at 15 %): the two RCCL runs take the same
# code path and must report IDENTICAL losses; the plain run computes the MoE clip norm another way (Gram matrices instead of the
# row-slab passes: last-bit differences in a scale Adam nearly cancels) and must agree to TOL (default 5e-3; measured 2e-3 after 13 iterations)
python3 - "$OUT" <<'PY' | tee "$OUT.verdict"
import json, sys
rows = [json.loads(l[len("LOSSES "):]) for l in open(sys.argv[1]) if l.startswith("LOSSES ")]
import os
tol = float(os.environ.get("TOL", "5e-3"))
assert len(rows) == 3, "expected three runs, saw %d" % len(rows)
exact = all(rows[0][k] == rows[1][k] for k in rows[0] if k != "pred_loss")
print("the two RCCL placements report %s losses" % ("IDENTICAL" if exact else "DIFFERENT"))
assert exact, (rows[0], rows[1])
worst = 0.0
for k, v in rows[2].items():
    for r in rows[:2]:
        worst = max(worst, abs(r[k] - v) / max(1.0, abs(v)))
        assert abs(r[k] - v) <= tol * max(1.0, abs(v)), ("losses differ from the plain single-GPU run", k, r[k], v)
print("losses of the three runs agree (worst relative difference to the plain run %.2e, tolerance %g)" % (worst, tol))
PY
cat "$OUT.verdict" >> "$OUT"; rm -f "$OUT.verdict"
