#!/bin/bash
# Spread of the "high" forward mode over weight draws at the LONG horizon: N times the real-size towers are trained for 512
# iterations in the default (non-deterministic: atomics) mode - every run ends on different weights - and the checkpoint is evaluated against the
# float64 oracle on the 4 evaluation videos (tests/_long_train.py::evaluate): teacher AND student columns, the shipped layout at B = 4 and inside a
# batch of 256, and the layout alternatives (no dithered layer; the student's L1 level on plain f16 as up to round 5).
#   bash scripts/precision_robustness_long.sh [draws=12] [out-file] [steps=512] [modes]
set -u
cd "$(dirname "$0")/.."
N=${1:-12}
OUT=${2:-gpurun_out/precision_robustness_long.txt}
STEPS=${3:-512}
MODES=${4:-"high;high@256;high:nodither;high:light"}
mkdir -p "$(dirname "$OUT")"
D=$(mktemp -d /tmp/evc_long.XXXXXX)
: > "$OUT.raw"
for i in $(seq 1 $N); do
  rm -f $D/*.pt
  EVC_DETERMINISTIC=0 timeout 600 python3 scripts/long_horizon.py train $D 16 1e-3 $STEPS >> "$OUT.raw" 2>&1
  timeout 600 python3 scripts/long_horizon.py eval $D "$MODES" 2>&1 | grep "^steps\|^   high\|^   split" | cut -c1-260 >> "$OUT.raw"
  echo "---" >> "$OUT.raw"
done
rm -rf $D
python3 - "$OUT.raw" "$N" "$STEPS" > "$OUT" <<'PY'
import collections, re, sys
d, order, mags = collections.defaultdict(lambda: {"t": [], "s": []}), [], []
for l in open(sys.argv[1]):
    m = re.match(r'\s+((?:high|split)\S*)\s+t_pred \S+ t_state (\S+) t_gate (\S+) t_expert (\S+) s_pred \S+ s_state (\S+) s_gate (\S+) s_expert (\S+)(.*)', l)
    if m:
        k = m.group(1)
        if k not in d:
            order.append(k)
        d[k]["t"].append(max(float(m.group(3)), float(m.group(4))))
        d[k]["s"].append(max(float(m.group(6)), float(m.group(7))))
        if "sat" in m.group(8) and "sat {} {}" not in m.group(8):
            d[k].setdefault("sat", []).append(m.group(8).strip())
    m = re.match(r'steps\s+(\d+)\s+\|z\| teacher (\S+) student (\S+)\s+\|state\| teacher (\S+) student (\S+)', l)
    if m:
        mags.append(tuple(float(m.group(i)) for i in range(2, 6)))
print("%s weight draws x %s training iterations (B = 16, lr 1e-3, non-deterministic mode); logit error (max over gates / experts) x 1e-4 vs float64, one column per draw" % (sys.argv[2], sys.argv[3]))
if mags:
    print("|z| teacher %.1f..%.1f student %.1f..%.1f; |state| teacher %.1f..%.1f student %.1f..%.1f" % (
        min(m[0] for m in mags), max(m[0] for m in mags), min(m[1] for m in mags), max(m[1] for m in mags),
        min(m[2] for m in mags), max(m[2] for m in mags), min(m[3] for m in mags), max(m[3] for m in mags)))
for k in order:
    for col, name in (("t", "teacher"), ("s", "student")):
        v = d[k][col]
        print("%-16s %-8s %s | mean %.2f max %.2f" % (k, name, " ".join("%4.1f" % (x * 1e4) for x in v), sum(v) / len(v) * 1e4, max(v) * 1e4))
    if d[k].get("sat"):
        print("%-16s saturated operands reported: %s" % (k, d[k]["sat"][:3]))
PY
cat "$OUT"
