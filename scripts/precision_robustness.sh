#!/bin/bash
# Spread of the "high" precision layouts over the weights the precision test can draw (its GPU training sums with atomics: every
# run ends on different weights).  N draws; per draw the emulated teacher-logit error (= the kernels' error, to the digit) of each
# candidate layout (scripts/precision_budget.py "ROBUST" configurations); summary = mean / max per layout.  DESIGN.md 7.
#   [ONLY="ROBUST FZ"] bash scripts/precision_robustness.sh [draws] [out-file]      (ONLY: substring filter of the layouts emulated)
set -u
cd "$(dirname "$0")/.."
N=${1:-8}
OUT=${2:-gpurun_out/precision_robustness.txt}
mkdir -p "$(dirname "$OUT")"
: > "$OUT.raw"
for i in $(seq 1 $N); do
  python3 scripts/precision_budget.py --gpu --only "${ONLY:-ROBUST}" 2>&1 | grep "ROBUST\|KERNELS high" | cut -c1-170 >> "$OUT.raw"
  echo "---" >> "$OUT.raw"
done
python3 - "$OUT.raw" > "$OUT" <<'PY'
import collections, re, sys
d, order = collections.defaultdict(list), []
for l in open(sys.argv[1]):
    m = re.match(r'(ROBUST \S+|KERNELS high)\s+(.*?)t state (\S+) gate (\S+) expert (\S+)', l)
    if m:
        key = m.group(1) + " " + m.group(2).strip()
        if key not in d:
            order.append(key)
        d[key].append(max(float(m.group(4)), float(m.group(5))))
print("teacher logit error (max over gates / experts) x 1e-4, one column per weight draw; KERNELS high = the shipped layout on the real kernels")
for k in order:
    v = d[k]
    print("%-78s %s | mean %.2f max %.2f" % (k[:78], " ".join("%4.1f" % (x * 1e4) for x in v), sum(v) / len(v) * 1e4, max(v) * 1e4))
PY
cat "$OUT"
