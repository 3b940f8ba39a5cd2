"""Per-kernel totals of ONE training step from a rocprofv3 --kernel-trace CSV of bench.py."""
import csv
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from trace_timeline import short  # noqa: E402

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
starts = [s for s, e, n, g in rows if n.startswith("l2norm_chunk")]       # first kernel of an H-LSTM training step
if len(starts) < 3:
    starts = [s for s, e, n, g in rows if n.startswith("dbof_gather")]      # ... of a DBoF step (bench.py --config dbof)
if len(starts) < 3:
    sys.exit("profile_digest: fewer than three training steps in the trace (no l2norm_chunk / dbof_gather launches)")
t0, t1 = starts[-3], starts[-2]
agg = defaultdict(lambda: [0, 0])
for s, e, n, g in rows:
    if t0 <= s < t1:
        agg[n][0] += 1
        agg[n][1] += e - s
tot = sum(v[1] for v in agg.values())
print("step wall %.2f ms; kernel-time sum %.2f ms" % ((t1 - t0) / 1e6, tot / 1e6))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 26]:
    print("%-58s x%3d %7.3f ms  avg %6.1f us" % (n, c, d / 1e6, d / c / 1e3))
