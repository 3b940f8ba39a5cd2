"""Per-kernel statistics of the STEADY-STATE steps of a bench.py run from a rocprofv3 --kernel-trace CSV: the window from the
third-last to the last l2norm_chunk launch of the trace (whole training steps; no graph construction, stream probing, settle pass
or secondary configuration), in the column layout of rocprofv3's own --stats file.
    python scripts/window_stats.py <run_kernel_trace.csv> <out.csv> [steps]"""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
starts = [s for s, e, n in rows if "l2norm_chunk_kernel" in n]            # first kernel of an H-LSTM training step
if len(starts) < nsteps + 1:
    starts = [s for s, e, n in rows if "dbof_gather" in n]                # ... of a DBoF step (bench.py --config dbof)
if len(starts) < nsteps + 1:
    sys.exit("window_stats: fewer than %d training steps in the trace (no l2norm_chunk / dbof_gather launches)" % (nsteps + 1))
t0, t1 = starts[-1 - nsteps], starts[-1]
agg = defaultdict(list)
for s, e, n in rows:
    if t0 <= s < t1:
        agg[n].append(e - s)
tot = sum(sum(v) for v in agg.values())
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "Window"])
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([n, len(v), sum(v), "%.1f" % (sum(v) / len(v)), "%.2f" % (100.0 * sum(v) / tot), min(v), max(v), "%d steps, %.3f ms wall" % (nsteps, (t1 - t0) / 1e6)])
print("window: %d steps, %.3f ms per step wall, %.3f ms per step kernel time" % (nsteps, (t1 - t0) / 1e6 / nsteps, tot / 1e6 / nsteps))
