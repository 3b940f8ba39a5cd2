"""Timeline digest of a rocprofv3 --kernel-trace CSV of bench.py: per-stream busy time, the idle
gaps of the stream carrying the teacher chain, and the kernels running when it is idle.

    python scripts/trace_timeline.py <kernel_trace.csv> [--last N]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+)(<.*)?\(", name)
    base = m.group(1) if m else name[:40]
    cfg = re.search(r"TileCfg[23]?<([0-9, a-z]+)>", name)
    return base.split("::")[-1] + ("<" + cfg.group(1).replace(" ", "") + ">" if cfg else "")


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Stream_Id"]), short(r["Kernel_Name"])))
    rows.sort()
    # one training step = from one l2norm_chunk_kernel to the next
    starts = [s for s, e, q, n in rows if n.startswith("l2norm_chunk_kernel")]
    if len(starts) < 3:
        print("need >= 3 steps in the trace")
        return
    t0, t1 = starts[-2], starts[-1]
    step = [(s, e, q, n) for s, e, q, n in rows if t0 <= s < t1]
    print("step wall: %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
    per_q = defaultdict(list)
    for s, e, q, n in step:
        per_q[q].append((s, e, n))
    for q, ks in sorted(per_q.items()):
        busy = sum(e - s for s, e, _ in ks)
        print("stream %d: %4d kernels, busy %.3f ms, first %.3f last %.3f" % (q, len(ks), busy / 1e6, (ks[0][0] - t0) / 1e6, (ks[-1][1] - t0) / 1e6))
        agg = defaultdict(lambda: [0, 0])
        for s, e, n in ks:
            agg[n][0] += 1
            agg[n][1] += e - s
        for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
            print("      %-60s x%3d  %.3f ms (avg %.1f us)" % (n, c, d / 1e6, d / c / 1e3))
    # union busy time (any stream) and concurrency histogram
    ev = []
    for s, e, q, n in step:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    conc, last, hist = 0, t0, defaultdict(int)
    for t, d in ev:
        hist[conc] += t - last
        last = t
        conc += d
    print("concurrency histogram (ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
    # gaps of the main stream (the one with the most lstm_fwd_step launches)
    mainq = max(per_q, key=lambda q: sum(1 for _, _, n in per_q[q] if n.startswith("lstm_fwd_step")))
    ks = per_q[mainq]
    gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2], ks[i + 1][2]) for i in range(len(ks) - 1)]
    tot = sum(g for g, _, _ in gaps if g > 0)
    print("main stream %d: sum of gaps %.3f ms over %d boundaries; gaps > 20us:" % (mainq, tot / 1e6, len(gaps)))
    for g, a, b in sorted(gaps, reverse=True)[:15]:
        print("      %.1f us between %s -> %s" % (g / 1e3, a, b))
    small = [g for g, _, _ in gaps if 0 < g <= 20000]
    print("      %d small gaps, avg %.1f us" % (len(small), sum(small) / max(1, len(small)) / 1e3))


if __name__ == "__main__":
    main()
