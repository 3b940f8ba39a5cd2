"""Per-stream timeline of ONE training step from HIP events at the engine's mark() points (ops.MARKS): no profiler, so the
streams overlap as they do in a real run (rocprofv3 --kernel-trace serialises the queues).  Prints, per stream, each mark's time
since the step's first mark, and the critical path candidates (last mark per stream).

    python scripts/step_timeline.py [--precision high] [--steps 12]          (schedule switches: EVC_* environment)
"""
import argparse
import os
import sys
import time
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import synthetic_inputs  # noqa: E402
from efficientvideoclassification_youtube8m_amd import ops  # noqa: E402
from efficientvideoclassification_youtube8m_amd.distill import DistillGraph  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
dev = "cuda:0"
g = DistillGraph(args.batch, every_n=10, device=dev, precision=args.precision)
batches = [synthetic_inputs(args.batch, 300, 1152, 4716, 100 + i, dev, False) for i in range(4)]
nhost = [b[1].cpu().numpy() for b in batches]
for i in range(6):
    x, n, y = batches[i % 4]
    g.step(x, y, n, num_frames_host=nhost[i % 4])
g.flush()
torch.cuda.synchronize()
K = args.steps
t0 = time.perf_counter()
marks = None
for i in range(K):
    x, n, y = batches[i % 4]
    if i == K - 3:
        ops.MARKS = []
        g.debug_marks = []
    g.step(x, y, n, num_frames_host=nhost[i % 4])
    if i == K - 2:                      # two steps of marks: the second one shows the deferred work of the first
        marks, ops.MARKS = ops.MARKS, None
        gm, g.debug_marks = g.debug_marks, None
g.flush()
torch.cuda.synchronize()
print("%.3f ms/step (%s; EVC_DEFER_UPDATES=%s EVC_STUDENT_EARLY=%s EVC_OPT_CU_MASK=%s)" % (
    (time.perf_counter() - t0) / K * 1e3, args.precision, os.environ.get("EVC_DEFER_UPDATES"), os.environ.get("EVC_STUDENT_EARLY"),
    os.environ.get("EVC_OPT_CU_MASK")))
base = gm[0][1]
names = {g._main.cuda_stream: "main", g._side.cuda_stream: "side", g._aux_t.cuda_stream: "aux_t", g._aux_s.cuda_stream: "aux_s"}
if g._opt_t is not None:
    names[g._opt_t.cuda_stream] = "opt_t"
    names[g._opt_s.cuda_stream] = "opt_s" if g._opt_s is not g._opt_t else "opt_t"
per = OrderedDict()
for name, st, ev in marks:
    per.setdefault(names.get(st, hex(st)), []).append((base.elapsed_time(ev), name))
for name, ev in gm:
    per.setdefault("graph", []).append((base.elapsed_time(ev), name))
for sname, ms in per.items():
    print("[%s]" % sname)
    for t, n in ms:
        print("   %8.3f  %s" % (t, n))
