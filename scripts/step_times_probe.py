"""Per-step wall times of the headline step from process start (each step synchronised) next to the caching allocator's segment
counters: shows one-off stalls (the allocator releasing / re-mapping segments) and which step they land on.
    python scripts/step_times_probe.py [steps] [pool]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
dev = "cuda:0"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
npool = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pool = [bench.synthetic_inputs(256, 300, 1152, 4716, 1234 + 1000 * i, dev, False) for i in range(npool)]
nh = [p[1].cpu().numpy() for p in pool]
g = DistillGraph(256, every_n=10, device=dev, seed=7)
t_start = time.perf_counter()
rows = []
for i in range(steps):
    x, n, y = pool[i % npool]
    t0 = time.perf_counter()
    g.step(x, y, n, num_frames_host=nh[i % npool])
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    rows.append(((time.perf_counter() - t0) * 1e3, st["segment.all.allocated"], st["segment.all.freed"], st["num_alloc_retries"],
                 st["reserved_bytes.all.current"] >> 20))
print("step: ms  segments allocated / freed  alloc retries  reserved MiB")
for i, r in enumerate(rows):
    if i < 14 or r[0] > 1.3 * rows[-1][0]:
        print("%3d: %6.1f  %d / %d  %d  %d" % ((i,) + r))
print("elapsed %.2f s" % (time.perf_counter() - t_start))
