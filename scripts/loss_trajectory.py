"""Debug helper: per-step losses of the teacher+student iteration on the GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from efficientvideoclassification_youtube8m_amd.distill import DistillGraph

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fresh = len(sys.argv) > 2 and sys.argv[2] == "fresh"
g = DistillGraph(B, every_n=10, device="cuda:0", seed=7)
x, n, labels = bench.synthetic_inputs(B, 300, 1152, 4716, 1234, "cuda:0", False)
for step in range(12):
    if fresh:
        x, n, labels = bench.synthetic_inputs(B, 300, 1152, 4716, 1234 + step, "cuda:0", False)
    out = g.step(x, labels, n)
    rep = g.loss_report()
    ts = out["teacher_state"]
    print(step, {k: round(v, 4) for k, v in rep.items()}, "|state| max %.3f rms %.4f  pmin %.3e" % (
        ts.abs().max().item(), ts.pow(2).mean().sqrt().item(), out["student_predictions"].min().item()))
