"""Gradients of one step under the current EVC_DETERMINISTIC setting -> file; with two files: compare them per tensor.
    EVC_DETERMINISTIC=0 python scripts/det_check.py /tmp/g0.pt; EVC_DETERMINISTIC=1 python scripts/det_check.py /tmp/g1.pt; python scripts/det_check.py /tmp/g0.pt /tmp/g1.pt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    for k in a:
        d = (a[k].double() - b[k].double()).norm() / (a[k].double().norm() + 1e-30)
        print("%-70s rel L2 %.3e  |a| %.3e |b| %.3e" % (k, float(d), float(a[k].double().norm()), float(b[k].double().norm())))
    sys.exit(0)
from bench import synthetic_inputs  # noqa: E402
from efficientvideoclassification_youtube8m_amd.distill import DistillGraph  # noqa: E402
B = int(os.environ.get("B", "64"))
g = DistillGraph(B, every_n=10, device="cuda:0", seed=7)
x, n, y = synthetic_inputs(B, 300, 1152, 4716, 1234, "cuda:0", False)
g.step(x, y, n, apply=False, num_frames_host=n.cpu().numpy())
torch.cuda.synchronize()
out = {}
for tw in (g.teacher, g.student):
    for k in tw.names:
        out[tw.scope + "/" + k] = tw.store.g(k).clone().cpu()
out["losses"] = g.losses.clone().cpu()
torch.save(out, sys.argv[1])
print("saved", sys.argv[1], "deterministic =", os.environ.get("EVC_DETERMINISTIC"))
