"""Is the training step launch-bound?  Host enqueue time per step vs GPU time per step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import synthetic_inputs  # noqa: E402
from efficientvideoclassification_youtube8m_amd.distill import DistillGraph  # noqa: E402

dev = "cuda:0"
g = DistillGraph(256, every_n=10, device=dev)
batches = [synthetic_inputs(256, 300, 1152, 4716, 100 + i, dev, False) for i in range(4)]
for i in range(4):
    x, n, y = batches[i % 4]
    g.step(x, y, n)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
host = []
for i in range(K):
    x, n, y = batches[i % 4]
    h0 = time.perf_counter()
    g.step(x, y, n)
    host.append(time.perf_counter() - h0)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue per step: %.2f ms (min %.2f max %.2f); wall per step incl. GPU: %.2f ms; host finished %.2f ms before the GPU"
      % (t_enq / K * 1e3, min(host) * 1e3, max(host) * 1e3, t_all / K * 1e3, (t_all - t_enq) * 1e3))
