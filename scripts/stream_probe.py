"""Step time with the measured-concurrent stream set (streams.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import synthetic_inputs  # noqa: E402
from efficientvideoclassification_youtube8m_amd import streams  # noqa: E402
from efficientvideoclassification_youtube8m_amd.distill import DistillGraph  # noqa: E402

dev = "cuda:0"
streams.concurrent_streams(dev, 4, verbose=True)
g = DistillGraph(256, every_n=10, device=dev)
batches = [synthetic_inputs(256, 300, 1152, 4716, 100 + i, dev, False) for i in range(4)]


def run(label, K=10):
    for i in range(3):
        x, n, y = batches[i % 4]
        g.step(x, y, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        x, n, y = batches[i % 4]
        g.step(x, y, n)
    torch.cuda.synchronize()
    print("%s: %.2f ms/step" % (label, (time.perf_counter() - t0) / K * 1e3), flush=True)


run("measured stream set")
run("measured stream set (again)")
