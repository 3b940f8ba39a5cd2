"""Which pairs of HIP streams actually run concurrently?  Two chains of small GEMMs (64 workgroups each,
far from filling the chip) on two streams: wall time ~1x the single-chain time if the hardware queues
overlap, ~2x if one queue starves the other."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from efficientvideoclassification_youtube8m_amd import ops  # noqa: E402

dev = "cuda:0"
N = 120
mk = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
A = [mk(1024, 8192) for _ in range(4)]
B = [mk(1024, 8192) for _ in range(4)]
C = [torch.empty(1024, 1024, device=dev) for _ in range(4)]


def chain(i):
    for _ in range(N):
        ops.gemm_nt(A[i], B[i], 1024, 1024, 8192, C[i])


def timed(streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            chain(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


default = torch.cuda.current_stream()
pool = [torch.cuda.Stream(dev) for _ in range(8)]
hp = [torch.cuda.Stream(dev, priority=-1) for _ in range(2)]
timed([default])
print("single chain: %.2f ms" % timed([default]))
print("default + pool[i]:", ["%.2f" % timed([default, s]) for s in pool])
print("pool[0] + pool[i]:", ["%.2f" % timed([pool[0], s]) for s in pool[1:]])
print("pool[1] + pool[i]:", ["%.2f" % timed([pool[1], s]) for s in pool[2:]])
print("default + hp[i]:", ["%.2f" % timed([default, s]) for s in hp])
print("3 chains default+pool0+pool1: %.2f   4 chains: %.2f" % (timed([default, pool[0], pool[1]]), timed([default, pool[0], pool[1], pool[2]])))
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
