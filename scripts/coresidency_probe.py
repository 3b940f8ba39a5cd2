"""Does a streaming kernel (clip+Adam over 58 M parameters) hide under a big GEMM on another stream?
EVC_FORCE_TILE=2 runs the GEMM on v1 128x128 tiles (138 registers/lane, 64 KB LDS: room left on every CU),
default = v2 256x256 tiles (241 registers x 2 waves/SIMD, 160 KB LDS: nothing else fits on the CU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from efficientvideoclassification_youtube8m_amd import ops, streams  # noqa: E402

dev = "cuda:0"
s1, s2 = streams.concurrent_streams(dev, 4)[:2]
M = N = 8192
K = 8192
A = (torch.randn(M, K, device=dev) * 0.1).to(torch.bfloat16)
B = (torch.randn(N, K, device=dev) * 0.1).to(torch.bfloat16)
C = torch.empty(M, N, device=dev)
n = 58_000_000
p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
v.abs_()
sums = torch.ones(2, device=dev)
pb = torch.empty(n, dtype=torch.bfloat16, device=dev)


def gemm(reps):
    for _ in range(reps):
        ops.gemm_nt(A, B, M, N, K, C)


def adam(reps):
    for _ in range(reps):
        ops.clip_adam_step(p, g, m, v, 0.0, sums, 1.0, 1e-3, p_bf16=pb)


def timed(f1, f2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if f1:
        with torch.cuda.stream(s1):
            f1()
    if f2:
        with torch.cuda.stream(s2):
            f2()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


gemm(2); adam(2)
tg = min(timed(lambda: gemm(6), None) for _ in range(3))
ta = min(timed(None, lambda: adam(20)) for _ in range(3))
tb = min(timed(lambda: gemm(6), lambda: adam(20)) for _ in range(3))
print("tile=%s: gemm x6 alone %.2f ms (%.0f TF/s) | adam x20 alone %.2f ms (%.2f TB/s) | both %.2f ms (sum %.2f)"
      % (os.environ.get("EVC_FORCE_TILE", "v2"), tg, 6 * 2.0 * M * N * K / tg / 1e9, ta, 20 * n * 30 / ta / 1e9, tb, tg + ta))
