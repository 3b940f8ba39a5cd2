"""NT GEMM timings at the MoE-head / hoist / dx shapes.  EVC_FORCE_TILE pins the old tile choice for comparison."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from efficientvideoclassification_youtube8m_amd import ops  # noqa: E402

dev = "cuda:0"
shapes = [(256, 14148, 4096, "moe fwd gates"), (256, 9432, 4096, "moe fwd experts"), (256, 4096, 14208, "moe dx gates"),
          (256, 4096, 9472, "moe dx experts"), (14148, 4096, 256, "moe dW gates"), (9432, 4096, 256, "moe dW experts"),
          (5120, 4096, 4096, "L2 hoist / dx"), (56640, 1024, 4096, "L1 dx"), (1280, 4096, 4096, "student L2 hoist")]
for M, N, K, what in shapes:
    A = (torch.randn(M, K, device=dev) * 0.1).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev) * 0.1).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev)
    for _ in range(2):
        ops.gemm_nt(A, B, M, N, K, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm_nt(A, B, M, N, K, C)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gb = (M * K * 2 + N * K * 2 + M * N * 4) / 1e9
    print("NT %-18s M=%d N=%d K=%d: %.1f us  %.0f TF/s  %.2f TB/s (compulsory bytes)" % (what, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, gb / ms))
