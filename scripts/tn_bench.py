"""TN GEMM (weight-gradient shapes) timings."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from efficientvideoclassification_youtube8m_amd import ops  # noqa: E402

dev = "cuda:0"
shapes = [(4096, 1152, 56640), (4096, 1024, 56640), (4096, 4096, 5120), (4096, 1024, 5120), (4096, 1152, 6144), (4096, 1024, 6144),
          (4096, 4096, 1280), (4096, 1024, 1280), (4096, 1152, 76800), (4096, 2176, 56640)]
for M, N, K in shapes:
    A = (torch.randn(K, M, device=dev) * 0.1).to(torch.bfloat16)
    B = (torch.randn(K, N, device=dev) * 0.1).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev)
    Cref = torch.empty(M, N, device=dev)
    ops.gemm_tn(A, B, M, N, K, Cref)
    for _ in range(2):
        ops.gemm_tn(A, B, M, N, K, C, accumulate=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm_tn(A, B, M, N, K, C, accumulate=True)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    err = ((C / 7.0) - Cref).abs().max().item() / (Cref.abs().max().item() + 1e-9)
    print("TN M=%d N=%d K=%d: %.1f us  %.0f TF/s  (accumulate mode, rel err of the 7-fold sum %.1e)" % (M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, err))


# two-segment form (evc_gemm_tn2) and the narrow strip of engine._wgrad_tn
for M, N1, N2, K in [(4096, 1024, 1024, 56640), (4096, 1024, 1024, 5120)]:
    A = (torch.randn(K, M, device=dev) * 0.1).to(torch.bfloat16)
    B1 = (torch.randn(K, N1, device=dev) * 0.1).to(torch.bfloat16)
    B2 = (torch.randn(K, N2, device=dev) * 0.1).to(torch.bfloat16)
    C = torch.zeros(M, N1 + N2, device=dev)
    for _ in range(2):
        ops.gemm_tn2(A, B1, N1, B2, N2, M, K, C, accumulate=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm_tn2(A, B1, N1, B2, N2, M, K, C, accumulate=True)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("TN2 M=%d N=%d+%d K=%d: %.1f us  %.0f TF/s" % (M, N1, N2, K, ms * 1e3, 2.0 * M * (N1 + N2) * K / ms / 1e9))
for M, N, K in [(4096, 128, 56640)]:
    A = (torch.randn(K, M, device=dev) * 0.1).to(torch.bfloat16)
    B = (torch.randn(K, N, device=dev) * 0.1).to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev)
    for _ in range(2):
        ops.gemm_tn(A, B, M, N, K, C, accumulate=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm_tn(A, B, M, N, K, C, accumulate=True)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("TN strip M=%d N=%d K=%d: %.1f us  %.0f TF/s" % (M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
