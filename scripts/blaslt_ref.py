"""What rocBLAS / hipBLASLt (through torch.matmul) reach on the plain GEMMs behind the hot kernels - a reference
point for the hand-written loops (the library kernels have no fused tails and no [x|h] two-segment contraction)."""
import torch

dev = "cuda:0"
shapes = [(3840, 4096, 2176, "L1 fwd step (NT)"), (5120, 4096, 2176, "L1 fwd step, all rows"), (3712, 1024, 4096, "L1 BPTT step (NT)"),
          (4096, 1152, 56640, "L1 dW (TN)"), (4096, 1024, 56640, "L1 dW h-part (TN)"), (8192, 8192, 8192, "square")]
for M, N, K, what in shapes:
    tn = "TN" in what
    A = (torch.randn((K, M) if tn else (M, K), device=dev) * 0.1).to(torch.bfloat16)
    B = (torch.randn((K, N) if tn else (N, K), device=dev) * 0.1).to(torch.bfloat16)
    f = (lambda: A.t() @ B) if tn else (lambda: A @ B.t())
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-24s M=%d N=%d K=%d: %.1f us  %.0f TF/s" % (what, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
