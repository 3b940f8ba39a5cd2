"""Ceiling table for the GEMM-shaped hot kernels: what rocBLAS / hipBLASLt (through torch.matmul) reach on the plain products behind them,
what this library's own loops reach on the SAME bare products (evc_gemm_nt / evc_gemm_tn: no fused tails), and the time the product would
take at the clock-limited MFMA rate (1.66 PF/s: the forward step's loop with everything but its MFMAs ablated) and at the
dense bf16 peak (2.5 PF/s).  The library kernels have no fused tails and no [x | h] two-segment contraction; the fused kernels' own times are
in the bench line (`rooflines`) and in profiles/r0N_digest_*.txt.

    python scripts/blaslt_ref.py            (on the GPU box; round 5: profiles/r05_blaslt_ref.txt)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from efficientvideoclassification_youtube8m_amd import ops  # noqa: E402

dev = "cuda:0"
CLOCK_LIMITED_TFLOPS, PEAK_TFLOPS = 1660.0, 2500.0
# (M, N, K, form, what): the five ring-tile shapes of the headline step + a square for scale
shapes = [(3840, 4096, 2176, "NT", "L1 forward step: [x_t | h] . W^T at 3 840 live rows (lstm_fwd_step_kernel, 240-row ring tile)"),
          (3712, 1024, 4096, "NT", "L1 BPTT step: dz_{t+1} . Wh^T (lstm_bwd_step_kernel, 128 x 128 ring tile)"),
          (56640, 1024, 4096, "NT", "L1 dX of the upper layer, all steps (gemm_nt_kernel, 224-row ring tile, bf16 out)"),
          (4096, 2048, 56640, "TN", "L1 weight gradient, upper layer: dz^T . [h0 | h1] (gemm_tn_kernel, 256 x 256, split-K)"),
          (4096, 1152, 56640, "TN", "L1 weight gradient, layer 0 input part: dz^T . x"),
          (256, 14148, 4096, "NT", "MoE gates: state . Wg^T (batch rows, gemm_nt_kernel 256 x 64, split-K)"),
          (8192, 8192, 8192, "NT", "square (for scale)")]


def timeit(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best * 1e3


print("%-96s %9s %9s %9s %9s   %s" % ("product (bf16 operands, f32 accumulation)", "library", "ours", "1.66 PF", "2.5 PF", "ours / library, ours as a share of the peak"))
for M, N, K, form, what in shapes:
    tn = form == "TN"
    A = (torch.randn((K, M) if tn else (M, K), device=dev) * 0.1).to(torch.bfloat16)
    B = (torch.randn((K, N) if tn else (N, K), device=dev) * 0.1).to(torch.bfloat16)
    lib = timeit((lambda: A.t() @ B) if tn else (lambda: A @ B.t()))
    out = torch.empty((M, N), dtype=torch.float32 if (tn or M <= 256) else torch.bfloat16, device=dev)
    if tn:
        ours = timeit(lambda: ops.gemm_tn(A, B, M, N, K, out))
    else:
        ours = timeit(lambda: ops.gemm_nt(A, B, M, N, K, out))
    fl = 2.0 * M * N * K
    print("%-96s %7.1f us %7.1f us %7.1f us %7.1f us   %.2f x, %.3f   [%s %d x %d x %d]" % (
        what, lib, ours, fl / CLOCK_LIMITED_TFLOPS / 1e6, fl / PEAK_TFLOPS / 1e6, ours / lib, fl / ours / 1e6 / PEAK_TFLOPS, form, M, N, K), flush=True)
    del A, B, out
