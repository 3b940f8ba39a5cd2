"""Micro-benchmark of the GEMM-shaped entry points on the shapes of the hot path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvideoclassification_youtube8m_amd import ops

dev = "cuda:0"
torch.manual_seed(0)


def bench(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


def gemm(M, N, K, name):
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    B = (torch.randn(N, K, device=dev) * 0.5).bfloat16()
    C = torch.empty(M, N, device=dev)
    ms = bench(lambda: ops.gemm_nt(A, B, M, N, K, C))
    ref = (A[:64].float() @ B.float().t())
    err = (C[:64] - ref).abs().max().item() / ref.abs().max().item()
    print("%-28s M=%6d N=%6d K=%6d  %8.3f ms  %7.1f TFLOP/s  relerr %.1e" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, err))


def gemm_tn(M, N, K, name):
    A = (torch.randn(K, M, device=dev) * 0.5).bfloat16()
    B = (torch.randn(K, N, device=dev) * 0.5).bfloat16()
    C = torch.empty(M, N, device=dev)
    ms = bench(lambda: ops.gemm_tn(A, B, M, N, K, C))
    ref = (A[:, :64].float().t() @ B.float())
    err = (C[:64] - ref).abs().max().item() / ref.abs().max().item()
    print("%-28s M=%6d N=%6d K=%6d  %8.3f ms  %7.1f TFLOP/s  relerr %.1e" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, err))


def lstm(M, T, Kin, H, hoist, name):
    x = (torch.randn(T, M, Kin, device=dev) * 0.3).bfloat16()
    wT = (torch.randn(4 * H, Kin + H, device=dev) * 0.03).bfloat16()
    w = wT.t().contiguous()
    b = torch.zeros(4 * H, device=dev)
    ln = torch.full((M,), T, dtype=torch.int32, device=dev)
    hbuf = torch.empty(T + 1, M, H, dtype=torch.bfloat16, device=dev)
    S = torch.empty(M, 2 * H, device=dev)
    gates = torch.empty(T, M, H, 2, dtype=torch.int32, device=dev)
    call = torch.empty(T + 1, M, H, device=dev)
    zx = torch.empty(T * M, 4 * H, device=dev) if hoist else None
    f = lambda: ops.lstm_layer_fwd(x, wT, b, ln, T, M, Kin, H, hbuf, S[:, :H], S[:, H:], 2 * H, gates, call, hoist=hoist, zx_ws=zx)
    ms = bench(f, 5)
    fl = 2.0 * M * 4 * H * (Kin * T + H * (T - 1))
    print("%-28s M=%6d T=%3d Kin=%5d H=%5d  %8.3f ms/layer %7.1f us/step %7.1f TFLOP/s" % (name, M, T, Kin, H, ms, ms / T * 1e3, fl / ms / 1e9))
    dS = torch.randn(M, 2 * H, device=dev)
    dz = torch.empty(T, M, 4 * H, dtype=torch.bfloat16, device=dev)
    dcw = torch.empty(M, H, device=dev)
    g = lambda: ops.lstm_layer_bwd(w, ln, T, M, Kin, H, gates, call, dS[:, :H], dS[:, H:], 2 * H, None, dcw, dz)
    ms = bench(g, 5)
    fl = 2.0 * M * 4 * H * H * (T - 1)
    print("%-28s bwd                           %8.3f ms/layer %7.1f us/step %7.1f TFLOP/s" % ("", ms, ms / T * 1e3, fl / ms / 1e9))


if __name__ == "__main__":
    gemm(4096, 4096, 4096, "square 4096")
    gemm(8192, 8192, 8192, "square 8192")
    gemm(76800, 4096, 1152, "x-projection (hoisted)")
    gemm(4096, 2176, 76800, "dW L1 layer0")
    gemm(76800, 1024, 4096, "dX L1 layer1")
    gemm_tn(4096, 4096, 4096, "TN square 4096")
    gemm_tn(4096, 1152, 76800, "TN dW L1 (x part)")
    gemm_tn(4096, 1024, 76800, "TN dW L1 (h part)")
    gemm(256, 14148, 4096, "MoE gates fwd")
    gemm(14148, 4096, 256, "MoE dW gates")
    gemm(256, 4096, 14208, "MoE dx")
    lstm(5120, 15, 1152, 1024, False, "teacher L1 layer0 (fused)")
    lstm(5120, 15, 1024, 1024, False, "teacher L1 layer1 (fused)")
    lstm(1280, 6, 1152, 1024, False, "student L1 layer0")
    lstm(256, 20, 4096, 1024, True, "teacher L2 layer0 (hoist)")
    lstm(256, 20, 1024, 1024, True, "teacher L2 layer1 (hoist)")
