"""BPTT of one M ~ batch two-layer stack (the L2 level: T = 20 steps, M = 256 rows, H = 1024, layer 0 input 4096) alone on the chip:
layer by layer (2 T skinny launches + the hoisted dX product of layer 1) against the wavefront pair launches (evc_lstm_stack2_bwd at
M <= 512: T + 1 launches, layer 0's steps contract layer 1's gradient in their own K walk).  us per chain; random tape (timing only).

    python scripts/l2_bwd_bench.py [--T 20] [--M 256]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from efficientvideoclassification_youtube8m_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=20)
ap.add_argument("--M", type=int, default=256)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = "cuda:0"
T, M, H, K0 = a.T, a.M, 1024, 4096
lens = torch.full((M,), T, dtype=torch.int32, device=dev)
w0 = (torch.randn(K0 + H, 4 * H, device=dev) * 0.02).to(torch.bfloat16)
w1 = (torch.randn(2 * H, 4 * H, device=dev) * 0.02).to(torch.bfloat16)
gates = [torch.randint(0, 2 ** 15, (T, M, H, 2), dtype=torch.int32, device=dev) for _ in range(2)]
c_all = [(torch.randn(T + 1, M, H, device=dev) * 0.3).to(torch.bfloat16) for _ in range(2)]
dS = torch.randn(M, 4 * H, device=dev) * 1e-3
dc = [torch.zeros(M, H, device=dev) for _ in range(2)]
dz = [torch.zeros(T, M, 4 * H, dtype=torch.bfloat16, device=dev) for _ in range(2)]
db = [torch.zeros(4 * H, device=dev) for _ in range(2)]
dx1 = torch.empty(T * M, H, dtype=torch.bfloat16, device=dev)


def layer_by_layer():
    ops.lstm_layer_bwd(w1, lens, T, M, H, H, gates[1], c_all[1], dS[:, 2 * H:], dS[:, 3 * H:], 4 * H, None, dc[1], dz[1], db=db[1])
    ops.gemm_nt(dz[1].view(T * M, 4 * H), w1, T * M, H, 4 * H, dx1)
    ops.lstm_layer_bwd(w0, lens, T, M, K0, H, gates[0], c_all[0], dS[:, 0:], dS[:, H:], 4 * H, dx1.view(T, M, H), dc[0], dz[0], db=db[0])


def pair():
    ops.lstm_stack2_bwd(w0, w1, lens, T, M, K0, H, gates, c_all, dS, dc, dz, db)


for name, fn in (("layer by layer", layer_by_layer), ("pair launches", pair), ("layer by layer", layer_by_layer), ("pair launches", pair)):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    n = 2 * T + 1 if fn is layer_by_layer else T + 1
    print("%-16s %7.1f us per chain (%d launches, %.1f us each)" % (name, best * 1e3, n, best * 1e3 / n), flush=True)
