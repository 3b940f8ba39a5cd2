"""Times the forward of one L1 layer at the teacher shape (row plan of a synthetic batch, 15 steps) in the operand layouts of the
precision modes: bf16, "high" with f16 K-extended weights (round-3 FZ layout) and "high" with the weights' low-order halves in fp8
(evc_lstm_layer_fwd_f16_fp8lo).  us per step; EVC_LIB=<other build> for A/B runs of loop variants.

    python scripts/fwd_high_bench.py [--layer 0|1] [--batch 256]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from efficientvideoclassification_youtube8m_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=8)
a = ap.parse_args()
dev = "cuda:0"
B, H, F, C, T = a.batch, 1024, 1152, 20, 15
rng = np.random.default_rng(0)
n = rng.integers(120, 301, size=B)
_, lens, _ = ops.host_frame_counts(n, 1, C, T)
M = C * B
ld = torch.from_numpy(lens.astype(np.int32)).to(dev)
plan = ops.RowPlan(ld, lens, T)
P, rows = plan.P, plan.rows
S = torch.zeros((M, 2 * H), device=dev)
b = torch.zeros(4 * H, device=dev)
hbf = torch.zeros((T + 1, P, H), dtype=torch.bfloat16, device=dev)


def timeit(fn):
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
    return best * 1e3 / T


for layer in (0, 1):
    Kin = F if layer == 0 else H
    w = torch.randn(4 * H, Kin + H, device=dev) * 0.02
    flops = sum(2.0 * r * 4 * H * (Kin + (H if t > 0 else 0)) for t, r in enumerate(rows)) / T
    res = {}
    # bf16
    x = (torch.randn(T, P, Kin, device=dev) * 0.05).to(torch.bfloat16)
    res["bf16"] = timeit(lambda: ops.lstm_layer_fwd(x, w.to(torch.bfloat16), b, plan.lens, T, P, Kin, H, hbf, S[:, :H], S[:, H:], 2 * H, plan=plan))
    # f16, weights K-extended by f16 low-order halves (layer 0: x in 3 segments, h in 2; layer 1: [h0 | h0/64 | h1 | h1/64])
    if layer == 0:
        x16 = (torch.randn(T, P, 3 * Kin, device=dev) * 0.05).half()
        w16 = torch.empty((4 * H, 3 * Kin + 2 * H), dtype=torch.float16, device=dev)
        ops.cast_f16_wide(w, Kin, H, 3, w16, h_ext=True)
        kx, ldx = 3 * Kin, 3 * Kin
    else:
        x16 = (torch.randn(T, P, 2 * Kin, device=dev) * 0.05).half()
        w16 = torch.empty((4 * H, 2 * (Kin + H)), dtype=torch.float16, device=dev)
        ops.cast_f16_wlo(w, Kin, H, w16)
        kx, ldx = 2 * Kin, 2 * Kin
    h16 = torch.zeros((T + 1, P, 2 * H), dtype=torch.float16, device=dev)
    res["f16 W-ext"] = timeit(lambda: ops.lstm_layer_fwd_f16(x16, w16, b, plan.lens, T, P, kx, H, h16, hbf, S[:, :H], S[:, H:], 2 * H, plan=plan, ldx=ldx, h_wide=True))
    # plain f16 (layer 0: x in 2 segments)
    kx1 = 2 * Kin if layer == 0 else Kin
    w16p = torch.empty((4 * H, kx1 + H), dtype=torch.float16, device=dev)
    ops.cast_f16_wide(w, Kin, H, kx1 // Kin, w16p, h_ext=False)
    h16p = torch.zeros((T + 1, P, H), dtype=torch.float16, device=dev)
    res["f16 plain"] = timeit(lambda: ops.lstm_layer_fwd_f16(x16, w16p, b, plan.lens, T, P, kx1, H, h16p, hbf, S[:, :H], S[:, H:], 2 * H, plan=plan, ldx=ldx))
    # f16 + fp8 low-order halves
    if Kin % 128 == 0:
        ldx8 = 2 * Kin if layer == 0 else 3 * H // 2
        x8 = (torch.randn(T, P, ldx8, device=dev) * 0.05).half()
        hi_cols = Kin if layer == 0 else 0          # layer 0: the input's low-order half against e4m3(Wx 2^6) as well
        w8 = torch.empty((4 * H, Kin + hi_cols + H), dtype=torch.uint8, device=dev)
        ops.cast_fp8_lo(w, w8, hi_cols=hi_cols)
        w16q = torch.empty((4 * H, Kin + H), dtype=torch.float16, device=dev)
        ops.cast_f16(w, w16q)
        h8 = torch.zeros((T + 1, P, 3 * H // 2), dtype=torch.float16, device=dev)
        args = (ldx8, Kin, 2 * Kin, 2 * Kin) if layer == 0 else (ldx8, H, 2 * H, H)
        res["f16 + fp8 lo"] = timeit(lambda: ops.lstm_layer_fwd_f16_fp8lo(x8, *args, w16q, w8, b, plan.lens, T, P, H, h8, hbf, S[:, :H], S[:, H:], 2 * H, plan=plan))
    print("layer %d (rows %d..%d, %.1f GFLOP per step algorithmic): " % (layer, rows[0], rows[-1], flops / 1e9) +
          " | ".join("%s %.1f us" % (k, v) for k, v in res.items()), flush=True)
