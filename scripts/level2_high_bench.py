"""The "high" mode's teacher L1 level (two layers, 15 steps, the row plan of a synthetic batch) alone on the GPU: one launch per layer and step
(evc_lstm_layer_fwd_f16_fp8lo + evc_lstm_layer_fwd_f16_dith) against the two-tile launches of evc_lstm_level2_fwd_high; ms per level, both input forms.

    python scripts/level2_high_bench.py [--batch 256] [--reps 8]
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
S = torch.zeros((M, 4 * H), device=dev)
b0 = torch.zeros(4 * H, device=dev)
b1 = torch.zeros(4 * H, device=dev)
k0 = torch.randn(4 * H, F + H, device=dev) * 0.02
k1 = torch.randn(4 * H, 2 * H, device=dev) * 0.02
w16 = torch.empty((4 * H, F + H), dtype=torch.float16, device=dev)
ops.cast_f16(k0, w16)
w8 = torch.empty((4 * H, 2 * F + 2 * H), dtype=torch.uint8, device=dev)
ops.cast_fp8_lo(k0, w8, hi_cols=F, hi_tail=True)
w16d = torch.empty((T, 4 * H, 2 * H), dtype=torch.float16, device=dev)
ops.cast_f16_dither(k1, w16d, 7)
h0 = torch.zeros((T + 1, P, 2 * H), dtype=torch.float16, device=dev)
h1 = torch.zeros((T + 1, P, H), dtype=torch.float16, device=dev)
hb = [torch.zeros((T + 1, P, H), dtype=torch.bfloat16, device=dev) for _ in range(2)]
gates = [torch.zeros((T, P, H, 2), dtype=torch.int32, device=dev) for _ in range(2)]
c_all = [torch.zeros((T + 1, P, H), dtype=torch.bfloat16, device=dev) for _ in range(2)]
flops = sum(2.0 * r * 4 * H * (k + (H if t > 0 else 0)) for k in (F, H) for t, r in enumerate(rows))


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
    return best


for form in ("uint8", "f32"):
    if form == "uint8":
        x = (torch.randn(T, P, 3 * F // 2, device=dev) * 0.05).half()
        x[:, :, :F] = torch.randint(-255, 256, (T, P, F), device=dev).half()
        rs = torch.full((T, P), 1e-3, device=dev)
        cc = torch.zeros(4 * H, device=dev)
        kw = dict(h_lo=True, x_int=(rs, cc), b8_gap=F)
        x8_off, kx8 = 2 * F, F
    else:
        x = (torch.randn(T, P, 2 * F, device=dev) * 0.05).half()
        kw = dict(h_lo=True)
        x8_off, kx8 = 2 * F, 2 * F

    def two_calls():
        ops.lstm_layer_fwd_f16_fp8lo(x, x.shape[-1], F, x8_off, kx8, w16, w8, b0, plan.lens, T, P, H, h0, hb[0], S[:, 0:], S[:, H:], 4 * H, gates[0], c_all[0], plan=plan, **kw)
        ops.lstm_layer_fwd_f16_dith(h0[1:], 2 * H, H, 0, 0, w16d, None, 0, 7 + ops.FP8_W_SCALE_EXP, b1, plan.lens, T, P, H, h1, hb[1], S[:, 2 * H:], S[:, 3 * H:], 4 * H,
                                    gates[1], c_all[1], plan=plan)

    def l0_only():
        ops.lstm_layer_fwd_f16_fp8lo(x, x.shape[-1], F, x8_off, kx8, w16, w8, b0, plan.lens, T, P, H, h0, hb[0], S[:, 0:], S[:, H:], 4 * H, gates[0], c_all[0], plan=plan, **kw)

    def walk():
        ops.lstm_level2_fwd_high(x, x.shape[-1], F, x8_off, kx8, w16, w8, b0, w16d, b1, plan.lens, T, P, H, h0, hb[0], h1, hb[1], S, gates, c_all, plan=plan, **kw)

    t2, t0, tw = timeit(two_calls), timeit(l0_only), timeit(walk)
    print("%s frames: rows %d..%d | layer by layer %.3f ms (layer 0 alone %.3f = %.1f us/step, layer 1 %.1f us/step; %.0f TF/s) | two-tile launches %.3f ms (%.0f TF/s)"
          % (form, rows[0], rows[-1], t2, t0, t0 * 1e3 / T, (t2 - t0) * 1e3 / T, flops / t2 / 1e9, tw, flops / tw / 1e9), flush=True)
