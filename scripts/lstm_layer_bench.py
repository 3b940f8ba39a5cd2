"""Times one LSTM layer (forward T steps, BPTT T steps) at the teacher / student L1 shapes with a row plan of
a synthetic batch.  EVC_FORCE_TILE=<n> pins the tile (see pick_fwd_tile / evc_lstm_layer_bwd).

    python scripts/lstm_layer_bench.py [--shape teacher|student|l2] [--noplan]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from efficientvideoclassification_youtube8m_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="teacher")
ap.add_argument("--noplan", action="store_true")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--hoist", action="store_true", help="x-projection as one GEMM up front (what the engine does for M < 1024)")
a = ap.parse_args()
dev = "cuda:0"
B, H = a.batch, 1024
rng = np.random.default_rng(0)
n = rng.integers(120, 301, size=B)
if a.shape == "teacher":
    C, T, Kin = 20, 15, 1152
    _, lens, _ = ops.host_frame_counts(n, 1, C, T)
elif a.shape == "student":
    C, T, Kin = 5, 6, 1152
    _, lens, _ = ops.host_frame_counts(n, 10, C, T)
else:
    C, T, Kin = 1, 20, 4096
    lens = ops.host_frame_counts(n, 1, 20, 15)[2]
M = C * B
ld = torch.from_numpy(lens.astype(np.int32)).to(dev)
plan = None if a.noplan else ops.RowPlan(ld, lens, T)
P = plan.P if plan else M
rows = plan.rows if plan else [M] * T
x = (torch.randn(T, P, Kin, device=dev) * 0.05).to(torch.bfloat16)
wT = (torch.randn(4 * H, Kin + H, device=dev) * 0.02).to(torch.bfloat16)
w_il = torch.empty((Kin + H, 4 * H), dtype=torch.bfloat16, device=dev)
ops.transpose_to_bf16(wT, 4 * H, Kin + H, w_il, 4 * H, interleave_H=H)
b = torch.zeros(4 * H, device=dev)
hbuf = torch.zeros((T + 1, P, H), dtype=torch.bfloat16, device=dev)
S = torch.zeros((M, 2 * H), device=dev)
gates = torch.empty((T, P, H, 2), dtype=torch.int32, device=dev)
c_all = torch.zeros((T + 1, P, H), dtype=torch.bfloat16, device=dev)
dz4 = torch.zeros((T, P, 4 * H), dtype=torch.bfloat16, device=dev)
dcw = torch.empty((P, H), device=dev)
dS = torch.randn(M, 2 * H, device=dev)
dha = (torch.randn(T, P, H, device=dev) * 0.1).to(torch.bfloat16)
lens_d = plan.lens if plan else ld


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


zx = torch.empty((T, P, 4 * H), device=dev) if a.hoist else None
fwd = timeit(lambda: ops.lstm_layer_fwd(x, wT, b, lens_d, T, P, Kin, H, hbuf, S[:, :H], S[:, H:], 2 * H, gates, c_all, hoist=a.hoist,
                                        zx_ws=zx, plan=plan))
bwd = timeit(lambda: ops.lstm_layer_bwd(w_il, lens_d, T, P, Kin, H, gates, c_all, dS[:, :H], dS[:, H:], 2 * H, dha, dcw, dz4, plan=plan))
ffl = sum(2.0 * r * 4 * H * (Kin + (H if t else 0)) for t, r in enumerate(rows))
bfl = sum(2.0 * r * 4 * H * H for t, r in enumerate(rows) if t < T - 1)
print("tile=%s shape=%s M=%d P=%d rows[0]=%d rows[-1]=%d | fwd %.3f ms (%.1f us/step, %.0f TF/s) | bwd %.3f ms (%.1f us/step, %.0f TF/s)"
      % (os.environ.get("EVC_FORCE_TILE", "auto"), a.shape, M, P, rows[0], rows[-1], fwd, fwd / T * 1e3, ffl / fwd / 1e9,
         bwd, bwd / T * 1e3, bfl / bwd / 1e9))

if a.shape == "l2":      # the two-layer wavefront form of the same stack (evc_lstm_stack2_fwd)
    w1T = (torch.randn(4 * H, 2 * H, device=dev) * 0.02).to(torch.bfloat16)
    hb1 = torch.zeros((T + 1, P, H), dtype=torch.bfloat16, device=dev)
    S4 = torch.zeros((M, 4 * H), device=dev)
    zx2 = torch.empty((T * P, 4 * H), device=dev)
    g2 = [gates, torch.empty_like(gates)]
    c2 = [c_all, torch.zeros_like(c_all)]
    ms = timeit(lambda: ops.lstm_stack2_fwd(x, wT, b, w1T, b, lens_d, T, P, Kin, H, zx2, hbuf, hb1, S4, g2, c2))
    print("stack2 wavefront fwd (2 layers, T=%d, M=%d): %.3f ms" % (T, M, ms))

if a.shape in ("teacher", "student"):      # the two-layer BPTT wavefront (evc_lstm_stack2_bwd) against two layer-wise passes + the hoisted dX product
    w1 = torch.empty((2 * H, 4 * H), dtype=torch.bfloat16, device=dev)
    ops.transpose_to_bf16((torch.randn(4 * H, 2 * H, device=dev) * 0.02).to(torch.bfloat16), 4 * H, 2 * H, w1, 4 * H, interleave_H=H)
    g2, c2 = [gates, gates.clone()], [c_all, c_all.clone()]
    dz2 = [dz4, torch.zeros_like(dz4)]
    dc2 = [dcw, torch.empty_like(dcw)]
    db2 = [torch.zeros(4 * H, device=dev), torch.zeros(4 * H, device=dev)]
    dS4 = torch.randn(M, 4 * H, device=dev)
    ms = timeit(lambda: ops.lstm_stack2_bwd(w_il, w1, lens_d, T, P, Kin, H, g2, c2, dS4, dc2, dz2, db2, plan=plan))
    dxl = torch.empty((T * P, H), dtype=torch.bfloat16, device=dev)

    def layerwise():
        ops.lstm_layer_bwd(w1, lens_d, T, P, H, H, g2[1], c2[1], dS4[:, 2 * H:], dS4[:, 3 * H:], 4 * H, None, dc2[1], dz2[1], plan=plan)
        ops.gemm_nt(dz2[1].view(T * P, 4 * H), w1, T * P, H, 4 * H, dxl)
        ops.lstm_layer_bwd(w_il, lens_d, T, P, Kin, H, g2[0], c2[0], dS4[:, :H], dS4[:, H:], 4 * H, dxl.view(T, P, H), dc2[0], dz2[0], plan=plan)
    ms2 = timeit(layerwise)
    fl = sum(2.0 * r * 4 * H * H * (2 if t < T - 1 else 0) for t, r in enumerate(rows)) + 2.0 * sum(rows) * 4 * H * H
    print("two-layer BPTT: wavefront %.3f ms (%.1f us/launch, %.0f TF/s) | layer-wise + hoisted dX %.3f ms" % (ms, ms / (T + 1) * 1e3, fl / ms / 1e9, ms2))
