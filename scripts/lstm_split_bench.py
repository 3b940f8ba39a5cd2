"""Is a recurrent layer faster as G independent row groups on G streams?  (Rows are independent through the
recurrence; while one group's step kernel drains its stores / relaunches, the other group's kernel computes.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from efficientvideoclassification_youtube8m_amd import ops, streams  # noqa: E402

dev = "cuda:0"
H, Kin, T, C, B = 1024, 1152, 15, 20, 256
rng = np.random.default_rng(0)
n = rng.integers(120, 301, size=B)
_, lens, _ = ops.host_frame_counts(n, 1, C, T)
order = np.argsort(-lens, kind="stable")
live = int((lens > 0).sum())
wT = (torch.randn(4 * H, Kin + H, device=dev) * 0.02).to(torch.bfloat16)
w_il = torch.empty((Kin + H, 4 * H), dtype=torch.bfloat16, device=dev)
ops.transpose_to_bf16(wT, 4 * H, Kin + H, w_il, 4 * H, interleave_H=H)
b = torch.zeros(4 * H, device=dev)


def make_group(sel):
    """One row group = its own plan and buffers over the selected (already length-sorted) rows."""
    l = lens[sel].astype(np.int32)
    ld = torch.from_numpy(l).to(dev)
    plan = ops.RowPlan(ld, l, T)
    P = plan.P
    return dict(plan=plan, P=P, x=(torch.randn(T, P, Kin, device=dev) * 0.05).to(torch.bfloat16),
                hbuf=torch.zeros((T + 1, P, H), dtype=torch.bfloat16, device=dev), S=torch.zeros((len(sel), 2 * H), device=dev),
                gates=torch.empty((T, P, H, 2), dtype=torch.int32, device=dev),
                c_all=torch.zeros((T + 1, P, H), dtype=torch.bfloat16, device=dev),
                dz=torch.zeros((T, P, 4 * H), dtype=torch.bfloat16, device=dev), dcw=torch.empty((P, H), device=dev),
                dS=torch.randn(len(sel), 2 * H, device=dev), dha=(torch.randn(T, P, H, device=dev) * 0.1).to(torch.bfloat16))


def fwd(g):
    ops.lstm_layer_fwd(g["x"], wT, b, g["plan"].lens, T, g["P"], Kin, H, g["hbuf"], g["S"][:, :H], g["S"][:, H:], 2 * H,
                       g["gates"], g["c_all"], plan=g["plan"])


def bwd(g):
    ops.lstm_layer_bwd(w_il, g["plan"].lens, T, g["P"], Kin, H, g["gates"], g["c_all"], g["dS"][:, :H], g["dS"][:, H:], 2 * H,
                       g["dha"], g["dcw"], g["dz"], plan=g["plan"])


ss = streams.concurrent_streams(dev, 4)
for G in (1, 2, 3, 4):
    # interleave the sorted live rows over the groups so that every group sees the same length mix
    groups = [make_group(order[:live][i::G]) for i in range(G)]
    for fn, name in ((fwd, "fwd"), (bwd, "bwd")):
        def run():
            ev = torch.cuda.Event()
            ev.record()
            for g, s in zip(groups, ss):
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    fn(g)
            for s in ss[:G]:
                torch.cuda.current_stream().wait_stream(s)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        e1.synchronize()
        print("groups=%d %s: %.3f ms per layer (%.1f us per step), P per group %s" % (G, name, e0.elapsed_time(e1) / 5,
              e0.elapsed_time(e1) / 5 / T * 1e3, [g["P"] for g in groups]), flush=True)
