"""evc_moe_grad_update (fused rank-B gradient + clip + Adam + shadows) vs HBM roofline."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from efficientvideoclassification_youtube8m_amd import ops  # noqa: E402

dev = "cuda:0"
K, rows = 4096, 256
for V in (14148, 9432):
    Vp = (V + 63) // 64 * 64
    dlog = torch.zeros(rows, Vp, dtype=torch.bfloat16, device=dev)
    dlog[:, :V] = (torch.randn(rows, V, device=dev) * 1e-3).to(torch.bfloat16)
    x = (torch.randn(rows, K, device=dev) * 0.1).to(torch.bfloat16)
    p, m, v = (torch.randn(V, K, device=dev) * 0.01 for _ in range(3))
    v.abs_()
    pb = torch.empty(V, K, dtype=torch.bfloat16, device=dev)
    pT = torch.zeros(K, Vp, dtype=torch.bfloat16, device=dev)
    sums = torch.zeros(2, device=dev)
    ws = torch.empty(2 * ((V + 127) // 128) * (K // 128), device=dev)

    def run():
        sums.zero_()
        ops.moe_grad_update(dlog, x, rows, V, K, p, m, v, pb, pT, 2e-8, sums, ws, 1.0, 1e-3)

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    n = V * K
    print("moe_grad_update V=%d: %.1f us for %.1f M params = %.2f TB/s at 34 B/param (p read twice, m, v read, p, m, v, 2 shadows written)"
          % (V, ms * 1e3, n / 1e6, n * 34 / ms / 1e9))
