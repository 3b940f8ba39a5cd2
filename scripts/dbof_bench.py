"""Micro-benchmark of the DBoF fused kernels at the BASELINE cfg-4 shapes (B=512, S=30, F=1152, C=8192)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from efficientvideoclassification_youtube8m_amd import ops

dev = "cuda:0"
torch.manual_seed(0)
B, S, F, C = int(os.environ.get("B", 512)), 30, 1152, 8192
Mp, P_in, P_cl = ops.dbof_workspace(B, S)


def bench(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


r_bn = (torch.randn(Mp, F, device=dev)).bfloat16()
xhat = (torch.randn(Mp, F, device=dev)).bfloat16()
W = (torch.randn(C, F, device=dev) / F ** 0.5)
Wb = W.bfloat16()
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
act = torch.empty(Mp, C, dtype=torch.bfloat16, device=dev)
part = torch.empty(P_cl, 2, C, device=dev)
xsel = torch.empty(B, C, device=dev); arg = torch.empty(B, C, dtype=torch.uint8, device=dev)
fl = 2.0 * B * S * F * C
which = sys.argv[1:] or ["fwd", "plain", "tn", "dact", "finish", "stats"]
if "fwd" in which:
    ms = bench(lambda: ops.dbof_cluster_pool_fwd(r_bn, Wb, B, S, F, C, gamma, xsel, arg, act=act, part=part))
    print("cluster_pool_fwd train (act + stats + select): %.3f ms  %.0f TF/s algorithmic (%.0f incl. padding rows)" % (ms, fl / ms / 1e9, fl * Mp / (B * S) / ms / 1e9))
    ms = bench(lambda: ops.dbof_cluster_pool_fwd(r_bn, Wb, B, S, F, C, gamma, xsel, arg, part=part))
    print("cluster_pool_fwd stats + select, no act store:  %.3f ms  %.0f TF/s" % (ms, fl / ms / 1e9))
    ms = bench(lambda: ops.dbof_cluster_pool_fwd(r_bn, Wb, B, S, F, C, gamma, xsel, arg))
    print("cluster_pool_fwd eval (select only):            %.3f ms  %.0f TF/s" % (ms, fl / ms / 1e9))
if "plain" in which:
    outb = torch.empty(Mp, C, dtype=torch.bfloat16, device=dev)
    ms = bench(lambda: ops.gemm_nt(r_bn, Wb, Mp, C, F, outb))
    print("plain gemm_nt bf16 out %dx%dx%d:           %.3f ms  %.0f TF/s (padded rows counted)" % (Mp, C, F, ms, 2.0 * Mp * C * F / ms / 1e9))
    outf = torch.empty(Mp, C, device=dev)
    ms = bench(lambda: ops.gemm_nt(r_bn, Wb, Mp, C, F, outf))
    print("plain gemm_nt f32 out:                          %.3f ms  %.0f TF/s" % (ms, 2.0 * Mp * C * F / ms / 1e9))
if "tn" in which:
    for n in (1, 2, 3, 4, 6, 8):
        slabs = torch.empty(n, C, F, device=dev)
        ms = bench(lambda: ops.gemm_tn_slabs(act, xhat, C, F, Mp, slabs, n))
        print("gemm_tn_slabs nslab=%d: %.3f ms  %.0f TF/s" % (n, ms, 2.0 * Mp * C * F / ms / 1e9))
    out = torch.empty(C, F, device=dev)
    ms = bench(lambda: ops.gemm_tn(act, xhat, C, F, Mp, out))
    print("gemm_tn (atomics): %.3f ms" % ms)
if "dact" in which:
    dpooled = torch.randn(B, C, device=dev); pooled = torch.rand(B, C, device=dev) * 3
    mean = torch.zeros(C, device=dev); var = torch.ones(C, device=dev)
    ws = torch.zeros(2 * C, dtype=torch.float64, device=dev)
    ms = bench(lambda: ops.dbof_dact(act, dpooled, pooled, arg, mean, var, gamma, ws, B * S, B, S, C))
    print("dact: %.3f ms  %.2f TB/s (r+w of the bf16 activation)" % (ms, 2.0 * Mp * C * 2 / ms / 1e9))
if "finish" in which:
    for n in (1, 3):
        slabs = torch.randn(n, C, F, device=dev)
        dW = torch.empty(C, F, device=dev); dg = torch.empty(F, device=dev); db = torch.empty(F, device=dev)
        gin = torch.ones(F, device=dev)
        ms = bench(lambda: ops.dbof_wgrad_finish(slabs, n, C, F, W, gin, dW, dg, db))
        print("wgrad_finish nslab=%d: %.3f ms  %.2f TB/s" % (n, ms, (n + 2) * C * F * 4 / ms / 1e9))
if "stats" in which:
    wsc = torch.empty(2 * C, dtype=torch.float64, device=dev)
    ms = bench(lambda: ops.bn_partials_reduce(part, P_cl, C, wsc))
    print("partials_reduce cluster [%d][2][%d]: %.3f ms" % (P_cl, C, ms))
    pin = torch.randn(P_in, 2, F, device=dev); wsi = torch.empty(2 * F, dtype=torch.float64, device=dev)
    ms = bench(lambda: ops.bn_partials_reduce(pin, P_in, F, wsi))
    print("partials_reduce input [%d][2][%d]: %.3f ms" % (P_in, F, ms))
