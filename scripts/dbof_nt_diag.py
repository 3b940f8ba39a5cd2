import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from efficientvideoclassification_youtube8m_amd import ops
DEV = "cuda:0"
for (B, F, C) in [(320, 128, 2304), (260, 256, 1024), (36, 192, 8192)]:
    S = 30
    rng = np.random.default_rng(B + C)
    Mp, P_in, P_cl = ops.dbof_workspace(B, S)
    rows = torch.from_numpy(ops.dbof_row_index(B, S).numpy().reshape(-1)).to(DEV)
    r_bn = torch.zeros((Mp, F), dtype=torch.bfloat16, device=DEV)
    r_bn[rows] = torch.from_numpy(rng.standard_normal((B * S, F)).astype(np.float32)).to(DEV).bfloat16()
    W = torch.from_numpy((rng.standard_normal((C, F)) / np.sqrt(F)).astype(np.float32)).to(DEV).bfloat16()
    ga = (1 + 0.3 * rng.standard_normal(C)).astype(np.float32); ga[::3] *= -1
    ga_d = torch.from_numpy(ga).to(DEV)
    ref = (r_bn.float() @ W.float().t())
    for rep in range(3):
        for mode in ("2", "0"):
            os.environ["EVC_DBOF_WALK"] = mode
            act = torch.full((Mp, C), 7.0, dtype=torch.bfloat16, device=DEV)
            part = torch.full((P_cl, 2, C), float("nan"), device=DEV)
            xsel = torch.full((B, C), float("nan"), device=DEV)
            arg = torch.full((B, C), 255, dtype=torch.uint8, device=DEV)
            ops.dbof_cluster_pool_fwd(r_bn, W, B, S, F, C, ga_d, xsel, arg, act=act, part=part)
            torch.cuda.synchronize()
            bad = ((act.float() - ref).abs() > 0.02 * (1 + ref.abs()))
            nb = int(bad.sum())
            msg = "B %d F %d C %d rep %d walk %s Mp %d: bad %d" % (B, F, C, rep, mode, Mp, nb)
            if nb:
                idx = bad.nonzero()
                r, c = idx[:, 0], idx[:, 1]
                msg += " rows %d..%d (distinct %d) cols %d..%d (distinct %d); first %s values %s ref %s" % (int(r.min()), int(r.max()), int(r.unique().numel()), int(c.min()), int(c.max()), int(c.unique().numel()),
                       idx[:6].tolist(), act[r[:6], c[:6]].float().tolist(), ref[r[:6], c[:6]].tolist())
                msg += " | sevens %d" % int((act[bad].float() == 7.0).sum())
            print(msg, flush=True)
