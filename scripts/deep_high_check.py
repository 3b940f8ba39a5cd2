import numpy as np, torch, sys
sys.path.insert(0, ".")
from oracle import model_math as mm
from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
DEV = "cuda:0"
for L in (1, 2, 3):
    B, F, V, H = 8, 384, 100, 384
    q, x, n, labels = mm.synthetic_batch(B, seed=35, feature_size=F, vocab_size=V, dtype=np.float32)
    xd, nd, yd = torch.from_numpy(q).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV)
    g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, lstm_layers=L, device=DEV, seed=4, precision="high")
    print("L", L, "dither layers", g.teacher.dither_layers(), g.teacher.precision_layout()["l1"][:120])
    for _ in range(2):
        out = g.step(xd, yd, nd, num_frames_host=n)
    torch.cuda.synchronize()
    rep = g.loss_report()
    assert all(np.isfinite(v) for v in rep.values()), rep
    assert torch.isfinite(out["predictions"]).all()
    # against the same graph with every weight corrected (fresh graph, same seed): predictions of the first step agree closely
    print("  losses", {k: round(float(v), 5) for k, v in rep.items()})
print("ok")
