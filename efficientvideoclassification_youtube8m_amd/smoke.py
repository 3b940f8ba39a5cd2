"""One tiny teacher+student training iteration on cuda:0, checked against the
CPU oracle (used by __graft_entry__.smoke and the GPU tests)."""
from __future__ import annotations

import numpy as np
import torch


def tower_params_numpy(tower):
    """TF-layout float64 copies keyed WITHOUT the scope prefix (oracle naming)."""
    sd = tower.state_dict()
    pre = tower.scope + "/"
    return {k[len(pre):]: v.detach().cpu().double().numpy() for k, v in sd.items()}


def tower_grads_numpy(tower):
    out = {}
    for k in tower.names:
        g = tower.store.g(k)
        out[k] = (g.t() if g.dim() == 2 else g).detach().cpu().double().numpy()
    return out


def run(batch=4, feature_size=64, lstm_cells=64, vocab_size=48, every_n=10, seed=0, verbose=True):
    from oracle import model_math as mm          # checker only
    from . import ops
    from .distill import DistillGraph

    ops.check_device(0)
    dev = "cuda:0"
    q, x, n, labels = mm.synthetic_batch(batch, seed=seed + 5, feature_size=feature_size, vocab_size=vocab_size,
                                         dtype=np.float32)
    g = DistillGraph(batch, every_n=every_n, feature_size=feature_size, vocab_size=vocab_size, lstm_cells=lstm_cells,
                     device=dev, seed=seed)
    teacher, student = tower_params_numpy(g.teacher), tower_params_numpy(g.student)
    out = g.step(torch.from_numpy(x).to(dev), torch.from_numpy(labels.astype(np.uint8)).to(dev),
                 torch.from_numpy(n).to(dev), apply=False)
    torch.cuda.synchronize()
    ref = mm.teacher_student_step(x.astype(np.float64), n, labels, teacher, student, every_n)
    rep = g.loss_report()
    err = {
        "teacher_state": float(np.abs(out["teacher_state"].cpu().numpy() - ref["teacher_state"]).max()),
        "teacher_pred": float(np.abs(out["predictions"].cpu().numpy() - ref["teacher_predictions"]).max()),
        "student_state": float(np.abs(out["student_state"].cpu().numpy() - ref["student_state"]).max()),
        "student_pred": float(np.abs(out["student_predictions"].cpu().numpy() - ref["student_predictions"]).max()),
    }
    for k, rk in (("label_loss", "label_loss"), ("student_loss_state", "student_loss_state"),
                  ("pred_loss", "pred_loss"), ("student_label_loss", "student_label_loss")):
        err[k] = abs(rep[k] - ref[rk]) / max(abs(ref[rk]), 1e-6)
    assert np.array_equal(out["num_frames_student"].cpu().numpy(), ref["num_frames_student"]), "frame counts differ"
    if verbose:
        print("smoke: max abs / rel errors vs float64 oracle:", {k: "%.2e" % v for k, v in err.items()})
    assert err["teacher_pred"] < 1e-3 and err["student_pred"] < 1e-3, err
    assert err["teacher_state"] < 2e-2 and err["student_state"] < 2e-2, err
    assert all(err[k] < 2e-3 for k in ("label_loss", "student_label_loss")), err
    return g, out, ref, err
