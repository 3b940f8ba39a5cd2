"""Helper process of the GPU tests / scripts/long_horizon.py: trains the REAL-SIZE teacher+student towers for hundreds of iterations
under EVC_DETERMINISTIC=1 (same weights on every run and every box) and saves a TF-named checkpoint at each requested step count.

    python tests/_long_train.py <out-dir> <batch> <lr> <steps,steps,...> [pool-batches]

Training data: a fixed pool of `pool-batches` synthetic batches (oracle.model_math.synthetic_batch, seeds 9100..) whose labels come
from a FIXED label function of the input (pool_batch below: the top-3 classes of a fixed random projection of the video's mean
l2-normalised frame, + class 0 for a third of the videos) - a learnable target, so the logits keep growing with the step count the
way they do on real data instead of collapsing onto the prior.  The first 4 videos of pool batch 0 are the ones the callers compare
against the float64 oracle (eval_videos below).  Learning rate / clip / l2 penalty: the reference's defaults (cs/train.py:71-94).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import model_math as mm  # noqa: E402

LABEL_SEED = 4242


def pool_batch(i, B, feature_size=1152, vocab_size=4716):
    """(x f32 [B,300,F] zero-padded, n int32 [B], labels bool [B,V]) of pool batch i; video 0 of batch 0 has all 300 frames."""
    q, x, n, _ = mm.synthetic_batch(B, seed=9100 + i, dtype=np.float32, feature_size=feature_size, vocab_size=vocab_size)
    if i == 0:
        n[0] = 300
        x[0] = mm.dequantize(q[0].astype(np.float32))
    x[np.arange(300)[None, :] >= n[:, None]] = 0.0
    xn = x / np.maximum(np.sqrt((x.astype(np.float64) ** 2).sum(-1, keepdims=True)), 1e-6)
    mean = xn.sum(1) / np.maximum(n, 1)[:, None]
    R = np.random.default_rng(LABEL_SEED).standard_normal((feature_size, vocab_size))
    score = mean @ R
    labels = np.zeros((B, vocab_size), bool)
    top = np.argsort(-score[:, 1:], axis=1)[:, :3] + 1
    labels[np.arange(B)[:, None], top] = True
    labels[:, 0] = score[:, 0] > np.quantile(score[:, 0], 0.7) if B > 1 else False
    return x.astype(np.float32), n, labels


def eval_videos_q(B):
    """The uint8 frames of eval_videos() (what the reader delivers; the input kernels dequantise and zero the padding rows themselves)."""
    q, _, n, _ = mm.synthetic_batch(B, seed=9100, dtype=np.float32)
    return q[:4].copy()


def eval_videos(B):
    """The 4 videos the long-horizon tests compare with the oracle: the head of pool batch 0 (seen in training: the towers' states
    and logits on them are as large as training has made them)."""
    x, n, labels = pool_batch(0, B)
    return x[:4].copy(), n[:4].copy(), labels[:4].copy()


def evaluate(ck, modes, B_train, dev="cuda:0"):
    """One checkpoint of main() against the float64 oracle on eval_videos(): ({magnitudes}, {mode: {quantity: max abs error}}) - predictions, states,
    gate / expert logits of both towers (+ the fp8_saturation() counts in "high")."""
    import numpy as np
    import torch
    from oracle import model_math as mm
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    x, n, labels = eval_videos(B_train)
    sd = ck["sd"]
    params = {sc: {k[len(sc) + 1:]: v.double().numpy() for k, v in sd.items() if k.startswith(sc + "/")} for sc in ("model", "model_student")}
    ref = mm.teacher_student_step(x.astype(np.float64), n, labels, params["model"], params["model_student"], 10, with_grads=False)
    ref_logits = {}
    for sc, st in (("model", ref["teacher_state"]), ("model_student", ref["student_state"])):
        ref_logits[sc] = (st @ params[sc]["classifier/gates/weights"], st @ params[sc]["classifier/experts/weights"] + params[sc]["classifier/experts/biases"])
    mags = dict(z_teacher=max(float(np.abs(a).max()) for a in ref_logits["model"]), z_student=max(float(np.abs(a).max()) for a in ref_logits["model_student"]),
                s_teacher=float(np.abs(ref["teacher_state"]).max()), s_student=float(np.abs(ref["student_state"]).max()),
                w_max=max(float(v.abs().max()) for v in sd.values()))
    xd, yd, nd = torch.from_numpy(x).to(dev), torch.from_numpy(labels.astype(np.uint8)).to(dev), torch.from_numpy(n).to(dev)
    res = {}
    from efficientvideoclassification_youtube8m_amd.engine import HLstmTower
    for mode in modes:
        # mode = precision[:option,...][@batch] - options of the "high" layout, switched in-process: nodither (every L1 layer on its weights' e4m3
        # low-order halves), light (the student's L1 level on plain f16: the default up to round 5), fixedrange (the head's input on the fixed 2^6 scale);
        # u8: the frames go in as the reader's uint8 quantisation (the integer-frame layer 0 of round 6; same values as the f32 frames);
        # @batch: the 4 evaluation videos at the head of a batch of that many (the rest: synthetic_batch(seed 92))
        spec, _, bs = mode.partition("@")
        prec, _, opts = spec.partition(":")
        opts = set(o for o in opts.split(",") if o)
        B = int(bs) if bs else 4
        saved = (HLstmTower.f16_dither_layers, os.environ.get("EVC_HIGH_STUDENT_LIGHT"))
        from efficientvideoclassification_youtube8m_amd.engine import MoeHead
        saved_dyn = MoeHead.dynamic_fp8_range
        HLstmTower.f16_dither_layers = () if "nodither" in opts else saved[0]
        os.environ["EVC_HIGH_STUDENT_LIGHT"] = "1" if "light" in opts else ("0" if "full" in opts else (saved[1] if saved[1] is not None else ""))
        if os.environ["EVC_HIGH_STUDENT_LIGHT"] == "":
            del os.environ["EVC_HIGH_STUDENT_LIGHT"]
        MoeHead.dynamic_fp8_range = saved_dyn and "fixedrange" not in opts
        xb, yb, nb, nhb = xd, yd, nd, n
        if "u8" in opts:
            xb = torch.from_numpy(eval_videos_q(B_train)).to(dev)
        if B > 4:
            qr, xr, nr, lr_ = mm.synthetic_batch(B - 4, seed=92, dtype=np.float32)
            xb = torch.cat([xb, torch.from_numpy(qr if "u8" in opts else xr).to(dev)])
            yb = torch.cat([yd, torch.from_numpy(lr_.astype(np.uint8)).to(dev)])
            nhb = np.concatenate([n, nr])
            nb = torch.from_numpy(nhb).to(dev)
        g = DistillGraph(B, every_n=10, device=dev, seed=3, precision=prec)
        g.teacher.load_state_dict({k: v.to(dev) for k, v in sd.items()})
        g.student.load_state_dict({k: v.to(dev) for k, v in sd.items()})
        out = g.step(xb, yb, nb, apply=False, num_frames_host=nhb)
        e = {}
        for name, tw, sc, kp, ks in (("teacher", g.teacher, "model", "predictions", "teacher_state"), ("student", g.student, "model_student", "student_predictions", "student_state")):
            e[name + "_pred"] = float(np.abs(out[kp][:4].cpu().numpy() - ref["teacher_predictions" if name == "teacher" else "student_predictions"]).max())
            e[name + "_state"] = float(np.abs(out[ks][:4].cpu().numpy() - ref[ks]).max())
            e[name + "_gate_logits"] = float(np.abs(tw.moe.gate_logits[:4].cpu().numpy() - ref_logits[sc][0]).max())
            e[name + "_expert_logits"] = float(np.abs(tw.moe.expert_logits[:4].cpu().numpy() - ref_logits[sc][1]).max())
            if prec == "high":
                sat = {k: v for k, v in tw.fp8_saturation(out[ks]).items() if v}
                e[name + "_saturated"] = sat
        res[mode] = e
        del g
        HLstmTower.f16_dither_layers = saved[0]
        MoeHead.dynamic_fp8_range = saved_dyn
        if saved[1] is None:
            os.environ.pop("EVC_HIGH_STUDENT_LIGHT", None)
        else:
            os.environ["EVC_HIGH_STUDENT_LIGHT"] = saved[1]
        torch.cuda.empty_cache()
    return mags, res


def main():
    import torch
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    out_dir, B, lr = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    marks = sorted(int(v) for v in sys.argv[4].split(","))
    pool_n = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    dev = "cuda:0"
    pool = []
    for i in range(pool_n):
        x, n, labels = pool_batch(i, B)
        pool.append((torch.from_numpy(x).to(dev), torch.from_numpy(labels.astype(np.uint8)).to(dev), torch.from_numpy(n).to(dev), n))
    g = DistillGraph(B, every_n=10, device=dev, seed=int(os.environ.get("EVC_LONG_SEED", "3")), base_learning_rate=lr)     # (EVC_LONG_SEED: another deterministic draw)
    os.makedirs(out_dir, exist_ok=True)
    for it in range(1, marks[-1] + 1):
        xd, yd, nd, nh = pool[(it - 1) % pool_n]
        o = g.step(xd, yd, nd, num_frames_host=nh)
        if it in marks:
            sd = {}
            sd.update(g.teacher.state_dict())
            sd.update(g.student.state_dict())
            torch.cuda.synchronize()
            info = dict(steps=it, losses=g.loss_report(), deterministic=os.environ.get("EVC_DETERMINISTIC"),
                        s_max=max(float(o["teacher_state"].abs().max()), float(o["student_state"].abs().max())),
                        z_max=max(float(g.teacher.moe.gate_logits.abs().max()), float(g.student.moe.gate_logits.abs().max()),
                                  float(g.teacher.moe.expert_logits.abs().max()), float(g.student.moe.expert_logits.abs().max())),
                        w_max=max(float(v.abs().max()) for v in sd.values()))
            torch.save({"sd": {k: v.cpu() for k, v in sd.items()}, **info}, os.path.join(out_dir, "step%d.pt" % it))
            print("step %d: |state| %.2f |logit| %.2f |W| %.3f losses %s" % (it, info["s_max"], info["z_max"], info["w_max"],
                                                                            {k: round(v, 3) for k, v in info["losses"].items()}), flush=True)


if __name__ == "__main__":
    main()
