"""Evidence horizon of the 1e-3 claim: the real-size towers after 16 / 128 / 512 deterministic training iterations
(tests/_long_train.py), each checkpoint evaluated in both forward modes against the float64 oracle on the 4 evaluation videos.
ANALYSIS TOOL (drives the oracle as the checker, like the tests).

    python scripts/long_horizon.py train <dir> [batch=16] [lr=1e-3] [steps=16,128,512]
    python scripts/long_horizon.py eval <dir> [modes=bf16,high]        -> one line per (checkpoint, mode)
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def evaluate(ck, modes, B_train, dev="cuda:0"):
    """{mode: {quantity: max abs error vs float64}} + magnitudes for one checkpoint."""
    import numpy as np
    import torch
    from oracle import model_math as mm
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    import _long_train as lt
    x, n, labels = lt.eval_videos(B_train)
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
    for prec in modes:
        g = DistillGraph(4, every_n=10, device=dev, seed=3, precision=prec)
        g.teacher.load_state_dict({k: v.to(dev) for k, v in sd.items()})
        g.student.load_state_dict({k: v.to(dev) for k, v in sd.items()})
        out = g.step(xd, yd, nd, apply=False, num_frames_host=n)
        e = {}
        for name, tw, sc, kp, ks in (("teacher", g.teacher, "model", "predictions", "teacher_state"), ("student", g.student, "model_student", "student_predictions", "student_state")):
            e[name + "_pred"] = float(np.abs(out[kp].cpu().numpy() - ref["teacher_predictions" if name == "teacher" else "student_predictions"]).max())
            e[name + "_state"] = float(np.abs(out[ks].cpu().numpy() - ref[ks]).max())
            e[name + "_gate_logits"] = float(np.abs(tw.moe.gate_logits.cpu().numpy() - ref_logits[sc][0]).max())
            e[name + "_expert_logits"] = float(np.abs(tw.moe.expert_logits.cpu().numpy() - ref_logits[sc][1]).max())
            if prec == "high":
                sat = {k: v for k, v in tw.fp8_saturation(out[ks]).items() if v}
                e[name + "_saturated"] = sat
        res[prec] = e
        del g
        torch.cuda.empty_cache()
    return mags, res


def main():
    cmd, d = sys.argv[1], sys.argv[2]
    if cmd == "train":
        B = sys.argv[3] if len(sys.argv) > 3 else "16"
        lr = sys.argv[4] if len(sys.argv) > 4 else "1e-3"
        steps = sys.argv[5] if len(sys.argv) > 5 else "16,128,512"
        env = dict(os.environ, EVC_DETERMINISTIC=os.environ.get("EVC_DETERMINISTIC", "1"))
        sys.exit(subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_long_train.py"), d, B, lr, steps], env=env).returncode)
    import torch
    modes = (sys.argv[3] if len(sys.argv) > 3 else "bf16,high").split(",")
    B_train = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    for f in sorted(glob.glob(os.path.join(d, "step*.pt")), key=lambda p: int(os.path.basename(p)[4:-3])):
        ck = torch.load(f, weights_only=False)
        mags, res = evaluate(ck, modes, B_train)
        print("steps %4d  |z| teacher %.2f student %.2f  |state| teacher %.2f student %.2f  |W| %.3f" % (ck["steps"], mags["z_teacher"], mags["z_student"], mags["s_teacher"], mags["s_student"], mags["w_max"]))
        for prec, e in res.items():
            print("   %-5s " % prec + " ".join("%s %.2e" % (k.replace("teacher", "t").replace("student", "s").replace("_logits", ""), v) for k, v in e.items() if not k.endswith("saturated"))
                  + ("  sat %s %s" % (e.get("teacher_saturated"), e.get("student_saturated")) if prec == "high" else ""), flush=True)


if __name__ == "__main__":
    main()
