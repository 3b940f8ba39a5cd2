"""Evidence horizon of the 1e-3 claim: the real-size towers after 16 / 128 / 512 deterministic training iterations
(tests/_long_train.py), each checkpoint evaluated in both forward modes against the float64 oracle on the 4 evaluation videos.
ANALYSIS TOOL (drives the oracle as the checker, like the tests).

    python scripts/long_horizon.py train <dir> [batch=16] [lr=1e-3] [steps=16,128,512]
    python scripts/long_horizon.py eval <dir> [modes=bf16,high] [train batch]       -> one line per (checkpoint, mode); modes separated by ";" or ",":
           precision[:nodither|light|full|fixedrange ...][@batch], e.g. "bf16;high;high:nodither;high@256;split"
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


from _long_train import evaluate  # noqa: E402  (shared with tests/test_gpu_step.py)


def main():
    cmd, d = sys.argv[1], sys.argv[2]
    if cmd == "train":
        B = sys.argv[3] if len(sys.argv) > 3 else "16"
        lr = sys.argv[4] if len(sys.argv) > 4 else "1e-3"
        steps = sys.argv[5] if len(sys.argv) > 5 else "16,128,512"
        env = dict(os.environ, EVC_DETERMINISTIC=os.environ.get("EVC_DETERMINISTIC", "1"))
        sys.exit(subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_long_train.py"), d, B, lr, steps], env=env).returncode)
    import torch
    arg = sys.argv[3] if len(sys.argv) > 3 else "bf16,high"
    modes = arg.split(";") if ";" in arg else ([arg] if ":" in arg else arg.split(","))
    B_train = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    for f in sorted(glob.glob(os.path.join(d, "step*.pt")), key=lambda p: int(os.path.basename(p)[4:-3])):
        ck = torch.load(f, weights_only=False)
        mags, res = evaluate(ck, modes, B_train)
        print("steps %4d  |z| teacher %.2f student %.2f  |state| teacher %.2f student %.2f  |W| %.3f" % (ck["steps"], mags["z_teacher"], mags["z_student"], mags["s_teacher"], mags["s_student"], mags["w_max"]))
        for prec, e in res.items():
            print("   %-22s " % prec + " ".join("%s %.2e" % (k.replace("teacher", "t").replace("student", "s").replace("_logits", ""), v) for k, v in e.items() if not k.endswith("saturated"))
                  + ("  sat %s %s" % (e.get("teacher_saturated"), e.get("student_saturated")) if prec.startswith("high") else ""), flush=True)


if __name__ == "__main__":
    main()
