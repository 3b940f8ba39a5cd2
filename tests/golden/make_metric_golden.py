"""Mint golden vectors for the host metrics by running the REFERENCE's own code.

Runs only in the build container (needs /root/reference).  It imports
code_student_uniform/eval_util.py (its single unused TensorFlow import,
eval_util.py:8 ``from tensorflow.python.platform import gfile``, is satisfied
by an empty stub module) and records inputs -> outputs into
tests/golden/metrics_golden.json.  Only data is committed; no reference source.

    python tests/golden/make_metric_golden.py
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/code_student_uniform"


def _import_reference():
    for name in ("tensorflow", "tensorflow.python", "tensorflow.python.platform",
                 "tensorflow.python.platform.gfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["tensorflow.python.platform"].gfile = sys.modules["tensorflow.python.platform.gfile"]
    sys.path.insert(0, REF)
    import eval_util  # noqa
    import average_precision_calculator as apc  # noqa
    return eval_util, apc


def make_case(rng, batch, classes, kind):
    pred = rng.random((batch, classes)).astype(np.float32)
    labels = np.zeros((batch, classes), np.float32)
    for b in range(batch):
        k = int(rng.integers(1, 5))
        labels[b, rng.choice(classes, size=k, replace=False)] = 1.0
    if kind == "ties":
        pred = np.round(pred * 8) / 8           # many exact ties
    elif kind == "some_zero":
        pred[pred < 0.3] = 0.0                  # PERR's "prediction > 0" branch
    elif kind == "peaked":
        pred = pred ** 8
    return pred, labels


def main():
    eval_util, apc = _import_reference()
    rng = np.random.default_rng(20260303)
    cases = []
    for (batch, classes, kind, top_k) in [(4, 30, "plain", 20), (7, 64, "ties", 20), (5, 25, "some_zero", 20),
                                          (16, 200, "peaked", 20), (3, 12, "plain", 20), (6, 50, "plain", 5)]:
        pred, labels = make_case(rng, batch, classes, kind)
        rec = {"kind": kind, "top_k": top_k,
               "predictions": pred.tolist(), "labels": labels.tolist(),
               "hit_at_one": float(eval_util.calculate_hit_at_one(pred, labels)),
               "perr": float(eval_util.calculate_precision_at_equal_recall_rate(pred, labels)),
               "gap": float(eval_util.calculate_gap(pred, labels, top_k))}
        # streaming EvaluationMetrics over two mini-batches (cs/eval_util.py:126-213)
        em = eval_util.EvaluationMetrics(classes, top_k)
        half = max(1, batch // 2)
        loss = rng.random(batch).astype(np.float32)
        em.accumulate(pred[:half], labels[:half], loss[:half])
        if half < batch:
            em.accumulate(pred[half:], labels[half:], loss[half:])
        got = em.get()
        rec["loss"] = loss.tolist()
        rec["stream"] = {"avg_hit_at_one": float(got["avg_hit_at_one"]), "avg_perr": float(got["avg_perr"]),
                         "avg_loss": float(got["avg_loss"]), "gap": float(got["gap"]),
                         "aps": [float(a) for a in got["aps"]]}
        cases.append(rec)
    # raw AP known answers
    ap_cases = []
    for n in (10, 37):
        p = rng.random(n)
        a = (rng.random(n) > 0.6).astype(np.float64)
        ap_cases.append({"predictions": p.tolist(), "actuals": a.tolist(),
                         "ap": float(apc.AveragePrecisionCalculator.ap(p, a)),
                         "ap_at_5": float(apc.AveragePrecisionCalculator.ap_at_n(p, a, n=5))})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "metrics_golden.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_metric_golden.py",
                   "reference": "code_student_uniform/eval_util.py, average_precision_calculator.py, "
                                "mean_average_precision_calculator.py",
                   "cases": cases, "ap_cases": ap_cases}, f)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
