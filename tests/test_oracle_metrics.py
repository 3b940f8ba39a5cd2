"""oracle/metrics.py against golden vectors minted from the reference's own
eval_util / average_precision_calculator (tests/golden/make_metric_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import metrics as om

GOLD = os.path.join(os.path.dirname(__file__), "golden", "metrics_golden.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_batch_metrics_match_reference(gold):
    for c in gold["cases"]:
        p = np.array(c["predictions"], np.float32)
        y = np.array(c["labels"], np.float32)
        assert om.hit_at_one(p, y) == pytest.approx(c["hit_at_one"], abs=0, rel=1e-15)
        assert om.precision_at_equal_recall_rate(p, y) == pytest.approx(c["perr"], abs=0, rel=1e-15)
        assert om.gap(p, y, c["top_k"]) == pytest.approx(c["gap"], abs=0, rel=1e-15)


def test_streaming_metrics_match_reference(gold):
    for c in gold["cases"]:
        p = np.array(c["predictions"], np.float32)
        y = np.array(c["labels"], np.float32)
        loss = np.array(c["loss"], np.float32)
        em = om.EvaluationMetrics(p.shape[1], c["top_k"])
        half = max(1, p.shape[0] // 2)
        em.accumulate(p[:half], y[:half], loss[:half])
        if half < p.shape[0]:
            em.accumulate(p[half:], y[half:], loss[half:])
        got = em.get()
        for k in ("avg_hit_at_one", "avg_perr", "avg_loss", "gap"):
            assert got[k] == pytest.approx(c["stream"][k], abs=0, rel=1e-15), k
        assert np.allclose(got["aps"], c["stream"]["aps"], rtol=1e-15, atol=0)


def test_ap_known_answers(gold):
    for c in gold["ap_cases"]:
        p, a = np.array(c["predictions"]), np.array(c["actuals"])
        assert om.AveragePrecisionCalculator.ap(p, a) == pytest.approx(c["ap"], rel=1e-15)
        assert om.AveragePrecisionCalculator.ap_at_n(p, a, n=5) == pytest.approx(c["ap_at_5"], rel=1e-15)
