"""numpy/pure-python restatement of the reference's host-side metrics (TEST ORACLE).

PINNED: checked in tests/test_oracle_metrics.py against golden vectors minted
by importing the reference's own metric modules in the build container
(tests/golden/make_metric_golden.py -> tests/golden/metrics_golden.json).

Follows cs/eval_util.py:17-124 (Hit@1, PERR, GAP, top_k_by_class),
cs/average_precision_calculator.py:77-240 (heap accumulate, ap_at_n incl. the
random.seed(0) shuffle before the descending sort) and
cs/mean_average_precision_calculator.py:60-99.
"""
from __future__ import annotations

import heapq
import random

import numpy as np


def hit_at_one(predictions, actuals):
    """cs/eval_util.py:17-31."""
    top = np.argmax(predictions, 1)
    hits = actuals[np.arange(actuals.shape[0]), top]
    return np.average(hits)


def precision_at_equal_recall_rate(predictions, actuals):
    """cs/eval_util.py:34-58."""
    agg = 0.0
    num_videos = actuals.shape[0]
    for row in np.arange(num_videos):
        num_labels = int(np.sum(actuals[row]))
        top_indices = np.argpartition(predictions[row], -num_labels)[-num_labels:]
        item = 0.0
        for li in top_indices:
            if predictions[row][li] > 0:
                item += actuals[row][li]
        item /= top_indices.size
        agg += item
    return agg / num_videos


def top_k_triplets(predictions, labels, k=20):
    """cs/eval_util.py:118-124."""
    m = len(predictions)
    k = min(k, m)
    indices = np.argpartition(predictions, -k)[-k:]
    return [(index, predictions[index], labels[index]) for index in indices]


def top_k_by_class(predictions, labels, k=20):
    """cs/eval_util.py:82-116."""
    if k <= 0:
        raise ValueError("k must be a positive integer.")
    k = min(k, predictions.shape[1])
    num_classes = predictions.shape[1]
    trip = []
    for v in range(predictions.shape[0]):
        trip.extend(top_k_triplets(predictions[v], labels[v], k))
    out_p = [[] for _ in range(num_classes)]
    out_l = [[] for _ in range(num_classes)]
    for t in trip:
        out_p[t[0]].append(t[1])
        out_l[t[0]].append(t[2])
    out_tp = [np.sum(labels[:, i]) for i in range(num_classes)]
    return out_p, out_l, out_tp


class AveragePrecisionCalculator(object):
    """cs/average_precision_calculator.py:47-240."""

    def __init__(self, top_n=None):
        if not ((isinstance(top_n, int) and top_n >= 0) or top_n is None):
            raise ValueError("top_n must be a positive integer or None.")
        self._top_n = top_n
        self._total_positives = 0
        self._heap = []

    @property
    def heap_size(self):
        return len(self._heap)

    def accumulate(self, predictions, actuals, num_positives=None):
        if len(predictions) != len(actuals):
            raise ValueError("the shape of predictions and actuals does not match.")
        if num_positives is not None:
            self._total_positives += num_positives
        else:
            self._total_positives += np.size(np.where(np.asarray(actuals) > 0))
        topk, heap = self._top_n, self._heap
        for i in range(np.size(predictions)):
            if topk is None or len(heap) < topk:
                heapq.heappush(heap, (predictions[i], actuals[i]))
            elif predictions[i] > heap[0][0]:
                heapq.heappop(heap)
                heapq.heappush(heap, (predictions[i], actuals[i]))

    def clear(self):
        self._heap = []
        self._total_positives = 0

    def peek_ap_at_n(self):
        if self.heap_size <= 0:
            return 0
        predlists = np.array(list(zip(*self._heap)))
        return self.ap_at_n(predlists[0], predlists[1], n=self._top_n,
                            total_num_positives=self._total_positives)

    @staticmethod
    def ap(predictions, actuals):
        return AveragePrecisionCalculator.ap_at_n(predictions, actuals, n=None)

    @staticmethod
    def ap_at_n(predictions, actuals, n=20, total_num_positives=None):
        if len(predictions) != len(actuals):
            raise ValueError("the shape of predictions and actuals does not match.")
        if n is not None and (not isinstance(n, int) or n <= 0):
            raise ValueError("n must be 'None' or a positive integer. It was '%s'." % n)
        ap = 0.0
        predictions = np.array(predictions)
        actuals = np.array(actuals)
        # shuffle (seeded) before the stable descending sort: :203-209,:235-240
        random.seed(0)
        suffidx = random.sample(range(len(predictions)), len(predictions))
        predictions = predictions[suffidx]
        actuals = actuals[suffidx]
        sortidx = sorted(range(len(predictions)), key=lambda k: predictions[k], reverse=True)
        if total_num_positives is None:
            numpos = np.size(np.where(actuals > 0))
        else:
            numpos = total_num_positives
        if numpos == 0:
            return 0
        if n is not None:
            numpos = min(numpos, n)
        delta_recall = 1.0 / numpos
        poscount = 0.0
        r = len(sortidx)
        if n is not None:
            r = min(r, n)
        for i in range(r):
            if actuals[sortidx[i]] > 0:
                poscount += 1
                ap += poscount / (i + 1) * delta_recall
        return ap


def flatten(l):
    return [item for sub in l for item in sub]


def gap(predictions, actuals, top_k=20):
    """cs/eval_util.py:61-79."""
    calc = AveragePrecisionCalculator()
    sp, sl, npos = top_k_by_class(predictions, actuals, top_k)
    calc.accumulate(flatten(sp), flatten(sl), sum(npos))
    return calc.peek_ap_at_n()


class EvaluationMetrics(object):
    """cs/eval_util.py:126-213 (+ MeanAveragePrecisionCalculator)."""

    def __init__(self, num_class, top_k):
        if not isinstance(num_class, int) or num_class <= 1:
            raise ValueError("num_class must be a positive integer.")
        self.sum_hit_at_one = 0.0
        self.sum_perr = 0.0
        self.sum_loss = 0.0
        self._aps = [AveragePrecisionCalculator() for _ in range(num_class)]
        self._gap = AveragePrecisionCalculator()
        self.top_k = top_k
        self.num_examples = 0

    def accumulate(self, predictions, labels, loss):
        bs = labels.shape[0]
        h1 = hit_at_one(predictions, labels)
        perr = precision_at_equal_recall_rate(predictions, labels)
        ml = np.mean(loss)
        sp, sl, npos = top_k_by_class(predictions, labels, self.top_k)
        for i in range(len(sp)):
            self._aps[i].accumulate(sp[i], sl[i], npos[i])
        self._gap.accumulate(flatten(sp), flatten(sl), sum(npos))
        self.num_examples += bs
        self.sum_hit_at_one += h1 * bs
        self.sum_perr += perr * bs
        self.sum_loss += ml * bs
        return {"hit_at_one": h1, "perr": perr, "loss": ml}

    def get(self):
        if self.num_examples <= 0:
            raise ValueError("total_sample must be positive.")
        return {"avg_hit_at_one": self.sum_hit_at_one / self.num_examples,
                "avg_perr": self.sum_perr / self.num_examples,
                "avg_loss": self.sum_loss / self.num_examples,
                "aps": [c.peek_ap_at_n() for c in self._aps],
                "gap": self._gap.peek_ap_at_n()}
