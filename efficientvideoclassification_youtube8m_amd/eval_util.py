"""Host-side evaluation metrics: Hit@1, PERR, GAP@k, per-class AP / mAP.

Same functions and results as the reference's cs/eval_util.py:17-213 (+
cs/average_precision_calculator.py, cs/mean_average_precision_calculator.py),
re-designed for throughput: the reference walks python loops over every video
and every pooled (score, label) pair through a heapq; here top-k selection is
one argpartition per batch and every average precision is a sort + cumsum.
Results are bit-identical to the reference whenever the pooled predictions are
distinct; when exact ties are present the reference's value depends on its
heap order and seeded shuffle, so that (rare) case is routed through an exact
emulation of those two steps.  Accepts numpy arrays or torch tensors (device
tensors are copied to the host first - the metrics are host arithmetic in
float64, as in the reference).
"""
from __future__ import annotations

import heapq
import random

import numpy as np


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def flatten(l):
    return [item for sub in l for item in sub]


def calculate_hit_at_one(predictions, actuals):
    """cs/eval_util.py:17-31."""
    predictions, actuals = _np(predictions), _np(actuals)
    top = np.argmax(predictions, 1)
    return np.average(actuals[np.arange(actuals.shape[0]), top])


def calculate_precision_at_equal_recall_rate(predictions, actuals):
    """cs/eval_util.py:34-58, rows grouped by their label count so each group is
    one argpartition."""
    predictions, actuals = _np(predictions), _np(actuals)
    num_videos, num_classes = actuals.shape
    counts = actuals.sum(axis=1).astype(np.int64)
    total = 0.0
    per_row = np.zeros(num_videos, dtype=np.float64)
    for k in np.unique(counts):
        rows = np.nonzero(counts == k)[0]
        p, a = predictions[rows], actuals[rows]
        if k == 0:
            # argpartition(x, -0)[-0:] is the whole row (python slice semantics): precision 0
            continue
        idx = np.argpartition(p, -int(k), axis=1)[:, -int(k):]
        pv = np.take_along_axis(p, idx, 1)
        av = np.take_along_axis(a, idx, 1).astype(np.float64)
        per_row[rows] = np.where(pv > 0, av, 0.0).sum(axis=1) / float(k)
    for v in per_row:            # same left-to-right float accumulation as the reference loop
        total += v
    return total / num_videos


def top_k_by_video(predictions, labels, k=20):
    """(scores [B,k], labels [B,k], class ids [B,k]) of each video's k best
    predictions, in np.argpartition order (what cs/eval_util.py:118-124 keeps)."""
    predictions, labels = _np(predictions), _np(labels)
    if k <= 0:
        raise ValueError("k must be a positive integer.")
    k = min(k, predictions.shape[1])
    idx = np.argpartition(predictions, -k, axis=1)[:, -k:]
    return np.take_along_axis(predictions, idx, 1), np.take_along_axis(labels, idx, 1), idx


def top_k_by_class(predictions, labels, k=20):
    """cs/eval_util.py:82-116 (same return structure: lists per class)."""
    predictions, labels = _np(predictions), _np(labels)
    if k <= 0:
        raise ValueError("k must be a positive integer.")
    num_classes = predictions.shape[1]
    pv, lv, idx = top_k_by_video(predictions, labels, k)
    out_p = [[] for _ in range(num_classes)]
    out_l = [[] for _ in range(num_classes)]
    fi, fp, fl = idx.reshape(-1), pv.reshape(-1), lv.reshape(-1)
    order = np.argsort(fi, kind="stable")       # class-major, video order preserved inside a class
    for c, p, l in zip(fi[order], fp[order], fl[order]):
        out_p[c].append(p)
        out_l[c].append(l)
    out_tp = [np.sum(labels[:, i]) for i in range(num_classes)]
    return out_p, out_l, out_tp


def _ap_sorted(actuals_sorted_desc, numpos, n=None):
    """Non-interpolated AP from labels already ordered by descending score
    (cs/average_precision_calculator.py:210-232)."""
    if numpos == 0:
        return 0
    if n is not None:
        numpos = min(numpos, n)
        actuals_sorted_desc = actuals_sorted_desc[:n]
    pos = actuals_sorted_desc > 0
    if not pos.any():
        return 0.0
    delta_recall = 1.0 / numpos
    ranks = np.nonzero(pos)[0] + 1
    ap = 0.0
    for j, r in enumerate(ranks):               # same accumulation order as the reference loop
        ap += (j + 1.0) / r * delta_recall
    return ap


def _exact_tie_order(predictions, actuals):
    """The reference's order when scores tie: heap array order, then
    random.seed(0) shuffle, then a stable descending sort."""
    heap = []
    for p, a in zip(predictions, actuals):
        heapq.heappush(heap, (p, a))
    pl = np.array(list(zip(*heap)))
    p, a = pl[0], pl[1]
    random.seed(0)
    perm = random.sample(range(len(p)), len(p))
    p, a = p[perm], a[perm]
    order = sorted(range(len(p)), key=lambda k: p[k], reverse=True)
    return a[order]


def average_precision(predictions, actuals, total_num_positives=None, n=None):
    """AP of a pooled (score, label) list; n=None uses every entry."""
    p = np.asarray(predictions)
    a = np.asarray(actuals)
    if len(p) == 0:
        return 0
    numpos = np.size(np.where(a > 0)) if total_num_positives is None else total_num_positives
    if len(np.unique(p)) == len(p):
        a_sorted = a[np.argsort(-p, kind="stable")]
    else:
        a_sorted = _exact_tie_order(p, a)
    return _ap_sorted(a_sorted, numpos, n)


def _num_positives(actuals):
    """sum(np.sum(actuals[:, i]) for i in classes) of cs/eval_util.py:73-77.  For 0/1 labels (what the data set has)
    every partial sum is an exact small integer, so one vectorised count gives the same number; anything else takes
    the reference's class-by-class route (4716 tiny reductions: 20 ms per batch)."""
    if actuals.dtype == np.bool_ or np.issubdtype(actuals.dtype, np.integer):
        return actuals.dtype.type(np.count_nonzero(actuals)) if actuals.dtype != np.bool_ else np.count_nonzero(actuals)
    flat = actuals.reshape(-1)
    if actuals.size < (1 << 24) and np.all((flat == 0) | (flat == 1)):
        return actuals.dtype.type(np.count_nonzero(flat))
    return sum(np.sum(actuals[:, i]) for i in range(actuals.shape[1]))


def calculate_gap(predictions, actuals, top_k=20):
    """cs/eval_util.py:61-79: global average precision over each video's top_k."""
    predictions, actuals = _np(predictions), _np(actuals)
    pv, lv, idx = top_k_by_video(predictions, actuals, top_k)
    order = np.argsort(idx.reshape(-1), kind="stable")       # the reference pools class-major
    num_pos = _num_positives(actuals)
    return average_precision(pv.reshape(-1)[order], lv.reshape(-1)[order], num_pos)


class EvaluationMetrics(object):
    """Streaming metrics over an epoch (cs/eval_util.py:126-213)."""

    def __init__(self, num_class, top_k):
        if not isinstance(num_class, int) or num_class <= 1:
            raise ValueError("num_class must be a positive integer.")
        self.num_class, self.top_k = num_class, top_k
        self.clear()

    def clear(self):
        self.sum_hit_at_one = 0.0
        self.sum_perr = 0.0
        self.sum_loss = 0.0
        self.num_examples = 0
        self._pool_p, self._pool_l, self._pool_c = [], [], []
        self._num_pos = 0
        self._class_pos = np.zeros(self.num_class, dtype=np.float64)

    def accumulate(self, predictions, labels, loss):
        predictions, labels = _np(predictions), _np(labels)
        loss = _np(loss)
        batch_size = labels.shape[0]
        mean_hit_at_one = calculate_hit_at_one(predictions, labels)
        mean_perr = calculate_precision_at_equal_recall_rate(predictions, labels)
        mean_loss = np.mean(loss)
        pv, lv, idx = top_k_by_video(predictions, labels, self.top_k)
        order = np.argsort(idx.reshape(-1), kind="stable")
        self._pool_p.append(pv.reshape(-1)[order])
        self._pool_l.append(lv.reshape(-1)[order])
        self._pool_c.append(idx.reshape(-1)[order])
        class_pos = labels.sum(axis=0)
        self._class_pos += class_pos
        self._num_pos += _num_positives(labels)
        self.num_examples += batch_size
        self.sum_hit_at_one += mean_hit_at_one * batch_size
        self.sum_perr += mean_perr * batch_size
        self.sum_loss += mean_loss * batch_size
        return {"hit_at_one": mean_hit_at_one, "perr": mean_perr, "loss": mean_loss}

    def get(self):
        if self.num_examples <= 0:
            raise ValueError("total_sample must be positive.")
        p = np.concatenate(self._pool_p)
        l = np.concatenate(self._pool_l)
        c = np.concatenate(self._pool_c)
        gap = average_precision(p, l, self._num_pos)
        aps = []
        order = np.argsort(c, kind="stable")       # per class, batches in arrival order
        cs, ps, ls = c[order], p[order], l[order]
        bounds = np.searchsorted(cs, np.arange(self.num_class + 1))
        for i in range(self.num_class):
            lo, hi = bounds[i], bounds[i + 1]
            aps.append(average_precision(ps[lo:hi], ls[lo:hi], self._class_pos[i]) if hi > lo else 0)
        return {"avg_hit_at_one": self.sum_hit_at_one / self.num_examples,
                "avg_perr": self.sum_perr / self.num_examples,
                "avg_loss": self.sum_loss / self.num_examples, "aps": aps, "gap": gap}
