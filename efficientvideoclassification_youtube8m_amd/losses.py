"""Label losses (cs/losses.py).  Only CrossEntropyLoss is on the hot path
(default --label_loss, cs/train.py:67); the other eight reference losses are
never selected by any launcher and are out of scope."""
from __future__ import annotations

import torch

from . import ops


class BaseLoss(object):
    """cs/losses.py:8-25."""

    def calculate_loss(self, unused_predictions, unused_labels, **unused_params):
        raise NotImplementedError()


class CrossEntropyLoss(BaseLoss):
    """mean_b sum_c -(y log(p+1e-5) + (1-y) log(1-p+1e-5))   (cs/losses.py:90-97).

    Returns a 0-d device tensor.  ``grad_out`` (optional [B,V] f32) receives
    dLoss/dpredictions in the same pass (the training graph feeds it straight to
    the model's backward instead of building an autograd tape)."""

    def calculate_loss(self, predictions, labels, grad_out=None, **unused_params):
        B, V = predictions.shape
        lab = labels if labels.dtype == torch.uint8 else labels.to(torch.uint8)
        loss = torch.zeros(1, dtype=torch.float32, device=predictions.device)
        ops.ce_loss(predictions, lab.contiguous(), loss, grad_out, grad_scale=1.0 / B)
        return loss[0]
