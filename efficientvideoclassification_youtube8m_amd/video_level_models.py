"""Video-level classifiers (cs/video_level_models.py).  MoeModel is the head
used by every frame-level model on the hot path (--video_level_classifier_model
default, cs/frame_level_models.py:33); LogisticModel is the plain baseline.
The other eleven reference heads are Kaggle-ensemble leftovers no launcher
selects: out of scope."""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

from . import models, ops
from .engine import F32, MoeHead, TowerBase
from .flags import FLAGS


class MoeTower(TowerBase):
    """Stand-alone MoE head (parameters + kernels) for a [B, K] input."""

    l2_names = (MoeHead.GATES, MoeHead.EXPERTS)

    def __init__(self, batch_size, input_size, vocab_size, num_mixtures=2, device="cuda:0", training=True,
                 scope="model", seed=0):
        if input_size % 64:
            raise ValueError("MoeModel input width must be a multiple of 64 for the MFMA GEMM tiles")
        self.device, self.training, self.scope = torch.device(device), training, scope
        self.K, self.V, self.Mx = input_size, vocab_size, num_mixtures
        self._setup_store(MoeHead.shapes(input_size, vocab_size, num_mixtures))
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed)
        for k, shp in self.store.shapes.items():
            if len(shp) == 2:
                lim = math.sqrt(6.0 / (shp[0] + shp[1]))
                self.store.p(k).copy_((torch.rand(shp, generator=gen, dtype=F32) * 2 - 1) * lim)
        self.refresh_shadows()
        self.moe = MoeHead(self, input_size, vocab_size, num_mixtures)
        self.moe.alloc(batch_size, training)

    def forward(self, x):
        if x.shape[0] != self.moe.B:
            self.moe.alloc(x.shape[0], self.training)
        return self.moe.forward(x.contiguous())

    def backward(self, dpred):
        return self.moe.backward(dpred)


class MoeModel(models.BaseModel):
    """A softmax over a mixture of logistic models (with L2 regularization)."""

    def __init__(self):
        self.tower = None

    def create_model(self, model_input, vocab_size, num_mixtures=None, l2_penalty=1e-8, **unused_params):
        """cs/video_level_models.py:397-448.  model_input [B, K] f32 on the GPU.
        Returns {"predictions": [B, vocab_size]}."""
        num_mixtures = num_mixtures or FLAGS.moe_num_mixtures
        B, K = model_input.shape
        if self.tower is None or (self.tower.K, self.tower.V, self.tower.Mx) != (K, vocab_size, num_mixtures):
            self.tower = MoeTower(B, K, vocab_size, num_mixtures, device=model_input.device,
                                  scope=unused_params.get("scope", "model"), seed=unused_params.get("seed", 0))
        self.l2_penalty = l2_penalty
        return {"predictions": self.tower.forward(model_input)}


class LogisticModel(models.BaseModel):
    """Logistic model with L2 regularization (cs/video_level_models.py:375-392)."""

    def __init__(self):
        self.W = self.b = None

    def create_model(self, model_input, vocab_size, l2_penalty=1e-8, **unused_params):
        B, K = model_input.shape
        if K % 64:
            raise ValueError("LogisticModel input width must be a multiple of 64 for the MFMA GEMM tiles")
        if self.W is None:
            gen = torch.Generator(device="cpu")
            gen.manual_seed(unused_params.get("seed", 0))
            lim = math.sqrt(6.0 / (K + vocab_size))
            self.W = ((torch.rand((vocab_size, K), generator=gen) * 2 - 1) * lim).to(model_input.device)
            self.b = torch.zeros(vocab_size, dtype=F32, device=model_input.device)
        out = torch.empty((B, vocab_size), dtype=F32, device=model_input.device)
        ops.gemm_nt(ops.cast_bf16(model_input.contiguous()), ops.cast_bf16(self.W), B, vocab_size, K, out, bias=self.b)
        return {"predictions": ops.sigmoid_(out)}
