"""Frame-level models (cs/frame_level_models.py) behind the reference's
``create_model`` plug-in interface, executing on the HIP kernels.

``model_input`` is a device tensor [batch, max_frames, num_features] (float32,
already l2-normalised by the caller as in cs/train.py:256) - or, on the fused
fast path used by ``train.build_graph``, a ``FrameBatch`` produced by the
l2norm/sub-sample/cast kernel.  Flags are read from ``flags.FLAGS`` exactly
where the reference reads them.  Variables live in a tower object per variable
scope (``scope=`` kwarg; the reference uses tf.variable_scope at the call
site: "model" / "model_student", cs/train.py:281,349).
"""
from __future__ import annotations

import torch

from . import models, ops, video_level_models
from .engine import HLstmTower
from .flags import FLAGS
from .towers import DbofGenericTower, DbofTower, LogisticTower, NetVladTower


class FrameBatch(object):
    """Output of the fused input kernel: bf16, l2-normalised, time-major chunked
    views of one batch for the teacher and (optionally) the student."""

    def __init__(self, teacher_view, student_view, batch, frames, features):
        self.teacher_view, self.student_view = teacher_view, student_view
        self.batch, self.frames, self.features = batch, frames, features


def _vl_check():
    if FLAGS.video_level_classifier_model != "MoeModel":
        getattr(video_level_models, FLAGS.video_level_classifier_model)   # AttributeError for unknown names, as getattr does
        raise NotImplementedError("only MoeModel is fused into the frame-level towers (SURVEY.md 8a row a5)")


class FrameLevelLogisticModel(models.BaseModel):
    def __init__(self):
        self.towers = {}

    def create_model(self, model_input, vocab_size, num_frames, **unused_params):
        """cs/frame_level_models.py:52-83: logistic classifier over the average of
        the frame features (sum over all padded rows / true frame count)."""
        scope = unused_params.get("scope", "model")
        B, T, F = model_input.shape
        tw = self.towers.get(scope)
        if tw is None:
            tw = self.towers[scope] = LogisticTower(B, T, F, vocab_size, device=model_input.device, scope=scope,
                                                    training=unused_params.get("is_training", True),
                                                    seed=unused_params.get("seed", 0))
        nf = num_frames.reshape(-1).to(torch.int32)
        return {"predictions": tw.forward(model_input.contiguous(), nf,
                                          normalize=unused_params.get("normalize_input", False))}


class DbofModel(models.BaseModel):
    def __init__(self):
        self.towers = {}

    def create_model(self, model_input, vocab_size, num_frames, iterations=None, add_batch_norm=None,
                     sample_random_frames=None, cluster_size=None, hidden_size=None, is_training=True,
                     **unused_params):
        """cs/frame_level_models.py:108-195 (Deep Bag of Frames)."""
        iterations = iterations or FLAGS.iterations
        add_batch_norm = add_batch_norm or FLAGS.dbof_add_batch_norm      # `x or FLAG`: cannot be switched off (:120)
        random_frames = sample_random_frames or FLAGS.sample_random_frames
        cluster_size = cluster_size or FLAGS.dbof_cluster_size
        hidden1_size = hidden_size or FLAGS.dbof_hidden_size
        method = FLAGS.dbof_pooling_method
        if method not in ("max", "average", "none"):
            raise ValueError("Unrecognized pooling method: %s" % method)                 # cs/model_utils.py:83
        if method == "none":
            raise NotImplementedError("dbof_pooling_method=none: FramePooling returns [batch*iterations, clusters] (cs/model_utils.py:79-81), so "
                                      "the predictions have batch*iterations rows against `batch` label rows - the reference's graph does not train")
        _vl_check()
        scope = unused_params.get("scope", "model")
        B, T, F = model_input.shape
        tw = self.towers.get(scope)
        default = bool(add_batch_norm) and bool(random_frames) and method == "max"
        if tw is None and default:
            tw = self.towers[scope] = DbofTower(B, T, F, vocab_size, iterations, cluster_size, hidden1_size,
                                                FLAGS.moe_num_mixtures, device=model_input.device, training=True,
                                                scope=scope, seed=unused_params.get("seed", 0))
        elif tw is None:     # a flag combination no launcher of the reference uses: the plain chain (towers.DbofGenericTower)
            tw = self.towers[scope] = DbofGenericTower(B, T, F, vocab_size, iterations, cluster_size, hidden1_size,
                                                       FLAGS.moe_num_mixtures, device=model_input.device, training=True,
                                                       scope=scope, seed=unused_params.get("seed", 0), pooling=method,
                                                       add_batch_norm=bool(add_batch_norm), random_frames=bool(random_frames))
        nf = num_frames.reshape(-1).to(torch.int32)
        u = unused_params.get("uniform")
        if u is None:
            u = torch.rand((B, iterations if random_frames else 1), dtype=torch.float32, device=model_input.device)
        return {"predictions": tw.forward(model_input.contiguous(), nf, u,
                                          normalize=unused_params.get("normalize_input", False), is_training=is_training)}


class HierarchicalLstmModel(models.BaseModel):
    def __init__(self):
        self.towers = {}

    def _tower(self, scope, B, T, C, F, vocab_size, device, training, seed):
        tw = self.towers.get(scope)
        if tw is None:
            tw = self.towers[scope] = HLstmTower(B, T, C, F, vocab_size, FLAGS.lstm_cells, FLAGS.lstm_layers,
                                                 FLAGS.moe_num_mixtures, device, training, scope, seed)
        return tw

    def create_model(self, model_input, vocab_size, num_frames, **unused_params):
        """cs/frame_level_models.py:200-267: num_inputs_to_lstm weight-shared L1
        LSTM chunks of max_num_frames/num_inputs_to_lstm frames, L2 LSTM over the L1
        final *states*, MoE classifier.  Returns (state, {"predictions": ...})."""
        _vl_check()
        scope = unused_params.get("scope", "model")
        C = FLAGS.num_inputs_to_lstm
        if isinstance(model_input, FrameBatch):
            view, B, T, F = model_input.teacher_view, model_input.batch, model_input.frames, model_input.features
        else:
            B, T, F = model_input.shape
            if T % C:
                raise ValueError("Dimension size must be evenly divisible by %d but is %d (tf.split)" % (C, T))
            view, _ = ops.l2norm_chunk(model_input.contiguous(), C, normalize=False)
        tw = self._tower(scope, B, T, C, F, vocab_size, view.device, unused_params.get("is_training", True),
                         unused_params.get("seed", 0))
        nf = num_frames.reshape(-1).to(torch.int32)
        _, l1, l2 = ops.frame_counts(nf, 1, C, T // C, T)
        state, pred = tw.forward(view, l1, l2)
        return state, {"predictions": pred}

    def create_model_inference(self, model_input, vocab_size, every_n, num_inputs_L1, num_frames, **unused_params):
        """cs/frame_level_models.py:269-338: the student view - model_input holds the
        max_num_frames/every_n retained frames, num_frames the (int64) student frame count."""
        _vl_check()
        scope = unused_params.get("scope", "model_student")
        C = num_inputs_L1
        if isinstance(model_input, FrameBatch):
            view, B, F = model_input.student_view, model_input.batch, model_input.features
            S = view.shape[0] * C
        else:
            B, S, F = model_input.shape
            if S != FLAGS.max_num_frames // every_n or S % C:
                raise ValueError("student input has %d frames; expected max_num_frames/every_n = %d split into %d chunks"
                                 % (S, FLAGS.max_num_frames // every_n, C))
            view, _ = ops.l2norm_chunk(model_input.contiguous(), C, normalize=False)
        tw = self._tower(scope, B, S, C, F, vocab_size, view.device, unused_params.get("is_training", True),
                         unused_params.get("seed", 1))
        nf = num_frames.reshape(-1).to(torch.int32)
        _, l1, l2 = ops.frame_counts(nf, 1, C, S // C, S)
        state, pred = tw.forward(view, l1, l2)
        return state, {"predictions": pred}


class NetVLADModel(models.BaseModel):
    """The reference's NetVLADModel is an empty stub (cs/frame_level_models.py:341-347: both methods `return`).  Here
    `create_model` builds the NetVLAD aggregation tower (towers.NetVladTower) - an EXTENSION with its own oracle
    (oracle/model_math.py::netvlad_fwd), there being no reference math to match; `create_model_inference` stays the
    reference's stub (the student path of the distillation graph exists for HierarchicalLstmModel only)."""

    def __init__(self):
        self.towers = {}

    def create_model(self, model_input, vocab_size, num_frames, iterations=None, cluster_size=None, hidden_size=None,
                     is_training=True, **unused_params):
        iterations = iterations or FLAGS.iterations
        cluster_size = cluster_size or FLAGS.netvlad_cluster_size
        hidden_size = hidden_size or FLAGS.netvlad_hidden_size
        _vl_check()
        scope = unused_params.get("scope", "model")
        B, T, F = model_input.shape
        tw = self.towers.get(scope)
        if tw is None:
            tw = self.towers[scope] = NetVladTower(B, T, F, vocab_size, iterations, cluster_size, hidden_size, FLAGS.moe_num_mixtures,
                                                   device=model_input.device, training=True, scope=scope,
                                                   seed=unused_params.get("seed", 0))
        nf = num_frames.reshape(-1).to(torch.int32)
        u = unused_params.get("uniform")
        if u is None:
            u = torch.rand((B, iterations), dtype=torch.float32, device=model_input.device)
        return {"predictions": tw.forward(model_input.contiguous(), nf, u, normalize=unused_params.get("normalize_input", False),
                                          is_training=is_training)}

    def create_model_inference(self, model_input, vocab_size, every_n, num_frames, **unused_params):
        return


class NeXtVLADModel(models.BaseModel):
    """Empty stub in the reference (cs/frame_level_models.py:349-355): returns None."""

    def create_model(self, model_input, vocab_size, num_frames, **unused_params):
        return

    def create_model_inference(self, model_input, vocab_size, every_n, num_frames, **unused_params):
        return
