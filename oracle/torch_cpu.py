"""CPU baseline of the teacher+student training iteration: PyTorch-CPU, float32, MKL/oneDNN sgemm, autograd.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ (parity against the float64 numpy oracle
``model_math``) and by bench.py's ``cpu_baseline`` leg (the "reference CPU path" of BASELINE.md section 3 - TensorFlow
1.x can be neither installed nor run here, so the reference graph is restated on the same class of BLAS a TF-CPU
build would call).  Never imported by the product.

Structured like the graph the reference builds, NOT like the GPU engine:
  * one ``dynamic_rnn`` loop per L1 chunk at batch B (20 loops for the teacher, 5 for the student, weights shared) -
    cs/frame_level_models.py:243-250 / :312-321 - each step one ``concat([x, h]) @ kernel + bias`` (BasicLSTMCell,
    gate order i, j, f, o, forget_bias 1.0), per-row copy-through beyond ``sequence_length``;
  * L2 over the stacked final STATES [B, C, 2*L*H] (:252-257), MoE head (cs/video_level_models.py:397-448);
  * CE (cs/losses.py:90-97), L_REP twice + L_PRED + CE for the student (cs/train.py:359-362,398-406), the l2
    regulariser, reverse mode by autograd (what tf.gradients builds), per-tensor clip_by_norm + TF-Adam
    (cs/train.py:329-334,413-418).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

CE_EPS = 10e-6


def to_torch(params: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.tensor(np.asarray(v), dtype=torch.float32, requires_grad=True) for k, v in params.items()}


def _layers(p, scope, L):
    return [(p["%s/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % (scope, l)],
             p["%s/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % (scope, l)]) for l in range(L)]


def dynamic_rnn(x, lengths, layers):
    """tf.nn.dynamic_rnn(MultiRNNCell([BasicLSTMCell]*L, state_is_tuple=False), x [B,T,F], sequence_length) ->
    final state concat([c0, h0, c1, h1], 1)."""
    B, T, _ = x.shape
    H = layers[0][1].shape[0] // 4
    c = [x.new_zeros((B, H)) for _ in layers]
    h = [x.new_zeros((B, H)) for _ in layers]
    for t in range(T):
        active = (lengths > t).unsqueeze(1)
        if not bool(active.any()):
            break                                         # (dynamic_rnn's while loop runs to max(sequence_length))
        inp = x[:, t]
        for l, (kernel, bias) in enumerate(layers):
            z = torch.cat([inp, h[l]], 1) @ kernel + bias
            i, j, f, o = z.split(H, 1)
            c_new = c[l] * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
            h_new = torch.tanh(c_new) * torch.sigmoid(o)
            c[l] = torch.where(active, c_new, c[l])
            h[l] = torch.where(active, h_new, h[l])
            inp = h_new
    return torch.cat([s for l in range(len(layers)) for s in (c[l], h[l])], 1)


def hlstm_fwd(x, num_frames, p, num_chunks, num_layers=2, num_mixtures=2):
    """HierarchicalLstmModel.create_model / create_model_inference: (state [B, 2LH], predictions [B, V])."""
    B, T, F = x.shape
    Lc = T // num_chunks
    n = torch.as_tensor(np.asarray(num_frames), dtype=torch.int64)
    l1, l2 = _layers(p, "RNN_L1", num_layers), _layers(p, "RNN_L2", num_layers)
    outs = []
    for i in range(num_chunks):                           # one dynamic_rnn per chunk, weights shared (reuse=True)
        ln = torch.clamp(n - Lc * i, 0, Lc)
        outs.append(dynamic_rnn(x[:, i * Lc:(i + 1) * Lc], ln, l1))
    l2_in = torch.stack(outs, 1)
    len2 = torch.ceil(n.to(torch.float32) / float(Lc)).to(torch.int64)
    state = dynamic_rnn(l2_in, len2, l2)
    M = num_mixtures
    V = p["classifier/experts/biases"].shape[0] // M
    ga = (state @ p["classifier/gates/weights"]).reshape(B * V, M + 1)
    ea = (state @ p["classifier/experts/weights"] + p["classifier/experts/biases"]).reshape(B * V, M)
    g = torch.softmax(ga, 1)
    pred = (g[:, :M] * torch.sigmoid(ea)).sum(1).reshape(B, V)
    return state, pred


def cross_entropy(p, y):
    return (-(y * torch.log(p + CE_EPS) + (1 - y) * torch.log(1 - p + CE_EPS))).sum(1).mean()


def pred_kl(pt, ps):
    P = pt / pt.sum(1, keepdim=True)
    Q = ps / ps.sum(1, keepdim=True)
    return (P * (torch.log(P) - torch.log(Q))).sum()


def l2_reg(p):
    return 1e-8 * 0.5 * (p["classifier/gates/weights"].square().sum() + p["classifier/experts/weights"].square().sum())


class Adam:
    """tf.train.AdamOptimizer (epsilon outside the bias correction) behind slim's per-tensor clip_by_norm."""

    def __init__(self, params, lr=1e-3, clip_norm=1.0):
        self.p, self.lr, self.clip, self.t = params, lr, clip_norm, 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    @torch.no_grad()
    def step(self, grads, beta1=0.9, beta2=0.999, eps=1e-8):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - beta2 ** self.t) / (1 - beta1 ** self.t)
        for k, p in self.p.items():
            g = grads[k]
            nrm = float(g.norm())
            if self.clip > 0:
                g = g * (self.clip / max(nrm, self.clip))
            self.m[k].mul_(beta1).add_(g, alpha=1 - beta1)
            self.v[k].mul_(beta2).addcmul_(g, g, value=1 - beta2)
            p.sub_(lr_t * self.m[k] / (self.v[k].sqrt() + eps))


def l2_normalize(x):
    return x * torch.rsqrt(torch.clamp(x.square().sum(2, keepdim=True), min=1e-12))


def teacher_student_iteration(x_raw, num_frames, labels, teacher, student, every_n, opt_t=None, opt_s=None,
                              regularization_penalty=2.0, mode="teacher_student"):
    """One sess.run([train_op, train_student_op, ...]) of cs/train.py:516-517 (or the teacher / student alone).
    x_raw [B,300,F] float32 tensor, labels [B,V] float32 tensor.  Returns a dict of python floats + the gradient
    dicts (before clipping); applies the updates when optimizers are given."""
    out = {}
    x = l2_normalize(x_raw)
    y = labels
    t_state = t_pred = None
    if mode != "student":
        t_state, t_pred = hlstm_fwd(x, num_frames, teacher, 20)
        out["label_loss"] = cross_entropy(t_pred, y)
        final = regularization_penalty * l2_reg(teacher) + out["label_loss"]
        names = list(teacher)
        tg = dict(zip(names, torch.autograd.grad(final, [teacher[k] for k in names])))
        out["teacher_grads"] = tg
    if mode != "teacher":
        S = 300 // every_n
        n_s = np.trunc(np.asarray(num_frames).astype(np.float64) / 300.0 * float(S)).astype(np.int64)
        xs = x[:, ::every_n][:, :S]
        s_state, s_pred = hlstm_fwd(xs, n_s, student, 5)
        out["student_label_loss"] = cross_entropy(s_pred, y)
        total = out["student_label_loss"] + regularization_penalty * l2_reg(student)
        if t_state is not None:
            out["student_loss_state"] = (t_state.detach() - s_state).square().sum(1).mean()
            out["pred_loss"] = pred_kl(t_pred.detach(), s_pred)
            total = total + 2.0 * out["student_loss_state"] + out["pred_loss"]
        names = list(student)
        sg = dict(zip(names, torch.autograd.grad(total, [student[k] for k in names])))
        out["student_grads"] = sg
    if opt_t is not None and "teacher_grads" in out:
        opt_t.step(out["teacher_grads"])
    if opt_s is not None and "student_grads" in out:
        opt_s.step(out["student_grads"])
    for k in ("label_loss", "student_label_loss", "student_loss_state", "pred_loss"):
        if k in out:
            out[k] = float(out[k].detach())
    return out
