"""DbofModel and FrameLevelLogisticModel towers over the C-ABI kernels.

Reference: cs/frame_level_models.py:85-195 (DBoF, add_batch_norm=True,
sample_random_frames=True, pooling 'max', MoE head), cs/model_utils.py:39-83,
and cs/frame_level_models.py:50-83 (logistic over the mean frame).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

from . import ops
from .engine import BF16, F32, MoeHead, TowerBase


def _maybe_allreduce(t, group):
    if torch.distributed.is_available() and torch.distributed.is_initialized() and \
            torch.distributed.get_world_size(group) > 1:
        torch.distributed.all_reduce(t, group=group)
        return torch.distributed.get_world_size(group)
    return 1


class BatchNorm:
    """slim.batch_norm(center=True, scale=True): training uses biased batch
    moments (eps 1e-3) and updates the moving averages (decay 0.999); eval uses
    the moving averages.  Under data parallelism the f64 partial sums are
    all-reduced (SyncBN) so statistics equal the single-device global batch."""

    def __init__(self, tower, scope, C):
        self.tw, self.scope, self.C = tower, scope, C
        dev = tower.device
        self.mean = torch.zeros(C, dtype=F32, device=dev)
        self.var = torch.ones(C, dtype=F32, device=dev)
        self.ws = torch.zeros(2 * C, dtype=torch.float64, device=dev)
        tower.buffers[scope + "/moving_mean"] = torch.zeros(C, dtype=F32, device=dev)
        tower.buffers[scope + "/moving_variance"] = torch.ones(C, dtype=F32, device=dev)

    @staticmethod
    def shapes(scope, C):
        return OrderedDict([(scope + "/beta", (C,)), (scope + "/gamma", (C,))])

    def gamma(self):
        return self.tw.store.p(self.scope + "/gamma")

    def beta(self):
        return self.tw.store.p(self.scope + "/beta")

    def stats(self, x, R, is_training):
        """Sets self.mean/var for this batch (training) or from the moving averages."""
        tw = self.tw
        if not is_training:
            self.mean.copy_(tw.buffers[self.scope + "/moving_mean"])
            self.var.copy_(tw.buffers[self.scope + "/moving_variance"])
            self.R_total = R
            return
        ops.bn_stats_partial(x, R, self.C, self.ws)
        world = _maybe_allreduce(self.ws, tw.pg)
        self.R_total = R * world
        ops.bn_stats_finalize(self.ws, self.R_total, self.C, self.mean, self.var)
        ops.ema_update(tw.buffers[self.scope + "/moving_mean"], self.mean)
        ops.ema_update(tw.buffers[self.scope + "/moving_variance"], self.var)

    def backward(self, x, dy, R, relu6, argmax=None, S=1, dx_f32=None, dx_bf16=None):
        tw = self.tw
        ops.bn_bwd_partial(x, dy, R, self.C, self.mean, self.var, self.gamma(), self.beta(), relu6, self.ws, argmax, S)
        _maybe_allreduce(self.ws, tw.pg)
        ops.bn_bwd_finalize(x, dy, R, self.R_total, self.C, self.mean, self.var, self.gamma(), self.beta(), relu6,
                            self.ws, argmax, S, dx_f32, dx_bf16,
                            tw.store.g(self.scope + "/gamma"), tw.store.g(self.scope + "/beta"))


class DbofTower(TowerBase):
    """Deep Bag of Frames: sample S frames -> input_bn -> .Wc -> cluster_bn ->
    relu6 -> max over frames -> .Wh -> hidden1_bn -> relu6 -> MoE."""

    CW, HW = "cluster_weights", "hidden1_weights"        # the reference's unnamed tf.Variable / Variable_1
    l2_names = (MoeHead.GATES, MoeHead.EXPERTS)
    # gradients that BatchNorm.backward already leaves summed over the ranks (SyncBN): not part of the gradient all-reduce
    global_grad_names = tuple("%s/%s" % (s, v) for s in ("input_bn", "cluster_bn", "hidden1_bn") for v in ("beta", "gamma"))

    def __init__(self, batch_size, max_frames=300, feature_size=1152, vocab_size=4716, iterations=30,
                 cluster_size=8192, hidden_size=1024, num_mixtures=2, device="cuda:0", training=True,
                 scope="model", seed=0, process_group=None):
        self.device, self.training, self.scope, self.pg = torch.device(device), training, scope, process_group
        self.T, self.F, self.V, self.S = max_frames, feature_size, vocab_size, iterations
        self.Cc, self.Hd, self.Mx = cluster_size, hidden_size, num_mixtures
        if feature_size % 64 or cluster_size % 64 or hidden_size % 64:
            raise ValueError("feature/cluster/hidden sizes must be multiples of 64 for the MFMA GEMM tiles")
        shapes = OrderedDict()
        shapes.update(BatchNorm.shapes("input_bn", feature_size))
        shapes[self.CW] = (cluster_size, feature_size)               # stored transposed [C][F]
        shapes.update(BatchNorm.shapes("cluster_bn", cluster_size))
        shapes[self.HW] = (hidden_size, cluster_size)                # stored transposed [Hd][C]
        shapes.update(BatchNorm.shapes("hidden1_bn", hidden_size))
        shapes.update(MoeHead.shapes(hidden_size, vocab_size, num_mixtures))
        self.buffers = OrderedDict()
        self._setup_store(shapes)
        self.bn_in = BatchNorm(self, "input_bn", feature_size)
        self.bn_cl = BatchNorm(self, "cluster_bn", cluster_size)
        self.bn_h = BatchNorm(self, "hidden1_bn", hidden_size)
        self.moe = MoeHead(self, hidden_size, vocab_size, num_mixtures)
        self._init_params(seed)
        self._alloc(batch_size)

    def _init_params(self, seed):
        """cs/frame_level_models.py:145-147,169-171: random_normal(stddev=1/sqrt(fan_in));
        BN gamma=1, beta=0; MoE glorot-uniform / zero bias."""
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed)
        for k, shp in self.store.shapes.items():
            p = self.store.p(k)
            if k in (self.CW, self.HW):
                p.copy_(torch.randn(shp, generator=gen, dtype=F32) / math.sqrt(shp[1]))
            elif len(shp) == 2:
                lim = math.sqrt(6.0 / (shp[0] + shp[1]))
                p.copy_((torch.rand(shp, generator=gen, dtype=F32) * 2 - 1) * lim)
            elif k.endswith("/gamma"):
                p.fill_(1.0)
        self.refresh_shadows()

    def _alloc(self, B):
        dev, F, S, Cc, Hd = self.device, self.F, self.S, self.Cc, self.Hd
        self.B = B
        R = B * S
        self.R = R
        self.r = torch.empty((R, F), dtype=F32, device=dev)
        self.idx = torch.empty((B, S), dtype=torch.int32, device=dev)
        self.r_bn = torch.empty((R, F), dtype=BF16, device=dev)
        self.act = torch.empty((R, Cc), dtype=F32, device=dev)
        self.pooled = torch.empty((B, Cc), dtype=F32, device=dev)
        self.pooled_bf = torch.empty((B, Cc), dtype=BF16, device=dev)
        self.argmax = torch.empty((B, Cc), dtype=torch.int32, device=dev)
        self.hid = torch.empty((B, Hd), dtype=F32, device=dev)
        self.h6 = torch.empty((B, Hd), dtype=F32, device=dev)
        self.moe.alloc(B, self.training)
        if self.training:
            self.Bp, self.Rp = ops.round_up(B, 64), ops.round_up(R, 64)
            self.dhid_bf = torch.empty((B, Hd), dtype=BF16, device=dev)
            self.dhidT = torch.empty((Hd, self.Bp), dtype=BF16, device=dev)
            self.pooledT = torch.empty((Cc, self.Bp), dtype=BF16, device=dev)
            self.dpooled = torch.empty((B, Cc), dtype=F32, device=dev)
            self.dact_bf = torch.empty((R, Cc), dtype=BF16, device=dev)
            self.dactT = torch.empty((Cc, self.Rp), dtype=BF16, device=dev)
            self.r_bnT = torch.empty((F, self.Rp), dtype=BF16, device=dev)
            self.dr_bn = torch.empty((R, F), dtype=F32, device=dev)

    def forward(self, x, num_frames, uniform, normalize=True, is_training=True):
        """x [B,T,F] f32 raw (normalize=True fuses tf.nn.l2_normalize of the sampled
        frames) ; uniform [B,S] f32 in [0,1): the tf.random_uniform draw of
        SampleRandomFrames, supplied by the caller so runs are reproducible."""
        B = x.shape[0]
        if B != self.B:
            self._alloc(B)
        R, F, S, Cc, Hd = self.R, self.F, self.S, self.Cc, self.Hd
        st = self.store
        ops.sample_frames_gather(x, uniform, num_frames, self.r, self.idx, normalize=normalize)
        self.bn_in.stats(self.r, R, is_training)
        high = self.precision == "high"
        if high:
            if not hasattr(self, "r_bn_f32") or self.r_bn_f32.shape[0] != R:
                self.r_bn_f32 = torch.empty((R, F), dtype=F32, device=self.device)
                self.r_bn_lo = torch.empty((R, F), dtype=BF16, device=self.device)
                self.pooled_lo = torch.empty((B, Cc), dtype=BF16, device=self.device)
            ops.bn_apply(self.r, R, F, self.bn_in.mean, self.bn_in.var, self.bn_in.gamma(), self.bn_in.beta(), False,
                         y_f32=self.r_bn_f32)
            ops.cast_bf16_split(self.r_bn_f32, self.r_bn, self.r_bn_lo)
            ops.gemm_nt_split(self.r_bn, self.r_bn_lo, self.shadow_fwd[self.CW], self.shadow_lo[self.CW], R, Cc, F, self.act)
        else:
            ops.bn_apply(self.r, R, F, self.bn_in.mean, self.bn_in.var, self.bn_in.gamma(), self.bn_in.beta(), False,
                         y_bf16=self.r_bn)
            ops.gemm_nt(self.r_bn, self.shadow_fwd[self.CW], R, Cc, F, self.act)
        self.bn_cl.stats(self.act, R, is_training)
        ops.bn_relu6_framepool_fwd(self.act, B, S, Cc, self.bn_cl.mean, self.bn_cl.var, self.bn_cl.gamma(),
                                   self.bn_cl.beta(), self.pooled, self.pooled_bf, self.argmax)
        if high:
            ops.cast_bf16_split(self.pooled, self.pooled_bf, self.pooled_lo)
            ops.gemm_nt_split(self.pooled_bf, self.pooled_lo, self.shadow_fwd[self.HW], self.shadow_lo[self.HW], B, Hd, Cc, self.hid)
        else:
            ops.gemm_nt(self.pooled_bf, self.shadow_fwd[self.HW], B, Hd, Cc, self.hid)
        self.bn_h.stats(self.hid, B, is_training)
        ops.bn_apply(self.hid, B, Hd, self.bn_h.mean, self.bn_h.var, self.bn_h.gamma(), self.bn_h.beta(), True,
                     y_f32=self.h6)
        return self.moe.forward(self.h6)

    def backward(self, dpred, on_moe_grads_ready=None):
        assert self.training
        B, R, F, S, Cc, Hd = self.B, self.R, self.F, self.S, self.Cc, self.Hd
        st = self.store
        dh6 = self.moe.backward(dpred)
        if on_moe_grads_ready is not None:
            on_moe_grads_ready()
        self.bn_h.backward(self.hid, dh6, B, True, dx_bf16=self.dhid_bf)
        # hidden1 weights: dWh^T [Hd][C] = dhid^T . pooled ; dpooled = dhid . Wh^T
        ops.transpose_to_bf16(self.dhid_bf, B, Hd, self.dhidT, self.Bp)
        ops.transpose_to_bf16(self.pooled_bf, B, Cc, self.pooledT, self.Bp)
        ops.gemm_nt(self.dhidT, self.pooledT, Hd, Cc, self.Bp, st.g(self.HW))
        ops.gemm_nt(self.dhid_bf, self.shadow_bwd[self.HW], B, Cc, Hd, self.dpooled)
        # max-pool routing + relu6 mask + cluster_bn backward in one pass over act
        self.bn_cl.backward(self.act, self.dpooled, R, True, argmax=self.argmax, S=S, dx_bf16=self.dact_bf)
        ops.transpose_to_bf16(self.dact_bf, R, Cc, self.dactT, self.Rp)
        ops.transpose_to_bf16(self.r_bn, R, F, self.r_bnT, self.Rp)
        ops.gemm_nt(self.dactT, self.r_bnT, Cc, F, self.Rp, st.g(self.CW))
        ops.gemm_nt(self.dact_bf, self.shadow_bwd[self.CW], R, F, Cc, self.dr_bn)
        self.bn_in.backward(self.r, self.dr_bn, R, False)

    @property
    def pred(self):
        return self.moe.pred


class LogisticTower(TowerBase):
    """FrameLevelLogisticModel: sigmoid(mean_frames(x) . W + b), with the
    reference's quirk that the sum runs over all (zero-padded) frames and is
    divided by the true frame count (cs/frame_level_models.py:72-78)."""

    W, Bn = "fully_connected/weights", "fully_connected/biases"
    l2_names = ("fully_connected/weights",)                      # weights_regularizer=slim.l2_regularizer(1e-8)

    def __init__(self, batch_size, max_frames=300, feature_size=1152, vocab_size=4716, device="cuda:0",
                 training=True, scope="model", seed=0):
        self.device, self.training, self.scope = torch.device(device), training, scope
        self.T, self.F, self.V = max_frames, feature_size, vocab_size
        if feature_size % 64:
            raise ValueError("feature_size must be a multiple of 64 for the MFMA GEMM tiles")
        shapes = OrderedDict([(self.W, (vocab_size, feature_size)), (self.Bn, (vocab_size,))])
        self._setup_store(shapes)
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed)
        lim = math.sqrt(6.0 / (vocab_size + feature_size))
        self.store.p(self.W).copy_((torch.rand((vocab_size, feature_size), generator=gen, dtype=F32) * 2 - 1) * lim)
        self.refresh_shadows()
        self._alloc(batch_size)

    def _alloc(self, B):
        dev, F, V = self.device, self.F, self.V
        self.B = B
        self.avg = torch.empty((B, F), dtype=F32, device=dev)
        self.avg_bf = torch.empty((B, F), dtype=BF16, device=dev)
        self.pred = torch.empty((B, V), dtype=F32, device=dev)
        if self.training:
            self.Bp = ops.round_up(B, 64)
            self.dz = torch.empty((B, V), dtype=BF16, device=dev)
            self.dzT = torch.empty((V, self.Bp), dtype=BF16, device=dev)
            self.avgT = torch.empty((F, self.Bp), dtype=BF16, device=dev)

    def forward(self, x, num_frames, normalize=True):
        B = x.shape[0]
        if B != self.B:
            self._alloc(B)
        ops.meanpool(x, num_frames, self.avg, self.avg_bf, normalize=normalize)
        if self.precision == "high":
            if not hasattr(self, "avg_lo") or self.avg_lo.shape != self.avg_bf.shape:
                self.avg_lo = torch.empty_like(self.avg_bf)
            ops.cast_bf16_split(self.avg, self.avg_bf, self.avg_lo)
            ops.gemm_nt_split(self.avg_bf, self.avg_lo, self.shadow_fwd[self.W], self.shadow_lo[self.W], B, self.V, self.F,
                              self.pred, bias=self.store.p(self.Bn))
        else:
            ops.gemm_nt(self.avg_bf, self.shadow_fwd[self.W], B, self.V, self.F, self.pred, bias=self.store.p(self.Bn))
        ops.sigmoid_(self.pred)
        return self.pred

    def backward(self, dpred, on_moe_grads_ready=None):
        B, V, F = self.B, self.V, self.F
        ops.sigmoid_bwd(self.pred, dpred, self.dz)
        ops.transpose_to_bf16(self.dz, B, V, self.dzT, self.Bp)
        ops.transpose_to_bf16(self.avg_bf, B, F, self.avgT, self.Bp)
        ops.gemm_nt(self.dzT, self.avgT, V, F, self.Bp, self.store.g(self.W))
        ops.rowsum_bf16(self.dzT, V, self.Bp, self.store.g(self.Bn))
        if on_moe_grads_ready is not None:
            on_moe_grads_ready()
