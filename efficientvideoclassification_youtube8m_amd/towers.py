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


_reducers = {}


def _maybe_allreduce(t, group):
    """SUM of the f64 batch-norm partial sums over the ranks (SyncBN), in place; returns the world size.  Goes through a
    distill.GradReducer of the group like every other collective of a step: in stream order on the current stream by default,
    funnelled through the process-wide serial stream under EVC_DP_SERIAL_COMM=1 (so that mode really has ONE collective in
    flight at a time, SyncBN included), and counted in GradReducer.stats."""
    if torch.distributed.is_available() and torch.distributed.is_initialized() and \
            torch.distributed.get_world_size(group) > 1:
        from .distill import GradReducer                     # (distill imports this module's towers lazily too)
        red = _reducers.get(group)               # keyed by the group object itself (an id() can be reused after a group is destroyed)
        if red is None or red.pg is not group:
            red = _reducers[group] = GradReducer(group)
        red.all_reduce_small(t)
        return red.world
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
    """Deep Bag of Frames: sample S frames -> input_bn -> .Wc -> cluster_bn -> relu6 -> max over frames -> .Wh ->
    hidden1_bn -> relu6 -> MoE (cs/frame_level_models.py:108-195).

    The [B*S, clusters] activation (503 MB of f32 at B=512) never exists in f32: the cluster GEMM's epilogue
    (evc_dbof_cluster_pool_fwd) leaves the batch-norm column sums and, per (video, cluster), the frame that the
    max-pool will select (max of sign(gamma)*x: batch-norm with a fixed-sign scale and relu6 are monotone); training
    keeps the activation as bf16 for the backward pass, which rewrites it in place as d(activation).
    Backward: the gradient wrt the batch-normalised input (a second 290 GFLOP product in the reference graph, whose
    only consumers are input_bn's gamma/beta) is not formed: with G = dact^T . xhat,  dWc = gamma_in * G,
    dgamma_in = sum_c Wc * G,  dbeta_in = 0 (evc_dbof_wgrad_finish)."""

    CW, HW = "cluster_weights", "hidden1_weights"        # the reference's unnamed tf.Variable / Variable_1
    l2_names = (MoeHead.GATES, MoeHead.EXPERTS)
    # gradients that the batch-norm backward passes already leave summed over the ranks (SyncBN: all-reduced f64 sums): not
    # part of the gradient all-reduce.  (input_bn's come from the rank's own G = dact^T . xhat and ARE all-reduced.)
    global_grad_names = tuple("%s/%s" % (s, v) for s in ("cluster_bn", "hidden1_bn") for v in ("beta", "gamma"))
    timing = None      # bench.py: set to a list to collect (start, end) events around the cluster GEMM launches

    def __init__(self, batch_size, max_frames=300, feature_size=1152, vocab_size=4716, iterations=30,
                 cluster_size=8192, hidden_size=1024, num_mixtures=2, device="cuda:0", training=True,
                 scope="model", seed=0, process_group=None):
        self.device, self.training, self.scope, self.pg = torch.device(device), training, scope, process_group
        self.T, self.F, self.V, self.S = max_frames, feature_size, vocab_size, iterations
        self.Cc, self.Hd, self.Mx = cluster_size, hidden_size, num_mixtures
        if feature_size % 64 or cluster_size % 64 or hidden_size % 64:
            raise ValueError("feature/cluster/hidden sizes must be multiples of 64 for the MFMA GEMM tiles")
        if iterations > 32:
            raise ValueError("iterations=%d: the fused cluster kernel holds at most 32 sampled frames per video" % iterations)
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
        self.R = B * S                                               # sampled frames of this rank's batch
        self.Mp, self.P_in, self.P_cl = ops.dbof_workspace(B, S)     # rows of the padded frame layout / partial-sum rows
        Mp = self.Mp
        self.r = torch.empty((Mp, F), dtype=F32, device=dev)         # sampled, l2-normalised frames (only live slots are written / read)
        self.idx = torch.empty((B, S), dtype=torch.int32, device=dev)
        self.part_in = torch.empty((self.P_in, 2, F), dtype=F32, device=dev)
        self.r_bn = torch.empty((Mp, F), dtype=BF16, device=dev)
        self.part_cl = torch.empty((self.P_cl, 2, Cc), dtype=F32, device=dev)
        self.xsel = torch.empty((B, Cc), dtype=F32, device=dev)
        self.arg = torch.empty((B, Cc), dtype=torch.uint8, device=dev)
        self.Bk = ops.round_up(B, 32)                                # batch rows as a TN contraction length (zero rows pad)
        self.pooled = torch.empty((B, Cc), dtype=F32, device=dev)
        self.pooled_bf = torch.zeros((self.Bk, Cc), dtype=BF16, device=dev)
        self.hid = torch.empty((B, Hd), dtype=F32, device=dev)
        self.h6 = torch.empty((B, Hd), dtype=F32, device=dev)
        self.moe.alloc(B, self.training)
        if self.training:
            self.xhat = torch.empty((Mp, F), dtype=BF16, device=dev)
            self.act = torch.empty((Mp, Cc), dtype=BF16, device=dev)  # bf16 activation, rewritten in place as d(activation)
            self.dhid_bf = torch.zeros((self.Bk, Hd), dtype=BF16, device=dev)
            self.dpooled = torch.empty((B, Cc), dtype=F32, device=dev)
            self.nslab = self._pick_nslab(Cc, F, Mp)
            self.slabs = torch.empty((self.nslab, Cc, F), dtype=F32, device=dev)
            self.wgrad_ws = torch.empty(((Cc + 7) // 8, F), dtype=F32, device=dev)

    # "high" precision on f16 + e4m3 operands (both operands' roundings corrected behind the f16 stages of the same launches: ops.gemm_nt_f16_fp8,
    # ops.dbof_cluster_pool_fwd_f16fp8) when every contraction length is a multiple of 128; EVC_HIGH_FP8_LO=0 or other sizes: the
    # separate hi / lo shadows and three split-bf16 products per contraction (also what precision "split" means here).
    def _alloc_high_shadows(self):
        import os
        dims_ok = self.F % 128 == 0 and self.F >= 256 and self.Cc % 128 == 0 and self.Cc >= 512 and self.Hd % 128 == 0 and self.Hd >= 512
        if self.precision == "high" and dims_ok and os.environ.get("EVC_HIGH_FP8_LO", "1") != "0":
            self.shadow_lo, self.shadow_w16, self.shadow_w8 = {}, {}, {}
            for k in (self.CW, self.HW, MoeHead.GATES, MoeHead.EXPERTS):
                shp = self.store.shapes[k]
                self.shadow_w16[k] = torch.zeros(shp, dtype=ops.F16, device=self.device)
                self.shadow_w8[k] = torch.zeros((shp[0], 2 * shp[1]), dtype=torch.uint8, device=self.device)
        else:
            super()._alloc_high_shadows()

    def _fp8_exps(self, k):
        return ops.FP8_DBOF_CLUSTER if k == self.CW else ops.FP8_MOE

    def _refresh_high(self, k):
        if k in getattr(self, "shadow_w8", {}):
            e = self._fp8_exps(k)
            p = self.store.p(k)
            ops.cast_f16(p, self.shadow_w16[k])
            ops.cast_fp8_lo(p, self.shadow_w8[k], hi_cols=p.shape[1], scale_exp=e["w_lo_exp"], hi_exp=e["w_hi_exp"])
        else:
            super()._refresh_high(k)

    @staticmethod
    def _pick_nslab(M, N, K):
        """Split-K factor of the cluster-weight gradient G [M][N] = dact^T . xhat over K rows: its 256x256 tiles rarely
        fill the 256 CUs (cfg 4: 32 x 5 = 160), so K is cut into slabs that the finishing pass adds on the way.  Cost
        model: rounds of 256 workgroups x (1/nslab) of a full-K tile, plus writing and reading nslab f32 slabs."""
        tiles = -(-M // 256) * -(-N // 256)
        nk = K // 32
        t_tile = 2.0 * 256 * 256 * K / 3.9e12                       # a CU at ~40 % of its MFMA peak
        best, best_cost = 1, None
        for n in range(1, 9):
            per = -(-nk // n)
            if per * (n - 1) >= nk or per < 8:
                continue
            cost = -(-tiles * n // 256) * t_tile / n + (n * 2.0 * M * N * 4 / 5e12 if n > 1 else 0.0)
            if best_cost is None or cost < best_cost * 0.97:
                best, best_cost = n, cost
        return best

    def _bn_train_stats(self, bn, part, P, width):
        """Column partial sums -> batch statistics of the GLOBAL batch (SyncBN) + moving averages."""
        ops.bn_partials_reduce(part, P, width, bn.ws)
        world = _maybe_allreduce(bn.ws, self.pg)
        bn.R_total = self.R * world
        ops.bn_finalize_ema(bn.ws, bn.R_total, width, bn.mean, bn.var, self.buffers[bn.scope + "/moving_mean"],
                            self.buffers[bn.scope + "/moving_variance"])

    def forward(self, x, num_frames, uniform, normalize=True, is_training=True):
        """x [B,T,F] raw frames, float32 or uint8 as the reader delivers them (Dequantize is fused; normalize=True
        fuses tf.nn.l2_normalize of the sampled frames); uniform [B,S] f32 in [0,1): the tf.random_uniform draw of
        SampleRandomFrames, supplied by the caller so runs are reproducible."""
        B = x.shape[0]
        if B != self.B:
            self._alloc(B)
        F, S, Cc, Hd = self.F, self.S, self.Cc, self.Hd
        high = self.precision != "bf16"
        tape = self.training and is_training
        ops.dbof_gather(x, uniform, num_frames, self.r, self.idx, self.part_in if is_training else None, normalize=normalize)
        if is_training:
            self._bn_train_stats(self.bn_in, self.part_in, self.P_in, F)
        else:
            self.bn_in.stats(None, self.R, False)
        fp8 = high and self.CW in getattr(self, "shadow_w8", {})      # "high" on f16 + e4m3 operands (else: three split-bf16 products)
        if fp8 and (not hasattr(self, "r_rows") or self.r_rows.shape[0] != self.r_bn.shape[0]):
            self.r_rows = torch.empty((self.Mp, 2 * F), dtype=ops.F16, device=self.r_bn.device)
            self.pooled_rows = torch.empty((B, 2 * Cc), dtype=ops.F16, device=self.r_bn.device)
        if high and not fp8 and (not hasattr(self, "r_bn_lo") or self.r_bn_lo.shape != self.r_bn.shape):
            self.r_bn_lo = torch.empty_like(self.r_bn)
            self.pooled_lo = torch.zeros_like(self.pooled_bf)
        if fp8:
            ops.dbof_input_bn_apply_f16fp8(self.r, B, S, F, self.bn_in.mean, self.bn_in.var, self.bn_in.gamma(), self.bn_in.beta(), self.r_rows,
                                           self.xhat if tape else None)
        else:
            ops.dbof_input_bn_apply(self.r, B, S, F, self.bn_in.mean, self.bn_in.var, self.bn_in.gamma(), self.bn_in.beta(), self.r_bn,
                                    self.r_bn_lo if high else None, self.xhat if tape else None)
        if self.timing is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if fp8:
            ops.dbof_cluster_pool_fwd_f16fp8(self.r_rows, self.shadow_w16[self.CW], self.shadow_w8[self.CW], B, S, F, Cc, self.bn_cl.gamma(),
                                             self.xsel, self.arg, act=self.act if tape else None, part=self.part_cl if is_training else None)
        else:
            ops.dbof_cluster_pool_fwd(self.r_bn, self.shadow_fwd[self.CW], B, S, F, Cc, self.bn_cl.gamma(), self.xsel, self.arg,
                                      act=self.act if tape else None, part=self.part_cl if is_training else None,
                                      r_bn_lo=self.r_bn_lo if high else None, wT_lo=self.shadow_lo[self.CW] if high else None)
        if self.timing is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.timing.append((e0, e1))
        if is_training:
            self._bn_train_stats(self.bn_cl, self.part_cl, self.P_cl, Cc)
        else:
            self.bn_cl.stats(None, self.R, False)
        ops.dbof_pool_finish(self.xsel, B, Cc, self.bn_cl.mean, self.bn_cl.var, self.bn_cl.gamma(), self.bn_cl.beta(), self.pooled,
                             self.pooled_bf, self.pooled_lo if (high and not fp8) else None)
        if fp8:          # pooled in [0, 6] (relu6): the MoE head's scales fit (6 * 2^6 = 384 < 448)
            ops.cast_f16_fp8x(self.pooled, self.pooled_rows)
            ops.gemm_nt_f16_fp8(self.pooled_rows, self.shadow_w16[self.HW], self.shadow_w8[self.HW], B, Hd, Cc, self.hid)
        elif high:
            ops.gemm_nt_split(self.pooled_bf, self.pooled_lo, self.shadow_fwd[self.HW], self.shadow_lo[self.HW], B, Hd, Cc, self.hid)
        else:
            ops.gemm_nt(self.pooled_bf, self.shadow_fwd[self.HW], B, Hd, Cc, self.hid)
        self.bn_h.stats(self.hid, B, is_training)
        ops.bn_apply(self.hid, B, Hd, self.bn_h.mean, self.bn_h.var, self.bn_h.gamma(), self.bn_h.beta(), True,
                     y_f32=self.h6)
        self._taped = tape
        return self.moe.forward(self.h6)

    def grad_stages(self):
        """Variable names in the order their gradients become final during backward(): (moe, hidden, cluster)."""
        moe = [MoeHead.GATES, MoeHead.EXPERTS, MoeHead.EBIAS]
        hidden = [self.HW, "hidden1_bn/beta", "hidden1_bn/gamma"]
        cluster = [k for k in self.names if k not in moe and k not in hidden]
        return moe, hidden, cluster

    def backward(self, dpred, on_moe_grads_ready=None, moe_weight_grads=True, on_stage=None):
        """moe_weight_grads=False: the two MoE weight gradients are not materialised (MoeHead.fused_update recomputes
        them from the factors inside the clip + Adam pass).  on_stage(i): called when the gradients of grad_stages()[i]
        are final (a data-parallel caller starts that stage's all-reduce there, under the rest of the backward pass)."""
        assert self.training and self._taped, "backward needs a training-mode forward"
        B, F, S, Cc, Hd = self.B, self.F, self.S, self.Cc, self.Hd
        st = self.store
        dh6 = self.moe.backward(dpred, weight_grads=moe_weight_grads)
        if on_moe_grads_ready is not None:
            on_moe_grads_ready()
        if on_stage is not None:
            on_stage(0)
        self.bn_h.backward(self.hid, dh6, B, True, dx_bf16=self.dhid_bf)
        # hidden1 weights: dWh^T [Hd][C] = dhid^T . pooled (TN over the batch rows); dpooled = dhid . Wh^T
        ops.gemm_tn(self.dhid_bf, self.pooled_bf, Hd, Cc, self.Bk, st.g(self.HW))
        if on_stage is not None:
            on_stage(1)
        ops.gemm_nt(self.dhid_bf, self.shadow_bwd[self.HW], B, Cc, Hd, self.dpooled)
        # max-pool routing + relu6 mask + cluster_bn backward: the two batch sums need only the [B][C] selected entries
        bn = self.bn_cl
        ops.bn_bwd_partial(self.xsel, self.dpooled, B, Cc, bn.mean, bn.var, bn.gamma(), bn.beta(), True, bn.ws)
        _maybe_allreduce(bn.ws, self.pg)
        ops.dbof_dact(self.act, self.dpooled, self.pooled, self.arg, bn.mean, bn.var, bn.gamma(), bn.ws, bn.R_total, B, S, Cc,
                      dgamma=st.g("cluster_bn/gamma"), dbeta=st.g("cluster_bn/beta"))
        # G = dact^T . xhat in split-K slabs, then cluster weights / input_bn gradients from G
        ops.gemm_tn_slabs(self.act, self.xhat, Cc, F, self.Mp, self.slabs, self.nslab)
        ops.dbof_wgrad_finish(self.slabs, self.nslab, Cc, F, st.p(self.CW), self.bn_in.gamma(), st.g(self.CW),
                              st.g("input_bn/gamma"), st.g("input_bn/beta"), part_ws=self.wgrad_ws)
        if on_stage is not None:
            on_stage(2)

    @property
    def pred(self):
        return self.moe.pred


class DbofGenericTower(TowerBase):
    """DbofModel with the flag values NO launcher of the reference selects (cs/frame_level_models.py:108-195): any combination of
    --dbof_pooling_method max | average (cs/model_utils.py:75-78), --dbof_add_batch_norm True | False (cluster_biases /
    hidden1_biases instead of the three slim.batch_norm, :158-161,181-185) and --sample_random_frames True | False
    (SampleRandomSequence, cs/model_utils.py:11-36).  The default combination runs on DbofTower (fused cluster kernel); this tower
    is the plain chain on the library's generic kernels - one launch per graph op group, the [B*S, clusters] activation in f32 -
    written for parity, not for speed.  ('none' pooling: FramePooling returns [B*S, C], so the predictions have B*S rows against B
    label rows: the reference's graph does not train with it - refused by DbofModel.create_model.)
    Precision (round 5): "high" = "split" here - the cluster, hidden and MoE products as split-bf16 (hi.hi + hi.lo + lo.hi, f32-operand
    accuracy: three launches per contraction on separate hi / lo shadows, TowerBase's default layout), so that these branches too have a
    mode inside north_star's 1e-3; the backward products stay bf16 as in every mode."""

    CW, CB, HW, HB = "cluster_weights", "cluster_biases", "hidden1_weights", "hidden1_biases"
    l2_names = (MoeHead.GATES, MoeHead.EXPERTS)

    def __init__(self, batch_size, max_frames=300, feature_size=1152, vocab_size=4716, iterations=30, cluster_size=8192,
                 hidden_size=1024, num_mixtures=2, device="cuda:0", training=True, scope="model", seed=0, process_group=None,
                 pooling="max", add_batch_norm=True, random_frames=True):
        if pooling not in ("max", "average"):
            raise ValueError("Unrecognized pooling method: %s" % pooling)                  # cs/model_utils.py:83
        self.device, self.training, self.scope, self.pg = torch.device(device), training, scope, process_group
        self.T, self.F, self.V, self.S = max_frames, feature_size, vocab_size, iterations
        self.Cc, self.Hd, self.Mx = cluster_size, hidden_size, num_mixtures
        self.pooling, self.bn, self.random_frames = pooling, bool(add_batch_norm), bool(random_frames)
        if feature_size % 64 or cluster_size % 64 or hidden_size % 64:
            raise ValueError("feature/cluster/hidden sizes must be multiples of 64 for the MFMA GEMM tiles")
        F, Cc, Hd = feature_size, cluster_size, hidden_size
        shapes = OrderedDict()
        if self.bn:
            shapes.update(BatchNorm.shapes("input_bn", F))
        shapes[self.CW] = (Cc, F)                                    # stored transposed [C][F]
        if self.bn:
            shapes.update(BatchNorm.shapes("cluster_bn", Cc))
        else:
            shapes[self.CB] = (Cc,)
        shapes[self.HW] = (Hd, Cc)
        if self.bn:
            shapes.update(BatchNorm.shapes("hidden1_bn", Hd))
        else:
            shapes[self.HB] = (Hd,)
        shapes.update(MoeHead.shapes(Hd, vocab_size, num_mixtures))
        self.buffers = OrderedDict()
        self._setup_store(shapes)
        if self.bn:
            self.bn_in, self.bn_cl, self.bn_h = BatchNorm(self, "input_bn", F), BatchNorm(self, "cluster_bn", Cc), BatchNorm(self, "hidden1_bn", Hd)
        # batch-norm gradients leave BatchNorm.backward already summed over the ranks (all-reduced f64 sums)
        self.global_grad_names = tuple("%s/%s" % (s_, v) for s_ in ("input_bn", "cluster_bn", "hidden1_bn") for v in ("beta", "gamma")) if self.bn else ()
        self.moe = MoeHead(self, Hd, vocab_size, num_mixtures)
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed)
        for k, shp in self.store.shapes.items():          # cs/frame_level_models.py:145-147,158-160,169-171,182-184
            p = self.store.p(k)
            if k in (self.CW, self.HW):
                p.copy_(torch.randn(shp, generator=gen, dtype=F32) / math.sqrt(shp[1]))
            elif k == self.CB:
                p.copy_(torch.randn(shp, generator=gen, dtype=F32) / math.sqrt(F))
            elif k == self.HB:
                p.copy_(torch.randn(shp, generator=gen, dtype=F32) * 0.01)
            elif len(shp) == 2:
                lim = math.sqrt(6.0 / (shp[0] + shp[1]))
                p.copy_((torch.rand(shp, generator=gen, dtype=F32) * 2 - 1) * lim)
            elif k.endswith("/gamma"):
                p.fill_(1.0)
        self.refresh_shadows()
        self._alloc(batch_size)

    def _alloc(self, B):
        dev, F, S, Cc, Hd = self.device, self.F, self.S, self.Cc, self.Hd
        self.B, self.R = B, B * S
        R = self.R
        if self.training and R % 32:
            raise ValueError("batch_size x iterations = %d must be a multiple of 32 (row count of the TN weight-gradient products)" % R)
        self.Bk = ops.round_up(B, 32)
        self.r = torch.empty((R, F), dtype=F32, device=dev)
        self.idx = torch.empty((B, S), dtype=torch.int32, device=dev)
        self.r_bn = torch.empty((R, F), dtype=BF16, device=dev)
        self.act = torch.empty((R, Cc), dtype=F32, device=dev)      # pre-activation (before cluster_bn, or with its bias)
        self.a6 = torch.empty((R, Cc), dtype=F32, device=dev)
        self.arg = torch.empty((B, Cc), dtype=torch.int32, device=dev)
        self.pooled = torch.empty((B, Cc), dtype=F32, device=dev)
        self.pooled_bf = torch.zeros((self.Bk, Cc), dtype=BF16, device=dev)
        self.hid = torch.empty((B, Hd), dtype=F32, device=dev)
        self.h6 = torch.empty((B, Hd), dtype=F32, device=dev)
        self.moe.alloc(B, self.training)
        if self.training:
            self.dhid_bf = torch.zeros((self.Bk, Hd), dtype=BF16, device=dev)
            self.dpooled = torch.empty((B, Cc), dtype=F32, device=dev)
            self.dy = torch.empty((R, Cc), dtype=F32, device=dev)
            self.dact_bf = torch.empty((R, Cc), dtype=BF16, device=dev)
            self.dx = torch.empty((R, F), dtype=F32, device=dev)

    def forward(self, x, num_frames, uniform, normalize=True, is_training=True):
        """x [B,T,F] float32 or uint8; uniform: [B,S] f32 in [0,1) (SampleRandomFrames) or [B] / [B,1] (SampleRandomSequence)."""
        B = x.shape[0]
        if B != self.B:
            self._alloc(B)
        F, S, Cc, Hd, R, st = self.F, self.S, self.Cc, self.Hd, self.R, self.store
        if self.random_frames:
            ops.sample_frames_gather(x, uniform, num_frames, self.r, self.idx, normalize=normalize)
        else:
            u1 = uniform.reshape(B, -1)[:, 0].contiguous()
            ops.sample_sequence_gather(x, u1, num_frames, S, self.r, self.idx, normalize=normalize)
        hp = self.precision != "bf16"              # split-bf16 forward products (the operands' low-order halves in r_lo / pooled_lo / shadow_lo)
        if hp and (not hasattr(self, "r_lo") or self.r_lo.shape != self.r_bn.shape):
            self.r_f32 = torch.empty((R, F), dtype=F32, device=x.device)
            self.r_lo = torch.empty_like(self.r_bn)
            self.pooled_lo = torch.zeros_like(self.pooled_bf)

        def product(a_hi, a_lo, a_f32, k, M, N, K, out, bias=None):
            if hp:
                ops.cast_bf16_split(a_f32, a_hi[:M], a_lo[:M])
                ops.gemm_nt_split(a_hi, a_lo, self.shadow_fwd[k], self.shadow_lo[k], M, N, K, out, bias=bias)
            else:
                ops.gemm_nt(a_hi, self.shadow_fwd[k], M, N, K, out, bias=bias)
        if self.bn:
            bi, bc, bh = self.bn_in, self.bn_cl, self.bn_h
            bi.stats(self.r, R, is_training)
            ops.bn_apply(self.r, R, F, bi.mean, bi.var, bi.gamma(), bi.beta(), False, y_f32=self.r_f32 if hp else None, y_bf16=self.r_bn)
            product(self.r_bn, self.r_lo if hp else None, self.r_f32 if hp else None, self.CW, R, Cc, F, self.act)
            bc.stats(self.act, R, is_training)
            ops.bn_apply(self.act, R, Cc, bc.mean, bc.var, bc.gamma(), bc.beta(), True, y_f32=self.a6)
        else:
            ops.cast_bf16(self.r, self.r_bn)
            product(self.r_bn, self.r_lo if hp else None, self.r, self.CW, R, Cc, F, self.act, bias=st.p(self.CB))
            ops.relu6_fwd(self.act, y_f32=self.a6)
        if self.pooling == "max":
            ops.framepool_max_fwd(self.a6, B, S, Cc, self.pooled, self.pooled_bf, self.arg)
        else:
            ops.framepool_mean_fwd(self.a6, B, S, Cc, self.pooled, self.pooled_bf)
        if self.bn:
            product(self.pooled_bf, self.pooled_lo if hp else None, self.pooled, self.HW, B, Hd, Cc, self.hid)
            bh.stats(self.hid, B, is_training)
            ops.bn_apply(self.hid, B, Hd, bh.mean, bh.var, bh.gamma(), bh.beta(), True, y_f32=self.h6)
        else:
            product(self.pooled_bf, self.pooled_lo if hp else None, self.pooled, self.HW, B, Hd, Cc, self.hid, bias=st.p(self.HB))
            ops.relu6_fwd(self.hid, y_f32=self.h6)
        self._taped = self.training and is_training
        return self.moe.forward(self.h6)

    def grad_stages(self):
        moe = [MoeHead.GATES, MoeHead.EXPERTS, MoeHead.EBIAS]
        hidden = [k for k in (self.HW, self.HB, "hidden1_bn/beta", "hidden1_bn/gamma") if k in self.names]
        return moe, hidden, [k for k in self.names if k not in moe and k not in hidden]

    def backward(self, dpred, on_moe_grads_ready=None, moe_weight_grads=True, on_stage=None):
        assert self.training and self._taped, "backward needs a training-mode forward"
        B, F, S, Cc, Hd, R, st = self.B, self.F, self.S, self.Cc, self.Hd, self.R, self.store
        dh6 = self.moe.backward(dpred, weight_grads=moe_weight_grads)
        if on_moe_grads_ready is not None:
            on_moe_grads_ready()
        if on_stage is not None:
            on_stage(0)
        if self.bn:
            self.bn_h.backward(self.hid, dh6, B, True, dx_bf16=self.dhid_bf)
        else:
            ops.relu6_bwd(self.hid, dh6, dx_bf16=self.dhid_bf[:B])
            ops.colsum_bf16(self.dhid_bf, self.Bk, Hd, st.g(self.HB))
        ops.gemm_tn(self.dhid_bf, self.pooled_bf, Hd, Cc, self.Bk, st.g(self.HW))          # dWh^T [Hd][C]
        if on_stage is not None:
            on_stage(1)
        ops.gemm_nt(self.dhid_bf, self.shadow_bwd[self.HW], B, Cc, Hd, self.dpooled)
        if self.pooling == "max":
            ops.framepool_max_bwd(self.dpooled, self.arg, B, S, Cc, self.dy)
        else:
            ops.framepool_mean_bwd(self.dpooled, B, S, Cc, self.dy)
        if self.bn:
            self.bn_cl.backward(self.act, self.dy, R, True, dx_bf16=self.dact_bf)
        else:
            ops.relu6_bwd(self.act, self.dy, dx_bf16=self.dact_bf)
            ops.colsum_bf16(self.dact_bf, R, Cc, st.g(self.CB))
        ops.gemm_tn(self.dact_bf, self.r_bn, Cc, F, R, st.g(self.CW))                        # dWc^T [C][F]
        if self.bn:
            ops.gemm_nt(self.dact_bf, self.shadow_bwd[self.CW], R, F, Cc, self.dx)
            self.bn_in.backward(self.r, self.dx, R, False)
        if on_stage is not None:
            on_stage(2)

    @property
    def pred(self):
        return self.moe.pred


class NetVladTower(TowerBase):
    """NetVLAD aggregation tower - an EXTENSION: the reference announces NetVLAD / NeXtVLAD teacher-student variants
    (README.md:126-127) but ships empty stubs (cs/frame_level_models.py:341-355), so there is no reference math; the
    definition is oracle/model_math.py::netvlad_fwd ("Learnable pooling with Context Gating" NetVLAD in the frame of the
    DBoF tower): sample S frames -> input_bn -> .Wc -> cluster_bn -> softmax over K clusters -> V[k] = sum_s a[s,k] (x_s - c2[k])
    -> intra-normalise -> l2-normalise -> .Wh -> hidden1_bn -> relu6 -> MoE.
    GEMM-shaped parts on the library's NT / TN kernels, the f32 pieces in csrc/evc_netvlad.hip.  V / Y are kept
    cluster-major [B][K][F]; hidden1_weights' TF row order f*K + k is restored in state_dict()."""

    PRECISIONS = ("bf16",)             # no split-bf16 forward (set_precision refuses anything else)

    CW, C2, HW = "cluster_weights", "cluster_weights2", "hidden1_weights"
    l2_names = (MoeHead.GATES, MoeHead.EXPERTS)
    # all three batch-norms go through BatchNorm.backward: their gamma/beta gradients are global (all-reduced f64 sums)
    global_grad_names = tuple("%s/%s" % (s_, v) for s_ in ("input_bn", "cluster_bn", "hidden1_bn") for v in ("beta", "gamma"))

    def __init__(self, batch_size, max_frames=300, feature_size=1152, vocab_size=4716, iterations=30, cluster_size=64,
                 hidden_size=1024, num_mixtures=2, device="cuda:0", training=True, scope="model", seed=0, process_group=None):
        self.device, self.training, self.scope, self.pg = torch.device(device), training, scope, process_group
        self.T, self.F, self.V, self.S = max_frames, feature_size, vocab_size, iterations
        self.Kc, self.Hd, self.Mx = cluster_size, hidden_size, num_mixtures
        if feature_size % 64 or cluster_size % 64 or hidden_size % 64:
            raise ValueError("feature / cluster / hidden sizes must be multiples of 64 for the MFMA GEMM tiles")
        if iterations > 64:
            raise ValueError("iterations=%d: the aggregation kernel holds at most 64 sampled frames per video" % iterations)
        F, K, H = feature_size, cluster_size, hidden_size
        shapes = OrderedDict()
        shapes.update(BatchNorm.shapes("input_bn", F))
        shapes[self.CW] = (K, F)                                     # stored transposed [K][F]
        shapes.update(BatchNorm.shapes("cluster_bn", K))
        shapes[self.C2] = (K, F)                                     # centres, stored [K][F] (tf: [1, F, K])
        shapes[self.HW] = (H, K * F)                                 # stored transposed, columns cluster-major (k*F + f)
        shapes.update(BatchNorm.shapes("hidden1_bn", H))
        shapes.update(MoeHead.shapes(H, vocab_size, num_mixtures))
        self.buffers = OrderedDict()
        self._setup_store(shapes)
        self.bn_in = BatchNorm(self, "input_bn", F)
        self.bn_cl = BatchNorm(self, "cluster_bn", K)
        self.bn_h = BatchNorm(self, "hidden1_bn", H)
        self.moe = MoeHead(self, H, vocab_size, num_mixtures)
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed)
        for k, shp in self.store.shapes.items():
            p = self.store.p(k)
            if k in (self.CW, self.C2):
                p.copy_(torch.randn(shp, generator=gen, dtype=F32) / math.sqrt(F))
            elif k == self.HW:
                p.copy_(torch.randn(shp, generator=gen, dtype=F32) / math.sqrt(K))
            elif len(shp) == 2:
                lim = math.sqrt(6.0 / (shp[0] + shp[1]))
                p.copy_((torch.rand(shp, generator=gen, dtype=F32) * 2 - 1) * lim)
            elif k.endswith("/gamma"):
                p.fill_(1.0)
        self.refresh_shadows()
        self._alloc(batch_size)

    # hidden1_weights: TF layout [F*K, H] with row f*K + k  <->  internal [H][k*F + f]
    def state_dict(self):
        out = super().state_dict()
        key = "%s/%s" % (self.scope, self.HW)
        w = out[key]                                                 # [K*F (k-major), H]
        out[key] = w.view(self.Kc, self.F, self.Hd).permute(1, 0, 2).reshape(self.F * self.Kc, self.Hd).contiguous()
        return out

    def load_state_dict(self, sd, prefix=None):
        prefix = self.scope if prefix is None else prefix
        key = "%s/%s" % (prefix, self.HW)
        if key not in sd:
            raise KeyError("NetVladTower.load_state_dict: %r is not in the checkpoint (scope prefix %r; it holds e.g. %s)"
                           % (key, prefix, sorted(sd)[:3]))
        sd = dict(sd)
        sd[key] = sd[key].view(self.F, self.Kc, self.Hd).permute(1, 0, 2).reshape(self.Kc * self.F, self.Hd).contiguous()
        super().load_state_dict(sd, prefix)

    def _alloc(self, B):
        dev, F, S, K, H = self.device, self.F, self.S, self.Kc, self.Hd
        self.B, self.R = B, B * S
        if self.training and self.R % 32:
            raise ValueError("batch_size x iterations = %d must be a multiple of 32 (row count of the TN weight-gradient products)" % self.R)
        R = self.R
        self.Bk = ops.round_up(B, 32)
        self.r = torch.empty((R, F), dtype=F32, device=dev)
        self.idx = torch.empty((B, S), dtype=torch.int32, device=dev)
        self.r_bn = torch.empty((R, F), dtype=BF16, device=dev)
        self.act = torch.empty((R, K), dtype=F32, device=dev)
        self.a = torch.empty((R, K), dtype=F32, device=dev)
        self.asum = torch.empty((B, K), dtype=F32, device=dev)
        self.Vv = torch.empty((B, K, F), dtype=F32, device=dev)
        self.n1 = torch.empty((B, K), dtype=F32, device=dev)
        self.n2 = torch.empty((B,), dtype=F32, device=dev)
        self.Y_bf = torch.zeros((self.Bk, K * F), dtype=BF16, device=dev)
        self.hid = torch.empty((B, H), dtype=F32, device=dev)
        self.h6 = torch.empty((B, H), dtype=F32, device=dev)
        self.moe.alloc(B, self.training)
        if self.training:
            self.dhid_bf = torch.zeros((self.Bk, H), dtype=BF16, device=dev)
            self.dY = torch.empty((B, K * F), dtype=F32, device=dev)
            self.dV = torch.empty((B, K, F), dtype=F32, device=dev)
            self.da = torch.empty((R, K), dtype=F32, device=dev)
            self.dz = torch.empty((R, K), dtype=F32, device=dev)
            self.dx = torch.empty((R, F), dtype=F32, device=dev)
            self.dact_bf = torch.empty((R, K), dtype=BF16, device=dev)

    def forward(self, x, num_frames, uniform, normalize=True, is_training=True):
        """x [B,T,F] float32 or uint8 raw frames; uniform [B,S] f32 in [0,1): the frame-sampling draw."""
        B = x.shape[0]
        if B != self.B:
            self._alloc(B)
        F, S, K, H, R = self.F, self.S, self.Kc, self.Hd, self.R
        st, bi, bc = self.store, self.bn_in, self.bn_cl
        if self.precision != "bf16":
            raise NotImplementedError("NetVladTower has no split-bf16 mode")
        ops.sample_frames_gather(x, uniform, num_frames, self.r, self.idx, normalize=normalize)
        bi.stats(self.r, R, is_training)
        ops.bn_apply(self.r, R, F, bi.mean, bi.var, bi.gamma(), bi.beta(), False, y_bf16=self.r_bn)
        ops.gemm_nt(self.r_bn, self.shadow_fwd[self.CW], R, K, F, self.act)
        bc.stats(self.act, R, is_training)
        ops.netvlad_softmax_fwd(self.act, R, K, bc.mean, bc.var, bc.gamma(), bc.beta(), self.a)
        ops.netvlad_aggregate_fwd(self.a, self.r, B, S, K, F, bi.mean, bi.var, bi.gamma(), bi.beta(), st.p(self.C2), self.Vv, self.asum)
        ops.netvlad_normalize_fwd(self.Vv, B, K, F, self.n1, self.n2, self.Y_bf)
        ops.gemm_nt(self.Y_bf, self.shadow_fwd[self.HW], B, H, K * F, self.hid)
        self.bn_h.stats(self.hid, B, is_training)
        ops.bn_apply(self.hid, B, H, self.bn_h.mean, self.bn_h.var, self.bn_h.gamma(), self.bn_h.beta(), True, y_f32=self.h6)
        self._taped = self.training and is_training
        return self.moe.forward(self.h6)

    def grad_stages(self):
        moe = [MoeHead.GATES, MoeHead.EXPERTS, MoeHead.EBIAS]
        hidden = [self.HW, "hidden1_bn/beta", "hidden1_bn/gamma"]
        return moe, hidden, [k for k in self.names if k not in moe and k not in hidden]

    def backward(self, dpred, on_moe_grads_ready=None, moe_weight_grads=True, on_stage=None):
        assert self.training and self._taped, "backward needs a training-mode forward"
        B, F, S, K, H, R = self.B, self.F, self.S, self.Kc, self.Hd, self.R
        st, bi = self.store, self.bn_in
        dh6 = self.moe.backward(dpred, weight_grads=moe_weight_grads)
        if on_moe_grads_ready is not None:
            on_moe_grads_ready()
        if on_stage is not None:
            on_stage(0)
        self.bn_h.backward(self.hid, dh6, B, True, dx_bf16=self.dhid_bf)
        ops.gemm_tn(self.dhid_bf, self.Y_bf, H, K * F, self.Bk, st.g(self.HW))          # dWh^T [H][K*F]
        if on_stage is not None:
            on_stage(1)
        ops.gemm_nt(self.dhid_bf, self.shadow_bwd[self.HW], B, K * F, H, self.dY)
        ops.netvlad_normalize_bwd(self.Vv, self.n1, self.n2, self.dY, B, K, F, self.dV)
        ops.netvlad_dcenters(self.asum, self.dV, B, K, F, st.g(self.C2))
        ops.netvlad_aggregate_bwd(self.a, self.r, B, S, K, F, bi.mean, bi.var, bi.gamma(), bi.beta(), st.p(self.C2), self.dV, self.da, self.dx)
        ops.netvlad_softmax_bwd(self.a, self.da, R, K, self.dz)
        self.bn_cl.backward(self.act, self.dz, R, False, dx_bf16=self.dact_bf)
        ops.gemm_tn(self.dact_bf, self.r_bn, K, F, R, st.g(self.CW))                   # dWc^T [K][F]
        ops.gemm_nt(self.dact_bf, self.shadow_bwd[self.CW], R, F, K, self.dx, accumulate=True)   # + dact . Wc^T
        bi.backward(self.r, self.dx, R, False)
        if on_stage is not None:
            on_stage(2)

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
        if self.precision != "bf16":
            if not hasattr(self, "avg_lo") or self.avg_lo.shape != self.avg_bf.shape:
                self.avg_lo = torch.empty_like(self.avg_bf)
            ops.cast_bf16_split(self.avg, self.avg_bf, self.avg_lo)
            ops.gemm_nt_split(self.avg_bf, self.avg_lo, self.shadow_fwd[self.W], self.shadow_lo[self.W], B, self.V, self.F,
                              self.pred, bias=self.store.p(self.Bn))
        else:
            ops.gemm_nt(self.avg_bf, self.shadow_fwd[self.W], B, self.V, self.F, self.pred, bias=self.store.p(self.Bn))
        ops.sigmoid_(self.pred)
        return self.pred

    def grad_stages(self):
        return ([self.W, self.Bn],)

    def backward(self, dpred, on_moe_grads_ready=None, moe_weight_grads=True, on_stage=None):
        B, V, F = self.B, self.V, self.F
        ops.sigmoid_bwd(self.pred, dpred, self.dz)
        ops.transpose_to_bf16(self.dz, B, V, self.dzT, self.Bp)
        ops.transpose_to_bf16(self.avg_bf, B, F, self.avgT, self.Bp)
        ops.gemm_nt(self.dzT, self.avgT, V, F, self.Bp, self.store.g(self.W))
        ops.rowsum_bf16(self.dzT, V, self.Bp, self.store.g(self.Bn))
        if on_moe_grads_ready is not None:
            on_moe_grads_ready()
        if on_stage is not None:
            on_stage(0)
