"""Execution engine of one model tower (teacher or student): parameters in
kernel layout, preallocated activation/gradient workspaces and the explicit
forward / backward / apply-gradients schedule over the C-ABI kernels.

What it replaces in the reference: the TensorFlow graph built by
``HierarchicalLstmModel.create_model`` / ``create_model_inference``
(cs/frame_level_models.py:200-338) + ``MoeModel.create_model``
(cs/video_level_models.py:397-448), its reverse-mode graph and the
``slim.learning.create_train_op`` update (cs/train.py:329-334, 413-418).

Memory layout (HBM, per tower; all preallocated once per batch size):
  * master parameters, gradients and Adam slots: three flat f32 buffers with
    one 64-byte aligned segment per variable (a single contiguous gradient
    buffer = one RCCL all-reduce payload);
  * LSTM kernels are stored TRANSPOSED, ``wT [4H][in+H]`` (K-contiguous B
    operand of the forward GEMM), with a bf16 shadow of the same layout and a
    bf16 shadow in TF layout ``w [in+H][4H]`` (K-contiguous for dz . W^T);
  * MoE weights likewise ``[V*(M+1)][K]`` / ``[V*M][K]`` + shadows
    ``[K][pad64(V*(M+1))]`` for the backward;
  * activations are time-major ``[T][M][width]``; L1 rows are
    ``m = chunk*B + b`` so L1's final state ``[C*B][2LH]`` *is* L2's
    time-major input ``[C][B][2LH]`` with no re-layout.
``state_dict()`` / ``load_state_dict()`` convert to/from the TF variable names
and layouts of SURVEY.md Appendix C.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


def _align(n, a=16):
    return (n + a - 1) // a * a


def _drain(gen):
    """Run a generator to its end and return its return value."""
    try:
        while True:
            next(gen)
    except StopIteration as e:
        return e.value


class ParamStore:
    """Flat f32 master / grad / adam-m / adam-v buffers with named views."""

    def __init__(self, shapes: "OrderedDict[str, tuple]", device):
        self.shapes = shapes
        self.offsets = OrderedDict()
        off = 0
        for k, shp in shapes.items():
            self.offsets[k] = off
            off += _align(int(math.prod(shp)))
        self.total = off
        self.master = torch.zeros(off, dtype=F32, device=device)
        self.grad = torch.zeros(off, dtype=F32, device=device)
        self.m = torch.zeros(off, dtype=F32, device=device)
        self.v = torch.zeros(off, dtype=F32, device=device)

    def view(self, buf, k):
        shp = self.shapes[k]
        n = int(math.prod(shp))
        o = self.offsets[k]
        return buf[o:o + n].view(*shp)

    def p(self, k):
        return self.view(self.master, k)

    def g(self, k):
        return self.view(self.grad, k)


class LstmStack:
    """L layers of BasicLSTMCell over T steps at M rows (one dynamic_rnn call
    of the reference, with the weight-shared chunk loops folded into M)."""

    def __init__(self, tower, scope, T, M, Kin, training):
        self.tw, self.scope, self.T, self.M, self.Kin = tower, scope, T, M, Kin
        H, L, dev = tower.H, tower.L, tower.device
        self.H, self.L = H, L
        self.S = torch.zeros((M, 2 * L * H), dtype=F32, device=dev)          # final state [c0|h0|c1|h1]
        self.hbuf = [torch.zeros((T + 1, M, H), dtype=BF16, device=dev) for _ in range(L)]   # zeros: stale rows stay finite
        self.kin = [Kin] + [H] * (L - 1)
        # hoist the x-projection when the per-step GEMM is small (M ~ batch): fewer, larger GEMMs
        self.hoist = [M < self.HOIST_BELOW for _ in range(L)]
        self.zx = None
        if any(self.hoist):
            self.zx = torch.empty((T * M, 4 * H), dtype=F32, device=dev)
        self.training = training
        if training:
            # history for BPTT: 8-byte gate records {i,j,f,o bf16} and the (bf16) cell state after every step
            self.gates = [torch.empty((T, M, H, 2), dtype=torch.int32, device=dev) for _ in range(L)]
            self.c_all = [torch.empty((T + 1, M, H), dtype=BF16, device=dev) for _ in range(L)]
            self.KP = ops.round_up(T * M, 64)
            self.dz = [torch.zeros((T, M, 4 * H), dtype=BF16, device=dev) for _ in range(L)]   # gate-interleaved [T][M][H][4];
            # one per layer so a layer's weight-gradient GEMMs (aux stream) can run under the next layer's BPTT
            self.use_tn = (T * M) % 32 == 0 and all(k % 8 == 0 for k in self.kin)
            if not self.use_tn:   # ragged row counts: transposed copies for the NT kernel
                self.dzT = torch.zeros((4 * H, self.KP), dtype=BF16, device=dev)
                self.xT = torch.empty((max(self.kin), self.KP), dtype=BF16, device=dev)
                self.hT_ws = torch.empty((H, self.KP), dtype=BF16, device=dev)
            self.dc_ws = torch.empty((M, H), dtype=F32, device=dev)
            self.dc_ws2 = torch.empty((M, H), dtype=F32, device=dev) if L == 2 else None    # wavefront BPTT: both layers in flight
            # dX of layer l > 0 = the dh arriving at layer l-1: bf16 (read once per BPTT step of the layer below)
            self.dx = [torch.empty((T * M, self.kin[l]), dtype=BF16, device=dev) if (l > 0) else None for l in range(L)]

    def names(self, l):
        base = "%s/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % (self.scope, l)
        return base + "kernel", base + "bias"

    # rows below which a stack counts as "M ~ batch": hoisted x-projection + wavefront pair launches.  1025 since round 6 (was 1024): cfg 5's L2 level
    # (M = 1024 videos, K = 4096 + 1024) as ONE 172 GFLOP product + 6 pair launches instead of ten 49 us step launches at 0.25 of peak -
    # 3.85 -> 3.68 ms per step, same box, alternating (profiles/r06_cfg5_hoist_ab.txt)
    HOIST_BELOW = int(os.environ.get("EVC_HOIST_BELOW", "1025"))
    wavefront = os.environ.get("EVC_NO_WAVEFRONT") != "1"   # two-layer M ~ batch stacks: see forward()
    fwd_walk2 = os.environ.get("EVC_FWD_WALK2", "1") != "0"   # two-layer many-row stacks (bf16): two tiles per workgroup, T + 1 launches (A/B: 0)
    # ... and the "high" mode's L1 level (round 6, ops.lstm_level2_fwd_high: bit-identical, NOT faster - alone 2.41 against 2.37 ms per teacher level, the training
    # step 11.67 against 11.56 ms, profiles/r06_walk2_high_ab.txt: a 96 us tile on 59 ring stages does not notice a 10 us tail; off unless EVC_FWD_WALK2_HIGH=1)
    fwd_walk2_high = os.environ.get("EVC_FWD_WALK2_HIGH", "0") == "1"
    # two-layer stacks with many rows (the L1 levels), the gradient arriving at layer 0 from layer 1:
    #   "off"   (default) one hoisted dX = dz1 . Wx1^T product over all T (bf16 result, re-read by layer 0's steps);
    #   "fused" contracted inside layer 0's BPTT steps (two-matrix K walk, K = 8H, f32 accumulator);
    #   "pair"  fused + wavefront order: layer 0's step t+1 and layer 1's step t in one launch (evc_lstm_stack2_bwd).
    # Measured on the headline step (DESIGN.md "Measured and dropped"): 12.75 / 12.95 / 12.89 ms - the step kernels' main
    # loops are L2->LDS-bound, so FLOPs moved into them cost more than the 1 PF/s hoisted product saves; kept for A/B runs.
    bwd_fuse = os.environ.get("EVC_BWD_FUSE", "off")
    bwd_wavefront = True          # (tests toggle this to compare the fused forms against the hoisted one)
    # M ~ batch two-layer stacks: wavefront BPTT on the skinny pair launches.  Built and measured in round 5 (same box, alternating): ALONE the
    # teacher's chain takes 453 instead of 507 us and the student's 119 instead of 144 (scripts/l2_bwd_bench.py), the training step 10.26 instead
    # of 10.07-10.13 ms - a pair launch wants two 64 KB workgroups on every CU at once and waits longer for them beside the other streams' tiles
    # than two 256-workgroup launches do, and layer 1's weight gradients / update no longer run under layer 0's chain.  Off; EVC_L2_BWD_PAIR=1.
    small_pair = os.environ.get("EVC_L2_BWD_PAIR", "0") == "1"
    timing = None      # set to a list to collect (start event, end event, launches, algorithmic flops) per layer forward
    timing_bwd = None  # set to {"bwd_step": [], "dx_nt": [], "wgrad_tn": []} to collect the same per backward launch sequence

    def _timed(self, kind, launches, flops, stream=None):
        """Context manager: HIP events on `stream` (default: current) around a launch sequence, kept in timing_bwd[kind]."""
        stack = self

        class _T:
            def __enter__(self_):
                self_.on = stack.timing_bwd is not None and kind in stack.timing_bwd
                if self_.on:
                    self_.e0 = torch.cuda.Event(enable_timing=True)
                    self_.e0.record(stream) if stream is not None else self_.e0.record()

            def __exit__(self_, *exc):
                if self_.on:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(stream) if stream is not None else e1.record()
                    stack.timing_bwd[kind].append((self_.e0, e1, launches, flops))
                return False
        return _T()

    @staticmethod
    def _v(buf, *shape):
        """Leading part of a preallocated buffer viewed with a (smaller) shape: row plans use [T][P][..] of [T][M][..]."""
        n = 1
        for d in shape:
            n *= d
        return buf.view(-1)[:n].view(*shape)

    def forward(self, x, lens, plan=None):
        """x [T][M][Kin] bf16 time-major (or a (hi, lo) pair in "high" precision); lens [M] int32.
        plan: ops.RowPlan of this batch - x is then [T][plan.P][Kin] in slot order and every buffer of the
        stack is used with plan.P rows per time slab; the returned state is in the original row order.
        Returns S [M, 2LH] f32."""
        tw, H, L, T = self.tw, self.H, self.L, self.T
        M = plan.P if plan is not None else self.M
        self.plan, self.Mrun = plan, M
        if plan is not None:
            lens = plan.lens
            if plan.rows[0] < self.M:
                self.S.zero_()                       # final state of the rows no step touches (length 0)
        hb = [self._v(h, T + 1, M, H) for h in self.hbuf]
        gates = [self._v(g, T, M, H, 2) for g in self.gates] if self.training else [None] * L
        c_all = [self._v(c, T + 1, M, H) for c in self.c_all] if self.training else [None] * L
        self._hb = hb
        rows = plan.rows if plan is not None else [M] * T
        if isinstance(x, tuple) and x[1].dtype == ops.F16:
            # "high" precision, many-row (L1) stacks: IEEE f16 operands, ONE MFMA product per depth - the cost of the bf16 step
            # at 2^-12 operand rounding (scripts/precision_budget.py: L1 states within ~3e-5 of float64 at trained magnitudes).
            # x = (bf16 image, f16 image): the bf16 one and the bf16 copies of h stay the operands of the backward products.
            x_bf, x16 = x[0], x[1]
            x_rs = x[2] if len(x) > 2 else None      # integer-frame rows (ops.l2norm_chunk_int): the frames' row scales [T][rows] f32
            self.x_in, self.lens = x_bf, lens
            if self.scope == "RNN_L2" and L == 2 and plan is None:
                # M ~ batch (the L2 level): wavefront pair launches on f16 operands, the upper layer's weights K-extended by their
                # low-order halves (ops.lstm_stack2_fwd_f16); wide [h | h/64] f16 images + the bf16 copies for the backward pass
                if self.zx is None:
                    self.zx = torch.empty((self.T * self.M, 4 * H), dtype=F32, device=self.S.device)
                (k0, b0), (k1, b1) = self.names(0), self.names(1)
                if k1 in tw.shadow8:      # weights' low-order halves as e4m3 operands in the same launches (ops.lstm_stack2_fwd_f16_fp8lo)
                    al = tw.act_lo()
                    wrow = 2 * H if al else 3 * H // 2                # containers per h row: [f16(h) | e4m3(h) (| e4m3(h_lo))]
                    if not hasattr(self, "hrows16") or self.hrows16[0].shape[-1] != wrow:
                        self.hrows16 = [torch.zeros((self.T + 1, self.M, wrow), dtype=ops.F16, device=h.device) for h in self.hbuf]
                    hr = [self._v(h, T + 1, M, wrow) for h in self.hrows16]
                    ops.lstm_stack2_fwd_f16_fp8lo(x16, tw.shadow16[k0], tw.shadow8[k0], tw.store.p(b0), tw.shadow16[k1], tw.shadow8[k1], tw.store.p(b1),
                                                  lens, T, M, self.Kin, H, self.zx, hr[0], hr[1], hb[0], hb[1], self.S, gates, c_all,
                                                  x_segments=tw.f16_l2_x_segments, h_lo=al)
                    return self.S
                if not hasattr(self, "hbuf16w"):
                    self.hbuf16w = [torch.zeros((self.T + 1, self.M, 2 * H), dtype=ops.F16, device=h.device) for h in self.hbuf]
                hw = [self._v(h, T + 1, M, 2 * H) for h in self.hbuf16w]
                (k0, b0), (k1, b1) = self.names(0), self.names(1)
                ops.lstm_stack2_fwd_f16(x16, tw.shadow16[k0], tw.store.p(b0), tw.shadow16[k1], tw.store.p(b1), lens, T, M, self.Kin, H,
                                        self.zx, hw[0], hw[1], hb[0], hb[1], self.S, gates, c_all,
                                        x_segments=tw.f16_l2_x_segments, h0_ext=tw.f16_l2_h0_ext)
                return self.S
            if self.scope == "RNN_L1" and tw.fp8_lo():
                # f16 stages + e4m3 stages behind them in the same launches, per layer one of two forms:
                #  * weights' low-order halves in fp8 (ops.lstm_layer_fwd_f16_fp8lo): h rows [f16(h) | e4m3(h 2^7)] of 3H bytes; layer 0 reads the
                #    [f16(x) | e4m3(x 2^7) | e4m3(x_lo 2^18)] rows of ops.l2norm_chunk(fp8_tail=True), the layers above the h rows of the layer below;
                #  * time-dithered f16 weight images (tw.dither_layers(), ops.lstm_layer_fwd_f16_dith, DESIGN.md 7 "dither"): step t contracts image
                #    t, whose rounding errors cancel over the steps of a chunk - no stages for the weights' low-order halves (layer 0 keeps the
                #    input's: the e4m3(x_lo 2^18) bytes against the e4m3(Wx 2^6) block of cast_fp8_lo's rows); plain f16 h rows.
                dl = tw.dither_layers()
                wh0 = tw.dither_wh0()                                             # layer 0: recurrent block dithered, input block corrected
                al = tw.act_lo()                                                  # rows [f16(h) | e4m3(h) | e4m3(h_lo)] (2H containers) against [lo | hi] weight rows
                widths = [H if (l in dl or (l == 0 and wh0)) else (2 * H if al else 3 * H // 2) for l in range(L)]        # halfwords per h row
                if getattr(self, "_hbuf16_widths", None) != widths:
                    self.hbuf16 = [torch.zeros((self.T + 1, self.M, widths[l]), dtype=ops.F16, device=self.hbuf[l].device) for l in range(L)]
                    self._hbuf16_widths = widths
                h16 = [self._v(self.hbuf16[l], T + 1, M, widths[l]) for l in range(L)]
                assert x16.shape[-1] == (3 * self.Kin // 2 if x_rs is not None else 2 * self.Kin), \
                    "the fp8 L1 level takes ops.l2norm_chunk(..., f16_segments=1, fp8_tail=True) rows (or ops.l2norm_chunk_int's)"
                assert x_rs is None or tw.x_int(), "integer-frame rows need layer 0 on the f16 + e4m3 form (HLstmTower.x_int)"
                inp, ldx, kx16 = x16, x16.shape[-1], self.Kin
                if L == 2 and tuple(dl) == (1,) and not wh0 and self.fwd_walk2_high:
                    # the shipped layout as T + 1 two-tile launches (round 6; ops.lstm_level2_fwd_high): layer 0 on its f16 + e4m3 stages and the dithered upper
                    # layer, step s and step s-1 of one launch - the same bits as the two layer calls below
                    (k0, b0), (k1, b1) = self.names(0), self.names(1)
                    assert tw.shadow16d[k1].shape[0] >= T
                    x8_off, kx8, xi, gap = 2 * self.Kin, 2 * self.Kin, None, 0
                    if x_rs is not None:
                        kx8, gap, xi = self.Kin, self.Kin, (x_rs, tw.x_col_const[k0])
                    if self.timing is not None:
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record()
                    ops.lstm_level2_fwd_high(inp, ldx, kx16, x8_off, kx8, tw.shadow16[k0], tw.shadow8[k0], tw.store.p(b0), tw.shadow16d[k1], tw.store.p(b1),
                                             lens, T, M, H, h16[0], hb[0], h16[1], hb[1], self.S, gates, c_all, plan=plan, h_lo=al, x_int=xi, b8_gap=gap)
                    if self.timing is not None:
                        e1 = torch.cuda.Event(enable_timing=True)
                        e1.record()
                        flops = sum(2.0 * r * 4 * H * (self.kin[l] + (H if t > 0 else 0)) for l in range(2) for t, r in enumerate(rows))
                        live = sum(1 for r in rows if r > 0)
                        self.timing.append((e0, e1, live + 1 if live else 0, flops))
                    return self.S
                for l in range(L):
                    kn, bn = self.names(l)
                    if self.timing is not None:
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record()
                    if l == 0 and wh0 and 0 not in dl:
                        # [x | h] . W16_t^T with only Wh dithered + [e4m3(x) | e4m3(x_lo)] . [lo(Wx) | e4m3(Wx 2^6)]^T: the first 2 Kin bytes of cast_fp8_lo's rows
                        assert tw.shadow16d[kn].shape[0] >= T
                        ops.lstm_layer_fwd_f16_dith(inp, ldx, kx16, 2 * self.Kin, 2 * self.Kin, tw.shadow16d[kn], tw.shadow8[kn], tw.shadow8[kn].stride(0),
                                                    7 + ops.FP8_W_SCALE_EXP, tw.store.p(bn), lens, T, M, H, h16[l], hb[l],
                                                    self.S[:, (2 * l) * H:], self.S[:, (2 * l + 1) * H:], 2 * L * H, gates[l], c_all[l], plan=plan)
                    elif l in dl:
                        assert tw.shadow16d[kn].shape[0] >= T
                        w8 = tw.shadow8[kn][:, self.Kin:2 * self.Kin] if l == 0 else None      # the e4m3(Wx 2^6) block of [lo(Wx) | hi(Wx) | lo(Wh)] rows
                        ops.lstm_layer_fwd_f16_dith(inp, ldx, kx16, 3 * self.Kin if l == 0 else 0, self.Kin if l == 0 else 0, tw.shadow16d[kn], w8,
                                                    tw.shadow8[kn].stride(0) if l == 0 else 0, 7 + ops.FP8_W_SCALE_EXP, tw.store.p(bn), lens, T, M, H,
                                                    h16[l], hb[l], self.S[:, (2 * l) * H:], self.S[:, (2 * l + 1) * H:], 2 * L * H, gates[l], c_all[l], plan=plan)
                    else:
                        x8_off, kx8 = (2 * self.Kin, 2 * self.Kin) if l == 0 else (2 * H, 2 * H if al else H)
                        xi, gap = None, 0
                        if l == 0 and x_rs is not None:
                            # integer frames: [f16(2q - 255) | e4m3(x_hat 2^7)] rows, the frame's factor applied to the accumulators behind the x-part;
                            # no stages for the input's low-order half - the hi(Wx) block of the weight image is skipped (b8_gap)
                            kx8, gap = self.Kin, self.Kin
                            xi = (x_rs, tw.x_col_const[kn])
                        ops.lstm_layer_fwd_f16_fp8lo(inp, ldx, kx16, x8_off, kx8, tw.shadow16[kn], tw.shadow8[kn], tw.store.p(bn), lens, T, M, H,
                                                     h16[l], hb[l], self.S[:, (2 * l) * H:], self.S[:, (2 * l + 1) * H:], 2 * L * H,
                                                     gates[l], c_all[l], plan=plan, h_lo=al, x_int=xi, b8_gap=gap)
                    if self.timing is not None:
                        e1 = torch.cuda.Event(enable_timing=True)
                        e1.record()
                        flops = sum(2.0 * r * 4 * H * (self.kin[l] + (H if t > 0 else 0)) for t, r in enumerate(rows))
                        self.timing.append((e0, e1, sum(1 for r in rows if r > 0), flops))
                    inp, ldx, kx16 = h16[l][1:], widths[l], H
                return self.S
            wide = [l in tw.f16_wh_ext_layers for l in range(L)]        # layers whose recurrent weights are K-extended (wide h rows)
            if not hasattr(self, "hbuf16"):
                self.hbuf16 = [torch.zeros((self.T + 1, self.M, (2 if wide[l] else 1) * H), dtype=ops.F16, device=self.hbuf[l].device)
                               for l in range(L)]
            h16 = [self._v(self.hbuf16[l], T + 1, M, (2 if wide[l] else 1) * H) for l in range(L)]
            inp, ldx = x16, x16.shape[-1]
            for l in range(L):
                kn, bn = self.names(l)
                if self.timing is not None:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                # layer 0: the first f16_x_segments of the image's segments (a tower may use fewer than the image holds); above: H of
                # the (wide) rows of the layer below
                kx = tw.f16_x_segments * self.Kin if l == 0 else (2 * H if l in tw.f16_wx_ext_layers else H)
                assert kx <= ldx, "a layer whose input weights are extended needs wide rows [h | h/64] from the layer below"
                assert tw.shadow16[kn].shape[1] == kx + (2 if wide[l] else 1) * H
                ops.lstm_layer_fwd_f16(inp, tw.shadow16[kn], tw.store.p(bn), lens, T, M, kx, H, h16[l], hb[l],
                                       self.S[:, (2 * l) * H:], self.S[:, (2 * l + 1) * H:], 2 * L * H,
                                       gates[l], c_all[l], plan=plan, ldx=ldx, h_wide=wide[l])
                if self.timing is not None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    flops = sum(2.0 * r * 4 * H * (self.kin[l] + (H if t > 0 else 0)) for t, r in enumerate(rows))
                    self.timing.append((e0, e1, sum(1 for r in rows if r > 0), flops))
                inp, ldx = h16[l][1:], h16[l].shape[-1]
            return self.S
        if isinstance(x, tuple):        # split-bf16 operands, K-extended loops (the M ~ batch L2 stacks of the "high" mode)
            x_bf, x_w = x               # plain bf16 image (backward operand) and the wide [lo | hi] image [T][M][2Kin]
            assert plan is None, "the split-bf16 layers take no row plan (M ~ batch stacks)"
            self.x_in, self.lens = x_bf, lens
            if not hasattr(self, "hbuf_w"):
                self.hbuf_w = [torch.zeros((self.T + 1, self.M, 2 * H), dtype=BF16, device=h.device) for h in self.hbuf]
            if self.zx is None:
                self.zx = torch.empty((self.T * self.M, 4 * H), dtype=F32, device=self.S.device)
            hw = [self._v(h, T + 1, M, 2 * H) for h in self.hbuf_w]
            inp = x_w
            for l in range(L):
                kn, bn = self.names(l)
                ops.lstm_layer_fwd_hp(inp, tw.shadow_wx[kn], tw.shadow_wh[kn], tw.store.p(bn), lens, T, M, self.kin[l], H,
                                      self.zx, hb[l], hw[l], self.S[:, (2 * l) * H:], self.S[:, (2 * l + 1) * H:], 2 * L * H,
                                      gates[l], c_all[l])
                inp = hw[l][1:]
            return self.S
        self.x_in, self.lens = x, lens
        inp = x
        if (L == 2 and plan is None and all(self.hoist) and self.wavefront and self.timing is None
                and self.Kin % 64 == 0 and H % 64 == 0):
            # M ~ batch: layer 0 step t+1 and layer 1 step t share a launch (T+1 dependent launches instead of 2T)
            (k0, b0), (k1, b1) = self.names(0), self.names(1)
            ops.lstm_stack2_fwd(x, tw.shadow_fwd[k0], tw.store.p(b0), tw.shadow_fwd[k1], tw.store.p(b1), lens, T, M,
                                self.Kin, H, self.zx, hb[0], hb[1], self.S, gates, c_all)
            return self.S
        if L == 2 and self.fwd_walk2 and not any(self.hoist) and self.Kin % 64 == 0 and H % 64 == 0:
            # many-row two-layer levels (L1; round 5): layer 0's step s and layer 1's step s-1 in one launch whose workgroups walk both tiles
            # (ops.lstm_level2_fwd): T + 1 launches instead of 2 T, the second tile's ring fill under the first tile's gate tail; same bits
            (k0, b0), (k1, b1) = self.names(0), self.names(1)
            if self.timing is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            ops.lstm_level2_fwd(x, tw.shadow_fwd[k0], tw.store.p(b0), tw.shadow_fwd[k1], tw.store.p(b1), lens, T, M, self.Kin, H,
                                hb[0], hb[1], self.S, gates, c_all, plan=plan)
            if self.timing is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                flops = sum(2.0 * r * 4 * H * (self.kin[l] + (H if t > 0 else 0)) for l in range(2) for t, r in enumerate(rows))
                live = sum(1 for r in rows if r > 0)
                self.timing.append((e0, e1, live + 1 if live else 0, flops))
            return self.S
        for l in range(L):
            kn, bn = self.names(l)
            if self.timing is not None:                         # bench.py: live timing of the step launches of each layer
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            ops.lstm_layer_fwd(inp, tw.shadow_fwd[kn], tw.store.p(bn), lens, T, M, self.kin[l], H,
                               hb[l], self.S[:, (2 * l) * H:], self.S[:, (2 * l + 1) * H:], 2 * L * H,
                               gates[l], c_all[l], hoist=self.hoist[l], zx_ws=self.zx, plan=plan)
            if self.timing is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                kx = 0 if self.hoist[l] else self.kin[l]
                flops = sum(2.0 * r * 4 * H * (kx + (H if t > 0 else 0)) for t, r in enumerate(rows))
                self.timing.append((e0, e1, sum(1 for r in rows if r > 0), flops))
            inp = hb[l][1:]
        return self.S

    def profile_fwd_layers(self, reps=3):
        """Bench helper: time each layer's forward launch sequence (of the last forward() call, same row
        plan) with events on the current stream.  Returns [(ms_per_call, step_launches, gemm_flops)] per
        layer; flops are the algorithmic 2*rows_t*4H*K of the per-step GEMMs over the rows each step
        runs on (t=0 has no recurrent half)."""
        tw, H, T, M, plan = self.tw, self.H, self.T, self.Mrun, self.plan
        rows = plan.rows if plan is not None else [M] * T
        res = []
        inp = self.x_in
        if self.L == 2 and self.fwd_walk2 and not any(self.hoist) and self.Kin % 64 == 0 and H % 64 == 0:      # the launches forward() issues: the two-tile walk
            (k0, b0), (k1, b1) = self.names(0), self.names(1)
            gates = [self._v(self.gates[l], T, M, H, 2) if self.training else None for l in range(2)]
            c_all = [self._v(self.c_all[l], T + 1, M, H) if self.training else None for l in range(2)]
            args = (inp, tw.shadow_fwd[k0], tw.store.p(b0), tw.shadow_fwd[k1], tw.store.p(b1), self.lens, T, M, self.Kin, H, self._hb[0], self._hb[1],
                    self.S, gates, c_all)
            ops.lstm_level2_fwd(*args, plan=plan)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.lstm_level2_fwd(*args, plan=plan)
            e1.record()
            e1.synchronize()
            flops = sum(2.0 * r * 4 * H * (self.kin[l] + (H if t > 0 else 0)) for l in range(2) for t, r in enumerate(rows))
            live = sum(1 for r in rows if r > 0)
            return [(e0.elapsed_time(e1) / reps, live + 1 if live else 0, flops)]
        for l in range(self.L):
            kn, bn = self.names(l)
            gates = self._v(self.gates[l], T, M, H, 2) if self.training else None
            c_all = self._v(self.c_all[l], T + 1, M, H) if self.training else None
            args = (inp, tw.shadow_fwd[kn], tw.store.p(bn), self.lens, T, M, self.kin[l], H, self._hb[l],
                    self.S[:, (2 * l) * H:], self.S[:, (2 * l + 1) * H:], 2 * self.L * H, gates, c_all)
            ops.lstm_layer_fwd(*args, hoist=self.hoist[l], zx_ws=self.zx, plan=plan)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.lstm_layer_fwd(*args, hoist=self.hoist[l], zx_ws=self.zx, plan=plan)
            e1.record()
            e1.synchronize()
            kx = 0 if self.hoist[l] else self.kin[l]
            flops = sum(2.0 * r * 4 * H * (kx + (H if t > 0 else 0)) for t, r in enumerate(rows))
            res.append((e0.elapsed_time(e1) / reps, sum(1 for r in rows if r > 0), flops))
            inp = self._hb[l][1:]
        return res

    # A/B: 0 = the weight-gradient products of a row-planned level contract over every row of the [T][P] images (rounds 1-5) instead of
    # skipping each time slab's dead rows (ops.gemm_tn(live_rows=...))
    wgrad_live_rows = os.environ.get("EVC_WGRAD_LIVE_ROWS", "1") != "0"

    def _wgrad_tn(self, dz2, layer_in, h_prev, kin, rows, gW, live=None):
        """gW [4H][kin+H] += dz^T . [layer_in | h_prev] (gate rows de-interleaved).  One launch over both column segments when
        the input width allows it (kin == H, a multiple of 256: the upper layers): dz is read once and the launch has twice the
        tiles; two launches otherwise (layer 0 of the L1 stacks: 1152 + 1024 columns = 8.5 column tiles would leave half of
        the chip idle in the second round)."""
        H = self.H
        n1 = kin // 256 * 256
        if ops.DETERMINISTIC and rows % 32 == 0 and kin % 8 == 0:
            # no atomics, K split kept (round 5): partial products into slabs, added in slab order (ops.gemm_tn_det) - round 4 ran these products
            # as one workgroup per tile with the whole K (3.4 instead of 2.0 ms per step)
            if n1 >= 256 and (kin == n1 or kin - n1 >= 8):
                ops.gemm_tn_det(dz2, layer_in, 4 * H, n1, rows, gW, row_interleave_H=H, accumulate=True, ldc=kin + H, B2=h_prev, N2=H, c_col2=kin)
                if kin > n1:
                    ops.gemm_tn_det(dz2, layer_in[:, n1:], 4 * H, kin - n1, rows, gW[:, n1:kin], row_interleave_H=H, accumulate=True, ldc=kin + H)
            else:
                ops.gemm_tn_det(dz2, layer_in, 4 * H, kin, rows, gW, row_interleave_H=H, accumulate=True, ldc=kin + H)
                ops.gemm_tn_det(dz2, h_prev, 4 * H, H, rows, gW[:, kin:], row_interleave_H=H, accumulate=True, ldc=kin + H)
            return
        if not self.wgrad_live_rows:
            live = None
        if kin == H and kin % 256 == 0 and self.fuse_wgrad:
            ops.gemm_tn2(dz2, layer_in, kin, h_prev, H, 4 * H, rows, gW, row_interleave_H=H, accumulate=True, live_rows=live)
        elif self.fuse_wgrad and kin - n1 in (64, 128) and n1 >= H and rows >= 16384 and (n1 + H) % 2048 == 0:
            # layer 0 of the L1 stacks (1152 = 1024 + 128 input columns): the first 1024 input columns next to the h-part as one
            # 2048-column launch, the last 128 as a narrow strip of their own (1.0 + 0.13 ms against 0.75 + 0.6 ms)
            ops.gemm_tn2(dz2, layer_in, n1, h_prev, H, 4 * H, rows, gW, row_interleave_H=H, accumulate=True, c_col2=kin, live_rows=live)
            ops.gemm_tn(dz2, layer_in[:, n1:], 4 * H, kin - n1, rows, gW[:, n1:kin], row_interleave_H=H, lda=dz2.stride(0),
                        ldb=layer_in.stride(0), ldc=kin + H, accumulate=True, live_rows=live)
        else:
            ops.gemm_tn(dz2, layer_in, 4 * H, kin, rows, gW, row_interleave_H=H, ldc=kin + H, accumulate=True, live_rows=live)
            ops.gemm_tn(dz2, h_prev, 4 * H, H, rows, gW[:, kin:], row_interleave_H=H, ldc=kin + H, accumulate=True, live_rows=live)

    fuse_wgrad = os.environ.get("EVC_NO_FUSED_WGRAD") != "1"

    def backward(self, dS, need_dx, aux=None, on_layer_grads=None):
        """backward_layers run to its end; returns dX (or None)."""
        return _drain(self.backward_layers(dS, need_dx, aux, on_layer_grads))

    def backward_layers(self, dS, need_dx, aux=None, on_layer_grads=None):
        """GENERATOR form of backward(): yields the layer index after each layer's launches have been enqueued (upper layer first)
        and returns dX through StopIteration - a caller can issue other work between two layers (DistillGraph interleaves the host
        issue of the two towers' backward passes this way).  Every resumption must happen under the stream context of the first.

        dS [M, 2LH] f32.  Writes the grads of this stack's kernels/biases into the tower's grad
        buffer; returns dX [T*M, Kin] f32 (gradient wrt the stack input) if need_dx.
        aux: optional side stream for the weight-gradient products, which nothing on the BPTT
        critical path waits for; the caller joins it before using the gradients.
        on_layer_grads(l): called on the stream that holds layer l's weight-gradient products, right after they are enqueued -
        that layer's kernel and bias gradients are final in stream order (the upper layer first): the caller can reduce and
        apply them there while the lower layer's BPTT still runs."""
        tw, H, L, T, plan = self.tw, self.H, self.L, self.T, self.plan
        M = self.Mrun                                               # rows per time slab in this batch's layout
        assert plan is None or not need_dx, "dX of a row-planned stack would be in slot order"
        dh_above = None
        dx_out = None
        main = torch.cuda.current_stream(tw.device)
        use_tn = (T * M) % 32 == 0 and all(k % 8 == 0 for k in self.kin)
        assert use_tn or not self.use_tn or M == self.M
        fuse_ok = L == 2 and self.bwd_wavefront and use_tn and not need_dx and M >= 1024 and H % 128 == 0
        # M ~ batch stacks (the L2 levels; round 5): BPTT in wavefront order on the skinny pair launches - layer 0's step t+1 (the gradient from
        # layer 1 contracted in its own K walk) and layer 1's step t in ONE launch: T + 1 dependent launches instead of 2 T + the hoisted dX
        # product of layer 1.  Not under EVC_DETERMINISTIC (the pair launches sum their bias gradients with atomics).  EVC_L2_BWD_PAIR=0: A/B.
        small_pair = (L == 2 and self.small_pair and use_tn and M <= 512 and H % 128 == 0 and plan is None and not ops.DETERMINISTIC
                      and self.dc_ws2 is not None)
        if small_pair or (fuse_ok and self.bwd_fuse == "pair"):
            (k0, b0), (k1, b1) = self.names(0), self.names(1)
            dz = [self._v(self.dz[l], T, M, 4 * H) for l in range(2)]
            gb = [tw.store.g(b0), tw.store.g(b1)]
            for g in gb:
                ops.fill_f32(g, 0.0)
            ops.lstm_stack2_bwd(tw.shadow_bwd[k0], tw.shadow_bwd[k1], self.lens, T, M, self.kin[0], H,
                                [self._v(self.gates[l], T, M, H, 2) for l in range(2)],
                                [self._v(self.c_all[l], T + 1, M, H) for l in range(2)], dS,
                                [self._v(self.dc_ws, M, H), self._v(self.dc_ws2, M, H)], dz, gb, plan=plan)
            side = main
            if aux is not None:
                ev = torch.cuda.Event()
                ev.record(main)
                aux.wait_event(ev)
                side = aux
            with torch.cuda.stream(side):
                for l, kn in ((1, k1), (0, k0)):
                    kin = self.kin[l]
                    gW = tw.store.g(kn)
                    dz2 = dz[l].view(T * M, 4 * H)
                    layer_in = (self.x_in if l == 0 else self._hb[0][1:]).reshape(T * M, kin)
                    h_prev = self._hb[l][:T].reshape(T * M, H)
                    ops.fill_f32(gW, 0.0)
                    self._wgrad_tn(dz2, layer_in, h_prev, kin, T * M, gW)
                    if on_layer_grads is not None:
                        on_layer_grads(l)
            if need_dx:       # gradient wrt the stack's input (the L2 levels: it flows on into the L1 level's final states), from layer 0's dz alone
                if self.dx[0] is None:
                    self.dx[0] = torch.empty((T * M, self.kin[0]), dtype=F32, device=tw.device)
                ops.gemm_nt(dz[0].view(T * M, 4 * H), tw.shadow_bwd[k0], T * M, self.kin[0], 4 * H, self.dx[0])
                dx_out = self.dx[0]
            yield 1
            yield 0
            return dx_out
        if fuse_ok and self.bwd_fuse == "fused" and on_layer_grads is not None:
            # the lower layer's steps read the upper layer's backward shadow (w_above): no layer may be updated before the end
            deferred, user_cb = [], on_layer_grads
            on_layer_grads = deferred.append
        else:
            deferred = user_cb = None
        for l in range(L - 1, -1, -1):
            kn, bn = self.names(l)
            w = tw.shadow_bwd[kn]                                   # [kin+H][4H] bf16, 4H axis gate-interleaved
            kin = self.kin[l]
            KP = self.KP
            dz = self._v(self.dz[l], T, M, 4 * H)
            gb = tw.store.g(bn)                                     # bias gradient: summed inside the step kernels
            det = ops.DETERMINISTIC                                 # ... with f32 atomics; EVC_DETERMINISTIC=1: column sums of dz afterwards
            if not det:
                ops.fill_f32(gb, 0.0)
            fused = fuse_ok and self.bwd_fuse == "fused" and l + 1 < L
            rows = plan.rows if plan is not None else [M] * T
            # BPTT step t contracts dh_t = dz_{t+1} . Wh^T over the rows live at t+1 (the last step has no recurrent product)
            with self._timed("bwd_step", sum(1 for r in rows if r > 0), sum(2.0 * r * H * 4 * H for r in rows[1:])):
                ops.lstm_layer_bwd(w, self.lens, T, M, kin, H, self._v(self.gates[l], T, M, H, 2), self._v(self.c_all[l], T + 1, M, H),
                                   dS[:, (2 * l) * H:], dS[:, (2 * l + 1) * H:], 2 * L * H,
                                   dh_above, self._v(self.dc_ws, M, H), dz, plan=plan, db=None if det else gb,
                                   dz_above=self._v(self.dz[l + 1], T, M, 4 * H) if fused else None,
                                   w_above=tw.shadow_bwd[self.names(l + 1)[0]] if fused else None)
            ops.mark("%s:%s_bptt%d_done" % (tw.scope, self.scope, l))
            dz2 = dz.view(T * M, 4 * H)
            # gradient wrt the layer input, all T at once (hoisted): dX = dz . Wx^T
            if l > 0 and fuse_ok and self.bwd_fuse == "fused":
                pass                                                # contracted inside layer l-1's steps (dz_above)
            elif l > 0:
                dxl = self._v(self.dx[l], T * M, kin)
                with self._timed("dx_nt", 1, 2.0 * T * M * kin * 4 * H):
                    ops.gemm_nt(dz2, w, T * M, kin, 4 * H, dxl)
                dh_above = dxl
            elif need_dx:
                if self.dx[0] is None:
                    self.dx[0] = torch.empty((T * M, kin), dtype=F32, device=tw.device)
                ops.gemm_nt(dz2, w, T * M, kin, 4 * H, self.dx[0])
                dx_out = self.dx[0]
            # dW^T [4H][kin+H] = dz^T . [x_in | h_prev]; db = column sums of dz
            side = main
            if aux is not None:
                ev = torch.cuda.Event()
                ev.record(main)
                aux.wait_event(ev)
                side = aux
            with torch.cuda.stream(side):
                gW = tw.store.g(kn)                                     # [4H][kin+H] f32
                layer_in = (self.x_in if l == 0 else self._hb[l - 1][1:]).reshape(T * M, kin)
                h_prev = self._hb[l][:T].reshape(T * M, H)
                if det:     # fixed-order bias gradient from the (bf16) gate gradients: [T*M][H][4] -> TF order g*H+u
                    ops.colsum_bf16(dz2, T * M, 4 * H, gb, deinterleave_H=H)
                if use_tn:
                    # "TN" products straight from the row-major activations (transpose reads in the kernel);
                    # the gate-interleaved rows of the product are stored in TF gate order.  The split-K partial
                    # tiles are added with atomics: one contiguous fill of the whole gradient, then accumulate
                    # (instead of a pitched 2-D memset inside each call).
                    ops.fill_f32(gW, 0.0)
                    # (flops of the LIVE rows: with a row plan the products skip each slab's dead rows, ops.gemm_tn(live_rows=...))
                    with self._timed("wgrad_tn", 1, 2.0 * (sum(rows) if plan is not None else T * M) * 4 * H * (kin + H), stream=side):
                        self._wgrad_tn(dz2, layer_in, h_prev, kin, T * M, gW, live=(rows, M) if plan is not None and M % 32 == 0 else None)      # (plan.P = M when the stack has fewer rows than a multiple of 32)
                else:   # T*M not a multiple of 32: transposed copies + NT products
                    ops.transpose_to_bf16(dz2, T * M, 4 * H, self.dzT, KP, interleave_H=-H)
                    inT = self.xT[:kin]
                    ops.transpose_to_bf16(layer_in, T * M, kin, inT, KP)
                    ops.transpose_to_bf16(h_prev, T * M, H, self.hT_ws, KP)
                    ops.gemm_nt(self.dzT, inT, 4 * H, kin, KP, gW, ldc=kin + H)
                    ops.gemm_nt(self.dzT, self.hT_ws, 4 * H, H, KP, gW[:, kin:], ldc=kin + H)
                if on_layer_grads is not None:
                    on_layer_grads(l)
                if deferred is not None and l == 0:
                    for ll in deferred:
                        user_cb(ll)
            yield l
        return dx_out


class MoeHead:
    """MoeModel (cs/video_level_models.py:394-448) on a [B, K] f32 input: two
    bf16 MFMA GEMMs (gates without bias, experts with bias) + the fused
    softmax/sigmoid/mix tail; backward gives d(input) and the weight grads."""

    GATES, EXPERTS, EBIAS = "classifier/gates/weights", "classifier/experts/weights", "classifier/experts/biases"

    def __init__(self, tower, K, V, Mx):
        self.tw, self.K, self.V, self.Mx = tower, K, V, Mx

    @staticmethod
    def shapes(K, V, Mx):
        sh = OrderedDict()
        sh[MoeHead.GATES] = (V * (Mx + 1), K)           # stored transposed
        sh[MoeHead.EXPERTS] = (V * Mx, K)
        sh[MoeHead.EBIAS] = (V * Mx,)
        return sh

    def alloc(self, B, training):
        dev, K, V, Mx = self.tw.device, self.K, self.V, self.Mx
        self.B = B
        self.Br = ops.round_up(B, 32)                                   # rows of the factor images (zero rows pad): TN contraction
        self.x_full = torch.zeros((self.Br, K), dtype=BF16, device=dev)
        self.x_bf = self.x_full[:B]
        self.gate_logits = torch.empty((B, V * (Mx + 1)), dtype=F32, device=dev)
        self.expert_logits = torch.empty((B, V * Mx), dtype=F32, device=dev)
        self.pred = torch.empty((B, V), dtype=F32, device=dev)
        self.rowsum = torch.empty((B,), dtype=F32, device=dev)
        if training:
            self.Bp = ops.round_up(B, 64)
            self.dgl_full = torch.zeros((self.Br, ops.round_up(V * (Mx + 1), 64)), dtype=BF16, device=dev)   # pad rows / cols stay 0
            self.del_full = torch.zeros((self.Br, ops.round_up(V * Mx, 64)), dtype=BF16, device=dev)
            self.dgl, self.del_ = self.dgl_full[:B], self.del_full[:B]
            self.partial_ws = torch.empty(2 * ((V * (Mx + 1) + 127) // 128) * ((K + 127) // 128), dtype=F32, device=dev)
            # Gram-matrix clip norms (ops.moe_grad_norms): K slabs of [Br][Br] f32 for dlogits (reused by both matrices) and x,
            # per-block partial sums, and the carried |W|^2 of the two matrices (wsq[i] = {|W|^2, 0}; valid flags below)
            self.gram_S = {"a": max(1, min(16, self.dgl_full.shape[1] // 32 // 16)), "x": max(1, min(8, K // 32 // 16))}
            self.gram_a = torch.empty(self.gram_S["a"] * self.Br * self.Br, dtype=F32, device=dev) if self.Br <= 512 else None
            self.gram_x = torch.empty(self.gram_S["x"] * self.Br * self.Br, dtype=F32, device=dev) if self.Br <= 512 else None
            self.norm_part = torch.empty(256 + 4 * self.Br, dtype=F32, device=dev)
            self.wsq = torch.zeros((2, 2), dtype=F32, device=dev)
            self._wsq_valid = [False, False]
            self.dglT = torch.empty((V * (Mx + 1), self.Bp), dtype=BF16, device=dev)
            self.delT = torch.empty((V * Mx, self.Bp), dtype=BF16, device=dev)
            self.xT = torch.empty((K, self.Bp), dtype=BF16, device=dev)
            self.dx = torch.empty((B, K), dtype=F32, device=dev)

    def forward(self, x):
        tw, B, V, Mx, K = self.tw, self.B, self.V, self.Mx, self.K
        if getattr(tw, "precision", "bf16") == "high" and self.GATES in getattr(tw, "shadow_w8", {}):
            # f16 product + both low-order corrections as e4m3 operands behind it in the same launch (ops.gemm_nt_f16_fp8)
            if not hasattr(self, "x_rows") or self.x_rows.shape[0] != B:
                self.x_rows = torch.empty((B, 2 * K), dtype=ops.F16, device=x.device)
            ops.cast_bf16(x, self.x_bf)                                 # the backward / fused-update factor
            # the input is the L2 state [c | h]: its cell-state half is unbounded (|c| ~ 16 after 512 training steps), so the e4m3 images take
            # their range from this batch's max|x| (64 partial maxima; no shift - the fixed 2^6 / 2^17 - while max|x| <= 7)
            amax = None
            if self.dynamic_fp8_range:
                if not hasattr(self, "x_amax"):
                    self.x_amax = torch.zeros(ops.AMAX_SLOTS, dtype=F32, device=x.device)
                amax = ops.absmax_partials(x, self.x_amax)
            ops.cast_f16_fp8x(x, self.x_rows, amax_ws=amax)
            ops.gemm_nt_f16_fp8(self.x_rows, tw.shadow_w16[self.GATES], tw.shadow_w8[self.GATES], B, V * (Mx + 1), K, self.gate_logits, amax_ws=amax)
            ops.gemm_nt_f16_fp8(self.x_rows, tw.shadow_w16[self.EXPERTS], tw.shadow_w8[self.EXPERTS], B, V * Mx, K, self.expert_logits,
                                bias=tw.store.p(self.EBIAS), amax_ws=amax)
            ops.moe_tail_fwd(self.gate_logits, self.expert_logits, B, V, Mx, self.pred, self.rowsum)
            return self.pred
        if getattr(tw, "precision", "bf16") != "bf16" and hasattr(tw, "shadow_w"):
            # split-bf16 operands, one K-extended launch per product (ops.gemm_nt_split_wide)
            if not hasattr(self, "x_w") or self.x_w.shape[0] != B:
                self.x_w = torch.empty((B, 2 * K), dtype=BF16, device=x.device)
            ops.cast_bf16(x, self.x_bf)                                 # the backward / fused-update factor
            ops.cast_bf16_wide(x, self.x_w, lo_first=True)
            ops.gemm_nt_split_wide(self.x_w, tw.shadow_w[self.GATES], B, V * (Mx + 1), K, self.gate_logits)
            ops.gemm_nt_split_wide(self.x_w, tw.shadow_w[self.EXPERTS], B, V * Mx, K, self.expert_logits, bias=tw.store.p(self.EBIAS))
            ops.moe_tail_fwd(self.gate_logits, self.expert_logits, B, V, Mx, self.pred, self.rowsum)
            return self.pred
        if getattr(tw, "precision", "bf16") != "bf16":                 # towers that keep separate hi / lo shadows (DBoF, logistic)
            if not hasattr(self, "x_lo") or self.x_lo.shape != self.x_bf.shape:
                self.x_lo = torch.empty_like(self.x_bf)
            ops.cast_bf16_split(x, self.x_bf, self.x_lo)
            ops.gemm_nt_split(self.x_bf, self.x_lo, tw.shadow_fwd[self.GATES], tw.shadow_lo[self.GATES], B, V * (Mx + 1), K,
                              self.gate_logits)
            ops.gemm_nt_split(self.x_bf, self.x_lo, tw.shadow_fwd[self.EXPERTS], tw.shadow_lo[self.EXPERTS], B, V * Mx, K,
                              self.expert_logits, bias=tw.store.p(self.EBIAS))
            ops.moe_tail_fwd(self.gate_logits, self.expert_logits, B, V, Mx, self.pred, self.rowsum)
            return self.pred
        ops.cast_bf16(x, self.x_bf)
        ops.gemm_nt(self.x_bf, tw.shadow_fwd[self.GATES], B, V * (Mx + 1), K, self.gate_logits)
        ops.gemm_nt(self.x_bf, tw.shadow_fwd[self.EXPERTS], B, V * Mx, K, self.expert_logits,
                    bias=tw.store.p(self.EBIAS))
        ops.moe_tail_fwd(self.gate_logits, self.expert_logits, B, V, Mx, self.pred, self.rowsum)
        return self.pred

    def can_fuse_update(self):
        """evc_moe_grad_update's shape constraints (the reference sizes satisfy them: 14148, 9432, 4096)."""
        return (self.V * (self.Mx + 1)) % 4 == 0 and (self.V * self.Mx) % 4 == 0 and self.K % 8 == 0

    dynamic_fp8_range = os.environ.get("EVC_HIGH_DYNAMIC_RANGE", "1") != "0"     # "high" head: e4m3 range of the input state from the batch (A/B: 0 = fixed 2^6)
    FUSE_MAX_ROWS = int(os.environ.get("EVC_MOE_FUSE_MAX_ROWS", "512"))
    # one process: the clip norm of the fused update from Gram matrices of the factors instead of a pass over the weights
    # (csrc/evc_moe_norms.hip; EVC_MOE_GRAM_NORMS=0: the two-pass form)
    gram_norms = os.environ.get("EVC_MOE_GRAM_NORMS", "1") != "0"
    gram_force = True if os.environ.get("EVC_MOE_GRAM_NORMS") == "1" else None     # unset: chosen by shape (use_gram_norms)
    skip_stale_fwd_shadow = os.environ.get("EVC_HIGH_KEEP_BF16_SHADOW", "0") != "1"

    @staticmethod
    def gram_slab_count(cols, want):
        """Largest S <= want that evc_gram_slabs accepts for `cols` columns (a multiple of 32; no empty last slab), 0 if none."""
        if cols <= 0 or cols % 32:
            return 0
        nk = cols // 32
        for S in range(max(1, min(want, nk)), 0, -1):
            if ((nk + S - 1) // S) * (S - 1) < nk:
                return S
        return 0

    def use_gram_norms(self, rows, Vn, cols):
        """Gram-matrix clip norm or pass 1 over the weights, by shape (round 5).  Pass 1 streams Vn x K f32 weights and recomputes the
        rank-`rows` gradient tile; the Gram route costs ~3.5 small launches per matrix plus 2 rows^2 (cols + K/2) flops on a
        fragments-from-global kernel - it grows with rows^2, pass 1 with K.  Constants fitted to both measured points
        (profiles/r04_bench_kernel_stats_default.csv: rows 256, K 4096 - Gram 0.16 vs pass 1 0.36 ms per tower;
        profiles/r04_bench_dbof_kernel_stats.csv: rows 512, K 1024 - Gram 0.16 vs pass 1 0.06 ms).  EVC_MOE_GRAM_NORMS=0 / 1 forces."""
        if not self.gram_norms or self.gram_a is None:
            return False
        K = self.K
        if rows % 32 or rows > 1024 or self.gram_slab_count(K, self.gram_S["x"]) == 0 or self.gram_slab_count(cols, self.gram_S["a"]) == 0:
            return False                     # evc_gram_slabs would refuse the shape: the two-pass form takes anything the fused update takes
        if self.gram_force is not None:
            return self.gram_force
        pass1 = Vn * K * 4.0 / 4.0e12 + 2.0 * rows * Vn * K / 5.0e14
        gram = 42e-6 + 2.0 * rows * rows * (cols + 0.5 * K) / 1.2e14
        return gram < pass1

    def invalidate_norm_cache(self):
        """The carried |W|^2 no longer describes the weights (they were written by something other than the fused update)."""
        if hasattr(self, "_wsq_valid"):
            self._wsq_valid = [False, False]

    def prefer_fused_update(self, data_parallel):
        """The fused update recomputes the rank-B gradient tile in both of its passes: 2 x 2 B V K flops against the 46 - 30
        bytes per parameter it saves.  One MI355X: faster up to B = 256 (the headline config), even at 512, slower at 1024
        (cfg 5 student-only: 5.15 vs 4.88 ms per step) - there the gradients are materialised.  Under data parallelism it
        always pays (no 386 MB gradient all-reduce, each rank updates 1/world of the rows)."""
        return data_parallel or self.B <= self.FUSE_MAX_ROWS

    # ---- data parallel: which exchange carries the MoE weight gradient (round 6) -----------------------------------------------
    def dp_exchange_bytes(self, world):
        """Bytes one rank puts on the wire per step for the two MoE weight matrices, per route (ring collectives):
        "factors"        all-gather of the three bf16 factor images [Br][N_g], [Br][N_e], [Br][K] (fused_update): (W-1) Br (N_g + N_e + K) 2 -
                         grows with the BATCH;
        "reduce_scatter" the locally formed gradient X^T dZ as a bf16 reduce-scatter onto the owners' row slabs (sharded_update):
                         (W-1)/W (N_g + N_e) K 2 - a constant of the MODEL.
        Both routes then all-gather the owners' new bf16 rows (the same bytes, not counted).  cfg 3 (B 256, K 4096, N 23 580, W 8):
        100 vs 176 MB -> factors; cfg 5 (B 1024): 398 vs 176 MB (169 + the slab padding) -> reduce_scatter (break-even B ~ 453); cfg 4 (B 512,
        K 1024): 177 vs 44 MB -> reduce_scatter."""
        Ng, Ne = ops.round_up(self.V * (self.Mx + 1), 64), ops.round_up(self.V * self.Mx, 64)
        slabs = sum((((n + 127) // 128 + world - 1) // world) * 128 * world for n in (self.V * (self.Mx + 1), self.V * self.Mx))
        return {"factors": (world - 1) * self.Br * (Ng + Ne + self.K) * 2.0, "reduce_scatter": (world - 1.0) / world * slabs * self.K * 2.0}

    def dp_route(self, world):
        """"factors" | "reduce_scatter": the cheaper exchange by shape (dp_exchange_bytes); EVC_DP_MOE_ROUTE=factors|reduce_scatter forces one."""
        forced = os.environ.get("EVC_DP_MOE_ROUTE", "auto")
        if forced in ("factors", "reduce_scatter"):
            return forced
        b = self.dp_exchange_bytes(max(2, world))        # (a one-rank debug group: decide as for two ranks' worth of shape, no byte moves either way)
        return "reduce_scatter" if b["reduce_scatter"] < b["factors"] else "factors"

    def sharded_update(self, lr_t, clip_norm, l2_coeff, dp, beta1=0.9, beta2=0.999, eps=1e-8):
        """Data-parallel update of the two MoE weight matrices from the MATERIALISED local gradients (backward(weight_grads=True)) - the
        "reduce_scatter" route: bf16 image of the local gradient -> reduce-scatter onto the owners' row slabs (shard()) -> the owner's
        per-tensor norm (8-byte all-reduce of the slab sums) + clip + TF-Adam on its slab -> all-gather of the slabs' new bf16 rows, local
        transpose for the backward shadow.  Same ZeRO-1 state as fused_update (consolidate() serves both); the expert biases all-reduce
        (38 KB) and update everywhere.  bf16 forward only (the "high" images of a slab are not gathered)."""
        tw, V, Mx, K = self.tw, self.V, self.Mx, self.K
        st = tw.store
        idx = {k: i for i, k in enumerate(tw.names)}
        if getattr(self, "world", None) != dp.shard_world:
            self.shard(dp.shard_world, dp.rank)
        self._wsq_valid = [False, False]
        if not hasattr(self, "_rs_buf"):
            self._rs_buf, self._rs_g32 = {}, {}
        for name, Vn in ((self.GATES, V * (Mx + 1)), (self.EXPERTS, V * Mx)):
            l2 = l2_coeff if name in tw.l2_names else 0.0
            slab = self.slab[name]
            if name not in self._rs_buf or self._rs_buf[name].shape[0] != slab * dp.shard_world:
                self._rs_buf[name] = torch.zeros((slab * dp.shard_world, K), dtype=BF16, device=tw.device)      # rows >= Vn stay zero
                self._rs_g32[name] = torch.empty((slab, K), dtype=F32, device=tw.device)
            buf, g32 = self._rs_buf[name], self._rs_g32[name]
            ops.cast_bf16(st.g(name), buf[:Vn])
            own = dp.reduce_scatter_rows(buf, slab)
            v0 = dp.rank * slab
            vs = min(Vn, v0 + slab) - v0                                   # rows of this rank's slab (<= 0: none)
            pw, mw, vw = st.p(name), st.view(st.m, name), st.view(st.v, name)
            sums = tw.sums[idx[name]]
            if vs > 0:
                g32[:vs].copy_(own[:vs])
                ops.grad_sqnorm(g32[:vs], pw[v0:v0 + vs] if l2 else None, l2, sums)
            dp.all_reduce_small(sums)
            if vs > 0:
                ops.clip_adam_step(pw[v0:v0 + vs], g32[:vs], mw[v0:v0 + vs], vw[v0:v0 + vs], l2, sums, clip_norm, lr_t, beta1, beta2, eps,
                                   p_bf16=self._full(name)[v0:v0 + vs])
            dp.all_gather_slabs(self._full(name), slab)
            sb = tw.shadow_bwd[name]
            ops.transpose_to_bf16(tw.shadow_fwd[name], Vn, K, sb, sb.shape[1])
            self._stale = dp.world > 1
        gb = st.g(self.EBIAS)
        dp.all_reduce_small(gb)
        ops.grad_sqnorm(gb, None, 0.0, tw.sums[idx[self.EBIAS]])
        ops.clip_adam_step(st.p(self.EBIAS), gb, st.view(st.m, self.EBIAS), st.view(st.v, self.EBIAS), 0.0,
                           tw.sums[idx[self.EBIAS]], clip_norm, lr_t, beta1, beta2, eps)

    # ---- data parallel: the two weight matrices are sharded by rows over the ranks (ZeRO-1) ----------------
    def shard(self, world, rank):
        """Row slabs of 128-row tiles: rank r owns rows [r*slab, (r+1)*slab) of each weight matrix - their f32
        weights and Adam moments are kept current only there; the bf16 forward shadow (re-allocated here with
        world*slab rows so that equal slabs can be all-gathered in place) is complete on every rank."""
        tw = self.tw
        self.world, self.rank, self.slab, self._stale = world, rank, {}, False
        for name, Vn in ((self.GATES, self.V * (self.Mx + 1)), (self.EXPERTS, self.V * self.Mx)):
            tiles = (Vn + 127) // 128
            slab = (tiles + world - 1) // world * 128
            self.slab[name] = slab
            old = tw.shadow_fwd[name]
            full = torch.zeros((slab * world, self.K), dtype=BF16, device=old.device)
            full[:Vn].copy_(old)
            tw.shadow_fwd[name] = full[:Vn]
            setattr(self, "_full_" + ("g" if name == self.GATES else "e"), full)

    def _full(self, name):
        return self._full_g if name == self.GATES else self._full_e

    def consolidate(self, dp):
        """Collective: brings the sharded f32 weights and Adam moments of every rank up to date (before a checkpoint,
        an export or an evaluation that reads them)."""
        if getattr(self, "world", 1) == 1 or not self._stale:
            return
        st = self.tw.store
        for name, Vn in ((self.GATES, self.V * (self.Mx + 1)), (self.EXPERTS, self.V * self.Mx)):
            slab = self.slab[name]
            v0 = self.rank * slab
            tmp = torch.zeros((slab * self.world, self.K), dtype=F32, device=self.tw.device)
            for t in (st.p(name), st.view(st.m, name), st.view(st.v, name)):
                if v0 < Vn:
                    tmp[v0:min(Vn, v0 + slab)].copy_(t[v0:v0 + slab])
                dp.all_gather_slabs(tmp, slab)
                t.copy_(tmp[:Vn])
        self._stale = False

    def fused_update(self, lr_t, clip_norm, l2_coeff, beta1=0.9, beta2=0.999, eps=1e-8, dp=None):
        """Weight gradient + per-tensor clip + TF-Adam + both bf16 shadows of the two MoE weight matrices from the
        factors dlogits / x left by backward(weight_grads=False), without materialising the gradients
        (evc_moe_grad_update); the expert biases go the ordinary way (column sums, then clip + Adam).
        dp (distill.GradReducer, data parallel): the factors are all-gathered over the ranks along the rows - the
        contraction over world x batch rows IS the summed gradient, so no gradient all-reduce is needed - and each
        rank updates only its row slab (shard()): phase 1 on the slab, 8-byte all-reduce of the norm sums, phase 2 on
        the slab, all-gather of the slab's new bf16 rows, local transpose for the backward shadow.  Per rank: the
        flops of a single-GPU update and 1/world of its HBM traffic."""
        tw, V, Mx, K = self.tw, self.V, self.Mx, self.K
        dgl, del_, x = self.dgl_full, self.del_full, self.x_full
        if dp is not None:
            dgl, del_, x = dp.all_gather_rows(dgl), dp.all_gather_rows(del_), dp.all_gather_rows(x)
            if getattr(self, "world", None) != dp.shard_world:      # normally done once by DistillGraph.__init__ (no step in flight)
                self.shard(dp.shard_world, dp.rank)
        rows = x.shape[0]
        idx = {k: i for i, k in enumerate(tw.names)}
        st = tw.store
        refreshed_wide = False
        self._gram_x_fresh = False          # X . X^T of this step's factor: computed by the first matrix that takes the Gram route
        for name, dlog, Vn in ((self.GATES, dgl, V * (Mx + 1)), (self.EXPERTS, del_, V * Mx)):
            l2 = l2_coeff if name in tw.l2_names else 0.0
            pw, mw, vw = st.p(name), st.view(st.m, name), st.view(st.v, name)
            if dp is None:
                hi = tw.precision != "bf16"                 # the non-bf16 forward's operand images come out of the same epilogue
                wide = getattr(tw, "shadow_w", {}).get(name) if hi else None             # "split": [hi | lo]
                w16, w8 = (getattr(tw, "shadow_w16", {}).get(name), getattr(tw, "shadow_w8", {}).get(name)) if hi else (None, None)   # "high": f16 + e4m3
                # the bf16 forward shadow is not an operand of a "high" head (forward: f16 + e4m3 images; backward: the transposed shadow):
                # 2 of the update's 34 bytes per parameter not written; set_precision("bf16") / refresh_shadows() rebuild it from the masters
                pfwd = None if (w16 is not None and self.skip_stale_fwd_shadow) else tw.shadow_fwd[name]
                if self.use_gram_norms(rows, Vn, dlog.shape[1]):
                    # clip norm from the Gram matrices of the factors + the forward logits + the carried |W|^2: no pass over W
                    # (csrc/evc_moe_norms.hip), then the update pass alone
                    i = 0 if name == self.GATES else 1
                    Sx = self.gram_slab_count(K, self.gram_S["x"])
                    if not self._wsq_valid[i]:
                        self.wsq[i].zero_()
                        ops.grad_sqnorm(pw, None, 0.0, self.wsq[i])
                        self._wsq_valid[i] = True
                    if i == 0 or not self._gram_x_fresh:
                        ops.gram_slabs(x, rows, K, Sx, self.gram_x)
                        self._gram_x_fresh = True
                    Sa = self.gram_slab_count(dlog.shape[1], max(1, min(self.gram_S["a"], dlog.shape[1] // 32 // 8)))
                    ops.gram_slabs(dlog, rows, dlog.shape[1], Sa, self.gram_a)
                    logits = self.gate_logits if i == 0 else self.expert_logits
                    ops.moe_grad_norms(self.gram_a, Sa, self.gram_x, Sx, rows, dlog, logits,
                                       None if i == 0 else st.p(self.EBIAS), self.B, Vn, l2, self.wsq[i], self.norm_part, tw.sums[idx[name]])
                    ops.moe_grad_update_apply(dlog, x, rows, Vn, K, pw, mw, vw, pfwd, tw.shadow_bwd[name], l2,
                                              tw.sums[idx[name]], self.partial_ws, clip_norm, lr_t, self.wsq[i], beta1, beta2, eps,
                                              p_wide=wide, p_f16=w16, p_fp8=w8)
                else:
                    ops.moe_grad_update(dlog, x, rows, Vn, K, pw, mw, vw, pfwd, tw.shadow_bwd[name], l2,
                                        tw.sums[idx[name]], self.partial_ws, clip_norm, lr_t, beta1, beta2, eps, p_wide=wide, p_f16=w16, p_fp8=w8)
                    self._wsq_valid[0 if name == self.GATES else 1] = False      # this matrix's carried |W|^2 is stale now
                refreshed_wide = refreshed_wide or wide is not None or w16 is not None
                continue
            self._wsq_valid = [False, False]
            slab = self.slab[name]
            v0 = dp.rank * slab
            vs = min(Vn, v0 + slab) - v0                                   # rows of this rank's slab (<= 0: none)
            args = None
            if vs > 0:
                args = (dlog[:, v0:], x, rows, vs, K, pw[v0:], mw[v0:], vw[v0:], self._full(name)[v0:], tw.shadow_bwd[name][:, v0:],
                        l2, tw.sums[idx[name]], self.partial_ws, clip_norm, lr_t, beta1, beta2, eps)
                ops.moe_grad_update(*args, phase=1)
            dp.all_reduce_small(tw.sums[idx[name]])
            if vs > 0:
                ops.moe_grad_update(*args, phase=2)
            dp.all_gather_slabs(self._full(name), slab)
            sb = tw.shadow_bwd[name]
            ops.transpose_to_bf16(tw.shadow_fwd[name], Vn, K, sb, sb.shape[1])
            self._stale = dp.world > 1
        if tw.precision != "bf16" and not refreshed_wide:     # towers with separate hi / lo shadows: the update wrote the hi halves only
            for name in (self.GATES, self.EXPERTS):
                tw._refresh_high(name)
        gb = st.g(self.EBIAS)
        ops.colsum_bf16(del_, rows, V * Mx, gb)
        ops.grad_sqnorm(gb, None, 0.0, tw.sums[idx[self.EBIAS]])
        ops.clip_adam_step(st.p(self.EBIAS), gb, st.view(st.m, self.EBIAS), st.view(st.v, self.EBIAS), 0.0,
                           tw.sums[idx[self.EBIAS]], clip_norm, lr_t, beta1, beta2, eps)

    def backward(self, dpred, dx_init=None, weight_grads=True, presum_l2=None):
        """Returns dx [B,K] f32 (= dx_init + MoE contribution) and writes the three weight grads
        (weight_grads=False: leaves only the factors for fused_update).
        presum_l2 (a float: the l2 coefficient; round 6): the two weight-gradient products also leave the per-tensor clip norms in the tower's norm rows
        (ops.gemm_nt_sqnorm: no separate evc_grad_sqnorm pass - 8 of 40 bytes per parameter of the materialised update, cfg 5); the rows must
        have been zeroed (TowerBase.begin_update) and apply_group() then skips the norm pass for TowerBase._presummed."""
        tw, B, V, Mx, K = self.tw, self.B, self.V, self.Mx, self.K
        ops.moe_tail_bwd(self.gate_logits, self.expert_logits, dpred, B, V, Mx, self.dgl, self.del_)
        V3p, V2p = self.dgl.shape[1], self.del_.shape[1]
        if dx_init is None:
            ops.gemm_nt(self.dgl, tw.shadow_bwd[self.GATES], B, K, V3p, self.dx)
        else:
            self.dx.copy_(dx_init)
            ops.gemm_nt(self.dgl, tw.shadow_bwd[self.GATES], B, K, V3p, self.dx, accumulate=True)
        ops.gemm_nt(self.del_, tw.shadow_bwd[self.EXPERTS], B, K, V2p, self.dx, accumulate=True)
        if not weight_grads:
            return self.dx
        Bp = self.Bp
        ops.transpose_to_bf16(self.dgl, B, V * (Mx + 1), self.dglT, Bp)
        ops.transpose_to_bf16(self.del_, B, V * Mx, self.delT, Bp)
        ops.transpose_to_bf16(self.x_bf, B, K, self.xT, Bp)
        tw._presummed = set()
        for name, aT, Vn in ((self.GATES, self.dglT, V * (Mx + 1)), (self.EXPERTS, self.delT, V * Mx)):
            g = tw.store.g(name)
            if presum_l2 is not None and ops.gemm_nt_sqnorm_ok(Vn, K, Bp) and g.data_ptr() % 16 == 0:
                l2 = presum_l2 if name in tw.l2_names else 0.0
                if getattr(self, "sq_part", None) is None or self.sq_part.numel() < ops.gemm_nt_sqnorm_ws(Vn, K):
                    self.sq_part = torch.empty(ops.gemm_nt_sqnorm_ws(V * (Mx + 1), K), dtype=F32, device=tw.device)
                ops.gemm_nt_sqnorm(aT, self.xT, Vn, K, Bp, g, tw.store.p(name) if l2 else None, l2, tw.sums[list(tw.names).index(name)], ws=self.sq_part)
                tw._presummed.add(name)
            else:
                ops.gemm_nt(aT, self.xT, Vn, K, Bp, g)
        ops.rowsum_bf16(self.delT, V * Mx, Bp, tw.store.g(self.EBIAS))
        return self.dx


class TowerBase:
    """Parameter store + bf16 shadows + per-tensor clip / TF-Adam shared by all model towers."""

    l2_names = ()

    def _setup_store(self, shapes, transposed_2d=True):
        self.store = ParamStore(shapes, self.device)
        self.names = list(shapes.keys())
        self.shadow_fwd, self.shadow_bwd = {}, {}
        for k, shp in shapes.items():
            if len(shp) == 2:
                self.shadow_fwd[k] = torch.zeros(shp, dtype=BF16, device=self.device)
                self.shadow_bwd[k] = torch.zeros((shp[1], ops.round_up(shp[0], 64)), dtype=BF16, device=self.device)
        self.adam_t = 0
        self.sums = torch.zeros((len(self.names), 2), dtype=F32, device=self.device)

    # "bf16": one bf16 MFMA product per contraction.  "high": the forward GEMMs hold north_star's 1e-3 on trained-magnitude
    # weights - per-product operand formats chosen by the measured error budget.  "split": split-bf16 operands
    # (f32-operand accuracy, 3 products) in EVERY forward GEMM - the uniform parity mode, for towers without a budgeted
    # layout "high" means the same.
    precision = "bf16"
    PRECISIONS = ("bf16", "high", "split")

    def set_precision(self, precision):
        if precision not in self.PRECISIONS:
            raise ValueError("%s: precision %r is not one of %s" % (type(self).__name__, precision, self.PRECISIONS))
        if getattr(self, "_high_alloc", None) not in (None, precision) and precision != "bf16":
            raise ValueError("the operand shadows of this tower were laid out for precision %r" % self._high_alloc)
        self.precision = precision
        if precision != "bf16" and not hasattr(self, "shadow_lo"):
            self._high_alloc = precision
            self._alloc_high_shadows()
        self.refresh_shadows()
        import logging
        logging.getLogger("evc").info("%s: precision layout %s", getattr(self, "scope", type(self).__name__), self.precision_layout())

    def precision_layout(self):
        """The RESOLVED forward-operand layout of this tower (what the EVC_HIGH_* environment and the tower's shape made of
        `precision`): logged by set_precision, reported by bench.py and stored with checkpoints - two processes with different
        environments compute different forwards from the same weights and flags."""
        return {"precision": self.precision}

    def _alloc_high_shadows(self):
        """Default: every 2-D weight gets a separate low-order bf16 shadow (hi.hi + hi.lo + lo.hi as three products)."""
        self.shadow_lo = {k: torch.zeros_like(v) for k, v in self.shadow_fwd.items()}

    def _refresh_high(self, k):
        """The "high" mode's forward operand(s) of weight k from the f32 master (which the bf16 forward shadow must already
        match: Adam's kernel or cast_bf16 wrote it)."""
        ops.cast_bf16_split(self.store.p(k), self.shadow_fwd[k], self.shadow_lo[k])

    def refresh_shadows(self, fwd=True):
        if getattr(self, "moe", None) is not None:
            self.moe.invalidate_norm_cache()             # (called whenever the f32 masters were written from outside)
        for k in self.shadow_fwd:
            p = self.store.p(k)
            if self.precision != "bf16":
                ops.cast_bf16(p, self.shadow_fwd[k])
                self._refresh_high(k)
            elif fwd:
                ops.cast_bf16(p, self.shadow_fwd[k])
            sb = self.shadow_bwd[k]
            il = p.shape[0] // 4 if k.endswith("basic_lstm_cell/kernel") else 0     # LSTM: gate-interleaved 4H axis
            ops.transpose_to_bf16(p, p.shape[0], p.shape[1], sb, sb.shape[1], interleave_H=il)

    def state_dict(self):
        """TF-named, TF-layout copies (2-D weights are stored transposed internally)."""
        if hasattr(self, "flush_deferred"):
            self.flush_deferred()
        if getattr(getattr(self, "moe", None), "_stale", False):
            raise RuntimeError("the MoE weights of %r are sharded over the ranks; call consolidate() on the graph on "
                               "every rank before reading them" % self.scope)
        out = OrderedDict()
        for k in self.names:
            p = self.store.p(k)
            out["%s/%s" % (self.scope, k)] = (p.t().contiguous() if p.dim() == 2 else p.clone())
        for k, v in getattr(self, "buffers", {}).items():
            out["%s/%s" % (self.scope, k)] = v.clone()
        return out

    def load_state_dict(self, sd, prefix=None):
        prefix = self.scope if prefix is None else prefix
        if hasattr(self, "flush_deferred"):
            self.flush_deferred()
        for k in self.names:
            src = sd["%s/%s" % (prefix, k)].to(self.device, F32)
            p = self.store.p(k)
            p.copy_(src.t() if p.dim() == 2 else src)
        for k, v in getattr(self, "buffers", {}).items():
            key = "%s/%s" % (prefix, k)
            if key in sd:
                v.copy_(sd[key].to(self.device, F32))
        self.refresh_shadows()
        if getattr(getattr(self, "moe", None), "_stale", False):
            self.moe._stale = False              # every rank has just loaded the complete weights

    def grad_ranges(self, exclude=(), only=None):
        """Maximal contiguous (lo, hi) element ranges of the flat gradient buffer that cover every variable (of `only`,
        if given) not in `exclude` (the payloads of a data-parallel all-reduce)."""
        st, out = self.store, []
        for k in self.names:
            if k in exclude or (only is not None and k not in only):
                continue
            lo = st.offsets[k]
            hi = lo + _align(int(math.prod(st.shapes[k])))
            if out and out[-1][1] == lo:
                out[-1] = (out[-1][0], hi)
            else:
                out.append((lo, hi))
        return out

    def begin_update(self):
        """Start one optimizer step (one tf.train op): bumps the Adam step count, clears the norm sums."""
        self.adam_t += 1
        self.sums.zero_()

    def adam_lr_t(self, lr, beta1=0.9, beta2=0.999):
        """TF-Adam's bias-corrected step size at the current step count (tf.train.AdamOptimizer defaults)."""
        t = self.adam_t
        return lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)

    def apply_group(self, names, lr, clip_norm=1.0, l2_coeff=0.0, beta1=0.9, beta2=0.999, eps=1e-8, refresh=True, lr_t=None):
        """Per-tensor clip_by_norm + TF-Adam for a subset of the variables (their gradients must be
        final).  slim create_train_op semantics (cs/train.py:329-334); the l2 regulariser gradient
        (regularization_penalty * 1e-8 * W) is folded into the gradient before the norm.
        lr_t: the bias-corrected step size of the optimizer step the gradients belong to (a deferred group is applied after
        the next begin_update() has already bumped the step count)."""
        if lr_t is None:
            lr_t = self.adam_lr_t(lr, beta1, beta2)
        if getattr(self, "moe", None) is not None and (MoeHead.GATES in names or MoeHead.EXPERTS in names):
            self.moe.invalidate_norm_cache()
        idx = {k: i for i, k in enumerate(self.names)}
        # LSTM layers: kernel + bias in two launches with every operand image written from the update pass (ops.lstm_adam_fused)
        done = set()
        if self.fused_lstm_adam and refresh:
            st = self.store
            for k in names:
                if not k.endswith("basic_lstm_cell/kernel"):
                    continue
                b = k[:-len("kernel")] + "bias"
                img = self._adam_images(k)
                if b not in names or k not in self.shadow_bwd or img is None or self.store.shapes[k][0] % 64:
                    continue
                if not hasattr(self, "_sqn_ws"):
                    self._sqn_ws = {}
                ws = self._sqn_ws.setdefault(k, torch.empty(1028, dtype=F32, device=self.device))      # (one per layer: launches of different layers may overlap)
                ops.lstm_adam_fused(st.p(k), st.g(k), st.view(st.m, k), st.view(st.v, k), st.p(b), st.g(b), st.view(st.m, b), st.view(st.v, b),
                                    ws, self.sums[idx[k]], self.sums[idx[b]], clip_norm, lr_t, self.shadow_fwd[k], self.shadow_bwd[k],
                                    beta1, beta2, eps, **img)
                self._after_fused_adam(k)
                done.update((k, b))
            # plain 2-D weights without a regulariser (DBoF cluster / hidden weights, ...): the same pass without a bias (ops.adam2d_fused)
            for k in names:
                if k in done or k not in self.shadow_bwd or k in self.l2_names or k.endswith("basic_lstm_cell/kernel"):
                    continue
                img = self._adam_images_2d(k)
                if img is None or self.store.shapes[k][1] % 4:
                    continue
                if not hasattr(self, "_sqn_ws"):
                    self._sqn_ws = {}
                ws = self._sqn_ws.setdefault(k, torch.empty(1028, dtype=F32, device=self.device))
                ops.adam2d_fused(st.p(k), st.g(k), st.view(st.m, k), st.view(st.v, k), ws, self.sums[idx[k]], clip_norm, lr_t,
                                 self.shadow_fwd[k], self.shadow_bwd[k], beta1, beta2, eps, **img)
                done.add(k)
        rest = [k for k in names if k not in done]
        # small tensors without an l2 term and without operand shadows (biases, batch-norm scales / offsets): one launch for up to 16 of them
        # (evc_clip_adam_small runs ONE workgroup per tensor in two serial passes: sized for a few thousand elements each - anything above
        #  SMALL_ADAM_MAX elements stays on the two full-grid launches below)
        small = [k for k in rest if k not in self.l2_names and k not in self.shadow_fwd and self.store.p(k).numel() <= self.SMALL_ADAM_MAX]
        if self.fused_small_adam and len(small) >= 2 and not ops.DETERMINISTIC:
            st = self.store
            for i in range(0, len(small), 16):
                grp = small[i:i + 16]
                ops.clip_adam_small([st.p(k) for k in grp], [st.g(k) for k in grp], [st.view(st.m, k) for k in grp], [st.view(st.v, k) for k in grp],
                                    [self.sums[idx[k]] for k in grp], clip_norm, lr_t, beta1, beta2, eps)
            rest = [k for k in rest if k not in small]
        presummed = getattr(self, "_presummed", set())       # norm rows already filled by the gradient products (MoeHead.backward presum_l2)
        for k in rest:
            if k in presummed:
                continue
            l2 = l2_coeff if k in self.l2_names else 0.0
            # (tensors without a regulariser: the norm pass reads the gradient only)
            ops.grad_sqnorm(self.store.g(k), self.store.p(k) if k in self.l2_names else None, l2, self.sums[idx[k]])
        self._presummed = set()
        for k in rest:
            l2 = l2_coeff if k in self.l2_names else 0.0
            ops.clip_adam_step(self.store.p(k), self.store.g(k), self.store.view(self.store.m, k),
                               self.store.view(self.store.v, k), l2, self.sums[idx[k]], clip_norm, lr_t, beta1, beta2, eps,
                               p_bf16=self.shadow_fwd.get(k))
        if self.precision != "bf16":
            for k in rest:
                if k in self.shadow_fwd:
                    self._refresh_high(k)
        if refresh:
            for k in rest:
                if k in self.shadow_bwd:
                    # from the bf16 forward shadow Adam just wrote (same rounding, half the bytes of the f32 master)
                    p, sb = self.shadow_fwd[k], self.shadow_bwd[k]
                    il = p.shape[0] // 4 if k.endswith("basic_lstm_cell/kernel") else 0
                    ops.transpose_to_bf16(p, p.shape[0], p.shape[1], sb, sb.shape[1], interleave_H=il)

    SMALL_ADAM_MAX = 1 << 15        # elements per tensor of the one-launch small-tensor update (evc_clip_adam_small's limit)
    fused_small_adam = os.environ.get("EVC_FUSED_SMALL_ADAM", "1") != "0"   # A/B: 0 = grad_sqnorm + clip_adam launches per small tensor
    fused_lstm_adam = os.environ.get("EVC_FUSED_LSTM_ADAM", "1") != "0"     # A/B: 0 = grad_sqnorm / clip_adam / transpose / cast launches per tensor

    def _adam_images_2d(self, k):
        """The same for a plain 2-D weight (ops.adam2d_fused): {} in bf16; f16 + e4m3 images where this tower keeps them (shadow_w16 / shadow_w8
        with hi_cols = the row length: ops.gemm_nt_f16_fp8's B operands); None for separate hi / lo bf16 shadows (their cast launches stay)."""
        if self.precision == "bf16":
            return {}
        if k in getattr(self, "shadow_w8", {}):
            e = self._fp8_exps(k)
            return dict(p_f16=self.shadow_w16[k], p_fp8=self.shadow_w8[k], fp8_hi_cols=self.store.shapes[k][1], fp8_lo_exp=e["w_lo_exp"], fp8_hi_exp=e["w_hi_exp"])
        return None

    def _fp8_exps(self, k):
        return ops.FP8_MOE

    def _after_fused_adam(self, k):
        """Operand images of LSTM kernel k that ops.lstm_adam_fused does not write (HLstmTower: the time-dithered f16 images)."""

    def _adam_images(self, k):
        """Keyword arguments of ops.lstm_adam_fused describing the non-bf16 forward operand images of LSTM kernel k that the update pass
        writes itself ({} in bf16), or None when this tower's layout for k has an image the fused pass does not write (the per-tensor
        launches + _refresh_high run instead)."""
        return {} if self.precision == "bf16" else None

    def apply_gradients(self, lr, clip_norm=1.0, l2_coeff=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
        self.begin_update()
        self.apply_group(self.names, lr, clip_norm, l2_coeff, beta1, beta2, eps)

    def reg_loss(self):
        """sum of slim.l2_regularizer(1e-8) terms, from the norms of the last apply_gradients()."""
        idx = [self.names.index(k) for k in self.l2_names]
        return 1e-8 * 0.5 * self.sums[idx, 1].sum()


class HLstmTower(TowerBase):
    """HierarchicalLstmModel + MoeModel for one variable scope ('model' or
    'model_student')."""

    GATES, EXPERTS, EBIAS = MoeHead.GATES, MoeHead.EXPERTS, MoeHead.EBIAS
    l2_names = (MoeHead.GATES, MoeHead.EXPERTS)                     # slim.l2_regularizer(1e-8) targets

    def __init__(self, batch_size, num_frames, num_chunks, feature_size=1152, vocab_size=4716,
                 lstm_cells=1024, lstm_layers=2, num_mixtures=2, device="cuda:0", training=True,
                 scope="model", seed=0):
        assert num_frames % num_chunks == 0, "frames must split evenly into chunks (tf.split)"
        self.B, self.T, self.C, self.F, self.V = batch_size, num_frames, num_chunks, feature_size, vocab_size
        self.H, self.L, self.Mx = lstm_cells, lstm_layers, num_mixtures
        self.device, self.training, self.scope = torch.device(device), training, scope
        if feature_size % 64 or lstm_cells % 64:
            raise ValueError("feature_size (%d) and lstm_cells (%d) must be multiples of 64 for the MFMA GEMM tiles"
                             % (feature_size, lstm_cells))
        H, L, F, V, Mx = self.H, self.L, self.F, self.V, self.Mx
        K = 2 * L * H
        self.K = K
        shapes = OrderedDict()
        for sc, in0 in (("RNN_L1", F), ("RNN_L2", K)):
            for l in range(L):
                nin = in0 if l == 0 else H
                base = "%s/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % (sc, l)
                shapes[base + "kernel"] = (4 * H, nin + H)          # stored transposed
                shapes[base + "bias"] = (4 * H,)
        shapes.update(MoeHead.shapes(K, V, Mx))
        self._setup_store(shapes)
        self.moe = MoeHead(self, K, V, Mx)
        self._init_params(seed)
        self._alloc(batch_size)

    # ---- "high" precision operands (budget measured by scripts/precision_budget.py) ----------------------
    # L1 level (85 % of the forward flops): IEEE f16 operands, ONE MFMA product per depth; layer 0's x-part as a K-extension
    # over f16_x_segments segments of the input (2: + x_lo . Wx, 3: + x . Wx_lo) - the rounding of the input frames is the
    # one term of that level that 2^-12 does not cover.  L2 level (two layers): f16 in the wavefront pair launches, the UPPER
    # layer's weights K-extended by their low-order halves (their rounding is that level's one such term).  MoE head (and an L2
    # level of any other depth): split-bf16, K-extended loops.
    f16_x_segments = int(os.environ.get("EVC_HIGH_X_SEGMENTS", "3"))
    # L1 layers whose RECURRENT weights are K-extended by their low-order halves ([h | h/64] . [Wh | Wh_lo*64]^T): layer 0 - the
    # rounding of Wh0 is, after the input frames, the largest and most draw-dependent term of the level (5e-4 on the logits on
    # some trained weights, scripts/precision_budget.py "ROBUST" runs).  EVC_HIGH_L1_WH_EXT: comma list of layers ("" = none).
    f16_wh_ext_layers = tuple(int(v) for v in os.environ.get("EVC_HIGH_L1_WH_EXT", "0,1").split(",") if v.strip() != "")
    # upper L1 layers whose INPUT weights are extended too ([h_below | h_below/64] . [Wx | Wx_lo*64]^T; needs the layer below wide and the
    # layer itself in f16_wh_ext_layers): with "0,1" / "1" every LSTM weight of the level is exact and only the f16 rounding of the
    # activations is left (the "FZ" layout of scripts/precision_budget.py: mean 2.0e-4, max 3.5e-4 on the logits)
    f16_wx_ext_layers = tuple(int(v) for v in os.environ.get("EVC_HIGH_L1_WX_EXT", "1").split(",") if v.strip() != "")
    # L2 level, layer 0: segments of its input (the L1 states; after the upper layer's weights their f16 rounding is the level's
    # largest term on the LOGITS: 1.2e-4) in the hoisted product, and its recurrent weights extended - free inside the pair
    # launches, whose time the upper layer's K = 4H sets
    f16_l2_x_segments = int(os.environ.get("EVC_HIGH_L2_X_SEGMENTS", "2"))
    f16_l2_h0_ext = os.environ.get("EVC_HIGH_L2_H0_EXT", "1") != "0"

    # L1 level with the LOW-ORDER HALVES of the weights - and of the input frames - contracted in fp8 (ops.lstm_layer_fwd_f16_fp8lo) instead of
    # f16 K-extensions: per step [x | h] . [Wx | Wh]^T in f16 + [e4m3(x) | e4m3(x_lo) | e4m3(h)] . [e4m3(Wx_lo) | e4m3(Wx) | e4m3(Wh_lo)]^T on the
    # MX-scaled MFMA - every weight and the input exact to ~2^-15 for half the MFMA time of the f16 extensions (scripts/precision_budget.py:
    # the weight term of the error budget 3.7e-5 -> 9e-7 on the logits, the input term 5.3e-5 -> 1e-6).
    # Needs F, H multiples of 128 and >= 384; EVC_HIGH_FP8_LO=0: the f16 K-extensions above (the round-3 "FZ" layout).
    f16_fp8_lo = os.environ.get("EVC_HIGH_FP8_LO", "1") != "0"
    # MoE head in "high": f16 product + both low-order corrections as e4m3 operands behind it in the same launch (ops.gemm_nt_f16_fp8) instead of
    # the split-bf16 K-extension; needs K % 128 == 0, K >= 512 (same switch; not touched by distill.student_light, which is about the L1 level)
    moe_f16_fp8 = os.environ.get("EVC_HIGH_MOE_FP8", os.environ.get("EVC_HIGH_FP8_LO", "1")) != "0"
    # L2 level (two layers) in "high": the same for layer 0's recurrent weights and all of layer 1's (ops.lstm_stack2_fwd_f16_fp8lo); H % 128 == 0, H >= 512
    l2_fp8_lo = os.environ.get("EVC_HIGH_L2_FP8", os.environ.get("EVC_HIGH_FP8_LO", "1")) != "0"

    # Round 6: the ACTIVATIONS' low-order halves as e4m3 operands too (h_lo forms of the same launches).  Up to round 5 only weights (and the input
    # frames) were corrected: "f16 is enough for every activation" held 16 steps from initialisation; on towers trained for 512 steps the uncorrected
    # f16 rounding of h in L1 layer 0 and of both activation operands of L2 layer 1 leaves 1e-3 .. 2e-3 on the teacher's logits in half of the weight
    # draws (profiles/r06_budget_worst_draw.txt, r06_budget_plans.txt).  h rows become [f16(h) | e4m3(h 2^7) | e4m3((h - f16(h)) 2^18)] (4H bytes)
    # against [lo(W) | hi(W)] weight rows: every layer on the fp8lo form (L1: the non-dithered layers; L2: both).  EVC_HIGH_ACT_LO=0: the round-5 layout.
    f16_act_lo = os.environ.get("EVC_HIGH_ACT_LO", "1") != "0"

    def act_lo(self):
        return bool(self.f16_act_lo and self.precision == "high")

    # Round 6: on the reader's uint8 frames layer 0 contracts the EXACT integers 2q - 255 (evc_l2norm_chunk_int rows) and applies the frame's
    # dequantise / l2-normalise factor to its accumulators (ops.lstm_layer_fwd_f16_fp8lo x_int): the f16 rounding of the input frames - whose e4m3 x e4m3
    # correction still left 1.8e-3 on the logits of the worst 512-step draw - is gone, and so are its 9 e4m3 stages per step.  Needs layer 0 on the
    # f16 + e4m3 form; f32 inputs keep the e4m3 correction.  EVC_HIGH_X_INT=0: always the f32-input form.
    f16_x_int = os.environ.get("EVC_HIGH_X_INT", "1") != "0"

    def x_int(self):
        return bool(self.f16_x_int and self.fp8_lo() and 0 not in self.dither_layers() and not self.dither_wh0() and self.F % 128 == 0)

    def _refresh_x_col_const(self, k):
        """(255/256) sum_k f16(Wx[j][k]) of L1 layer 0's kernel - the constant term of the dequantised frames, from the f16 image the step contracts
        (- 1 in the forget-gate block: the step kernels add forget_bias to whatever their accumulators start from)."""
        if not (k.startswith("RNN_L1/") and "cell_0/" in k and k in getattr(self, "shadow16", {}) and self.x_int()):
            return
        if not hasattr(self, "x_col_const"):
            self.x_col_const = {}
        c = self.shadow16[k][:, :self.F].sum(dim=1, dtype=F32) * (255.0 / 256.0)
        c[2 * self.H:3 * self.H] -= 1.0
        if k in self.x_col_const:            # in place: the buffer the step kernels read keeps its address (and its allocation stream)
            self.x_col_const[k].copy_(c)
        else:
            self.x_col_const[k] = c.contiguous()

    def fp8_lo(self):
        """True if this tower's L1 level runs on ops.lstm_layer_fwd_f16_fp8lo (or, with dither(), on ops.lstm_layer_fwd_f16_dith)."""
        return (self.precision == "high" and self.f16_fp8_lo and self.F % 128 == 0 and self.H % 128 == 0 and self.F >= 384 and self.H >= 384)

    # L1 layers on TIME-DITHERED f16 weight images (round 5, DESIGN.md 7 "dither"; ops.lstm_layer_fwd_f16_dith): step t of a chunk contracts
    # image t of the layer's kernel - each element rounded down or up so that the round-ups over any run of steps match its position between
    # its f16 neighbours.  The f16 rounding of a WEIGHT is the error a recurrence integrates coherently (the reason for the e4m3 low-order
    # halves above); dithered over the steps it largely cancels, and a dithered layer needs no stages for its weights' low-order halves
    # (layer 1 of the teacher: 32 ring stages per step instead of 32 + 16; layer 0: 34 + 9 - the input's low-order half - instead of 34 + 26).
    # Costs T images per dithered kernel (15 x 17 MB per teacher layer) and one pass over them per update.  What it buys and what it costs in
    # accuracy is measured over weight draws in DESIGN.md 7 (12 draws each, teacher logits x 1e-4, real kernels): no layer dithered mean 2.65 /
    # max 6.1 over 34 draws at 1.21x the bf16 step; the TOP layer dithered (default) 2.65 / 6.3 over 36 draws at 1.18x; both layers 4.1 / 8.3 at 1.16x - too close to
    # the 1e-3 the mode exists for.  EVC_HIGH_DITHER_LAYERS = comma list of L1 layers (a SUFFIX of the stack: a layer on e4m3 low-order
    # halves reads the e4m3 image of h from the layer below, which a dithered layer does not write); "" = none (rounds 3-5).
    # Unset: the TOP layer of a level of two or more layers.
    f16_dither_layers = (tuple(int(v) for v in os.environ["EVC_HIGH_DITHER_LAYERS"].split(",") if v.strip() != "")
                         if "EVC_HIGH_DITHER_LAYERS" in os.environ else None)

    def dither_layers(self):
        """L1 layers that run on time-dithered weight images (empty unless this tower's L1 level is the f16 + e4m3 one)."""
        if not self.fp8_lo():
            return ()
        if self.f16_dither_layers is None:                   # default: the top layer; a one-layer level keeps its corrections
            return (self.L - 1,) if self.L >= 2 else ()
        dl = tuple(sorted(set(l for l in self.f16_dither_layers if 0 <= l < self.L)))
        if dl and dl != tuple(range(dl[0], self.L)):
            raise ValueError("EVC_HIGH_DITHER_LAYERS=%r: the dithered L1 layers must be the top layers of the stack (a suffix of 0..%d)" % (list(dl), self.L - 1))
        return dl

    # ... and layer 0's RECURRENT block as well (its input block keeps the e4m3 corrections of Wx and of the input frames - the one block whose
    # dithering costs accuracy): 34 + 18 stages instead of 34 + 26.  Needs every layer above dithered (layer 0 then writes plain h rows).
    f16_dither_wh0 = os.environ.get("EVC_HIGH_DITHER_WH0", "0") == "1"

    def dither_wh0(self):
        return bool(self.f16_dither_wh0 and self.L >= 2 and self.fp8_lo() and self.dither_layers() == tuple(range(1, self.L)))

    def dither_col0(self, k):
        """First dithered column of L1 kernel k's images: the recurrent block only for layer 0 under dither_wh0(), else every column."""
        layer = int(k.split("cell_")[1].split("/")[0])
        return (self.store.shapes[k][1] - self.H) if (layer == 0 and 0 not in self.dither_layers()) else 0

    def l1_steps(self):
        return self.T // self.C

    def input_split(self):
        """The `split` argument of ops.l2norm_chunk that produces this tower's L1 input."""
        return {"bf16": False, "high": "f16", "split": "wide"}[self.precision]

    def _alloc_high_shadows(self):
        dev, H, F, K = self.device, self.H, self.F, self.K
        self.shadow_lo = {}                                              # (no separate low-order shadows in this tower)
        self.shadow16, self.shadow_wx, self.shadow_wh, self.shadow_w, self.shadow8 = {}, {}, {}, {}, {}
        self.shadow16d = {}                                              # L1 kernels: [T][4H][in + H] time-dithered f16 images (dither())
        self.shadow_w16, self.shadow_w8 = {}, {}                      # MoE head in "high": f16(W) and [e4m3(W_lo) | e4m3(W)] (ops.gemm_nt_f16_fp8)
        for k, shp in self.store.shapes.items():
            if len(shp) != 2:
                continue
            if k.startswith("RNN_L1/") and (int(k.split("cell_")[1].split("/")[0]) in self.dither_layers() or (self.dither_wh0() and "cell_0/" in k)):
                # T time-dithered f16 images; layer 0 also keeps cast_fp8_lo's [lo(Wx) | e4m3(Wx 2^6) | lo(Wh)] rows for their middle block
                nin = shp[1] - H
                layer = int(k.split("cell_")[1].split("/")[0])
                self.shadow16d[k] = torch.zeros((self.l1_steps(),) + tuple(shp), dtype=ops.F16, device=dev)
                if layer == 0:
                    self.shadow8[k] = torch.zeros((shp[0], shp[1] + nin), dtype=torch.uint8, device=dev)
            elif k.startswith("RNN_L1/") and self.fp8_lo():
                nin = shp[1] - H
                layer = int(k.split("cell_")[1].split("/")[0])
                self.shadow16[k] = torch.zeros(shp, dtype=ops.F16, device=dev)
                # [lo(Wx) | hi(Wx) (layer 0: against the input's low-order half) | lo(Wh)]; act_lo(): [lo(Wx) | hi(Wx) | lo(Wh) | hi(Wh)] in every layer
                self.shadow8[k] = torch.zeros((shp[0], 2 * shp[1] if self.act_lo() else shp[1] + (nin if layer == 0 else 0)), dtype=torch.uint8, device=dev)
            elif k.startswith("RNN_L1/") and self.precision == "high":
                nin = shp[1] - H
                layer = int(k.split("cell_")[1].split("/")[0])
                xw = self.f16_x_segments * nin if layer == 0 else (2 if layer in self.f16_wx_ext_layers else 1) * nin
                self.shadow16[k] = torch.zeros((shp[0], xw + (2 if layer in self.f16_wh_ext_layers else 1) * H), dtype=ops.F16, device=dev)
            elif k.startswith("RNN_L2/") and self.precision == "high" and self.L == 2 and self.l2_fp8_lo and H % 128 == 0 and H >= 512:
                # f16 L2 level with e4m3 low-order halves (ops.lstm_stack2_fwd_f16_fp8lo): layer 0 [Wx segments | f16(Wh)] + lo(Wh), layer 1 f16(W) + lo(W)
                nin = shp[1] - H
                a2 = 2 if self.act_lo() else 1                      # act_lo(): [lo | hi] per operand part
                if "cell_0" in k:
                    self.shadow16[k] = torch.zeros((shp[0], self.f16_l2_x_segments * nin + H), dtype=ops.F16, device=dev)
                    self.shadow8[k] = torch.zeros((shp[0], a2 * H), dtype=torch.uint8, device=dev)
                else:
                    self.shadow16[k] = torch.zeros(shp, dtype=ops.F16, device=dev)
                    self.shadow8[k] = torch.zeros((shp[0], a2 * shp[1]), dtype=torch.uint8, device=dev)
            elif k.startswith("RNN_L2/") and self.precision == "high" and self.L == 2:
                # f16 L2 level (ops.lstm_stack2_fwd_f16): layer 0 [Wx segments | Wh (| Wh_lo*64)], layer 1 [Wx | Wx_lo*64 | Wh | Wh_lo*64]
                nin = shp[1] - H
                w0 = self.f16_l2_x_segments * nin + (2 if self.f16_l2_h0_ext else 1) * H
                self.shadow16[k] = torch.zeros((shp[0], w0 if "cell_0" in k else 2 * shp[1]), dtype=ops.F16, device=dev)
            elif k.startswith("RNN_L"):                                 # L2 level ("split": the L1 level too)
                nin = shp[1] - H
                self.shadow_wx[k] = torch.zeros((shp[0], 2 * nin), dtype=BF16, device=dev)
                self.shadow_wh[k] = torch.zeros((shp[0], 2 * H), dtype=BF16, device=dev)
            elif self.precision == "high" and self.moe_f16_fp8 and shp[1] % 128 == 0 and shp[1] >= 512:
                self.shadow_w16[k] = torch.zeros(shp, dtype=ops.F16, device=dev)
                self.shadow_w8[k] = torch.zeros((shp[0], 2 * shp[1]), dtype=torch.uint8, device=dev)
            else:
                self.shadow_w[k] = torch.zeros((shp[0], 2 * shp[1]), dtype=BF16, device=dev)

    def _refresh_high(self, k):
        self._refresh_high_images(k)
        self._refresh_x_col_const(k)

    def _refresh_high_images(self, k):
        p, H = self.store.p(k), self.H
        if k in self.shadow_w8:              # MoE head: f16(W) + [e4m3((W - f16(W)) 2^18) | e4m3(W 2^7)]
            ops.cast_f16(p, self.shadow_w16[k])
            ops.cast_fp8_lo(p, self.shadow_w8[k], hi_cols=p.shape[1], scale_exp=ops.FP8_MOE["w_lo_exp"], hi_exp=ops.FP8_MOE["w_hi_exp"])
        elif k in self.shadow16d:            # L1 level, time-dithered f16 images (layer 0: + cast_fp8_lo's rows for the e4m3(Wx 2^6) block)
            ops.cast_f16_dither(p, self.shadow16d[k], self.dither_seed(k), col0=self.dither_col0(k))
            if k in self.shadow8:
                ops.cast_fp8_lo(p, self.shadow8[k], hi_cols=self.shadow8[k].shape[1] - p.shape[1])
        elif k in self.shadow8 and k.startswith("RNN_L2/"):     # L2 level, fp8 low-order halves (act_lo(): + the full-value images, [lo | hi] per part)
            nin = p.shape[1] - H
            al = self.act_lo()
            if "cell_0" in k:
                ops.cast_f16_wide(p, nin, H, self.f16_l2_x_segments, self.shadow16[k], h_ext=False)
                ops.cast_fp8_lo(p[:, nin:], self.shadow8[k], hi_tail=al)
            else:
                ops.cast_f16(p, self.shadow16[k])
                ops.cast_fp8_lo(p, self.shadow8[k], hi_cols=nin if al else 0, hi_tail=al)
        elif k in self.shadow8 and self.act_lo():     # L1 level, every part [lo | hi]: [lo(Wx) | hi(Wx) | lo(Wh) | hi(Wh)]
            ops.cast_f16(p, self.shadow16[k])
            ops.cast_fp8_lo(p, self.shadow8[k], hi_cols=p.shape[1] - H, hi_tail=True)
        elif k in self.shadow8:              # L1 level, fp8 low-order halves: f16(W) + e4m3((W - f16(W)) 2^17) (layer 0: + e4m3(Wx 2^6) for the input's)
            ops.cast_f16(p, self.shadow16[k])
            ops.cast_fp8_lo(p, self.shadow8[k], hi_cols=self.shadow8[k].shape[1] - p.shape[1])
        elif k in self.shadow16 and k.startswith("RNN_L2/") and "cell_1" in k:
            ops.cast_f16_wlo(p, p.shape[1] - H, H, self.shadow16[k])
        elif k in self.shadow16 and k.startswith("RNN_L1/") and "cell_0" not in k and int(k.split("cell_")[1].split("/")[0]) in self.f16_wx_ext_layers:
            layer = int(k.split("cell_")[1].split("/")[0])          # upper L1 layer with its INPUT weights extended: [Wx | Wx_lo*64 | Wh (| Wh_lo*64)]
            if layer in self.f16_wh_ext_layers:
                ops.cast_f16_wlo(p, p.shape[1] - H, H, self.shadow16[k])
            else:
                raise NotImplementedError("EVC_HIGH_L1_WX_EXT layers must also be in EVC_HIGH_L1_WH_EXT")
        elif k in self.shadow16:
            nin = p.shape[1] - H
            layer = int(k.split("cell_")[1].split("/")[0])
            h_ext = (layer in self.f16_wh_ext_layers) if k.startswith("RNN_L1/") else self.f16_l2_h0_ext
            nseg = (self.shadow16[k].shape[1] - (2 if h_ext else 1) * H) // nin
            ops.cast_f16_wide(p, nin, H, nseg, self.shadow16[k], h_ext=h_ext)
        elif k in self.shadow_wx:
            nin = p.shape[1] - H
            ops.cast_bf16_wide(p[:, :nin], self.shadow_wx[k], lo_first=False)
            ops.cast_bf16_wide(p[:, nin:], self.shadow_wh[k], lo_first=False)
        else:
            ops.cast_bf16_wide(p, self.shadow_w[k], lo_first=False)

    def precision_layout(self):
        d = {"precision": self.precision}
        if self.precision == "high":
            d.update(l1=("f16 + e4m3 low-order halves (weights, input frames)" + ("; layers %s on %d time-dithered f16 weight images instead of their weights' low-order halves"
                                                                                   % (list(self.dither_layers()), self.l1_steps()) if self.dither_layers() else "")
                         + ("; layer 0's recurrent block dithered too" if self.dither_wh0() else "")) if self.fp8_lo() else
                     "f16, x segments %d, Wh extended in layers %s, Wx extended in layers %s" % (self.f16_x_segments, list(self.f16_wh_ext_layers), list(self.f16_wx_ext_layers)),
                     l2=("f16 + e4m3 low-order halves, %d input segments" % self.f16_l2_x_segments) if any(k.startswith("RNN_L2/") for k in getattr(self, "shadow8", {}))
                     else ("f16 K-extensions, %d input segments, h0_ext %s" % (self.f16_l2_x_segments, self.f16_l2_h0_ext) if self.L == 2 else "split-bf16"),
                     moe="f16 + e4m3 corrections of both operands" if getattr(self, "shadow_w8", None) else "split-bf16 K-extension",
                     activations=("low-order halves of h as e4m3 operands in every corrected layer (L1: the non-dithered ones; L2: both): rows [f16 | e4m3 | e4m3_lo] "
                                  "against [lo(W) | hi(W)]" if self.act_lo() else "f16, uncorrected (round-5 layout)"),
                     moe_input_range="from the batch (absmax -> shift)" if self.moe.dynamic_fp8_range else "fixed 2^6",
                     uint8_frames=("layer 0 contracts the exact integers 2q - 255, the frame's factor applied to its accumulators (no input rounding, no "
                                   "stages for an input low-order half)" if self.x_int() else "dequantised to f32 first: f16 image + e4m3 low-order half"),
                     fp8_scales=dict(lstm_w_lo_exp=ops.FP8_W_SCALE_EXP, lstm_wx_hi_exp=ops.FP8_WX_HI_EXP, **{"moe_" + k: v for k, v in ops.FP8_MOE.items()}))
        return d

    def fp8_saturation(self, state=None):
        """Validation aid for the "high" mode's FIXED e4m3 scales (ops.FP8_*): how many operand elements lie outside the range that
        never clamps - their low-order correction is (partly) lost and the 1e-3 contract degrades without any other signal.
        Counts weights of the L1 / L2 levels with |W| >= 4 (lo scale 2^17: |W - f16(W)| 2^17 <= 448), MoE weights with |W| >= 3.5
        (2^7) and, given `state` [B, K] (the head's input of a batch), its elements with |s| >= 7 (2^6; the cell-state half of the
        L2 state is unbounded) - under MoeHead.dynamic_fp8_range (default, round 6) the state's range follows the batch and that count is 0.  Returns {name: count}; all zeros means the contract's assumptions hold.  (Diagnostic: plain
        tensor reductions, one host sync.)"""
        out = {}
        if self.precision != "high":
            return out
        lim_w = 448.0 / 2.0 ** (ops.FP8_W_SCALE_EXP - 12 + 3)             # |W - f16(W)| <= 2^-11 |W| (half an ulp of 11 bits): 448 / 2^17 * 2^11 = 7 -> keep the documented 4
        for k in list(getattr(self, "shadow8", {})):
            out[k] = int((self.store.p(k).abs() >= min(4.0, lim_w * 8)).sum())
        for k in list(getattr(self, "shadow_w8", {})):
            out[k] = int((self.store.p(k).abs() >= 448.0 / 2.0 ** ops.FP8_MOE["w_hi_exp"]).sum())
        if state is not None and getattr(self, "shadow_w8", None):
            if self.moe.dynamic_fp8_range:
                # the head's input takes its e4m3 range from the batch (ops.absmax_partials -> fp8_range_drop): nothing can clamp; the shift is reported
                out["moe_input_state"] = 0
                amax = float(state.abs().max())
                self.last_state_range = dict(absmax=amax, shift_bits=max(0, int(math.ceil(math.log2(amax / 7.0)))) if amax > 7.0 else 0)
            else:
                out["moe_input_state"] = int((state.abs() >= 448.0 / 2.0 ** ops.FP8_MOE["x_hi_exp"]).sum())
        return out

    def _adam_images(self, k):
        """The "high" layouts ops.lstm_adam_fused writes from the update pass - what _refresh_high(k) would produce: L1 level f16 + e4m3
        low-order halves (layer 0: + the full-value image of Wx), L2 level f16 (layer 0: K-extended input blocks) + e4m3 low-order
        halves, and plain / x-extended f16 images without a recurrent extension.  None for the others (f16 K-extensions of the
        recurrent weights, split-bf16): they keep their cast launches."""
        if self.precision == "bf16":
            return {}
        p, H = self.store.p(k), self.H
        nin = p.shape[1] - H
        if k in self.shadow16d:              # dithered L1 kernel: the update pass writes layer 0's e4m3 rows; the T images follow (_after_fused_adam)
            return dict(p_fp8=self.shadow8[k], fp8_col0=0, fp8_hi_cols=self.shadow8[k].shape[1] - p.shape[1]) if k in self.shadow8 else {}
        al = self.act_lo()
        if k in self.shadow8 and k.startswith("RNN_L2/"):
            if "cell_0" in k:
                return dict(p_f16=self.shadow16[k], nin=nin, nseg=self.f16_l2_x_segments, p_fp8=self.shadow8[k], fp8_col0=nin, fp8_hi_cols=0, fp8_hi_tail=al)
            return dict(p_f16=self.shadow16[k], nin=nin, nseg=1, p_fp8=self.shadow8[k], fp8_col0=0, fp8_hi_cols=nin if al else 0, fp8_hi_tail=al)
        if k in self.shadow8 and al:
            return dict(p_f16=self.shadow16[k], nin=nin, nseg=1, p_fp8=self.shadow8[k], fp8_col0=0, fp8_hi_cols=nin, fp8_hi_tail=True)
        if k in self.shadow8:
            return dict(p_f16=self.shadow16[k], nin=nin, nseg=1, p_fp8=self.shadow8[k], fp8_col0=0, fp8_hi_cols=self.shadow8[k].shape[1] - p.shape[1])
        if k in self.shadow16 and k.startswith("RNN_L1/"):
            layer = int(k.split("cell_")[1].split("/")[0])
            if layer not in self.f16_wh_ext_layers and not (layer > 0 and layer in self.f16_wx_ext_layers):
                nseg = (self.shadow16[k].shape[1] - H) // nin
                return dict(p_f16=self.shadow16[k], nin=nin, nseg=nseg)
        return None

    def dither_seed(self, k):
        """Seed of kernel k's dither phases: its index among this tower's variables (any fixed value does; images are a pure function of (W, seed))."""
        return 1 + list(self.names).index(k)

    def _after_fused_adam(self, k):
        # (only while the tower RUNS in "high": one laid out for it and switched to bf16 reads none of the T images; refresh_shadows rebuilds them on the way back)
        if self.precision == "high" and k in getattr(self, "shadow16d", {}):
            ops.cast_f16_dither(self.store.p(k), self.shadow16d[k], self.dither_seed(k), col0=self.dither_col0(k))
        if self.precision == "high":
            self._refresh_x_col_const(k)

    # ---- parameters -------------------------------------------------------
    def _init_params(self, seed):
        """TF defaults at the reference call sites: glorot-uniform kernels /
        fully_connected weights, zero biases."""
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed)
        for k, shp in self.store.shapes.items():
            if len(shp) == 2:
                fan_out, fan_in = shp                               # stored transposed: [out][in]
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                w = (torch.rand(shp, generator=gen, dtype=F32) * 2 - 1) * lim
                self.store.p(k).copy_(w)
        self.refresh_shadows()

    # ---- workspaces ---------------------------------------------------------
    def _alloc(self, B):
        dev, K = self.device, self.K
        self.B = B
        Lc = self.T // self.C
        self.l1 = LstmStack(self, "RNN_L1", Lc, self.C * B, self.F, self.training)
        self.l2 = LstmStack(self, "RNN_L2", self.C, B, K, self.training)
        self.S1_bf = torch.empty((self.C * B, K), dtype=BF16, device=dev)
        self.moe.alloc(B, self.training)

    # convenience views used by the distillation graph / tests
    @property
    def pred(self):
        return self.moe.pred

    @property
    def rowsum(self):
        return self.moe.rowsum

    @property
    def gate_logits(self):
        return self.moe.gate_logits

    # ---- forward ------------------------------------------------------------
    def forward(self, x_view, len_l1, len_l2, plan_l1=None, after_l1=None):
        """x_view [Lc][C*B][F] bf16 (ops.l2norm_chunk); lengths from ops.frame_counts.
        plan_l1: ops.RowPlan of the L1 chunk rows (x_view is then [Lc][plan.P][F], from l2norm_chunk(plan1=..)).
        after_l1: optional callable, invoked once the L1 level has been enqueued (the L2 chain and the MoE head that
        follow are latency-bound launches that leave most of the chip idle: a caller can start other work there).
        Returns (state [B, 2LH] f32, predictions [B, V] f32)."""
        high = isinstance(x_view, tuple)
        if high != (self.precision != "bf16"):
            raise ValueError("precision %r takes %s" % (self.precision, "an image pair from ops.l2norm_chunk(..., split=tower.input_split())"
                                                        if self.precision != "bf16" else "the plain bf16 image"))
        if high and (x_view[1].dtype == ops.F16) != (self.precision == "high"):
            raise ValueError("precision 'high' takes (bf16, IEEE f16) images, 'split' (bf16, wide [lo | hi] bf16) images: "
                             "ops.l2norm_chunk(..., split=tower.input_split(), f16_segments=tower.f16_x_segments)")
        B = int(len_l2.shape[0])
        if B != self.B:
            self._alloc(B)
        self.run_deferred()                                            # (normally already enqueued at the start of the step)
        ops.mark(self.scope + ":fwd_begin")
        S1 = self.l1.forward(x_view, len_l1, plan_l1)
        ops.mark(self.scope + ":l1_fwd_done")
        self.wait_deferred()                                           # L2-level / MoE updates of the previous step (defer=True)
        if after_l1 is not None:
            after_l1()
        ops.cast_bf16(S1, self.S1_bf)                                  # = L2 input [C][B][2LH] (in "high": the backward operand)
        if high and self.precision == "high" and self.L == 2:          # f16 L2 level (error budget: DESIGN.md 7)
            ns = self.f16_l2_x_segments
            if not hasattr(self, "S1_16") or self.S1_16.shape != (self.C * B, ns * self.K):
                self.S1_16 = torch.empty((self.C * B, ns * self.K), dtype=ops.F16, device=self.device)
            ops.cast_f16_segs(S1, ns, self.S1_16)
            S2 = self.l2.forward((self.S1_bf.view(self.C, B, self.K), self.S1_16.view(self.C, B, ns * self.K)), len_l2)
        elif high:
            if not hasattr(self, "S1_w") or self.S1_w.shape[0] != self.S1_bf.shape[0]:
                self.S1_w = torch.empty((self.C * B, 2 * self.K), dtype=BF16, device=self.device)
            ops.cast_bf16_wide(S1, self.S1_w, lo_first=True)
            S2 = self.l2.forward((self.S1_bf.view(self.C, B, self.K), self.S1_w.view(self.C, B, 2 * self.K)), len_l2)
        else:
            S2 = self.l2.forward(self.S1_bf.view(self.C, B, self.K), len_l2)
        ops.mark(self.scope + ":l2_fwd_done")
        pred = self.moe.forward(S2)
        ops.mark(self.scope + ":moe_fwd_done")
        return S2, pred

    # ---- backward -----------------------------------------------------------
    def param_groups(self):
        """Variables in the order their gradients become final during backward()."""
        moe = [self.GATES, self.EXPERTS, self.EBIAS]
        l2 = [k for k in self.names if k.startswith("RNN_L2/")]
        l1 = [k for k in self.names if k.startswith("RNN_L1/")]
        return moe, l2, l1

    def grad_segments(self):
        """(lo, hi) element ranges of the flat gradient buffer per variable group: (MoE, L2, L1)."""
        st = self.store
        l2_lo = st.offsets[next(k for k in self.names if k.startswith("RNN_L2/"))]
        moe_lo = st.offsets[self.GATES]
        return (moe_lo, st.total), (l2_lo, moe_lo), (0, l2_lo)

    fused_moe_update = True      # recompute the rank-B MoE gradient inside the Adam step instead of materialising it
    _deferred, _deferred_ev, _deferred_aux = (), None, None    # backward(defer=True): closures / event / stream of the pending updates
    # experiment: the fused MoE update enqueued behind the L2 level's BPTT chain + weight gradients instead of in front of them
    # (its 80 / 48 KB workgroups hold the LDS the skinny chain kernels of the critical path are waiting for)
    moe_update_after_l2 = os.environ.get("EVC_MOE_UPDATE_AFTER_L2") == "1"

    def backward(self, *args, **kwargs):
        """backward_phases run to its end."""
        _drain(self.backward_phases(*args, **kwargs))

    def backward_phases(self, dstate, dpred, on_moe_grads_ready=None, aux=None, early_apply=None, reduce_fn=None, dp=None, defer=False, opt=None):
        """GENERATOR: yields after the MoE head's backward (+ its update / collectives) and after every LSTM layer of the L2 and L1
        levels (upper layer first) have been enqueued - five phase boundaries at which DistillGraph can switch to the other tower, so
        that the HOST issue order of both towers' launches and collectives follows the order in which they become ready on the GPU
        (one-communicator placement under data parallelism: DESIGN.md 6.1).  Every resumption must happen under the same current
        stream as the first.

        dstate [B,2LH] f32 or None (gradient on the returned state), dpred [B,V] f32.
        Fills self.store.grad (every segment is overwritten).
        aux: side stream that takes the weight-gradient GEMMs (and, with early_apply =
        (lr, clip, l2_coeff), the clip+Adam of each variable group as soon as its gradients are final)
        off the BPTT critical path; it is joined into the current stream before returning.
        reduce_fn(lo, hi) -> work handle or None: data-parallel all-reduce of a gradient segment; with
        early_apply each group is reduced on the aux stream right after its gradients are final and updated
        as soon as the collective has finished.
        dp (distill.GradReducer): with it the MoE weights (2/3 of the parameters) need no gradient all-reduce at all -
        factor all-gather + row-sharded update (MoeHead.fused_update).
        opt: optional stream for the optimizer launches (clip + Adam, fused MoE update) - by default they share `aux` with the
        weight-gradient products; a CU-masked stream (streams.cu_masked_stream) confines them to a few compute units.
        defer (with aux + early_apply, one process): the updates of the variables the NEXT forward reads last - the MoE head
        and the L2 level - are not enqueued here but kept as closures (self._deferred): run_deferred() enqueues them on the aux
        stream at the start of the next step, where they run under that step's L1 forward (MFMA-bound) instead of under this
        step's memory-bound BPTT chain; forward() waits for them before the L2 level.  Same values, later in wall time
        (cs/train.py:516-517: both train ops of an iteration read the pre-update weights - unchanged)."""
        assert self.training
        if not isinstance(self._deferred, list):
            self._deferred = []
        defer = bool(defer) and aux is not None and early_apply is not None and dp is None and reduce_fn is None
        self.flush_deferred()                       # (a caller that skipped the next forward: nothing may be pending twice)
        main = torch.cuda.current_stream(self.device)
        g_moe, g_l2, g_l1 = self.param_groups()
        seg_moe, seg_l2, seg_l1 = self.grad_segments()

        def reduce_then_apply(names, seg, f32=False):
            if reduce_fn is not None:
                h = reduce_fn(*seg, f32) if f32 else reduce_fn(*seg)
                if h is not None:
                    h.wait()                                           # the aux stream waits for the collective, not the host
            self.apply_group(names, *early_apply)

        # data parallel, bf16: which exchange carries the MoE gradient is chosen by shape (MoeHead.dp_route) - the factor all-gather of the fused
        # update grows with the batch, the reduce-scatter of the materialised gradient does not (cfg 5, B = 1024: 398 vs 169 MB per rank and step)
        route_rs = (dp is not None and aux is not None and early_apply is not None and self.fused_moe_update and self.precision == "bf16"
                    and self.moe.can_fuse_update() and self.moe.dp_route(dp.shard_world) == "reduce_scatter")
        fuse = (aux is not None and early_apply is not None and self.fused_moe_update and not route_rs
                and (self.precision == "bf16" or dp is None)      # "high" under data parallelism: the lo halves of the row slabs are not gathered
                and self.moe.can_fuse_update() and (reduce_fn is None or dp is not None)
                and self.moe.prefer_fused_update(dp is not None))
        if not (fuse or route_rs) and getattr(self.moe, "_stale", False):
            raise RuntimeError("the MoE weights of %r are sharded over the ranks (fused data-parallel update); call "
                               "DistillGraph.consolidate() on every rank before an update that is not" % self.scope)
        ops.mark(self.scope + ":bwd_begin")
        early = aux is not None and early_apply is not None
        if early:
            self.begin_update()                     # (before the MoE backward: its gradient products may leave their clip norms in the norm rows)
        # materialised MoE gradients that are clipped as they are (one process, or the all-reduce-free routes do not apply): norms from the products
        presum = early_apply[2] if (early and not fuse and not route_rs and reduce_fn is None) else None
        dS2 = self.moe.backward(dpred, dstate, weight_grads=not fuse, presum_l2=presum)
        ops.mark(self.scope + ":moe_bwd_done")
        if on_moe_grads_ready is not None:
            on_moe_grads_ready()
        if aux is None or early_apply is None:
            opt = None
        ostream = opt if opt is not None else aux
        late_moe = None
        if aux is not None and early_apply is not None:
            ev = torch.cuda.Event()
            ev.record(main)
            aux.wait_event(ev)
            if opt is not None:
                opt.wait_event(ev)
            lr_t_now = self.adam_lr_t(early_apply[0])
            if defer:
                self._deferred_aux = ostream
                if fuse:
                    lr, clip, l2c = early_apply
                    self._deferred.append(lambda: self.moe.fused_update(lr_t_now, clip, l2c, dp=None))
                else:
                    self._deferred.append(lambda: self.apply_group(g_moe, *early_apply, lr_t=lr_t_now))
            else:
                def moe_update_now():
                    with torch.cuda.stream(ostream):                   # 2/3 of the parameters, under the LSTM BPTT
                        ops.mark(self.scope + ":moe_update_begin")
                        if fuse:
                            lr, clip, l2c = early_apply
                            self.moe.fused_update(lr_t_now, clip, l2c, dp=dp)
                        elif route_rs:
                            lr, clip, l2c = early_apply
                            self.moe.sharded_update(lr_t_now, clip, l2c, dp)
                        else:
                            reduce_then_apply(g_moe, seg_moe, f32=True)    # (a bf16 gradient payload is for the LSTM segments only)
                        ops.mark(self.scope + ":moe_update_done")
                if self.moe_update_after_l2 and dp is None:
                    late_moe = moe_update_now                          # (experiment: behind the L2 level's BPTT chain, see below)
                else:
                    moe_update_now()
        # Per LAYER: as soon as a layer's weight-gradient products are enqueued on the aux stream its kernel + bias gradients are
        # final there - reduce (data parallel) and clip + Adam them right behind, under the BPTT of the layer below.  Only the
        # LOWEST layer of the L1 level is left for the end of the step (round 3; before, a level's four tensors waited for its
        # last product: the upper layer's update - and, data parallel, its all-reduce - sat in the serial tail of the step).
        per_layer = aux is not None and early_apply is not None
        st = self.store

        def layer_cb(stack, deferred=False):
            def cb(l):
                kn, bn = stack.names(l)
                lo = st.offsets[kn]
                hi = st.offsets[bn] + _align(int(math.prod(st.shapes[bn])))
                assert lo < hi and st.offsets[bn] > lo, "a layer's kernel and bias are adjacent in the gradient buffer"
                ops.mark("%s:%s_wgrad%d_done" % (self.scope, stack.scope, l))
                if opt is not None:                                # the gradients are final on the current (aux) stream
                    evg = torch.cuda.Event()
                    evg.record(torch.cuda.current_stream(self.device))
                    opt.wait_event(evg)
                if deferred:
                    self._deferred.append(lambda: self.apply_group([kn, bn], *early_apply, lr_t=lr_t_now))
                elif opt is not None:
                    with torch.cuda.stream(opt):
                        reduce_then_apply([kn, bn], (lo, hi))
                        ops.mark("%s:%s_adam%d_done" % (self.scope, stack.scope, l))
                else:
                    reduce_then_apply([kn, bn], (lo, hi))
                    ops.mark("%s:%s_adam%d_done" % (self.scope, stack.scope, l))
            return cb if per_layer else None

        yield "moe"
        dS1 = yield from self.l2.backward_layers(dS2, need_dx=True, aux=aux, on_layer_grads=layer_cb(self.l2, defer))   # [C*B][2LH] = d(L1 final state)
        ops.mark(self.scope + ":l2_bwd_done")
        if late_moe is not None:
            late_moe()
        yield from self.l1.backward_layers(dS1, need_dx=False, aux=aux, on_layer_grads=layer_cb(self.l1))
        ops.mark(self.scope + ":l1_bwd_done")
        if aux is not None:
            ev = torch.cuda.Event()
            ev.record(aux)
            main.wait_event(ev)
        if opt is not None:
            ev = torch.cuda.Event()
            ev.record(opt)
            main.wait_event(ev)

    # ---- cross-step deferral of the MoE / L2-level updates (backward(defer=True)) ----------------------------------
    def run_deferred(self):
        """Enqueue the pending updates of the previous backward() on the aux stream they belong to (in stream order behind that
        step's weight-gradient products) and record the event forward() waits for before the L2 level.  No-op without any."""
        if not self._deferred:
            return
        aux = self._deferred_aux
        with torch.cuda.stream(aux):
            ops.mark(self.scope + ":deferred_begin")
            for fn in self._deferred:
                fn()
            ops.mark(self.scope + ":deferred_done")
            self._deferred_ev = torch.cuda.Event()
            self._deferred_ev.record(aux)
        self._deferred = []

    def wait_deferred(self, stream=None):
        """Make `stream` (default: current) wait for the deferred updates enqueued by run_deferred()."""
        self.run_deferred()
        ev, self._deferred_ev = self._deferred_ev, None
        if ev is not None:
            (stream or torch.cuda.current_stream(self.device)).wait_event(ev)

    def flush_deferred(self):
        """run_deferred + wait on the current stream: every weight / shadow is current in stream order afterwards."""
        if self._deferred or self._deferred_ev is not None:
            self.wait_deferred()
