"""The training-iteration graph of the reference's train.py / train_finetune.py
as an explicit schedule over the HIP kernels.

Reference: ``build_graph`` cs/train.py:185-427 (teacher + student in one
``sess.run``, cs/train.py:516-517) and cs/train_finetune.py:185-331
(student only).  Quirks kept on purpose: L_REP counted
twice in the student objective (cs/train.py:406), global_step += 2 per
iteration (one increment per train op, :332,:416), per-tensor gradient
clipping, teacher tensors constant in the student loss, both updates computed
from the pre-update weights of the same step.

Data parallel (new; the reference is single-device): one process per GPU,
replicated parameters, per-rank batch, gradients summed with RCCL all-reduce
(``torch.distributed`` backend "nccl") on the flat gradient buffer - the MoE
segment is reduced as soon as it is final so it overlaps the LSTM BPTT, the
teacher's reduce overlaps the student's forward/backward.  Batch-mean losses
(CE, L_REP) are scaled by 1/world so the summed gradient equals the
single-device gradient at the global batch; L_PRED is a batch *sum*
(cs/train.py:402) and is not scaled.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import ops
from .streams import concurrent_streams
from .engine import HLstmTower

F32 = torch.float32


def exponential_decay(lr0, global_step, batch_size, decay_examples, rate):
    """tf.train.exponential_decay(lr0, global_step*batch, decay_examples, rate,
    staircase=True) - cs/train.py:223-236."""
    return lr0 * rate ** math.floor(global_step * batch_size / decay_examples)


def every_n_indices(every_n, max_frames=300):
    """cs/train.py:265-269 (host-side restatement of the static index list)."""
    idx, k = [], 0
    while every_n * k <= max_frames - 1:
        idx.append(every_n * k)
        k += 1
    return idx


def validate_every_n(every_n, num_inputs_l1=5, max_frames=300):
    """The reference's graph only builds when len(index list) == int(300/every_n)
    and that count splits into 5 chunks (cs/train.py:262-272,
    cs/frame_level_models.py:286,307).  Same failure, as a ValueError."""
    if every_n <= 0:
        raise ValueError("every_n must be a positive integer")
    s = max_frames // every_n
    if len(every_n_indices(every_n, max_frames)) != s or s == 0 or s % num_inputs_l1 != 0:
        raise ValueError("every_n=%d: %d gathered frames cannot be split into %d equal chunks of the %d-frame "
                         "student input" % (every_n, len(every_n_indices(every_n, max_frames)), num_inputs_l1, s))


def dp_loss_scales(world):
    """Per-rank gradient scale factors such that an all-reduce SUM over `world`
    ranks (each holding B_local videos, losses normalised by B_local) equals the
    single-device gradient at the global batch: CE (cs/losses.py:97) and L_REP
    (cs/train.py:362) are batch means -> 1/world; L_PRED is a batch sum
    (cs/train.py:402) -> 1.  The l2 regulariser is added once, after the reduce."""
    return {"ce": 1.0 / world, "rep": 1.0 / world, "kl": 1.0}


class GradReducer:
    """Gradient all-reduce (SUM) on slices of a flat buffer and the collectives of the row-sharded MoE update.
    backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

    Two ways to place the collectives (both give the same numbers; chosen per process by EVC_DP_SERIAL_COMM):

    * default - every collective is a synchronous c10d op issued from inside the compute stream it belongs to, so the
      RCCL kernel sits in stream order between the kernels that produce and consume its data; the student tower uses a
      communicator of its own (DistillGraph), two communicators may be in flight on two streams at once.
    * EVC_DP_SERIAL_COMM=1 - the conservative form for a first run on real xGMI: ONE communicator for everything and ONE
      collective in flight at a time: every collective still runs on the stream it belongs to, but first waits for the end
      of the previous one (event chain in host issue order, identical on all ranks), and DistillGraph issues the two
      towers' backward phases in readiness order so that the chain follows the order in which the data becomes ready.
      RCCL kernels of this process then execute strictly one after the other; compute on the other streams overlaps them."""

    _last_ev = None                 # serial placement: the event behind the last collective this process issued (any stream)
    # Accounting for bench.py's N > 1 line (class-wide: every reducer of the process adds to it):
    #   stats[kind] = [collectives issued, payload bytes handed to them]     (always on: two integer adds per call)
    #   timing = None | list of (kind, bytes, start event, end event) recorded on the stream the collective runs on
    stats = {}
    timing = None

    @staticmethod
    def _sim():
        """EVC_DP_SIM=<busbw GB/s>:<blocks>:<LDS KB per block>[:<world>] -> (busbw, blocks, lds_kb, world) or None."""
        v = os.environ.get("EVC_DP_SIM")
        if not v:
            return None
        f = v.split(":")
        return float(f[0]), int(f[1]), int(f[2]), int(f[3]) if len(f) > 3 else 8

    @staticmethod
    def wire_bytes(kind, payload, world):
        """Bytes one rank sends (= receives) for a ring collective over `world` ranks: all-reduce 2 (w-1)/w of the payload,
        all-gather (w-1) times its own part (payload = the part this rank contributes)."""
        if world <= 1:
            return 0.0
        if kind.startswith("all_reduce"):
            return 2.0 * (world - 1) / world * payload
        if kind.startswith("reduce_scatter"):          # payload = the whole buffer every rank contributes (world slabs)
            return (world - 1.0) / world * payload
        return float(world - 1) * payload

    def __init__(self, process_group=None, grad_dtype=None):
        self.pg = process_group
        # gradient all-reduce payload type for the LSTM segments: "f32" (default) or "bf16" (EVC_DP_GRAD_DTYPE=bf16): half the
        # bytes on xGMI - for configurations whose step is shorter than their f32 all-reduce (cfg 5: 4.4 ms step, 187 MB per
        # tower) - at one bf16 rounding of each rank's gradient (2^-9 relative; the sum itself runs in RCCL's bf16 arithmetic)
        self.grad_dtype = grad_dtype or os.environ.get("EVC_DP_GRAD_DTYPE", "f32")
        if self.grad_dtype not in ("f32", "bf16"):
            raise ValueError("EVC_DP_GRAD_DTYPE / grad_dtype must be f32 or bf16, not %r" % self.grad_dtype)
        self._bf16_ws = None
        self.world = 1
        init = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.rank = 0
        if init:
            self.world = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)
        # collectives are issued when there is more than one rank; EVC_DP_FORCE=1 (debug) also issues them on a
        # one-rank group, which runs the whole RCCL path of a step on a single-GPU box (scripts/rccl_one_rank.sh)
        self.active = self.world > 1 or (init and os.environ.get("EVC_DP_FORCE") == "1")
        # TIMING AID (scripts/dp_occupancy_sim.sh, round 6): EVC_DP_SIM_WORLD=W on a ONE-rank forced group - this process does the per-rank WORK
        # of rank 0 of W: MoE row slabs of 1/W of the rows (MoeHead.shard), factor all-gathers replicated to W x batch rows, slab gathers /
        # reduce-scatters on the own slab only.  The other W - 1 slabs are never updated: the step's numbers are meaningless, its kernel
        # times are those of one rank of a W-GPU node with zero time on the wire (EVC_DP_SIM adds the wire time).  Never set in a real run.
        self.shard_world = self.world
        if self.world == 1 and self.active and os.environ.get("EVC_DP_SIM_WORLD"):
            self.shard_world = max(1, int(os.environ["EVC_DP_SIM_WORLD"]))
        self.serial = self.active and serial_comm() and torch.cuda.is_available()
        self._pending = []

    def _run(self, fn, *tensors, kind="collective", nbytes=0):
        """Issue one collective: in stream order on the current stream, or through the process-wide serial stream."""
        st = GradReducer.stats.setdefault(kind, [0, 0])
        st[0] += 1
        st[1] += int(nbytes)
        sim = GradReducer._sim()
        if sim is not None and self.world == 1 and torch.cuda.is_available():
            # one-GPU stand-in for the fabric (EVC_DP_SIM, scripts/dp_occupancy_sim.sh): after the (empty) one-rank collective a kernel
            # with an RCCL-like footprint holds `blocks` CUs for the time the bytes would spend on the wire at `busbw`
            busbw, blocks, lds_kb, w = sim
            if kind == "all_gather_slabs" and self.shard_world == 1:
                wire = (w - 1.0) / w * nbytes          # (at one rank the "slab" is the whole matrix)
            else:
                wire = GradReducer.wire_bytes(kind, nbytes, w)
            us = wire / (busbw * 1e9) * 1e6 + 20.0       # + a launch / rendezvous latency
            real = fn

            def fn():
                out = real()
                from . import _lib
                _lib.call("evc_debug_occupy", blocks, 256, lds_kb * 1024, us, torch.cuda.current_stream().cuda_stream)
                return out
        if GradReducer.timing is not None and torch.cuda.is_available():
            inner = fn

            def fn():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = inner()
                e1.record()
                GradReducer.timing.append((kind, int(nbytes), e0, e1))
                return out
        if not self.serial:
            return fn()
        # ONE collective of this process at a time, in host issue order, each on the stream it belongs to: the collective waits for
        # the end of the previous one (an event recorded behind it on ITS stream) and leaves an event for the next.  (Round 3
        # funnelled them through a fifth stream instead - which shares one of the four hardware queues with a compute stream
        # (streams.py) and serialised with that stream's kernels: 16.5 ms per step in the stand-in runs of DESIGN.md 6.1.)
        cur = torch.cuda.current_stream()
        if GradReducer._last_ev is not None:
            cur.wait_event(GradReducer._last_ev)
        out = fn()
        ev = torch.cuda.Event()
        ev.record(cur)
        GradReducer._last_ev = ev
        return out

    def reduce(self, flat, lo, hi, f32=True):
        """All-reduce of flat[lo:hi] as an async c10d op, joined by wait() (the schedule without early apply).  The payload type
        follows grad_dtype only where the caller allows it (f32=False: LSTM segments); with a bf16 payload, and under the serial
        placement, the call goes through reduce_async (stream order) - one implementation of the dtype switch."""
        if not self.active or hi <= lo:
            return
        if self.serial or (self.grad_dtype == "bf16" and not f32):
            self.reduce_async(flat, lo, hi, f32=f32)
            return
        self._pending.append(torch.distributed.all_reduce(flat[lo:hi], op=torch.distributed.ReduceOp.SUM,
                                                          group=self.pg, async_op=True))

    def reduce_async(self, flat, lo, hi, f32=False):
        """f32=True: this segment crosses the fabric as f32 whatever grad_dtype says (the MoE segment when its update is not the
        fused one: EVC_DP_GRAD_DTYPE=bf16 is an option for the LSTM segments only).
        All-reduce of flat[lo:hi] in stream order: issued as a *synchronous* c10d op, which ProcessGroupNCCL
        enqueues on the CURRENT stream (the RCCL kernel sits between the kernels that produce the gradients and the
        ones that consume them; the host does not wait).  An async_op=True collective runs on the process group's
        own stream instead, which shares one of the 4 hardware queues with a compute stream: its event then waits
        for every packet already queued there - measured on a one-rank communicator, where no byte moves: 14.4
        instead of 13.5 ms per step (scripts/dp_host_probe.py).  Returns None (nothing left to wait for)."""
        if not self.active or hi <= lo:
            return None
        seg = flat[lo:hi]
        if self.grad_dtype == "bf16" and not f32:
            return self._reduce_bf16(seg)
        self._run(lambda: torch.distributed.all_reduce(seg, op=torch.distributed.ReduceOp.SUM, group=self.pg, async_op=False), seg,
                  kind="all_reduce_grad_f32", nbytes=seg.numel() * 4)
        return None

    def _reduce_bf16(self, seg):
        """seg (f32) <- SUM over the ranks of bf16(seg): the payload crosses the fabric as bf16.  (gloo has no bfloat16: the CPU
        tests reduce the bf16-rounded values in f32 - the same per-rank rounding, an exact sum.)"""
        n = seg.numel()
        if self._bf16_ws is None or self._bf16_ws.numel() < n or self._bf16_ws.device != seg.device:
            self._bf16_ws = torch.empty(n, dtype=torch.bfloat16, device=seg.device)
        ws = self._bf16_ws[:n]
        ws.copy_(seg)
        if torch.distributed.get_backend(self.pg) == "gloo":
            seg.copy_(ws)
            self._run(lambda: torch.distributed.all_reduce(seg, op=torch.distributed.ReduceOp.SUM, group=self.pg, async_op=False), seg,
                      kind="all_reduce_grad_bf16", nbytes=n * 2)
            return None
        self._run(lambda: torch.distributed.all_reduce(ws, op=torch.distributed.ReduceOp.SUM, group=self.pg, async_op=False), ws,
                  kind="all_reduce_grad_bf16", nbytes=n * 2)
        seg.copy_(ws)
        return None

    def all_gather_rows(self, t):
        """[rows, cols] bf16 (contiguous) of every rank stacked along the rows, in rank order (the collective
        moves raw bytes: gloo has neither bfloat16 nor int16).  Stream-ordered like reduce_async."""
        if not self.active:
            return t
        t = t.contiguous()

        def go():
            out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            torch.distributed.all_gather_into_tensor(out.view(torch.uint8), t.view(torch.uint8), group=self.pg)
            if self.shard_world != self.world:           # EVC_DP_SIM_WORLD: the contraction length of a W-rank gather
                out = out.repeat(self.shard_world, *([1] * (out.dim() - 1)))
            return out
        return self._run(go, t, kind="all_gather_factors", nbytes=t.numel() * t.element_size())

    def all_reduce_small(self, t):
        """In-place SUM of a few floats (the partial norm sums of a sharded tensor), stream-ordered."""
        if self.active:
            self._run(lambda: torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=self.pg, async_op=False), t,
                      kind="all_reduce_small", nbytes=t.numel() * t.element_size())

    def all_gather_slabs(self, full, slab_rows):
        """full [world * slab_rows, cols] (contiguous): rank r owns rows [r*slab_rows, (r+1)*slab_rows); every
        rank's slab is written into every rank's `full`, stream-ordered."""
        if not self.active:
            return
        assert full.is_contiguous() and full.shape[0] == self.shard_world * slab_rows

        def go():
            own = full[self.rank * slab_rows:(self.rank + 1) * slab_rows].clone()     # (24 MB at world 8: no aliasing of in / out)
            dst = full if self.shard_world == self.world else full[:self.world * slab_rows]      # (EVC_DP_SIM_WORLD: the own slab only)
            torch.distributed.all_gather_into_tensor(dst.view(torch.uint8), own.view(torch.uint8), group=self.pg)
            return own
        self._run(go, full, kind="all_gather_slabs", nbytes=slab_rows * full.shape[1] * full.element_size())

    def reduce_scatter_rows(self, full, slab_rows):
        """full [world * slab_rows, cols] bf16 (contiguous): this rank's contribution to every rank's slab.  Returns [slab_rows, cols] bf16 = the
        SUM over the ranks of slab `rank` (stream-ordered).  RCCL sums in bf16 along the ring; gloo has neither bfloat16 nor reduce_scatter:
        the CPU / shared-GPU tests all-reduce the bf16-rounded values in f32 (the same per-rank rounding, an exact sum) and round the own slab once."""
        assert full.is_contiguous() and full.dtype == torch.bfloat16 and full.shape[0] == self.shard_world * slab_rows
        if not self.active:
            return full[:slab_rows]

        def go():
            if torch.distributed.get_backend(self.pg) == "gloo":
                f = full.float()
                torch.distributed.all_reduce(f, op=torch.distributed.ReduceOp.SUM, group=self.pg)
                return f[self.rank * slab_rows:(self.rank + 1) * slab_rows].to(torch.bfloat16)
            own = torch.empty((slab_rows, full.shape[1]), dtype=full.dtype, device=full.device)
            src = full if self.shard_world == self.world else full[:self.world * slab_rows]      # (EVC_DP_SIM_WORLD: the own slab only)
            torch.distributed.reduce_scatter_tensor(own, src, op=torch.distributed.ReduceOp.SUM, group=self.pg)
            return own
        return self._run(go, full, kind="reduce_scatter_grad_bf16", nbytes=full.numel() * 2)

    def wait(self):
        for w in self._pending:
            w.wait()
        self._pending = []


def serial_comm():
    """EVC_DP_SERIAL_COMM=1: one communicator, one collective at a time (see GradReducer)."""
    return os.environ.get("EVC_DP_SERIAL_COMM") == "1"


def dp_timeout():
    """Process-group timeout for init_process_group (bench.py, train.py): a collective that has not completed after this long
    makes the watchdog abort the process - a wedged first contact with the fabric ends non-zero in minutes instead of sitting in
    the launcher's window (c10d's default is 10 minutes).  EVC_DP_TIMEOUT_S, default 120 s; the longest legitimate wait is a
    peer's checkpoint write or its first-step allocations."""
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("EVC_DP_TIMEOUT_S", "120")))


def student_light(student, precision):
    """The "high" precision layout of the STUDENT tower: plain f16 on its L1 level (no K-extensions).  It runs 6 steps per chunk
    over 5 chunks where the teacher runs 15 over 20, so the rounding terms the teacher's layout extends have no time
    to build up: all-f16 leaves 1e-4 on its logits (the teacher: up to 1e-3) - and the extensions would cost its six L1 layer-0
    launches 2.5x their depth.  Only for students of at most 30 frames (every_n >= 10).  It reads the first segment of the shared K-extended input image.
    OFF BY DEFAULT since round 6 (EVC_HIGH_STUDENT_LIGHT=1 switches it on): true 16 steps from initialisation (9e-5 on the student's logits), not on
    towers trained for 512 steps - 7.8e-4 .. 1.64e-3 over 12 weight draws against 3.7 .. 9.1e-4 with the teacher's layout
    (profiles/r06_precision_robustness_long.txt), for 0.14 ms of the 11.7 ms "high" step."""
    if (student is not None and precision == "high" and student.T <= 30       # (every_n >= 10; a longer student takes the teacher's layout)
            and os.environ.get("EVC_HIGH_STUDENT_LIGHT", "0") == "1"):
        student.f16_x_segments = 1
        student.f16_wh_ext_layers = ()
        student.f16_wx_ext_layers = ()
        student.f16_fp8_lo = False


def input_image_args(*towers):
    """f16_segments / fp8_tail of ops.l2norm_chunk for the towers that share one input image: the widest K-extension any of them
    contracts (a tower may read fewer segments than the rows hold); with a tower on the fp8 L1 level the rows are its 5F-byte ones."""
    tw = [t for t in towers if t is not None]
    fp8 = any(t.fp8_lo() for t in tw)
    if fp8:
        assert all(t.fp8_lo() or t.f16_x_segments == 1 for t in tw), "towers sharing an fp8 input image read its plain f16 part"
        return dict(f16_segments=1, fp8_tail=True)
    return dict(f16_segments=max(t.f16_x_segments for t in tw))


def input_views(g, x_raw, num_frames, tp, sp, need_student):
    """The L1 input images of graph ``g``'s towers for one batch (ops.l2norm_chunk): (teacher view, student view), each the plain bf16 image or - in
    the non-bf16 modes - a tuple (bf16 image, second image[, row scales]).  uint8 frames into "high" towers whose layer 0 is on the f16 + e4m3 form
    take the INTEGER-frame images (ops.l2norm_chunk_int, HLstmTower.x_int: the input exact, round 6)."""
    towers = [t for t in (g.teacher, g.student if need_student else None) if t is not None]
    p1, p2 = (tp[2] if tp else None), (sp[3] if sp else None)
    if x_raw.dtype == torch.uint8 and towers and all(t.precision == "high" and t.x_int() for t in towers):
        return ops.l2norm_chunk_int(x_raw, num_frames, g.C1, g.every_n if need_student else None, g.C2, plan1=p1, plan2=p2, teacher_view=g.teacher is not None)
    return ops.l2norm_chunk(x_raw, g.C1, g.every_n if need_student else None, g.C2, num_frames=num_frames if x_raw.dtype == torch.uint8 else None,
                            split=towers[0].input_split(), plan1=p1, plan2=p2, teacher_view=g.teacher is not None,
                            **input_image_args(g.teacher, g.student if need_student else None))


def frame_counts_and_plans(g, num_frames, nh, need_teacher, need_student):
    """Frame counts (device) and the L1 row plans of both towers of graph ``g`` (DistillGraph / EvalGraph): rows
    sorted by length so the padding rows drop out of every L1 kernel (ops.RowPlan).  Host twins of the counts
    give the launch geometry.  Returns ((len_l1, len_l2, plan) | None, (n_student, len_l1, len_l2, plan) | None)."""
    t = s = None
    if need_teacher:
        _, l1, l2 = ops.frame_counts(num_frames, 1, g.C1, g.max_frames // g.C1, g.max_frames)
        plan = None
        if g.row_plans:
            _, l1h, _ = ops.host_frame_counts(nh, 1, g.C1, g.max_frames // g.C1, g.max_frames)
            plan = ops.RowPlan(l1, l1h, g.max_frames // g.C1)
        t = (l1, l2, plan)
    if need_student:
        n_s, l1s, l2s = ops.frame_counts(num_frames, g.every_n, g.C2, g.S // g.C2, g.max_frames, subsampled=True)
        plan = None
        if g.row_plans:
            _, l1h, _ = ops.host_frame_counts(nh, g.every_n, g.C2, g.S // g.C2, g.max_frames, subsampled=True)
            plan = ops.RowPlan(l1s, l1h, g.S // g.C2)
        s = (n_s, l1s, l2s, plan)
    return t, s


class DistillGraph:
    """mode: 'teacher_student' (train.py), 'teacher' (teacher only, BASELINE cfg 2),
    'student' (train_finetune.py)."""

    LOSS_SLOTS = ("label_loss", "student_loss_state", "pred_loss", "student_label_loss")
    # host issue orders of the two towers' backward phases (HLstmTower.backward_phases: MoE head, L2 layer 1, L2 layer 0, L1 layer 1,
    # L1 layer 0 each): letters name the tower whose next phase is issued; whatever is left afterwards is drained student first
    ISSUE_ORDERS = {"sequential": "sssss", "interleaved": "tsstsstst"}

    def __init__(self, batch_size, every_n=10, mode="teacher_student", feature_size=1152, vocab_size=4716,
                 max_frames=300, num_inputs_to_lstm=20, num_inputs_l1_student=5, lstm_cells=1024, lstm_layers=2,
                 num_mixtures=2, base_learning_rate=0.001, learning_rate_decay=1.0,
                 learning_rate_decay_examples=4000000, regularization_penalty=2.0, clip_gradient_norm=1.0,
                 count_rep_twice=True, device="cuda:0", seed=7, process_group=None, overlap_towers=True,
                 precision="bf16"):
        assert mode in ("teacher_student", "teacher", "student")
        self.mode, self.B, self.every_n = mode, batch_size, every_n
        self.max_frames, self.C1, self.C2 = max_frames, num_inputs_to_lstm, num_inputs_l1_student
        self.lr0, self.lr_decay, self.lr_decay_examples = base_learning_rate, learning_rate_decay, learning_rate_decay_examples
        self.reg_pen, self.clip = regularization_penalty, clip_gradient_norm
        self.rep_w = 2.0 if count_rep_twice else 1.0
        self.device = torch.device(device)
        self.pg = process_group
        self.reducer = GradReducer(process_group)
        self.world, self.dp = self.reducer.world, self.reducer.active
        # The student's collectives get a communicator of their own: one process group executes its collectives in
        # issue order, so on a shared group the teacher's early factor all-gather (host-issued after the student's
        # backward) would queue behind the student's last gradient all-reduce and hold the teacher's whole update
        # chain back.  (new_group is collective: every rank constructs the graph.)
        self.reducer_s = self.reducer
        if self.dp and mode == "teacher_student" and not serial_comm():
            ranks = list(range(torch.distributed.get_world_size(process_group))) if process_group is None else None
            self.reducer_s = GradReducer(torch.distributed.new_group(ranks) if ranks is not None else process_group)
        self.global_step = 0
        self.teacher = self.student = None
        if mode != "student":
            self.teacher = HLstmTower(batch_size, max_frames, num_inputs_to_lstm, feature_size, vocab_size, lstm_cells,
                                      lstm_layers, num_mixtures, device, True, "model", seed)
        if mode != "teacher":
            validate_every_n(every_n, num_inputs_l1_student, max_frames)
            self.S = max_frames // every_n
            self.student = HLstmTower(batch_size, self.S, num_inputs_l1_student, feature_size, vocab_size, lstm_cells,
                                      lstm_layers, num_mixtures, device, True, "model_student", seed + 1)
        if self.dp:                      # row-shard the MoE optimizer state now, while nothing is in flight on any stream
            for tw, red in ((self.teacher, self.reducer), (self.student, self.reducer_s)):
                if tw is not None:
                    tw.moe.shard(red.shard_world, red.rank)
        self.precision = precision       # engine.TowerBase.precision: "bf16" | "high" (1e-3 at trained magnitudes) | "split" (uniform)
        student_light(self.student, precision)
        if precision != "bf16":
            for tw in (self.teacher, self.student):
                if tw is not None:
                    tw.set_precision(precision)
        self.losses = torch.zeros(8, dtype=F32, device=self.device)
        self._losses_reduced = torch.zeros(8, dtype=F32, device=self.device)   # data parallel: SUM over the ranks, per step
        self._dp_t = self._dp_s = self._ds_s = None
        self.overlap_towers = overlap_towers
        self.row_plans = precision != "split"   # sort the L1 chunk rows by length and skip the padding rows (ops.RowPlan);
        # (the uniform split-bf16 layers run on every row)
        # True: the student's forward starts next to the teacher's forward instead of after it.  Measured 0.1 ms/step
        # faster, but the teacher's fused forward steps then share the chip (67 -> 84 us per launch): off by default so
        # that the step's dominant kernel runs - and is measured - alone.
        self.student_forward_early = False
        # True: the student's forward starts when the teacher's L1 level has been enqueued, next to the teacher's L2 chain /
        # MoE head (latency-bound launches).  Measured 13.01 -> 12.88 ms/step with the teacher's L1 steps unaffected.
        self.student_forward_after_l1 = os.environ.get("EVC_STUDENT_AFTER_L1", "1") == "1"
        if os.environ.get("EVC_STUDENT_EARLY") is not None:
            self.student_forward_early = os.environ["EVC_STUDENT_EARLY"] == "1"
        # Cross-step deferral (one process): the MoE-head and L2-level updates of step k are enqueued at the start of step k+1,
        # under its L1 forward, instead of under step k's BPTT chain (HLstmTower.backward(defer=True)).  Whoever reads the
        # weights between two steps calls flush() first (state_dict() / consolidate() / apply_gradients() do).
        self.defer_updates = os.environ.get("EVC_DEFER_UPDATES", "0") == "1"
        # one communicator for both towers (EVC_DP_SERIAL_COMM=1): issue the backward phases in readiness order, so that collectives
        # funnelled through one stream do not wait behind the other tower's later ones; EVC_ISSUE_ORDER overrides (A/B runs)
        self.issue_order = os.environ.get("EVC_ISSUE_ORDER", "interleaved" if (self.dp and serial_comm()) else "sequential")
        if self.issue_order not in self.ISSUE_ORDERS:
            raise ValueError("EVC_ISSUE_ORDER must be one of %s" % sorted(self.ISSUE_ORDERS))
        self._opt_t = self._opt_s = None
        if self.device.type == "cuda":
            # four streams that measurably overlap (streams.py); the step never runs on the default stream
            self._main, self._side, self._aux_t, self._aux_s = concurrent_streams(self.device, 4)
            if os.environ.get("EVC_SINGLE_STREAM") == "1":     # profiling aid: the same launches, all on ONE stream (solo kernel times)
                self._side = self._aux_t = self._aux_s = self._main
            # experiment: the optimizer launches on a CU-masked stream of their own - EVC_OPT_CU_MASK=<CUs per XCD>[:<first>]
            m = os.environ.get("EVC_OPT_CU_MASK")
            if m:
                from .streams import cu_masked_stream
                f = [int(v) for v in m.split(":")]
                self._opt_t = cu_masked_stream(self.device, f[0], f[1] if len(f) > 1 else 0)
                self._opt_s = self._opt_t if os.environ.get("EVC_OPT_CU_MASK_SHARED", "1") == "1" else cu_masked_stream(self.device, f[0], f[1] if len(f) > 1 else 0)
            # Single-tower graphs (cfg 2 teacher only, cfg 5 student only) use two of the four streams: the tower's collectives + optimizer launches
            # take a THIRD one (the other tower's idle aux stream) instead of queueing on the aux stream between the weight-gradient products - under
            # data parallelism that stream is the step's critical path and every byte on the wire was exposed (cfg 5 as rank 0 of 8 with stand-in
            # collectives at 300 GB/s: 5.67 ms; profiles/r06_dp_sim_world.txt).  EVC_OPT_SPARE_STREAM=0 / 1 forces it off / on (default: under DP).
            spare = os.environ.get("EVC_OPT_SPARE_STREAM")
            if (spare == "1" or (spare is None and self.dp)) and os.environ.get("EVC_SINGLE_STREAM") != "1" and not m:
                if mode == "student":
                    self._opt_s = self._aux_t
                elif mode == "teacher":
                    self._opt_t = self._aux_s
            self._ev_fwd, self._ev_student, self._ev_in = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()

    # ---- data-parallel gradient reduction -------------------------------------
    def _reduce_tower(self, tower, moe_first):
        st = tower.store
        moe_lo = st.offsets[tower.GATES]
        if moe_first:
            self.reducer.reduce(st.grad, moe_lo, st.total, f32=True)
        else:
            self.reducer.reduce(st.grad, 0, moe_lo, f32=False)

    # ---- one training iteration -----------------------------------------------
    def step(self, x_raw, labels_u8, num_frames, apply=True, num_frames_host=None):
        """Runs ``_step`` on the graph's own main stream, ordered after the caller's current stream on entry
        and before it on exit (so callers see ordinary single-stream semantics).

        num_frames_host: the same frame counts on the host (numpy / CPU tensor / list), as the input pipeline
        has them before the H2D copy.  The launch geometry of the row-planned L1 stacks depends on them; without
        it they are read back from the device, which stalls the host on everything queued before."""
        if num_frames_host is None:
            num_frames_host = num_frames.cpu()
        nh = np.asarray(num_frames_host, dtype=np.int64).reshape(-1)
        caller = torch.cuda.current_stream(self.device)
        if caller == self._main:
            return self._step(x_raw, labels_u8, num_frames, apply, nh)
        self._main.wait_stream(caller)
        with torch.cuda.stream(self._main):
            out = self._step(x_raw, labels_u8, num_frames, apply, nh)
        for t in (x_raw, labels_u8, num_frames):
            t.record_stream(self._main)
        caller.wait_stream(self._main)
        return out

    debug_marks = None      # set to a list to collect (name, timing event) pairs of one step (scripts/step_marks.py)

    def _mark(self, name, stream):
        if self.debug_marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            self.debug_marks.append((name, ev))

    def _step(self, x_raw, labels_u8, num_frames, apply=True, nh=None):
        """x_raw [B,300,F] f32 (or uint8), labels_u8 [B,V] uint8, num_frames [B] int32.
        Returns a dict mirroring the graph collections the reference's loop
        fetches (cs/train.py:336-344,420-425,516-517); tensors stay on device.

        Schedule: the teacher's backward and the whole student tower are independent once the
        teacher's forward is done (teacher tensors are constants in the student loss), so the
        student runs on a second HIP stream: its many small launches (M = batch L2 steps, 30-frame
        L1) fill the CUs that the teacher's under-filled launches leave idle."""
        B = x_raw.shape[0]
        V = labels_u8.shape[1]
        dev = self.device
        if self.dp and B != self.B:
            # every rank must step on the same number of videos: the factor all-gathers move round_up(B, 32) rows per
            # rank and the per-rank loss scales 1/(world*B) only add up to the global-batch mean for equal B
            raise ValueError("data-parallel step on %d videos, the graph was built for %d per rank (ragged batches are "
                             "not allowed under data parallelism: drop the remainder)" % (B, self.B))
        if self._dp_t is None or self._dp_t.shape[0] != B:
            self._dp_t = torch.empty((B, V), dtype=F32, device=dev)
            self._dp_s = torch.empty((B, V), dtype=F32, device=dev)
        need_student = self.student is not None
        main = torch.cuda.current_stream(dev)
        tp, sp = frame_counts_and_plans(self, num_frames, nh, self.teacher is not None, need_student)
        xt, xs = input_views(self, x_raw, num_frames, tp, sp, need_student)     # (student only: the sub-sampled frames alone are read)
        for tw in (self.teacher, self.student):     # step k-1's deferred MoE / L2-level updates: now, under this step's L1 forward
            if tw is not None:
                tw.run_deferred()
        self.losses.zero_()
        out = {}
        mark = self._mark
        mark("start", main)
        sc = dp_loss_scales(self.world)
        # both train ops read the same global_step / learning rate within one iteration (cs/train.py:223-236)
        lr = exponential_decay(self.lr0, self.global_step, B * self.world, self.lr_decay_examples, self.lr_decay)
        l2c = self.reg_pen * 1e-8
        self._teacher_applied = self._student_applied = False
        two_streams = self.teacher is not None and need_student and self.overlap_towers
        side = self._side if two_streams else main
        early_student = two_streams and self.student_forward_early
        # "after_l1": the student's forward starts when the teacher's L1 level is done - next to the teacher's L2 chain
        # and MoE head (small launches), not next to its L1 steps (the roofline kernel keeps the chip to itself)
        mid_student = two_streams and not early_student and self.student_forward_after_l1 and self.teacher is not None and need_student
        s_state = s_pred = n_s = l1s = l2s = plan_s = gen_s = early_s = None
        if need_student and early_student:
            # The student's forward needs only its own inputs and weights: it starts right away, next to the
            # teacher's forward (whose L2 / MoE tail is a chain of small launches that leaves most CUs idle);
            # only its distillation losses wait for the teacher's outputs.
            self._ev_in.record(main)
            side.wait_event(self._ev_in)
            with torch.cuda.stream(side):
                mark("student_start", side)
                n_s, l1s, l2s, plan_s = sp
                s_state, s_pred = self.student.forward(xs, l1s, l2s, plan_s)
                ops.ce_loss(s_pred, labels_u8, self.losses[3:4], self._dp_s, grad_scale=sc["ce"] / B)
                mark("student_fwd_done", side)
        def student_forward_mid():
            nonlocal s_state, s_pred, n_s, l1s, l2s, plan_s
            self._ev_in.record(main)
            side.wait_event(self._ev_in)
            with torch.cuda.stream(side):
                mark("student_start", side)
                n_s, l1s, l2s, plan_s = sp
                s_state, s_pred = self.student.forward(xs, l1s, l2s, plan_s)
                ops.ce_loss(s_pred, labels_u8, self.losses[3:4], self._dp_s, grad_scale=sc["ce"] / B)
                mark("student_fwd_done", side)

        if self.teacher is not None:
            l1, l2, plan_t = tp
            t_state, t_pred = self.teacher.forward(xt, l1, l2, plan_t, after_l1=student_forward_mid if mid_student else None)
            ops.ce_loss(t_pred, labels_u8, self.losses[0:1], self._dp_t, grad_scale=sc["ce"] / B)
            if two_streams:
                self._ev_fwd.record(main)
            mark("teacher_fwd_done", main)
        if need_student:
            if two_streams:
                side.wait_event(self._ev_fwd)
            with torch.cuda.stream(side):
                if not early_student and not mid_student:
                    mark("student_start", side)
                    n_s, l1s, l2s, plan_s = sp
                    s_state, s_pred = self.student.forward(xs, l1s, l2s, plan_s)
                    ops.ce_loss(s_pred, labels_u8, self.losses[3:4], self._dp_s, grad_scale=sc["ce"] / B)
                    mark("student_fwd_done", side)
                ds = None
                if self.teacher is not None:
                    if self._ds_s is None or self._ds_s.shape != s_state.shape:
                        self._ds_s = torch.empty_like(s_state)
                    ops.kl_pred_loss(t_pred, self.teacher.rowsum, s_pred, self.student.rowsum, self.losses[2:3], self._dp_s,
                                     grad_scale=sc["kl"], accumulate_grad=True)
                    ops.rep_loss(t_state, s_state, self.losses[1:2], self._ds_s, grad_scale=self.rep_w * sc["rep"])
                    ds = self._ds_s
                early_s = (lr, self.clip, l2c) if (apply and self.overlap_towers and self._aux_s is not None) else None
                st_s = self.student.store
                gen_s = self.student.backward_phases(
                    ds, self._dp_s, on_moe_grads_ready=None if early_s else (lambda: self._reduce_tower(self.student, True)),
                    aux=self._aux_s if self.overlap_towers else None, early_apply=early_s,
                    reduce_fn=(lambda lo, hi, f32=False: self.reducer_s.reduce_async(st_s.grad, lo, hi, f32=f32)) if (early_s and self.dp) else None,
                    dp=self.reducer_s if (early_s and self.dp) else None, defer=self.defer_updates and not self.dp, opt=self._opt_s)
        gen_t = None
        if self.teacher is not None:
            # weight-gradient GEMMs and the per-group clip+Adam go to an aux stream, under the BPTT chain
            # (the tower's outputs t_state / t_pred are separate buffers, untouched by the update)
            early_t = (lr, self.clip, l2c) if (apply and self.overlap_towers and self._aux_t is not None) else None
            st_t = self.teacher.store
            gen_t = self.teacher.backward_phases(
                None, self._dp_t, on_moe_grads_ready=None if early_t else (lambda: self._reduce_tower(self.teacher, True)),
                aux=self._aux_t if self.overlap_towers else None, early_apply=early_t,
                reduce_fn=(lambda lo, hi, f32=False: self.reducer.reduce_async(st_t.grad, lo, hi, f32=f32)) if (early_t and self.dp) else None,
                dp=self.reducer if (early_t and self.dp) else None, defer=self.defer_updates and not self.dp, opt=self._opt_t)
        # Host issue order of the two backward passes (same launches, same streams, same results): "sequential" = the student's
        # whole backward, then the teacher's; "interleaved" = phase by phase in the order in which the phases become ready on the
        # GPU (measured timeline, profiles/r04_timeline_default.txt) - what a ONE-communicator placement of the collectives needs
        # so that no tower's early collective queues behind the other tower's late one.
        gens = {"s": (gen_s, side), "t": (gen_t, main)}
        live = {k for k, (g_, _) in gens.items() if g_ is not None}
        order = self.ISSUE_ORDERS[self.issue_order] if (gen_s is not None and gen_t is not None) else ""
        def resume(who):
            g_, st_ = gens[who]
            with torch.cuda.stream(st_):
                if next(g_, self) is self:             # exhausted
                    live.discard(who)
        for who in order:                              # the scripted part (written for two LSTM layers per level) ...
            if who in live:
                resume(who)
        for who in ("s", "t"):                         # ... then each tower to its end, whatever --lstm_layers says: a tower needs
            while who in live:                         # 2 * lstm_layers + 2 resumptions, and an unfinished generator would silently
                resume(who)                            # skip the lowest layers' BPTT, weight gradients, Adam and the final stream joins
        assert not live
        if need_student:
            with torch.cuda.stream(side):
                if not early_s:
                    self._reduce_tower(self.student, False)
                self._student_applied = early_s is not None
                mark("student_done", side)
                if two_streams:
                    self._ev_student.record(side)
                    used = (xs if isinstance(xs, tuple) else (xs,)) + (n_s, l1s, l2s)
                    if plan_s is not None:
                        used += (plan_s.pos, plan_s.inv, plan_s.lens)
                    for t in used:                                          # allocated on `main`, consumed on `side`
                        if t is not None:
                            t.record_stream(side)
            out.update(student_predictions=s_pred, student_state=s_state, num_frames_student=n_s,
                       student_loss_state=self.losses[1], pred_loss=self.losses[2], student_label_loss=self.losses[3])
        if self.teacher is not None:
            if not early_t:
                self._reduce_tower(self.teacher, False)
            mark("teacher_bwd_done", main)
            out.update(predictions=t_pred, teacher_state=t_state, loss=self.losses[0])
            self._teacher_applied = early_t is not None
        if two_streams:
            main.wait_event(self._ev_student)
        self.reducer.wait()
        if self.dp:
            # the loss values of the global batch, as part of the step (8 floats, in stream order behind the teacher's
            # collectives): loss_report() then needs no collective and may be called by any one rank, at any time
            self._losses_reduced.copy_(self.losses)
            self.reducer.all_reduce_small(self._losses_reduced)
        if apply:
            self._apply_gradients(B, lr)
        out["global_step"] = self.global_step
        return out

    def flush(self):
        """Enqueue and join the deferred updates of the last step (defer_updates): afterwards every weight, moment and operand
        shadow of both towers is current in the order of the caller's stream.  Cheap no-op when nothing is pending."""
        if self.device.type != "cuda":
            return
        cur = torch.cuda.current_stream(self.device)
        for tw in (self.teacher, self.student):
            if tw is not None:
                tw.wait_deferred(cur)

    def apply_gradients(self, batch_size, lr=None):
        """Runs whichever train op has not been applied inside step(); each one increments
        global_step (cs/train.py:332,416 -> += 2 per iteration, README.md:116,121)."""
        self.flush()
        self._apply_gradients(batch_size, lr)

    def _apply_gradients(self, batch_size, lr=None):
        l2c = self.reg_pen * 1e-8
        if lr is None:
            lr = exponential_decay(self.lr0, self.global_step, batch_size * self.world, self.lr_decay_examples, self.lr_decay)
        if self.teacher is not None:
            if not getattr(self, "_teacher_applied", False):
                self.teacher.apply_gradients(lr, self.clip, l2c)
            self.global_step += 1
        if self.student is not None:
            if not getattr(self, "_student_applied", False):
                self.student.apply_gradients(lr, self.clip, l2c)
            self.global_step += 1
        self._teacher_applied = self._student_applied = False

    def consolidate(self):
        """Collective (every rank calls it; a no-op on one rank): the row-sharded f32 MoE weights and Adam moments
        (MoeHead.shard) are all-gathered so that every rank holds the complete model again - before a checkpoint,
        state_dict(), or any update that does not go through the fused data-parallel path."""
        self.flush()
        for tw, red in ((self.teacher, self.reducer), (self.student, self.reducer_s)):
            if tw is not None:
                tw.moe.consolidate(red)

    @property
    def losses_for_report(self):
        """Device tensor loss_report() reads (a caller that logs one step behind the GPU clones it after the step)."""
        return self._losses_reduced if self.dp else self.losses

    def loss_report(self, losses=None):
        """Host floats in the order the reference logs them (cs/train.py:528-533), for the global batch.  Not a
        collective (under data parallelism step() has already summed the values over the ranks).  losses: a copy of
        losses_for_report taken after an earlier step (device or host tensor)."""
        v = (self.losses_for_report if losses is None else losses).tolist()
        rep = {k: v[i] for i, k in enumerate(self.LOSS_SLOTS)}
        if self.dp:   # sums over the ranks -> global-batch means (L_PRED is a batch sum)
            rep = {"label_loss": v[0] / self.world, "student_loss_state": v[1] / self.world, "pred_loss": v[2],
                   "student_label_loss": v[3] / self.world}
        return rep


class EvalGraph:
    """Forward-only graph of cs/validate.py:109-189 (teacher built too, so that the
    student_state_loss ||teacher_state - student_state||^2 can be logged) and of
    cs/eval_finetune.py:108-175 (``student_only``).  Towers are built with
    training=False (no backward tape); ``restore`` takes a TF-named state dict."""

    def __init__(self, batch_size, every_n=10, student_only=False, feature_size=1152, vocab_size=4716, max_frames=300,
                 num_inputs_to_lstm=20, num_inputs_l1_student=5, lstm_cells=1024, lstm_layers=2, num_mixtures=2,
                 device="cuda:0", precision="bf16"):
        validate_every_n(every_n, num_inputs_l1_student, max_frames)
        self.every_n, self.max_frames, self.C1, self.C2 = every_n, max_frames, num_inputs_to_lstm, num_inputs_l1_student
        self.S = max_frames // every_n
        self.device = torch.device(device)
        self.teacher = None
        if not student_only:
            self.teacher = HLstmTower(batch_size, max_frames, num_inputs_to_lstm, feature_size, vocab_size, lstm_cells,
                                      lstm_layers, num_mixtures, device, False, "model", 7)
        self.student = HLstmTower(batch_size, self.S, num_inputs_l1_student, feature_size, vocab_size, lstm_cells,
                                  lstm_layers, num_mixtures, device, False, "model_student", 8)
        self.precision = precision
        student_light(self.student, precision)
        if precision != "bf16":
            for tw in (self.teacher, self.student):
                if tw is not None:
                    tw.set_precision(precision)
        self.losses = torch.zeros(4, dtype=F32, device=self.device)
        self._main, self._side = concurrent_streams(self.device, 4)[:2]
        self._ev_in, self._ev_out = torch.cuda.Event(), torch.cuda.Event()
        self.row_plans = precision != "split"

    def restore(self, state_dict):
        """saver_teacher / saver_student .restore (cs/validate.py:350-384): the 11 variables of each tower by name."""
        for tw in (self.teacher, self.student):
            if tw is not None:
                tw.load_state_dict(state_dict)

    def step(self, x_raw, labels_u8, num_frames, num_frames_host=None):
        if num_frames_host is None:
            num_frames_host = num_frames.cpu()
        nh = np.asarray(num_frames_host, dtype=np.int64).reshape(-1)
        caller = torch.cuda.current_stream(self.device)
        self._main.wait_stream(caller)
        with torch.cuda.stream(self._main):
            out = self._step(x_raw, labels_u8, num_frames, nh)
        for t in (x_raw, labels_u8, num_frames):
            t.record_stream(self._main)
        caller.wait_stream(self._main)
        return out

    def _step(self, x_raw, labels_u8, num_frames, nh):
        """Returns predictions (student), student_label_loss, student_state_loss (teacher_student only) - the
        tensors cs/validate.py:240 fetches.  The two towers are independent: they run on two streams."""
        main = torch.cuda.current_stream(self.device)
        split = self.student.input_split()
        u8 = x_raw.dtype == torch.uint8
        tp, sp = frame_counts_and_plans(self, num_frames, nh, self.teacher is not None, True)
        xt, xs = input_views(self, x_raw, num_frames, tp, sp, True)
        self.losses.zero_()
        out = {}
        self._ev_in.record(main)
        self._side.wait_event(self._ev_in)
        n_s, l1s, l2s, plan_s = sp
        with torch.cuda.stream(self._side):
            s_state, s_pred = self.student.forward(xs, l1s, l2s, plan_s)
            ops.ce_loss(s_pred, labels_u8, self.losses[0:1])
            self._ev_out.record(self._side)
            used = (xs if isinstance(xs, tuple) else (xs,)) + (n_s, l1s, l2s)
            if plan_s is not None:
                used += (plan_s.pos, plan_s.inv, plan_s.lens)
            for t in used:
                t.record_stream(self._side)
        if self.teacher is not None:
            l1, l2, plan_t = tp
            t_state, t_pred = self.teacher.forward(xt, l1, l2, plan_t)
            out.update(teacher_state=t_state, teacher_predictions=t_pred)
        main.wait_event(self._ev_out)
        if self.teacher is not None:
            ops.rep_loss(t_state, s_state, self.losses[1:2])
            out["student_state_loss"] = self.losses[1]
        out.update(predictions=s_pred, student_state=s_state, num_frames=n_s, student_label_loss=self.losses[0],
                   loss=self.losses[0])
        return out


class SingleTowerGraph:
    """Teacher-only training step for dict-returning models (DbofModel,
    FrameLevelLogisticModel).  The reference's train.py cannot run these
    (it unpacks the H-LSTM tuple, cs/train.py:282 - SURVEY.md Appendix D-8);
    this follows the upstream starter-code semantics it was forked from:
    final_loss = regularization_penalty*reg + CE, one train op, global_step += 1.

    Update: the MoE head's two weight matrices go through MoeHead.fused_update (gradient recomputed from its
    rank-B factors inside the clip + Adam pass) whenever the step also applies; the other variables through
    clip_by_norm + TF-Adam on the materialised gradients.
    Data parallel: the gradients of each backward stage (MoE bias -> hidden layer -> cluster layer) are all-reduced on
    a side stream as soon as the stage is final, under the rest of the backward pass; the MoE weights need no gradient
    all-reduce at all (factor all-gather + row-sharded update, as in DistillGraph); batch-norm statistics and the
    batch-norm scale/offset gradients of cluster_bn / hidden1_bn are global through the all-reduced f64 sums."""

    def __init__(self, tower, base_learning_rate=0.001, learning_rate_decay=1.0, learning_rate_decay_examples=4000000,
                 regularization_penalty=2.0, clip_gradient_norm=1.0, process_group=None):
        self.tower, self.device = tower, tower.device
        self.lr0, self.lr_decay, self.lr_decay_examples = base_learning_rate, learning_rate_decay, learning_rate_decay_examples
        self.reg_pen, self.clip, self.pg = regularization_penalty, clip_gradient_norm, process_group
        self.reducer = GradReducer(process_group)
        self.world, self.dp = self.reducer.world, self.reducer.active
        self.global_step = 0
        self.losses = torch.zeros(4, dtype=F32, device=self.device)
        self._dp = None
        self.moe = getattr(tower, "moe", None)
        self.fused_moe_update = True
        if self.dp and self.moe is not None and self.moe.can_fuse_update():
            self.moe.shard(self.reducer.shard_world, self.reducer.rank)      # while nothing is in flight
        self._aux = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def consolidate(self):
        """Collective (no-op on one rank): complete f32 MoE weights / moments on every rank (before a checkpoint)."""
        if self.moe is not None:
            self.moe.consolidate(self.reducer)

    def step(self, x_raw, labels_u8, num_frames, uniform=None, apply=True):
        """x_raw [B,T,F] float32 or uint8 (as the reader delivers it: Dequantize is fused into the input kernels)."""
        from .towers import DbofGenericTower, DbofTower, NetVladTower
        B, V = labels_u8.shape
        if self._dp is None or self._dp.shape[0] != B:
            self._dp = torch.empty((B, V), dtype=F32, device=self.device)
        tw = self.tower
        if self.dp and B != tw.B:
            raise ValueError("data-parallel step on %d videos, the tower was built for %d per rank (ragged batches are not "
                             "allowed under data parallelism: drop the remainder)" % (B, tw.B))
        if isinstance(tw, (DbofTower, NetVladTower, DbofGenericTower)):
            if uniform is None:      # (SampleRandomSequence draws one start per video: column 0 is used)
                uniform = torch.rand((B, tw.S), dtype=F32, device=self.device)
            self.last_uniform = uniform              # the tf.random_uniform draw of this step (SampleRandomFrames)
            pred = tw.forward(x_raw, num_frames, uniform)
        else:
            pred = tw.forward(x_raw, num_frames)
        self.losses.zero_()
        ops.ce_loss(pred, labels_u8, self.losses[0:1], self._dp, grad_scale=1.0 / (B * self.world))
        fuse = (apply and self.fused_moe_update and self.moe is not None and self.moe.can_fuse_update()
                and self.moe.prefer_fused_update(self.dp)
                and tw.precision == "bf16")
        if self.moe is not None and not fuse and getattr(self.moe, "_stale", False):
            raise RuntimeError("the MoE weights are sharded over the ranks (fused data-parallel update); call consolidate() "
                               "on every rank before a step that does not apply through it")
        # data parallel: the exchange that carries the MoE gradient is chosen by shape (MoeHead.dp_route): factor all-gather (fused update) or
        # bf16 reduce-scatter of the materialised gradient onto the owners' slabs (sharded_update) - either way no all-reduce of those segments
        route_rs = bool(fuse and self.dp and self.moe.dp_route(self.reducer.shard_world) == "reduce_scatter")
        main = torch.cuda.current_stream(self.device)
        stages = tw.grad_stages()
        fused_names = (self.moe.GATES, self.moe.EXPERTS, self.moe.EBIAS) if fuse else ()    # (the bias gradient comes from the same factors)
        skip = tuple(getattr(tw, "global_grad_names", ())) + fused_names

        def on_stage(i):
            # data parallel: SUM of this stage's per-rank gradients on the side stream, in stream order behind the kernels
            # that produced them; batch-norm gradients that are already global and the fused MoE weights stay out
            if not self.dp:
                return
            self._aux.wait_stream(main)
            with torch.cuda.stream(self._aux):
                for lo, hi in tw.grad_ranges(exclude=skip, only=stages[i]):
                    self.reducer.reduce_async(tw.store.grad, lo, hi)

        tw.backward(self._dp, moe_weight_grads=(not fuse) or route_rs, on_stage=on_stage)
        if self.dp:
            main.wait_stream(self._aux)
        if apply:
            lr = exponential_decay(self.lr0, self.global_step, B * self.world, self.lr_decay_examples, self.lr_decay)
            l2c = self.reg_pen * 1e-8
            if fuse:
                tw.begin_update()
                if route_rs:
                    self.moe.sharded_update(tw.adam_lr_t(lr), self.clip, l2c, self.reducer)
                else:
                    self.moe.fused_update(tw.adam_lr_t(lr), self.clip, l2c, dp=self.reducer if self.dp else None)
                tw.apply_group([k for k in tw.names if k not in fused_names], lr, self.clip, l2c)
            else:
                tw.apply_gradients(lr, self.clip, l2c)
            self.global_step += 1
        return {"predictions": pred, "loss": self.losses[0], "global_step": self.global_step}
