#!/usr/bin/env python3
"""Headline benchmark: frames/sec of the H-LSTM teacher+student training
iteration (BASELINE.json metric) on synthetic [B,300,1152] inputs.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full training iteration of cs/train.py:516-517 (teacher
fwd+bwd+update and student fwd+bwd+update) on one batch of B=256 videos x 300
frames x 1152 features PER GPU (weak scaling).  Inputs are resident in HBM when
the timed region starts.  Rank 0 prints ONE JSON line.

The headline (`value`, `ms_per_step`, `roofline`) is BASELINE cfg 3 in plain bf16 (one MFMA product per
contraction, as north_star prescribes).  At N=1 the same line also carries, each timed the same way on a short run:
  precision_modes   the same step in the "high" forward mode (IEEE f16 operands + low-order corrections on the MX-scaled MFMA,
                    DESIGN.md 7: the mode that holds the 1e-3 logit tolerance on trained-magnitude weights,
                    tests/test_gpu_step.py) next to the bf16 figure;
  other_configs     BASELINE cfg 2 (teacher only), cfg 5 (student only, every_n=30, B=1024), cfg 4 (DBoF + MoE,
                    B=512, with the roofline of its cluster GEMM) and the all-300-frames worst case;
  cpu_baseline      the PyTorch-CPU float32 restatement of the reference graph (oracle/torch_cpu.py) on the host cores.
`--config dbof` makes cfg 4 the timed workload of the line instead (its own metric string).

Every timed figure carries `ms_per_step` (mean over the K timed steps: wall clock between the two barriers, what the
driver's own clock checks), `ms_per_step_median` / `ms_per_step_max` (HIP events on the caller's stream after every step) and
`stall_suspected` (max > 3 x median: one queue stall inside the window moved the mean - read the median).

`--gpus N` means N: under a launcher (WORLD_SIZE set) it must equal the world; without one and N > 1 this process - which
never touches the GPU - starts N rank supervisors itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, each supervising its
benchmark child exactly as under a launcher).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

DTYPE_OF = {"bf16": "bf16", "high": "f16", "split": "bf16"}     # MFMA operand type of the forward products (accumulation f32; backward products bf16 in every mode)
PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
T_FRAMES, F_FEAT, V_CLS, H_CELLS = 300, 1152, 4716, 1024


def step_stats(per_step_ms, wall_ms_per_step=None, span_ms_per_step=None):
    """Robust companions of the mean: median and maximum of the per-step times and a flag for a window that contains a stall
    (one step > 3 x the median: e.g. the ~30-65 ms the driver's unmapping work can hold the queue right after tens of GB were
    freed - such a step moves the mean of a 20-step window of 2 ms steps by 2x and says nothing about the kernels; or the wall-clock
    mean more than 5 % above the event span: time lost before the first or behind the last event - seen once as +4 ms behind a
    20-step window of the cfg-4 high run).  span_ms_per_step = (event after the window's LAST enqueued work, i.e. after graph.flush() -
    first event) / steps: the deferred updates of the last step sit behind the last per-step event by design and are not a stall;
    wall_ms_per_step is this rank's own wall clock (before the MAX over the ranks)."""
    a = sorted(float(v) for v in per_step_ms)
    if not a:
        return {"ms_per_step_median": None, "ms_per_step_max": None, "stall_suspected": False}
    n = len(a)
    med = a[n // 2] if n % 2 else 0.5 * (a[n // 2 - 1] + a[n // 2])
    ref = span_ms_per_step if span_ms_per_step is not None else sum(a) / n
    stall = a[-1] > 3.0 * med or (wall_ms_per_step is not None and wall_ms_per_step > 1.05 * ref)
    return {"ms_per_step_median": round(med, 4), "ms_per_step_max": round(a[-1], 4), "stall_suspected": bool(stall)}


STAT_KEYS = ("ms_per_step_median", "ms_per_step_max", "stall_suspected")


class StepClock:
    """HIP events on the caller's stream: one before the first timed step, one after every step (graph.step() joins the graph's
    streams back into the caller's before it returns).  Recording costs ~2 us of host time per step and no GPU work."""

    def __init__(self):
        self.ev = []

    def tick(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.ev.append(e)

    def close(self):
        """One more event behind everything the window enqueued after its last step's tick (graph.flush())."""
        self.end = torch.cuda.Event(enable_timing=True)
        self.end.record()

    def per_step_ms(self):              # after a synchronize
        return [a.elapsed_time(b) for a, b in zip(self.ev[:-1], self.ev[1:])]

    def span_ms(self):                  # first tick -> close() (or the last tick)
        return self.ev[0].elapsed_time(getattr(self, "end", self.ev[-1]))


def synthetic_inputs(B, T, F, V, seed, device, all_full, as_uint8=False):
    """uint8-uniform features dequantised as cs/utils.py:22-25,
    n ~ U{120..300} with padded rows zeroed, ~3 positives per video (+class 0 w.p. 0.3).
    as_uint8: the frames stay the reader's uint8 quantisation (cs/readers.py:146-174 keeps them so up to the GPU; the input kernel
    dequantises and zeroes the padding rows itself) - 88 MB instead of 354 MB per batch of 256."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    q = torch.randint(0, 256, (B, T, F), generator=g, device=device, dtype=torch.uint8)
    if all_full:
        n = torch.full((B,), T, dtype=torch.int32, device=device)
    else:
        n = torch.randint(120, T + 1, (B,), generator=g, device=device, dtype=torch.int32)
    if as_uint8:
        x = q
    else:
        x = q.float() * (4.0 / 255.0) + (4.0 / 512.0 - 2.0)
        x[torch.arange(T, device=device)[None, :] >= n[:, None]] = 0.0
    labels = torch.zeros((B, V), dtype=torch.uint8, device=device)
    idx = torch.randint(0, V, (B, 3), generator=g, device=device)
    labels.scatter_(1, idx, 1)
    labels[torch.rand(B, generator=g, device=device) < 0.3, 0] = 1
    return x.contiguous(), n, labels


# ---------------------------------------------------------------------------------------------------------------
# GEMM FLOPs of one H-LSTM training iteration: nominal (every video counted at 300 frames,
# train = 3 x forward) and executed (what the launches of this batch actually contract over: the length-sorted L1
# stacks skip the padding rows - ops.RowPlan - and the hoisted backward products run on P = live rows)
# ---------------------------------------------------------------------------------------------------------------
def hlstm_gflop(n_host, mode, every_n, B, row_plans=True):
    from efficientvideoclassification_youtube8m_amd import ops
    H, F, V, K2 = H_CELLS, F_FEAT, V_CLS, 4 * H_CELLS
    nominal = executed = 0.0
    towers = []
    if mode != "student":
        towers.append((1, 20, 15, False))
    if mode != "teacher":
        S = 300 // every_n
        towers.append((every_n, 5, S // 5, True))
    for ev, C, Lc, sub in towers:
        _, l1, _ = ops.host_frame_counts(n_host, ev, C, Lc, 300, subsampled=sub)
        M = C * B
        rows = [int((l1 > t).sum()) for t in range(Lc)] if row_plans else [M] * Lc
        P = min(M, max(32, (rows[0] + 31) // 32 * 32)) if row_plans else M
        fwd_l1 = sum(2.0 * r * 4 * H * ((F + H) + (H if t > 0 else 0) + (H if t > 0 else 0)) for t, r in enumerate(rows))
        bptt_l1 = sum(2.0 * r * H * 4 * H * 2 for t, r in enumerate(rows) if t < Lc - 1)        # dh = dz . Wh^T, both layers
        hoisted_l1 = 2.0 * Lc * P * 4 * H * (H + (F + H) + 2 * H)                                 # dX of layer 1 + both dW
        fwd_l2 = C * 2.0 * B * 4 * H * ((K2 + H) + 2 * H)
        bwd_l2 = 2.0 * fwd_l2
        moe = 2.0 * B * K2 * V * 5
        executed += fwd_l1 + bptt_l1 + hoisted_l1 + fwd_l2 + bwd_l2 + moe * 4                    # fwd, dx, fused update: 2 passes
        per_video = Lc * C * 2.0 * 4 * H * (F + 3 * H) + C * 2.0 * 4 * H * (K2 + 3 * H) + 2.0 * K2 * V * 5
        nominal += 3.0 * B * per_video
    return nominal / 1e9, executed / 1e9


def run_hlstm(device, rank, world, B, mode, every_n, steps, warmup, all_full=False, precision="bf16", pool=8,
              overlap=True, fused_moe=True, student_forward_early=False, roofline=False, input_u8=False):
    """Times `steps` training iterations of one DistillGraph configuration (after `warmup`), inputs resident in HBM.
    Returns a dict; with roofline=True also the live timing of the teacher's L1 forward step launches."""
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    pool_in = [synthetic_inputs(B, T_FRAMES, F_FEAT, V_CLS, 1234 + rank + 1000 * i, device, all_full, as_uint8=input_u8) for i in range(pool)]
    # the frame counts also live on the host, as an input pipeline has them before the H2D copy (the launch
    # geometry of the length-sorted L1 stacks is derived from them, see ops.RowPlan)
    n_host = [p[1].cpu().numpy() for p in pool_in]
    graph = DistillGraph(B, every_n=every_n, mode=mode, device=device, seed=7, overlap_towers=overlap, precision=precision)
    if student_forward_early:
        graph.student_forward_early = True
    if not fused_moe:
        for tw in (graph.teacher, graph.student):
            if tw is not None:
                tw.fused_moe_update = False

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    it = 0
    # Settle (before the W warm-up steps, never inside the timed region): ONE pass over the pool.  The launch geometry follows the
    # batch (row plans: tile height per launch from the live rows), so a batch can be the first to use a kernel instantiation -
    # whose first launch costs 40-80 ms once (code-object load + hipFuncSetAttribute for its LDS size; seen as a single 40-84 ms
    # step somewhere among the first handful of steps of every process, scripts/step_times_probe.py).  After every batch of the
    # pool has run once nothing in the timed region is a first use.
    for i in range(pool):
        graph.step(pool_in[i][0], pool_in[i][2], pool_in[i][1], num_frames_host=n_host[i])
    torch.cuda.synchronize()
    for _ in range(warmup):
        x, n, labels = pool_in[it % pool]
        graph.step(x, labels, n, num_frames_host=n_host[it % pool])
        it += 1
    barrier()
    from efficientvideoclassification_youtube8m_amd.distill import GradReducer
    dp_on = graph.dp
    if dp_on:                                 # collectives of the TIMED steps only: payload bytes and HIP-event times per kind
        GradReducer.stats.clear()
        GradReducer.timing = []
    l1_stack = (graph.teacher if graph.teacher is not None else graph.student).l1
    if roofline:
        l1_stack.timing = []      # HIP events around the L1 forward launch sequences of the timed steps (launch stream)
        l1_stack.timing_bwd = {"bwd_step": [], "dx_nt": [], "wgrad_tn": []}      # ... and around the backward ones
    gf = [hlstm_gflop(n_host[(it + i) % pool], mode, every_n, B) for i in range(steps)]
    clock = StepClock()
    clock.tick()
    t0 = time.perf_counter()
    for _ in range(steps):
        x, n, labels = pool_in[it % pool]
        graph.step(x, labels, n, num_frames_host=n_host[it % pool])
        it += 1
        clock.tick()
    graph.flush()              # defer_updates: the last step's MoE / L2-level updates are enqueued and joined INSIDE the timed region
    clock.close()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0          # this rank's own wall clock (stall check only)
    barrier()
    dt = time.perf_counter() - t0
    per_step = clock.per_step_ms()
    if os.environ.get("EVC_BENCH_STEP_EVENTS") == "1":
        sys.stderr.write("[bench] per-step ms (events on the caller's stream): %s\n" % " ".join("%.1f" % v for v in per_step))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    res = {"ms_per_step": dt / steps * 1e3, "frames_per_sec": world * B * T_FRAMES * steps / dt, "steps": steps,
           "warmup": warmup, "batch_per_gpu": B, **step_stats(per_step, dt_local / steps * 1e3, clock.span_ms() / steps),
           "nominal_tflop_per_step": round(float(np.mean([g[0] for g in gf])) / 1e3, 3),
           "executed_tflop_per_step": round(float(np.mean([g[1] for g in gf])) / 1e3, 3),
           "losses": {k: round(v, 4) for k, v in graph.loss_report().items()},
           "schedule": {"defer_updates": bool(graph.defer_updates and not graph.dp), "student_forward_early": bool(graph.student_forward_early),
                        "student_forward_after_l1": bool(graph.student_forward_after_l1), "opt_cu_mask": os.environ.get("EVC_OPT_CU_MASK")}}
    res["executed_tflops"] = round(res["executed_tflop_per_step"] / (res["ms_per_step"] * 1e-3), 1)
    if precision != "bf16":
        res["high_layout"] = {tw.scope: tw.precision_layout() for tw in (graph.teacher, graph.student) if tw is not None}
    if dp_on:
        from efficientvideoclassification_youtube8m_amd.distill import serial_comm
        timing, GradReducer.timing = GradReducer.timing, None
        w = max(world, 1)
        sim = GradReducer._sim() if world == 1 else None
        if sim is not None:
            w = sim[3]
        elif graph.reducer.shard_world > 1:
            w = graph.reducer.shard_world
        per_kind = {}
        for kind, (calls, nbytes) in sorted(GradReducer.stats.items()):
            ms_k = sum(e0.elapsed_time(e1) for k, _, e0, e1 in timing if k == kind)
            wire = GradReducer.wire_bytes(kind, nbytes / steps, w)
            if sim is not None and kind == "all_gather_slabs" and graph.reducer.shard_world == 1:
                wire = (w - 1.0) / w * nbytes / steps
            per_kind[kind] = {"calls_per_step": round(calls / steps, 2), "payload_mb_per_step": round(nbytes / steps / 1e6, 3),
                              "wire_mb_per_rank_per_step": round(wire / 1e6, 3), "event_ms_per_step": round(ms_k / steps, 4)}
        res["dp"] = {"placement": ("EVC_DP_SERIAL_COMM=1: one communicator, every collective funnelled through one stream, backward phases issued in %s order"
                                   % graph.issue_order) if serial_comm() else "stream order, teacher and student towers on two communicators",
                     "attempt": int(os.environ.get("EVC_BENCH_ATTEMPT", "0")), "grad_dtype": graph.reducer.grad_dtype, "world": world,
                     "process_group": {"backend": torch.distributed.get_backend(), "ranks": torch.distributed.get_world_size(),
                                       "what": "nccl = RCCL (one rank per GPU); gloo only under the EVC_BENCH_SHARED_GPU test hook",
                                       "launched_by": "bench.py --gpus N itself" if os.environ.get("EVC_BENCH_SELF_LAUNCHED") == "1" else "launcher"},
                     "collectives": per_kind,
                     "wire_mb_per_rank_per_step": round(sum(v["wire_mb_per_rank_per_step"] for v in per_kind.values()), 2),
                     "collective_event_ms_per_step": round(sum(v["event_ms_per_step"] for v in per_kind.values()), 3),
                     "sim_world": graph.reducer.shard_world if graph.reducer.shard_world != max(world, 1) else None,
                     "sim": None if sim is None else {"busbw_gbps": sim[0], "blocks": sim[1], "lds_kb_per_block": sim[2], "world": sim[3],
                                                      "what": "one GPU, one-rank communicator: after every collective a stand-in kernel holds `blocks` "
                                                              "workgroups (256 threads, LDS as given) for wire bytes / busbw + 20 us"},
                     "note": "event_ms = HIP events around each collective on the stream it runs on (rank 0): kernel time incl. waiting for "
                             "the peers; the collectives of the two towers and the compute streams overlap, so the sum is not a share of the step"}
    if roofline and not l1_stack.timing:
        l1_stack.timing = l1_stack.timing_bwd = None
    elif roofline:
        # lstm_fwd_step_kernel<TileCfg3<BM,4,64,..>> (30 launches per iteration, the largest FLOP share of the
        # recurrent path).  Live timing with HIP events on the launch stream around each layer's 15-step launch
        # sequence in every timed step; algorithmic FLOPs = 2*rows_t*4H*K of each step GEMM over the rows that
        # step runs on.
        tower = graph.teacher if graph.teacher is not None else graph.student
        timing, l1_stack.timing = l1_stack.timing, None
        timing_bwd, l1_stack.timing_bwd = l1_stack.timing_bwd, None
        ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in timing)          # over the timed region, as the kernel ran there
        launches = sum(nl for _, _, nl, _ in timing)
        flops = sum(fl for _, _, _, fl in timing)
        ms_i = launches_i = flops_i = 0.0                                 # the same launch sequences alone on the chip
        if precision == "bf16":
            for (m, nl, fl) in tower.l1.profile_fwd_layers(reps=3):
                ms_i, launches_i, flops_i = ms_i + m, launches_i + nl, flops_i + fl
        traffic = mfma_busy = pmc_src = None
        for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):    # HBM bytes / MFMA busy from the committed --pmc passes
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    pmc = json.load(f)
                traffic = pmc["hbm_bytes_per_launch"]
                mfma_busy = round(pmc["mfma"]["mfma_busy_fraction"], 4)
                pmc_src = "profiles/%s (builder-run rocprofv3 --pmc passes of this command, not measured in this run)" % name
                break
            except Exception:
                pass
        achieved = flops / (ms * 1e-3) / 1e12
        f16 = precision != "bf16"
        res["roofline"] = {
            "bound": "mfma", "kernel": (("lstm_fwd_walk2_kernel<TileCfg3<BM,4,64,2,4,2>> (teacher L1; one launch = layer 0's step s + layer 1's step s-1, every workgroup walks "
                                         "both tiles; BM = 224..256 per launch from the active rows; lstm_fwd_step_kernel for 288/320-row launches)")
                                        if (not f16 and getattr(l1_stack, "fwd_walk2", False) and l1_stack.L == 2) else
                                        ("lstm_fwd_step_kernel<TileCfg3<BM,4,64,2,4,2>%s> (teacher L1; BM = 224..256 per launch, TileCfg2 for 288/320, from the "
                                         "active rows)" % (", F16" if f16 else ""))) if graph.teacher is not None else "lstm_fwd_step_kernel (student L1)",
            "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
            "traffic": None if f16 else traffic, "traffic_source": None if f16 else pmc_src, "mfma_busy_pmc": None if f16 else mfma_busy,
            "avg_launch_ms": round(ms / launches, 4),
            "launches_per_step": int(round(launches / max(1, steps))), "algorithmic_gflop_per_launch": round(flops / launches / 1e9, 2)}
        if f16 and tower.fp8_lo():
            res["roofline"]["kernel"] = ("lstm_fwd_step_kernel<TileCfg3<BM,4,64,2,4,2>, F16, FP8> (teacher L1; BM = 160..256 per launch from the active rows): "
                                         "f16 stages, then e4m3 stages on v_mfma_scale_f32_16x16x128_f8f6f4")
            k8_0 = (1.0 if input_u8 and tower.x_int() else 2.0) * F_FEAT + (2.0 if tower.act_lo() else 1.0) * H_CELLS     # e4m3 depth of layer 0
            k8_1 = 0.0 if 1 in tower.dither_layers() else (2.0 if tower.act_lo() else 1.0) * 2 * H_CELLS               # ... of layer 1 (dithered: none)
            res["roofline"]["note"] = ("algorithmic FLOPs count K = Kin + H once, priced against the dense bf16/f16 peak; executed: that contraction on IEEE f16 "
                                       "operands plus e4m3 correction stages at twice the MFMA rate - layer 0: K8 = %d (low-order halves of the weights, of h%s), "
                                       "layer 1: %s - i.e. %.2fx the algorithmic MFMA time over both layers (round 5: 1.31x, round 3's f16 K-extensions: 2.27x)"
                                       % (k8_0, ", of the f32 input frames" if not (input_u8 and tower.x_int()) else "; the uint8 frames are contracted as exact integers",
                                          "time-dithered f16 weight images, no e4m3 stages" if k8_1 == 0 else "K8 = %d" % k8_1,
                                          1.0 + 0.5 * (k8_0 + k8_1) / (F_FEAT + 3.0 * H_CELLS)))
        elif f16:
            res["roofline"]["note"] = ("IEEE f16 operands (same MFMA rate as bf16: priced against the same dense peak); algorithmic FLOPs count "
                                       "K = Kin + H once - layer 0 executes its input part %dx (K-extension by the low-order half), i.e. "
                                       "%.2fx the algorithmic MFMA work over both layers" % (
                                           tower.f16_x_segments, 1.0 + (tower.f16_x_segments - 1) * F_FEAT / (F_FEAT + 3.0 * H_CELLS)))
        else:
            res["roofline"]["isolated"] = {"achieved": round(flops_i / (ms_i * 1e-3) / 1e12, 2), "avg_launch_ms": round(ms_i / launches_i, 4),
                                           "note": "same launch sequences re-run alone after the timed loop (no other stream active)"}
        # The other GEMM-shaped launch sequences of the teacher's L1 level, timed the same way inside the same steps (events on the
        # stream each sequence is launched on; the student tower and the optimizer run beside them on the other streams, so
        # these are the rates inside the real step, not solo rates): the fused BPTT step, the hoisted dX product of the upper
        # layer and the TN weight-gradient products.
        who = "teacher" if graph.teacher is not None else "student"
        names = {"bwd_step": "lstm_bwd_step_kernel<TileCfg3<128,1,128,2,4,4>> (%s L1 BPTT: dh = dz_{t+1} . Wh^T + gate derivative tail)" % who,
                 "dx_nt": "gemm_nt_kernel (%s L1 upper layer: dX = dz . Wx^T over all steps, bf16 out)" % who,
                 "wgrad_tn": "gemm_tn_kernel<TileCfg2<256,1,256,2,4,5>> (%s L1: dW^T = dz^T . [x | h_prev] per layer, split-K, 1-2 launches per layer)" % who}
        rl = {"fwd_step": res["roofline"]}
        for kind, rows in timing_bwd.items():
            if not rows:
                continue
            ms_k = sum(e0.elapsed_time(e1) for e0, e1, _, _ in rows)
            nl_k = sum(nl for _, _, nl, _ in rows)
            fl_k = sum(fl for _, _, _, fl in rows)
            ach = fl_k / (ms_k * 1e-3) / 1e12
            rl[kind] = {"bound": "mfma", "kernel": names[kind], "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_BF16_TFLOPS, 4), "avg_launch_ms": round(ms_k / nl_k, 4),
                        "launches_per_step": int(round(nl_k / max(1, steps))), "algorithmic_gflop_per_launch": round(fl_k / nl_k / 1e9, 2),
                        "traffic": None}
        if "bwd_step" in rl:
            rl["bwd_step"]["note"] = ("memory-side kernel: ~113 MB of tape / dc / dz per launch are algorithmic (DESIGN.md 4.4); "
                                      "profiles/r0N_pmc_kernels.json has its HBM-side bytes")
        res["rooflines"] = rl
    # GAP@20 (cs/eval_util.py:61-79) of the last step's predictions, outside the timed regions: the second half of
    # BASELINE's metric name; on synthetic labels it only shows that the metric path runs on the step's outputs.
    from efficientvideoclassification_youtube8m_amd import eval_util
    gap_tower = graph.student if graph.student is not None else graph.teacher
    last_labels = pool_in[(it - 1) % pool][2]
    res["gap_at_20_last_batch"] = round(float(eval_util.calculate_gap(gap_tower.pred.float().cpu().numpy(),
                                                                      last_labels.float().cpu().numpy(), top_k=20)), 6)
    del graph, pool_in
    torch.cuda.empty_cache()
    return res


def run_dbof(device, rank, world, B, steps, warmup, pool=4, precision="bf16"):
    """BASELINE cfg 4: DbofModel (cluster 8192, hidden 1024, 30 sampled frames) + MoE(2), one training step per
    batch of B videos (cs/frame_level_models.py:108-195).  uint8 inputs resident in HBM."""
    from efficientvideoclassification_youtube8m_amd.distill import SingleTowerGraph
    from efficientvideoclassification_youtube8m_amd.towers import DbofTower
    g = torch.Generator(device=device)
    g.manual_seed(99 + rank)
    pool_in = []
    for i in range(pool):
        _, n, labels = synthetic_inputs(B, T_FRAMES, F_FEAT, V_CLS, 1234 + rank + 1000 * i, device, False)
        q = torch.randint(0, 256, (B, T_FRAMES, F_FEAT), generator=g, device=device, dtype=torch.uint8)   # as the reader delivers it
        pool_in.append((q, n, labels, torch.rand((B, 30), generator=g, device=device)))
    pg = None
    tw = DbofTower(B, T_FRAMES, F_FEAT, V_CLS, 30, 8192, 1024, 2, device=device, process_group=pg)
    if precision != "bf16":
        tw.set_precision(precision)
    graph = SingleTowerGraph(tw)
    # settle (as run_hlstm; before the warm-up, never inside the timed region): one pass over the pool, so that no timed step is the
    # first use of a kernel instantiation (40-80 ms once: code-object load + hipFuncSetAttribute) or of an allocator block
    for i in range(pool):
        x, n, labels, u = pool_in[i]
        graph.step(x, labels, n, uniform=u)
    torch.cuda.synchronize()
    for i in range(warmup):
        x, n, labels, u = pool_in[i % pool]
        graph.step(x, labels, n, uniform=u)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    if hasattr(tw, "timing"):
        tw.timing = []
    clock = StepClock()
    clock.tick()
    t0 = time.perf_counter()
    for i in range(steps):
        x, n, labels, u = pool_in[(warmup + i) % pool]
        graph.step(x, labels, n, uniform=u)
        clock.tick()
    clock.close()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_step = clock.per_step_ms()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    gflop_fwd = (2.0 * B * 30 * F_FEAT * 8192 + 2.0 * B * 8192 * 1024 + 2.0 * B * 1024 * V_CLS * 5) / 1e9
    res = {"ms_per_step": dt / steps * 1e3, "videos_per_sec": world * B * steps / dt,
           "frames_per_sec": world * B * T_FRAMES * steps / dt, "steps": steps, "warmup": warmup, "batch_per_gpu": B,
           **step_stats(per_step, dt_local / steps * 1e3, clock.span_ms() / steps),
           "nominal_tflop_per_step": round(3 * gflop_fwd / 1e3, 4), "loss": round(float(graph.losses[0]), 4)}
    res["nominal_tflops"] = round(res["nominal_tflop_per_step"] / (res["ms_per_step"] * 1e-3), 1)
    timing = getattr(tw, "timing", None)
    if timing:
        ms = sum(e0.elapsed_time(e1) for e0, e1 in timing) / len(timing)
        flops = 2.0 * B * 30 * F_FEAT * 8192
        ach = flops / (ms * 1e-3) / 1e12
        res["roofline"] = {"bound": "mfma", "kernel": ("dbof_cluster_pool_walk_kernel (cluster GEMM + BN statistics + per-video max/min; one workgroup per CU walks the row "
                                                       "tiles of one W_c column panel)" if precision == "bf16" else
                                                       "dbof_cluster_pool_kernel<FP8> (f16 stages + e4m3 correction stages, cluster GEMM + BN statistics + per-video max/min)"),
                           "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                           "traffic": None, "avg_launch_ms": round(ms, 4), "algorithmic_gflop_per_launch": round(flops / 1e9, 2)}
    del graph, tw, pool_in
    torch.cuda.empty_cache()
    return res


def retime_on_stall(run):
    """Secondary configurations only (the headline times exactly the K steps it was asked for): a window with a stall in it is
    measured once more and the first attempt's figures stay in the line next to the second's."""
    r = run()
    if r.get("stall_suspected"):
        first = {k: r[k] for k in ("ms_per_step",) + STAT_KEYS}
        time.sleep(0.5)
        r = run()
        r["retimed_after_stall"] = first
    return r


def self_launch(n, argv, script=None):
    """`python bench.py --gpus N` without a launcher (N > 1, WORLD_SIZE unset): this process - it has not touched the GPU and never
    will - starts one rank supervisor per GPU as fresh children (never an exec) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; each
    of them supervises its benchmark child exactly as under torch.distributed.run (supervise_ranks).  Returns the worst exit code."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as sk:                      # a free port: +16..+18 (fallback rendezvous) and +40..+42 (agreement stores) follow it
            sk.bind(("127.0.0.1", 0))
            port = str(min(sk.getsockname()[1], 65000))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port, EVC_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env))
    worst = 0
    for pr in procs:
        rc = pr.wait()
        worst = worst if rc == 0 else (rc if worst == 0 else worst)
    return worst


def host_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU boxes show 256
    logical CPUs but run under a 16-CPU quota: 256 BLAS threads on 16 CPUs' worth of time spin in their barriers)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:     # cgroup v1
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                cores = min(cores, max(1, quota // period))
        except Exception:
            pass
    return cores


def cpu_baseline(every_n, batch=256, budget_s=84.0, threads=0):
    """the reference graph restated on PyTorch-CPU float32 (oracle/torch_cpu.py: one dynamic_rnn
    per chunk, autograd BPTT, per-tensor clip, TF-Adam), all host cores, on a BOUNDED sample of the headline workload:
    `batch` synthetic videos x 300 x 1152 per iteration (256 = the batch the metric is quoted on; the rate depends on it:
    1.7 k frames/s at 64, 5.3 k at 256 on 16 cores, profiles/r03_cpu_baseline_b256.json), one warm-up iteration + up to
    three timed ones inside `budget_s` seconds."""
    from oracle import model_math as mm
    from oracle import torch_cpu as tc
    cores = threads if threads > 0 else host_cores()
    torch.set_num_threads(cores)
    rng = np.random.default_rng(7)
    teacher = tc.to_torch(mm.init_hlstm_params(rng, dtype=np.float32))
    student = tc.to_torch(mm.init_hlstm_params(rng, dtype=np.float32))
    _, x, n, labels = mm.synthetic_batch(batch, seed=1234, dtype=np.float32)
    xt, yt = torch.from_numpy(x), torch.from_numpy(labels.astype(np.float32))
    opt_t, opt_s = tc.Adam(teacher), tc.Adam(student)
    t_start = time.perf_counter()
    tc.teacher_student_iteration(xt, n, yt, teacher, student, every_n, opt_t, opt_s)         # warm-up
    warm = time.perf_counter() - t_start
    times = []
    while len(times) < 3 and (not times or time.perf_counter() - t_start + np.mean(times) < budget_s):
        t0 = time.perf_counter()
        tc.teacher_student_iteration(xt, n, yt, teacher, student, every_n, opt_t, opt_s)
        times.append(time.perf_counter() - t0)
    dtm = float(np.mean(times))
    return {"value": batch * 300 / dtm, "unit": "frames/sec", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "%d timed full teacher+student training iterations (after 1 warm-up of %.1f s) of the PyTorch-CPU float32 "
                      "restatement of the reference graph (oracle/torch_cpu.py: 20+1 / 5+1 dynamic_rnn loops at batch B, autograd, "
                      "per-tensor clip, TF-Adam of 2x143M parameters) on %d synthetic videos x 300 x 1152, %.2f s per iteration"
                      % (len(times), warm, batch, dtm),
            "batch": batch, "iterations_timed": len(times), "sec_per_iteration": round(dtm, 3)}


def _agree_on_attempt(attempt, rc, rank, world, port, wait_s):
    """CPU-side agreement of the rank supervisors on one attempt's outcome (no GPU, no process group): every rank publishes its
    child's exit code in a TCPStore hosted by rank 0's supervisor and reads all of them; returns the worst one (non-zero when any
    rank failed - then EVERY rank retries, also those whose own child succeeded) or 1 when the exchange itself fails."""
    import datetime
    try:
        from torch.distributed import TCPStore
        store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), port, world, is_master=(rank == 0),
                         timeout=datetime.timedelta(seconds=wait_s), wait_for_workers=False)
        store.set("rc/%d/%d" % (attempt, rank), str(int(rc)))
        keys = ["rc/%d/%d" % (attempt, r) for r in range(world)]
        store.wait(keys, datetime.timedelta(seconds=wait_s))
        worst = max(abs(int(store.get(k))) for k in keys)
        store.set("seen/%d/%d" % (attempt, rank), "1")
        if rank == 0:                       # the host of the store leaves last
            store.wait(["seen/%d/%d" % (attempt, r) for r in range(world)], datetime.timedelta(seconds=30))
        return worst
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("[bench supervisor rank %d] could not agree on attempt %d with the other ranks (%s): treating it as failed\n" % (rank, attempt, e))
        return 1


def supervise_ranks(argv, script=None):
    """N > 1 only, one supervisor per rank, BEFORE anything touches the GPU: runs the benchmark in a child process and, if that
    child dies or exceeds its wall limit on ANY rank (a collective wedged on first contact with the fabric: the process group's
    watchdog aborts it after distill.dp_timeout()), starts the next placement of the collectives on the next rendezvous port:
      attempt 0  stream order, teacher and student towers on two communicators (the fastest on the one-rank probes)
      attempt 1  EVC_DP_SERIAL_COMM=1: ONE communicator, ONE communication stream, backward phases issued in readiness order
                 (DistillGraph.issue_order "interleaved": no collective queues behind the other tower's later ones)
      attempt 2  the same with the sequential issue order (round 3's conservative placement)
    The ranks AGREE on the outcome of each attempt through a CPU-side TCPStore (rank 0's supervisor hosts it on MASTER_PORT + 40 +
    attempt): a rank whose own child exited 0 retries with the others when any peer failed, so nobody starts a retry alone.
    A child is a fresh process (never an exec of one that has initialised the GPU).  Returns the exit code for this rank."""
    import subprocess
    attempts = [({}, "stream order, two communicators")]
    if os.environ.get("EVC_DP_SERIAL_COMM") != "1":
        attempts.append(({"EVC_DP_SERIAL_COMM": "1"}, "EVC_DP_SERIAL_COMM=1, readiness issue order"))
    if os.environ.get("EVC_ISSUE_ORDER") != "sequential":
        attempts.append(({"EVC_DP_SERIAL_COMM": "1", "EVC_ISSUE_ORDER": "sequential"}, "EVC_DP_SERIAL_COMM=1, sequential issue order"))
    port = int(os.environ.get("MASTER_PORT", "29500"))
    limit = float(os.environ.get("EVC_BENCH_ATTEMPT_S", "600"))
    rank = os.environ.get("RANK", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rc = 1
    for i, (extra, name) in enumerate(attempts):
        env = dict(os.environ, EVC_BENCH_CHILD="1", EVC_BENCH_ATTEMPT=str(i))
        env.update(extra)
        if i > 0:      # the launcher's store on MASTER_PORT belongs to attempt 0: the fallback ranks rendezvous among themselves
            env["MASTER_PORT"] = str(port + 16 + i)
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        t0 = time.perf_counter()
        child = subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env)
        try:
            rc = child.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            child.kill()
            child.wait()
            rc = 124
        own = rc
        if world > 1 and os.environ.get("EVC_BENCH_NO_AGREEMENT") != "1":
            rc = _agree_on_attempt(i, rc, int(rank), world, port + 40 + i, limit + 120.0)
        if rc == 0:
            return 0
        sys.stderr.write("[bench supervisor rank %s] attempt %d (%s) ended with code %s here, %s over all ranks, after %.0f s%s\n" % (
            rank, i, name, own, rc, time.perf_counter() - t0, "; retrying with the next placement" if i + 1 < len(attempts) else ""))
        sys.stderr.flush()
    return rc or 1


def _log(msg):
    if os.environ.get("EVC_BENCH_VERBOSE") == "1":
        sys.stderr.write("[bench %.1fs] %s\n" % (time.perf_counter() - _T0, msg))
        sys.stderr.flush()


_T0 = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="hlstm", choices=["hlstm", "dbof"], help="hlstm: BASELINE cfg 3 (the metric); dbof: cfg 4")
    ap.add_argument("--batch", type=int, default=None, help="videos per GPU (default 256; 512 for --config dbof)")
    ap.add_argument("--global_batch", type=int, default=None, help="STRONG scaling (SURVEY.md 8(d): 'also report global-256'): this many videos per step over all "
                    "GPUs, global_batch / gpus each (must divide); the line says \"scaling\": \"strong\".  Default: --batch per GPU, weak")
    ap.add_argument("--every_n", type=int, default=10)
    ap.add_argument("--mode", default="teacher_student", choices=["teacher_student", "teacher", "student"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "high", "split"])
    ap.add_argument("--all_full", action="store_true", help="every video has 300 frames (no padding)")
    ap.add_argument("--input", default="f32", choices=["f32", "uint8"], help="frame tensor resident in HBM: f32 (dequantised, the rounds 1-5 input) or the reader's "
                    "uint8 quantisation (cs/readers.py:146-174: what the pipeline delivers; Dequantize fused into the input kernel)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_secondary", action="store_true", help="skip the precision_modes / other_configs runs (N=1 only anyway)")
    ap.add_argument("--cpu_videos", type=int, default=256, help="batch of the CPU leg: 256 = the batch the metric is quoted on (SURVEY 8(d)); "
                    "64 was the round 1-2 sample and runs 3x slower per frame (1.7 k vs 5.3 k frames/s on 16 cores: the BLAS sees a smaller M)")
    ap.add_argument("--cpu_baseline_only", action="store_true", help="print only the cpu_baseline object (e.g. --cpu_videos 256 --cpu_budget 400: "
                    "the batch SURVEY 8(d) specifies; profiles/r03_cpu_baseline_b256.json)")
    ap.add_argument("--cpu_threads", type=int, default=0, help="threads of the CPU leg (default: every core this process may use)")
    ap.add_argument("--cpu_budget", type=float, default=84.0, help="seconds the CPU leg may take (1 warm-up + up to 3 timed iterations; "
                    "at B = 256 on 16 cores: 17 s + 3 x 16 s)")
    ap.add_argument("--no_fused_moe", action="store_true", help="debug: materialise the MoE weight gradients (A/B of evc_moe_grad_update)")
    ap.add_argument("--student_forward_early", action="store_true", help="A/B: student forward next to the teacher forward")
    ap.add_argument("--no_overlap", action="store_true", help="debug: everything on one stream (solo kernel times for profiling)")
    ap.add_argument("--pool", type=int, default=8, help="distinct synthetic batches cycled through (HBM resident)")
    args = ap.parse_args()

    if args.cpu_baseline_only:
        print(json.dumps({"cpu_baseline": cpu_baseline(args.every_n, args.cpu_videos, args.cpu_budget, args.cpu_threads)}))
        return
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))     # (this process never touches the GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: the line's n_gpus would not be what was asked "
                         "for; start it with --nproc-per-node %d (or without a launcher: bench.py starts the ranks itself)\n"
                         % (args.gpus, world, args.gpus))
        sys.exit(2)
    if world > 1 and os.environ.get("EVC_BENCH_CHILD") != "1" and os.environ.get("EVC_BENCH_NO_SUPERVISOR") != "1":
        sys.exit(supervise_ranks(sys.argv[1:]))            # (this process never touches the GPU)
    # test hook (tests/test_gpu_dp.py): several ranks on ONE GPU over gloo, to exercise this file's multi-rank
    # path on a single-GPU box (RCCL refuses two ranks per device).  Never set in a real run.
    if os.environ.get("EVC_BENCH_SHARED_GPU") == "1":
        local_rank = 0
    # debug (scripts/rccl_one_rank.sh): EVC_DP_FORCE=1 under a one-process launcher runs the step's collectives
    # on a one-rank RCCL communicator
    one_rank_dp = world == 1 and os.environ.get("EVC_DP_FORCE") == "1" and "MASTER_PORT" in os.environ
    if world > 1 or one_rank_dp:
        # the process group comes first: nothing has touched the GPU yet (and nothing below ever re-executes this process)
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from efficientvideoclassification_youtube8m_amd.distill import dp_timeout
        if os.environ.get("EVC_BENCH_SHARED_GPU") == "1":
            torch.distributed.init_process_group("gloo", timeout=dp_timeout())
        else:
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=dp_timeout())
    n_gpus = world if world > 1 else 1
    device = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)

    from efficientvideoclassification_youtube8m_amd import ops
    ops.check_device(local_rank)

    if args.config == "dbof":
        B = args.batch or 512
        r = run_dbof(device, rank, world, B, args.steps, args.warmup, precision=args.precision)
        if rank == 0:
            res = {"metric": "frames/sec (whole node) DBoF(8192,1024)+MoE(2) training step B=512x300x1152 (BASELINE cfg 4)",
                   "value": r["frames_per_sec"], "unit": "frames/sec", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": r["ms_per_step"], **{k: r[k] for k in STAT_KEYS}, "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": DTYPE_OF[args.precision], "precision_mode": args.precision, "data": "synthetic",
                   "config": {"workload": "DbofModel cluster 8192, hidden 1024, 30 sampled frames, MoE(2), batch %d x 300 x 1152 per GPU" % B,
                              "global_batch": B * n_gpus, "frames_per_video": T_FRAMES, "parallelism": "dp%d" % n_gpus},
                   "videos_per_sec": r["videos_per_sec"], "nominal_tflop_per_step": r["nominal_tflop_per_step"], "loss": r["loss"]}
            if "roofline" in r:
                res["roofline"] = r["roofline"]
            print(json.dumps(res))
        if world > 1 or one_rank_dp:
            torch.distributed.destroy_process_group()
        return

    B = args.batch or 256
    if args.global_batch:
        if args.global_batch % n_gpus or args.batch:
            raise SystemExit("--global_batch %d must be a multiple of --gpus %d (and excludes --batch)" % (args.global_batch, n_gpus))
        B = args.global_batch // n_gpus
    head = run_hlstm(device, rank, world, B, args.mode, args.every_n, args.steps, args.warmup, args.all_full, args.precision,
                     args.pool, not args.no_overlap, not args.no_fused_moe, args.student_forward_early, roofline=True, input_u8=args.input == "uint8")
    _log("headline done: %.2f ms/step" % head["ms_per_step"])
    secondary = n_gpus == 1 and not args.no_secondary and not one_rank_dp
    extra = {}
    if secondary:
        s_steps, s_warm = 5, 2
        keep = ("ms_per_step",) + STAT_KEYS + ("frames_per_sec", "steps", "warmup", "batch_per_gpu", "nominal_tflop_per_step",
                                                 "executed_tflop_per_step", "executed_tflops", "retimed_after_stall")
        pm = {args.precision: {k: head[k] for k in keep if k in head}}
        other = "high" if args.precision == "bf16" else "bf16"
        r = retime_on_stall(lambda: run_hlstm(device, rank, world, B, args.mode, args.every_n, 10, 3, args.all_full, other, 4, roofline=True,
                                              input_u8=args.input == "uint8"))
        pm[other] = {k: r[k] for k in keep if k in r}
        if "roofline" in r:
            pm[other]["roofline"] = r["roofline"]
        _log("precision mode %s done: %.2f ms/step" % (other, r["ms_per_step"]))
        pm["bf16"]["logits_within_1e-3_of_f64_oracle"] = "at the reference's initialisation (|logit| <~ 1); ~1e-3*|logit| on trained weights"
        pm["high"]["logits_within_1e-3_of_f64_oracle"] = "also on towers trained for 16 / 128 / 512 steps (tests/test_gpu_step.py::test_high_mode_holds_1e3_after_long_training)"
        pm["high"]["what"] = ("forward operands chosen by a measured error budget (scripts/precision_budget.py, DESIGN.md 5): every forward product on IEEE "
                              "f16 operands (one MFMA product per depth) with the low-order halves of its weights AND of its activations h (round 6) as OCP e4m3 "
                              "operands on the MX-scaled MFMA behind the f16 stages of the same launch - L1 level (evc_lstm_layer_fwd_f16_fp8lo h_lo: + the low-order "
                              "half of f32 input frames; uint8 frames are contracted as exact integers, other_configs.cfg3_uint8_input_b256), L2 level wavefront pair "
                              "launches (evc_lstm_stack2_fwd_f16_fp8lo h_lo), MoE head (evc_gemm_nt_f16_fp8_dyn: both operands' corrections, the e4m3 range of the "
                              "input state from the batch); the TOP L1 layer of both towers on time-dithered f16 weight images (evc_lstm_layer_fwd_f16_dith); all "
                              "operand images written by the optimizer kernels' epilogues; backward products bf16 as in the bf16 mode.  Towers trained for 512 "
                              "steps, 12 weight draws: teacher logits max 5.2e-4 (uint8 frames) / 8.5e-4 (f32 frames), student 3.5e-4 "
                              "(profiles/r06_precision_robustness_long.txt; the round-5 layout: 2.8e-3)")
        pm["high"]["layout"] = r.get("high_layout") if other == "high" else head.get("high_layout")
        extra["precision_modes"] = pm
        oc = {}
        if args.input == "f32":      # the same step fed the reader's uint8 frames ("keep uint8 to the GPU"), both forward modes
            u8 = {}
            for prec in ("bf16", "high"):
                r = retime_on_stall(lambda: run_hlstm(device, rank, world, B, args.mode, args.every_n, 10, 3, args.all_full, prec, 4, input_u8=True))
                u8[prec] = {k: r[k] for k in keep if k in r}
            u8["high_over_bf16"] = round(u8["high"]["ms_per_step"] / u8["bf16"]["ms_per_step"], 4)
            oc["cfg3_uint8_input_b256"] = u8
            _log("uint8 input done: %.2f / %.2f ms/step" % (u8["bf16"]["ms_per_step"], u8["high"]["ms_per_step"]))
        # dominant = the launch sequence with the largest share of the step's kernel time in that configuration's rocprofv3 digest
        # (profiles/r06_digest_cfg2.txt / _cfg5.txt), timed live with HIP events like the headline's forward step - by time, not by habit
        for name, kw in (("cfg2_teacher_only_b256", dict(B=256, mode="teacher", every_n=10, dominant="bwd_step")),
                         ("cfg5_student_only_every_n30_b1024", dict(B=1024, mode="student", every_n=30, dominant="wgrad_tn")),
                         ("cfg3_all_300_frames_b256", dict(B=256, mode="teacher_student", every_n=10, all_full=True))):
            # (cfg 2 / cfg 5: 10 steps - a 5-step window of 4 ms steps read 3-5 % above the 20-step figure of the same box)
            r = retime_on_stall(lambda: run_hlstm(device, rank, world, kw["B"], kw["mode"], kw["every_n"], 10 if "dominant" in kw else s_steps,
                                                  3 if "dominant" in kw else s_warm, kw.get("all_full", False), "bf16", 4, roofline="dominant" in kw))
            oc[name] = {k: r[k] for k in keep if k in r}
            if "dominant" in kw and kw["dominant"] in r.get("rooflines", {}):
                oc[name]["roofline"] = dict(r["rooflines"][kw["dominant"]], dominant_by="share of the step's kernel time (profiles/r06_digest_%s.txt)" % name.split("_")[0])
                oc[name]["rooflines"] = {k: {kk: v[kk] for kk in ("frac", "avg_launch_ms", "launches_per_step", "algorithmic_gflop_per_launch")}
                                         for k, v in r["rooflines"].items()}
            _log("%s done: %.2f ms/step" % (name, r["ms_per_step"]))
        # (a 2 ms step: 20 steps and a pause first - right after the tens of GB of the previous configuration are freed the
        #  driver's unmapping work can stall the queue for ~65 ms once, which a 5-step window reported as 15 ms per step)
        time.sleep(0.5)
        r = retime_on_stall(lambda: run_dbof(device, rank, world, 512, 20, 6))
        oc["cfg4_dbof_8192_1024_moe2_b512"] = r
        _log("dbof done: %.2f ms/step" % r["ms_per_step"])
        time.sleep(1.0)
        rh = retime_on_stall(lambda: run_dbof(device, rank, world, 512, 20, 10, precision="high"))   # the mode that holds 1e-3 on its predictions (bf16: 2.3e-3)
        oc["cfg4_dbof_8192_1024_moe2_b512"]["high"] = {k: rh[k] for k in ("ms_per_step",) + STAT_KEYS + ("videos_per_sec", "frames_per_sec", "steps", "warmup",
                                                                                                      "loss", "retimed_after_stall") if k in rh}
        oc["cfg4_dbof_8192_1024_moe2_b512"]["high"]["what"] = ("IEEE f16 operands with both operands' low-order corrections as OCP e4m3 stages behind them in the same launch "
                                                               "(cluster, hidden and MoE products; EVC_HIGH_FP8_LO=0: three split-bf16 products per contraction): "
                                                                  "predictions 5.6e-6 from the float64 oracle at these dims (tests/test_gpu_dbof_logistic.py)")
        _log("dbof high done: %.2f ms/step" % rh["ms_per_step"])
        extra["other_configs"] = oc

    if rank == 0:
        res = {
            "metric": "frames/sec (whole node) H-LSTM teacher+student B=256x300x1152; GAP@20",
            "value": head["frames_per_sec"], "unit": "frames/sec", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], **{k: head[k] for k in STAT_KEYS}, "higher_is_better": True,
            "scaling": "strong" if args.global_batch else "weak",
            "vs_baseline": None, "dtype": DTYPE_OF[args.precision], "data": "synthetic", "precision_mode": args.precision,
            "config": {"workload": "HierarchicalLstmModel %s every_n=%d, lstm_cells=1024x2, MoE(2), batch %d x 300 x 1152 per GPU"
                                   % (args.mode, args.every_n, B),
                       "global_batch": B * n_gpus, "frames_per_video": T_FRAMES, "parallelism": "dp%d" % n_gpus,
                       "num_frames": "all 300" if args.all_full else "U{120..300}",
                       "input": "f32 frames resident in HBM (uint8-uniform values dequantised as cs/utils.py:22-25)" if args.input == "f32" else
                                "the reader's uint8 frames resident in HBM (Dequantize + padding zeroing fused into the input kernel)",
                       "tflop_per_step_per_gpu": head["nominal_tflop_per_step"],
                       "executed_tflop_per_step_per_gpu": head["executed_tflop_per_step"]},
            "executed_tflops_per_gpu": head["executed_tflops"],
            "losses": head["losses"], "gap_at_20_last_batch": head["gap_at_20_last_batch"],
            "roofline": head.get("roofline"), "rooflines": head.get("rooflines"), "schedule": head.get("schedule"),
        }
        if "dp" in head:
            res["dp"] = head["dp"]
        res.update(extra)
    # The CPU leg runs on rank 0 AFTER the process group is gone (no peer waits inside a collective while rank 0 spends a minute on the
    # host cores; the other ranks have exited by then and left the cores to it), so the N > 1 line carries cpu_baseline next to roofline too.
    if world > 1 or one_rank_dp:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.every_n, args.cpu_videos, args.cpu_budget, args.cpu_threads)
            if n_gpus > 1:
                res["cpu_baseline"]["sample"] += "; timed on rank 0 after destroy_process_group() (the other ranks have exited)"
        print(json.dumps(res))


if __name__ == "__main__":
    main()
