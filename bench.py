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

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def synthetic_inputs(B, T, F, V, seed, device, all_full):
    """uint8-uniform features dequantised as cs/utils.py:22-25,
    n ~ U{120..300} with padded rows zeroed, ~3 positives per video (+class 0 w.p. 0.3)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    q = torch.randint(0, 256, (B, T, F), generator=g, device=device, dtype=torch.uint8)
    if all_full:
        n = torch.full((B,), T, dtype=torch.int32, device=device)
    else:
        n = torch.randint(120, T + 1, (B,), generator=g, device=device, dtype=torch.int32)
    x = q.float() * (4.0 / 255.0) + (4.0 / 512.0 - 2.0)
    x[torch.arange(T, device=device)[None, :] >= n[:, None]] = 0.0
    labels = torch.zeros((B, V), dtype=torch.uint8, device=device)
    idx = torch.randint(0, V, (B, 3), generator=g, device=device)
    labels.scatter_(1, idx, 1)
    labels[torch.rand(B, generator=g, device=device) < 0.3, 0] = 1
    return x.contiguous(), n, labels


def cpu_baseline(every_n, sample_videos=8):
    """The oracle (numpy port of the reference graph, float32, BLAS threads = host
    cores) timed on a bounded sample of the same workload: one full training
    iteration (fwd + bwd of both towers + clip/Adam) on `sample_videos` videos."""
    from oracle import model_math as mm
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    rng = np.random.default_rng(7)
    dt = np.float32
    teacher = mm.init_hlstm_params(rng, dtype=dt)
    student = mm.init_hlstm_params(rng, dtype=dt)
    _, x, n, labels = mm.synthetic_batch(sample_videos, seed=1234, dtype=dt)
    t0 = time.perf_counter()
    out = mm.teacher_student_step(x, n, labels, teacher, student, every_n)
    t_fb = time.perf_counter() - t0
    t0 = time.perf_counter()
    mm.apply_train_op(teacher, out["teacher_grads"], {}, 1, 1e-3, 1.0)
    mm.apply_train_op(student, out["student_grads"], {}, 1, 1e-3, 1.0)
    t_opt = time.perf_counter() - t0
    dtm = t_fb + t_opt
    return {"value": sample_videos * 300 / dtm, "unit": "frames/sec", "cores": int(cores), "kind": "port",
            "extrapolated_b256": 256 * 300 / (t_fb * 256 / sample_videos + t_opt),
            "sample": "1 full teacher+student training iteration (float32 numpy/OpenBLAS oracle) on %d synthetic "
                      "videos x 300 x 1152: fwd+bwd %.1f s (scales with batch) + clip/Adam of 2x143M params %.1f s "
                      "(fixed per step); extrapolated_b256 = 76800/(fwd+bwd*256/%d + Adam)"
                      % (sample_videos, t_fb, t_opt, sample_videos)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="videos per GPU")
    ap.add_argument("--every_n", type=int, default=10)
    ap.add_argument("--mode", default="teacher_student", choices=["teacher_student", "teacher", "student"])
    ap.add_argument("--all_full", action="store_true", help="every video has 300 frames (no padding)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_videos", type=int, default=8)
    ap.add_argument("--no_fused_moe", action="store_true", help="debug: materialise the MoE weight gradients (A/B of evc_moe_grad_update)")
    ap.add_argument("--student_forward_early", action="store_true", help="A/B: student forward next to the teacher forward")
    ap.add_argument("--no_overlap", action="store_true", help="debug: everything on one stream (solo kernel times for profiling)")
    ap.add_argument("--pool", type=int, default=8, help="distinct synthetic batches cycled through (HBM resident)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tests/test_gpu_dp.py): several ranks on ONE GPU over gloo, to exercise this file's multi-rank
    # path on a single-GPU box (RCCL refuses two ranks per device).  Never set in a real run.
    if os.environ.get("EVC_BENCH_SHARED_GPU") == "1":
        local_rank = 0
    # debug (scripts/rccl_one_rank.sh): EVC_DP_FORCE=1 under a one-process launcher runs the step's collectives
    # on a one-rank RCCL communicator
    one_rank_dp = world == 1 and os.environ.get("EVC_DP_FORCE") == "1" and "MASTER_PORT" in os.environ
    if world > 1 or one_rank_dp:
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("EVC_BENCH_SHARED_GPU") == "1":
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world if world > 1 else 1
    device = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)

    from efficientvideoclassification_youtube8m_amd import ops
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph

    ops.check_device(local_rank)
    B, T, F, V = args.batch, 300, 1152, 4716
    # a pool of distinct synthetic batches, all resident in HBM before the timed region
    pool = [synthetic_inputs(B, T, F, V, 1234 + rank + 1000 * i, device, args.all_full) for i in range(args.pool)]
    # the frame counts also live on the host, as an input pipeline has them before the H2D copy (the launch
    # geometry of the length-sorted L1 stacks is derived from them, see ops.RowPlan)
    n_host = [p[1].cpu().numpy() for p in pool]
    graph = DistillGraph(B, every_n=args.every_n, mode=args.mode, device=device, seed=7, overlap_towers=not args.no_overlap)
    graph.student_forward_early = args.student_forward_early
    if args.no_fused_moe:
        for tw in (graph.teacher, graph.student):
            if tw is not None:
                tw.fused_moe_update = False

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        x, n, labels = pool[it % len(pool)]
        graph.step(x, labels, n, num_frames_host=n_host[it % len(pool)])
        it += 1
    barrier()
    l1_stack = (graph.teacher if graph.teacher is not None else graph.student).l1
    l1_stack.timing = []          # HIP events around the L1 forward launch sequences of the timed steps (launch stream)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x, n, labels = pool[it % len(pool)]
        graph.step(x, labels, n, num_frames_host=n_host[it % len(pool)])
        it += 1
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    losses = graph.loss_report()

    # ---- roofline of the dominant kernel: the fused LSTM forward step of the teacher's L1 ---------
    # lstm_fwd_step_kernel<TileCfg2<BM,4,64,..>> (30 launches per iteration, the largest FLOP share of the
    # recurrent path).  Live timing with HIP events on the launch stream around each layer's 15-step launch
    # sequence in every timed step; algorithmic FLOPs = 2*rows_t*4H*K of each step GEMM over the rows that
    # step runs on.
    tower = graph.teacher if graph.teacher is not None else graph.student
    timing, l1_stack.timing = l1_stack.timing, None
    ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in timing)              # over the timed region, as the kernel ran there
    launches = sum(nl for _, _, nl, _ in timing)                          # (next to the student's forward on another stream)
    flops = sum(fl for _, _, _, fl in timing)
    avg_ms = ms / launches
    achieved = flops / (ms * 1e-3) / 1e12
    ms_i = launches_i = flops_i = 0.0                                     # the same launch sequences alone on the chip
    for (m, nl, fl) in tower.l1.profile_fwd_layers(reps=3):
        ms_i, launches_i, flops_i = ms_i + m, launches_i + nl, flops_i + fl
    achieved_isolated = flops_i / (ms_i * 1e-3) / 1e12
    traffic = mfma_busy = None
    try:   # HBM bytes per launch / MFMA busy fraction from the committed rocprofv3 --pmc passes (profiles/)
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            pmc = json.load(f)
        traffic = pmc["hbm_bytes_per_launch"]
        mfma_busy = round(pmc["mfma"]["mfma_busy_fraction"], 4)
    except Exception:
        pass
    # GAP@20 (cs/eval_util.py:61-79) of the last step's predictions, outside the timed regions (after the live kernel
    # timing above, which must run on a busy chip: a host-side pause first lets the clocks drop): the second half
    # of BASELINE's metric name; on synthetic labels it only shows that the metric path runs on the step's outputs.
    from efficientvideoclassification_youtube8m_amd import eval_util
    gap_tower = graph.student if graph.student is not None else graph.teacher
    last_labels = pool[(it - 1) % len(pool)][2]
    gap20 = float(eval_util.calculate_gap(gap_tower.pred.float().cpu().numpy(), last_labels.float().cpu().numpy(), top_k=20))
    roofline = {"bound": "mfma", "kernel": "lstm_fwd_step_kernel<TileCfg2<BM,4,64,2,4,..>> (teacher L1; BM = 224..320 per launch from the active rows)"
                if graph.teacher is not None else "lstm_fwd_step_kernel (student L1)", "achieved": round(achieved, 2),
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                "traffic": traffic, "mfma_busy_pmc": mfma_busy, "avg_launch_ms": round(avg_ms, 4),
                "launches_per_step": int(round(launches / max(1, args.steps))),
                "algorithmic_gflop_per_launch": round(flops / launches / 1e9, 2),
                "isolated": {"achieved": round(achieved_isolated, 2), "avg_launch_ms": round(ms_i / launches_i, 4),
                             "note": "same launch sequences re-run alone after the timed loop (no other stream active)"}}

    if rank == 0:
        frames = n_gpus * B * T * args.steps
        res = {
            "metric": "frames/sec (whole node) H-LSTM teacher+student B=256x300x1152; GAP@20",
            "value": frames / dt, "unit": "frames/sec", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "HierarchicalLstmModel %s every_n=%d, lstm_cells=1024x2, MoE(2), batch %d x 300 x 1152 per GPU"
                                   % (args.mode, args.every_n, B),
                       "global_batch": B * n_gpus, "frames_per_video": T, "parallelism": "dp%d" % n_gpus,
                       "num_frames": "all 300" if args.all_full else "U{120..300}",
                       "tflop_per_step_per_gpu": round(3 * B * ((11.748 if graph.teacher else 0) + (
                           (1.525 if args.every_n == 10 else 0.833) if graph.student else 0)) / 1e3, 3)},
            "losses": {k: round(v, 4) for k, v in losses.items()}, "gap_at_20_last_batch": round(gap20, 6),
            "roofline": roofline,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.every_n, args.cpu_videos)
        print(json.dumps(res))
    if world > 1 or one_rank_dp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
