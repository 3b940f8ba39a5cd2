"""Training binary with the reference's flag surface (cs/train.py, and
cs/train_finetune.py via ``--finetune``):

    python -m efficientvideoclassification_youtube8m_amd.train \
        --train_data_pattern synthetic --train_dir ./model_HLSTM_TeaStud_every10_train/ \
        --frame_features True --feature_names "rgb, audio" --feature_sizes "1024, 128" \
        --model "HierarchicalLstmModel" --gpu 0 --batch_size 256 --num_inputs_to_lstm 20 \
        --lstm_layers 2 --start_new_model True --num_epochs 1 --every_n 10      # = run_train.sh:6

What is kept: flag names/defaults/syntax (flags.py), class lookup by name
(``find_class_by_name``, cs/train.py:179-182), the teacher+student graph
(``build_graph`` -> distill.DistillGraph), the per-step log line
(cs/train.py:528-533), global_step += 2, resume-unless-``--start_new_model``.
What is replaced: TF Supervisor/queue runners -> a plain loop; TF checkpoints ->
``torch.save`` of a TF-named state dict (model.ckpt-<step>.pt, max_to_keep=1);
TFRecord input -> readers.py (native parser, uint8 feed, pinned staging); the pattern
"synthetic" generates random uint8 videos on the device instead.
Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N``;
``--gpu`` is then ignored in favour of LOCAL_RANK.
"""
from __future__ import annotations

import glob
import logging
import os
import sys
import time

import numpy as np
import torch

from . import eval_util, frame_level_models, losses, ops, readers, video_level_models
from .distill import DistillGraph, SingleTowerGraph
from .flags import FLAGS, GetListOfFeatureNamesAndSizes
from .towers import DbofTower, LogisticTower, NetVladTower

NUM_CLASSES = 4716       # readers.YT8MFrameFeatureReader default num_classes (cs/readers.py:121)


def find_class_by_name(name, modules):
    """Searches the provided modules for the named class and returns it (cs/train.py:179-182)."""
    found = [getattr(module, name, None) for module in modules]
    return next(a for a in found if a)


def _apply_precision(tw):
    """--precision for the single-tower models: towers with a split-bf16 forward take it (DbofTower, LogisticTower); a tower
    without one (NetVladTower) refuses anything but 'bf16' instead of silently ignoring the flag."""
    if FLAGS.precision != "bf16":
        if "high" not in getattr(tw, "PRECISIONS", ()):
            raise ValueError("--precision %s: %s has no such forward mode" % (FLAGS.precision, type(tw).__name__))
        tw.set_precision(FLAGS.precision)


def build_graph(model, label_loss_fn, feature_size, batch_size, every_n, device, finetune=False, process_group=None):
    """Equivalent of cs/train.py:185-427 (and cs/train_finetune.py:185-331 when
    finetune): returns the graph object whose ``step`` runs one iteration."""
    if not isinstance(label_loss_fn, losses.CrossEntropyLoss):
        raise NotImplementedError("only CrossEntropyLoss is fused into the training graph (SURVEY.md 8a row a6)")
    common = dict(base_learning_rate=FLAGS.base_learning_rate, learning_rate_decay=FLAGS.learning_rate_decay,
                  learning_rate_decay_examples=FLAGS.learning_rate_decay_examples,
                  regularization_penalty=FLAGS.regularization_penalty, clip_gradient_norm=FLAGS.clip_gradient_norm,
                  process_group=process_group)
    if isinstance(model, frame_level_models.HierarchicalLstmModel):
        # every_n == 1 (the reference's default, cs/train.py:100-101) still builds and trains model_student, on all 300
        # frames in 5 chunks of 60 (cs/train.py:262-272,349-356): global_step += 2 and the checkpoint holds both scopes.
        # Teacher-only training (BASELINE cfg 2) is not a reference mode: it is asked for with --teacher_only.
        mode = "student" if finetune else ("teacher" if getattr(FLAGS, "teacher_only", False) else "teacher_student")
        return DistillGraph(batch_size, every_n=every_n, mode=mode, feature_size=feature_size, vocab_size=NUM_CLASSES,
                            max_frames=FLAGS.max_num_frames, num_inputs_to_lstm=FLAGS.num_inputs_to_lstm,
                            lstm_cells=FLAGS.lstm_cells, lstm_layers=FLAGS.lstm_layers,
                            num_mixtures=FLAGS.moe_num_mixtures, device=device, precision=FLAGS.precision, **common)
    if isinstance(model, frame_level_models.DbofModel):
        tw = DbofTower(batch_size, FLAGS.max_num_frames, feature_size, NUM_CLASSES, FLAGS.iterations,
                       FLAGS.dbof_cluster_size, FLAGS.dbof_hidden_size, FLAGS.moe_num_mixtures, device=device,
                       process_group=process_group)
        _apply_precision(tw)
        return SingleTowerGraph(tw, **common)
    if isinstance(model, frame_level_models.NetVLADModel):          # extension (the reference's class is an empty stub)
        tw = NetVladTower(batch_size, FLAGS.max_num_frames, feature_size, NUM_CLASSES, FLAGS.iterations, FLAGS.netvlad_cluster_size,
                          FLAGS.netvlad_hidden_size, FLAGS.moe_num_mixtures, device=device, process_group=process_group)
        _apply_precision(tw)
        return SingleTowerGraph(tw, **common)
    if isinstance(model, frame_level_models.FrameLevelLogisticModel):
        tw = LogisticTower(batch_size, FLAGS.max_num_frames, feature_size, NUM_CLASSES, device=device)
        _apply_precision(tw)
        return SingleTowerGraph(tw, **common)
    raise NotImplementedError("model %s has no training graph (NeXtVLAD is an empty stub in the reference too)"
                              % type(model).__name__)


def synthetic_batches(batch_size, feature_size, device, videos_per_epoch, num_epochs, seed, drop_remainder=False):
    """Synthetic stand-in for get_input_data_tensors (cs/train.py:129-176): uint8
    features dequantised by the input kernel, n ~ U{120..300}, ~3 labels/video.
    Yields (features, labels, num_frames, num_frames on the host).  drop_remainder (data parallel): no smaller
    final batch."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    T = FLAGS.max_num_frames
    for _ in range(num_epochs):
        left = videos_per_epoch
        while left > 0:
            b = min(batch_size, left)                  # allow_smaller_final_batch=True (cs/train.py:175)
            left -= b
            if b < batch_size and drop_remainder:
                break
            q = torch.randint(0, 256, (b, T, feature_size), generator=g, device=device, dtype=torch.uint8)
            n = torch.randint(min(120, T), T + 1, (b,), generator=g, device=device, dtype=torch.int32)
            labels = torch.zeros((b, NUM_CLASSES), dtype=torch.uint8, device=device)
            labels.scatter_(1, torch.randint(0, NUM_CLASSES, (b, 3), generator=g, device=device), 1)
            yield q, labels, n, n.cpu().numpy()


def get_reader():
    """cs/train.py:618-630: the reader the flags select."""
    feature_names, feature_sizes = GetListOfFeatureNamesAndSizes(FLAGS.feature_names, FLAGS.feature_sizes)
    if FLAGS.frame_features:
        return readers.YT8MFrameFeatureReader(num_classes=NUM_CLASSES, feature_names=feature_names, feature_sizes=feature_sizes,
                                              max_frames=FLAGS.max_num_frames)
    return readers.YT8MAggregatedFeatureReader(num_classes=NUM_CLASSES, feature_names=feature_names, feature_sizes=feature_sizes)


LAST_BATCH = {"ids": None}


def get_input_data(data_pattern, batch_size, feature_size, device, num_epochs, seed, rank=0, world=1):
    """get_input_data_tensors (cs/train.py:129-176) -> iterator of (features uint8, labels uint8, num_frames int32)
    device tensors; ``batch_size`` is per GPU (cs/train.py:205 batch_size * num_towers)."""
    if data_pattern in ("", "synthetic"):
        return synthetic_batches(batch_size, feature_size, device, FLAGS.synthetic_videos, num_epochs, seed,
                                 drop_remainder=world > 1), None
    logging.info("Using batch size of %d for training.", batch_size)
    pipe = readers.get_input_data_tensors(get_reader(), data_pattern, batch_size=batch_size, num_epochs=num_epochs,
                                          num_readers=FLAGS.num_readers, seed=seed, device=device, rank=rank, world_size=world,
                                          with_host_counts=True)
    logging.info("Number of training files / records on this rank: %d / %d.", len(pipe.index), pipe.num_records)
    def batches():
        for b in pipe:
            LAST_BATCH["ids"] = b[0]                     # video ids of the batch being handed out (tests, debugging)
            yield b[1:]
    return batches(), pipe.num_batches


def _ckpt_step(path):
    name = os.path.basename(path)
    return int(name[len("model.ckpt-"):-3]) if name.startswith("model.ckpt-") else 0


def latest_checkpoint(train_dir):
    """tf.train.latest_checkpoint: the highest-numbered model.ckpt-<step>.pt; the un-numbered model.ckpt.pt that
    train_convert_model writes counts as step 0."""
    cks = glob.glob(os.path.join(train_dir, "model.ckpt-*.pt")) + glob.glob(os.path.join(train_dir, "model.ckpt.pt"))
    return max(cks, key=_ckpt_step) if cks else None


def save_checkpoint(graph, train_dir, rank):
    if hasattr(graph, "consolidate"):
        graph.consolidate()                 # collective: sharded optimizer state -> complete on every rank
    if rank != 0:
        return
    os.makedirs(train_dir, exist_ok=True)
    sd = {"global_step": graph.global_step}
    for tw in (getattr(graph, "teacher", None), getattr(graph, "student", None), getattr(graph, "tower", None)):
        if tw is not None:
            sd.update({k: v.cpu() for k, v in tw.state_dict().items()})
            sd["%s/adam" % tw.scope] = {"t": tw.adam_t, "m": tw.store.m.cpu(), "v": tw.store.v.cpu()}
            sd["%s/precision_layout" % tw.scope] = tw.precision_layout()      # metadata: the forward-operand layout these weights were trained under
    path = os.path.join(train_dir, "model.ckpt-%d.pt" % graph.global_step)
    # like tf.train.Saver: write to a temporary name, flush to disk, rename (a validate.py polling the directory never
    # sees a half-written file); the temporary name does not match the model.ckpt*.pt glob
    tmp = os.path.join(train_dir, ".tmp-%d-model.ckpt-%d" % (os.getpid(), graph.global_step))
    with open(tmp, "wb") as f:
        torch.save(sd, f)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)
    for old in glob.glob(os.path.join(train_dir, "model.ckpt*.pt")):       # max_to_keep=1 (cs/train.py:651)
        if old != path:
            try:
                os.remove(old)
            except FileNotFoundError:
                pass
    return path


def restore_checkpoint(graph, path):
    sd = torch.load(path, map_location="cpu")
    graph.global_step = int(sd["global_step"])
    for tw in (getattr(graph, "teacher", None), getattr(graph, "student", None), getattr(graph, "tower", None)):
        if tw is not None and any(k.startswith(tw.scope + "/") for k in sd):
            tw.load_state_dict(sd)
            ad = sd.get("%s/adam" % tw.scope)
            if ad:
                tw.adam_t = ad["t"]
                tw.store.m.copy_(ad["m"])
                tw.store.v.copy_(ad["v"])


def agree_step_limit(max_steps, num_batches, world, device=None):
    """Number of iterations every rank will run: None = until the data ends, an int (0 INCLUDED) = exactly that many.

    Ranks own different files, so under data parallelism they agree on MIN(whole batches) up front - no rank may wait in
    a gradient all-reduce for a peer whose input has run dry.  A rank with fewer records than one batch reports 0 whole
    batches (drop_remainder): the agreed limit is then 0 and NO rank enters the loop (a rank that did would sit in a
    gradient all-reduce while the empty one is already in save_checkpoint's consolidate(): mismatched collectives)."""
    limit = int(max_steps) if max_steps else None
    if world > 1 and num_batches is not None:
        nb = torch.tensor([int(num_batches)], device=device)
        torch.distributed.all_reduce(nb, op=torch.distributed.ReduceOp.MIN)
        limit = int(nb) if limit is None else min(limit, int(nb))
    return limit


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    finetune = "--finetune" in argv
    argv = [a for a in argv if a != "--finetune"]
    FLAGS.parse(argv)
    for k, v in FLAGS.flag_values_dict().items():
        print("Key: %s Value: %s" % (k, v))
    logging.basicConfig(level=logging.INFO, format="INFO:evc:%(message)s")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(FLAGS.gpu)))
    # test hook (tests/test_gpu_dp.py): several ranks on ONE GPU over gloo, to run this file's multi-rank path on a
    # single-GPU box (RCCL refuses two ranks per device).  Never set in a real run.
    shared_gpu = os.environ.get("EVC_TRAIN_SHARED_GPU") == "1"
    if shared_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = "cuda:%d" % local
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from .distill import dp_timeout
        if shared_gpu:
            torch.distributed.init_process_group("gloo", timeout=dp_timeout())
        else:
            torch.distributed.init_process_group("nccl", device_id=torch.device(device), timeout=dp_timeout())
    ops.check_device(local)
    task = "/job:master/task:%d" % rank
    _, feature_sizes = GetListOfFeatureNamesAndSizes(FLAGS.feature_names, FLAGS.feature_sizes)
    feature_size = sum(feature_sizes)
    model = find_class_by_name(FLAGS.model, [frame_level_models, video_level_models])()
    label_loss_fn = find_class_by_name(FLAGS.label_loss, [losses])()
    if FLAGS.optimizer != "AdamOptimizer":
        raise NotImplementedError("only AdamOptimizer (the reference default, cs/train.py:91) is built")
    graph = build_graph(model, label_loss_fn, feature_size, FLAGS.batch_size, FLAGS.every_n, device, finetune)
    logging.info("%s: Built graph.", task)
    ck = None if FLAGS.start_new_model else latest_checkpoint(FLAGS.train_dir)
    if FLAGS.start_new_model:
        logging.info("%s: Flag 'start_new_model' is set. Building a new model.", task)
    elif ck is None:
        logging.info("%s: No checkpoint file found. Building a new model.", task)
    else:
        logging.info("%s: Restoring from %s", task, ck)
        restore_checkpoint(graph, ck)
    data, num_batches = get_input_data(FLAGS.train_data_pattern, FLAGS.batch_size, feature_size, device, FLAGS.num_epochs,
                                       1234 + rank, rank, world)
    step_limit = agree_step_limit(FLAGS.max_steps, num_batches, world, device)
    if step_limit == 0:
        logging.warning("%s: a rank has fewer than one whole batch of %d records: no training step on any rank "
                        "(give every rank at least batch_size records)", task, FLAGS.batch_size)
    logging.info("%s: Entering training loop.", task)
    start, last_save, it = time.time(), time.time(), 0
    is_distill = isinstance(graph, DistillGraph)
    steps_per_it = 2 if is_distill and graph.teacher and graph.student else 1
    copy_stream = torch.cuda.Stream(device=device)
    host_bufs = {}                       # pinned staging, two alternating sets (one may still be read while the next fills)

    def snapshot(out, labels, it):
        """What the reference's sess.run fetch returns at a logging step (cs/train.py:515-526), taken WITHOUT stopping
        the GPU: device clones in stream order (the step's output buffers are overwritten by the next step), then D2H
        on a copy stream into pinned memory.  finish_log() reads it after the NEXT step has been enqueued, so the host
        metrics of step k are computed while the GPU runs step k+1; the log lines are the same, one step late."""
        pred = out.get("predictions", out.get("student_predictions")).clone()
        lab = labels.clone()
        loss_dev = graph.losses_for_report.clone() if is_distill else out["loss"].detach().clone().reshape(1)
        cur = torch.cuda.current_stream(device)
        copy_stream.wait_stream(cur)
        slot = host_bufs.setdefault(it % 2, {})
        with torch.cuda.stream(copy_stream):
            # the loss values travel to the host on EVERY rank (8 floats, already summed over the ranks inside the
            # step): the non-finite check below must stop all ranks together, not leave the others in a collective
            for key, t in ((("pred", pred), ("lab", lab), ("loss", loss_dev)) if rank == 0 else (("loss", loss_dev),)):
                if key not in slot or slot[key].shape != t.shape or slot[key].dtype != t.dtype:
                    slot[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                slot[key].copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        for t in (pred, lab, loss_dev):
            t.record_stream(copy_stream)
        return {"slot": slot, "event": ev, "global_step": out["global_step"], "batch": labels.shape[0]}

    last_log_time = [time.time(), 0]     # (wall time, iteration) of the previous log line: rates over the interval in between

    def finish_log(snap, it_now):
        if rank != 0 and not is_distill:
            return
        snap["event"].synchronize()              # nothing here touches a stream that step it_now+1 has been queued on
        r = graph.loss_report(losses=snap["slot"]["loss"]) if is_distill else None
        if is_distill and not all(np.isfinite(v) for v in r.values()):          # slim.learning.create_train_op's check_numerics;
            raise FloatingPointError("LossTensor is inf or nan : %s" % r)      # same (reduced) values on every rank: all stop
        if rank != 0:
            return
        p, y = snap["slot"]["pred"].numpy(), snap["slot"]["lab"].numpy().astype(np.float32)
        hit, perr, gap = (eval_util.calculate_hit_at_one(p, y), eval_util.calculate_precision_at_equal_recall_rate(p, y),
                          eval_util.calculate_gap(p, y))
        history.append((snap["global_step"], dict(r) if is_distill else {"loss": float(snap["slot"]["loss"][0])},
                        {"hit_at_one": float(hit), "perr": float(perr), "gap": float(gap)}))
        if is_distill:
            logging.info("%s: training step %d| Hit@1: %.2f| PERR: %.2f| GAP: %.2f| Teacher_Loss: %s| L_REP: %s| L_PRED: %s"
                         "| L_CE: %s", task, snap["global_step"], hit, perr, gap, round(r["label_loss"], 2),
                         round(r["student_loss_state"], 2), round(r["pred_loss"], 2), round(r["student_label_loss"], 2))
        else:
            logging.info("%s: training step %d| Hit@1: %.2f| PERR: %.2f| GAP: %.2f| Loss: %s", task, snap["global_step"],
                         hit, perr, gap, round(float(snap["slot"]["loss"][0]), 2))
        now = time.time()
        dt = max(now - last_log_time[0], 1e-9) / max(1, it_now - last_log_time[1])
        last_log_time[0], last_log_time[1] = now, it_now
        logging.info("global_step/sec: %g  Examples/Second: %g", steps_per_it / dt, snap["batch"] * world / dt)

    history = []                         # (global_step, loss dict) of every logged step, returned to the caller
    pending = None
    for q, labels, n, n_host in (data if step_limit != 0 else ()):
        out = graph.step(q, labels, n, num_frames_host=n_host) if is_distill else graph.step(q, labels, n)    # uint8 features: Dequantize is fused into every input kernel
        it += 1
        graph.last_batch_ids = LAST_BATCH["ids"]
        logging_step = it % max(1, FLAGS.log_every) == 0
        snap = snapshot(out, labels, it) if logging_step else None
        if pending is not None:
            finish_log(pending, it - 1)                                        # step it-1's metrics, under step it
        pending = snap
        # the checkpoint decision is collective under data parallelism: rank 0's clock decides, every 64 iterations
        save_due = time.time() - last_save > 30 * 60                           # save_model_secs (cs/train.py:500)
        if world > 1:
            save_due = False
            if it % 64 == 0:                                                   # (a host sync: not at every step)
                flag = torch.tensor([1 if time.time() - last_save > 30 * 60 else 0], device=device)
                torch.distributed.broadcast(flag, src=0)
                save_due = bool(flag.item())
        if save_due:
            save_checkpoint(graph, FLAGS.train_dir, rank)
            last_save = time.time()
        if step_limit is not None and it >= step_limit:
            break
    if pending is not None:
        finish_log(pending, it)
    logging.info("%s: Done training -- epoch limit reached.", task)
    save_checkpoint(graph, FLAGS.train_dir, rank)
    logging.info("%s: Exited training loop.", task)
    print("Total time taken is " + str(time.time() - start))
    if world > 1:
        torch.distributed.destroy_process_group()
    return {"graph": graph, "history": history, "iterations": it}


if __name__ == "__main__":
    main()
