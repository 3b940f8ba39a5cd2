"""Evaluation binary with the reference's flag surface: cs/validate.py (teacher +
student restored from a teacher-student checkpoint, student metrics + the
representation loss logged) and, with ``student_only``, cs/eval_finetune.py
(student restored from a finetune checkpoint).

    python -m efficientvideoclassification_youtube8m_amd.validate \
        --eval_data_pattern "./yt8m/validate*.tfrecord" --train_dir ./model_HLSTM_TeaStud_every10_train/ \
        --frame_features True --feature_names "rgb, audio" --feature_sizes "1024, 128" \
        --model "HierarchicalLstmModel" --gpu 0 --batch_size 512 --num_inputs_to_lstm 20 --lstm_layers 2 \
        --every_n 10 --top_k 20 --run_once True                                     # = run_validate.sh

Kept: flags, IOError texts, the per-batch and per-epoch log lines (cs/utils.py:35-126), the
EvaluationMetrics accumulation (Hit@1 / PERR / mAP / GAP@top_k, host float64), "skip this checkpoint"
when the global step has not moved, looping until --run_once.  Replaced: TF session / queue runners
-> readers.get_input_evaluation_tensors + distill.EvalGraph; events file -> events.jsonl.
``--eval_data_pattern synthetic`` evaluates ``--synthetic_videos`` random videos (no data set on the box).
"""
from __future__ import annotations

import logging
import sys
import time

import numpy as np
import torch

from . import eval_util, frame_level_models, losses, ops, readers, utils, video_level_models
from .distill import EvalGraph
from .flags import FLAGS
from .train import NUM_CLASSES, find_class_by_name, get_reader, latest_checkpoint, synthetic_batches


def get_input_evaluation_tensors(reader, data_pattern, batch_size=1024, num_readers=1, device=None):
    """cs/validate.py:70-104."""
    logging.info("Using batch size of " + str(batch_size) + " for evaluation.")
    try:
        pipe = readers.get_input_evaluation_tensors(reader, data_pattern, batch_size=batch_size, num_readers=num_readers, device=device,
                                                    with_host_counts=True)
    except IOError as e:
        if "Unable to find" in str(e):
            raise IOError("Unable to find the evaluation files.")
        raise
    logging.info("number of evaluation files: " + str(len(pipe.index)))
    return pipe


def build_graph(reader, model, batch_size, device, student_only=False):
    """cs/validate.py:107-189 / cs/eval_finetune.py:108-175."""
    if not isinstance(model, frame_level_models.HierarchicalLstmModel):
        raise NotImplementedError("validate.py unpacks the H-LSTM (state, result) pair (cs/validate.py:150,157); "
                                  "model %s cannot be evaluated by the reference either" % type(model).__name__)
    return EvalGraph(batch_size, every_n=FLAGS.every_n, student_only=student_only, feature_size=sum(reader.feature_sizes),
                     vocab_size=reader.num_classes, max_frames=FLAGS.max_num_frames, num_inputs_to_lstm=FLAGS.num_inputs_to_lstm,
                     lstm_cells=FLAGS.lstm_cells, lstm_layers=FLAGS.lstm_layers, num_mixtures=FLAGS.moe_num_mixtures, device=device,
                     precision=FLAGS.precision)


def _batches(reader, device):
    if FLAGS.eval_data_pattern == "synthetic":
        for i, (q, y, n, nh) in enumerate(synthetic_batches(FLAGS.batch_size, sum(reader.feature_sizes), device, FLAGS.synthetic_videos, 1, 4321)):
            yield ["syn%06d" % (i * FLAGS.batch_size + j) for j in range(q.shape[0])], q, y, n, nh
    else:
        for b in get_input_evaluation_tensors(reader, FLAGS.eval_data_pattern, FLAGS.batch_size, FLAGS.num_readers, device):
            yield b


def evaluation_loop(graph, reader, label_loss_fn, summary_writer, evl_metrics, last_global_step_val, device):
    """Run the evaluation loop once (cs/validate.py:192-303).  Returns (global_step_val, epoch_info_dict or None)."""
    ck = latest_checkpoint(FLAGS.train_dir)
    if not ck:
        logging.info("No checkpoint file found.")
        return -1, None
    logging.info("Loading checkpoint for eval: " + ck)
    try:
        sd = torch.load(ck, map_location="cpu")
    except FileNotFoundError as e:
        # the trainer replaced the file between the directory listing and the load (max_to_keep=1) - the only race an
        # atomic writer (train.save_checkpoint: temp file, fsync, os.replace) leaves; look again at the next poll.
        # Anything else (a corrupt or incompatible file) is an error of THIS checkpoint and propagates: polling it
        # forever - or ending a --run_once evaluation with no result and no failure - would hide it.
        logging.info("checkpoint %s disappeared before it could be loaded (%s); will look again.", ck, e)
        return last_global_step_val, None
    graph.restore(sd)
    global_step_val = int(sd.get("global_step", 0))
    if global_step_val == last_global_step_val:
        logging.info("skip this checkpoint global_step_val=%s (same as the previous one).", global_step_val)
        return global_step_val, None
    logging.info("enter eval_once loop global_step_val = %s. ", global_step_val)
    evl_metrics.clear()
    examples_processed, total_example_per_sec = 0, []
    fused_ce = isinstance(label_loss_fn, losses.CrossEntropyLoss)
    fetcher = utils.AsyncFetcher(device)
    last_time = [time.time()]

    def account(handle):
        """The host side of one batch (cs/validate.py:240-282), run while the GPU already works on the next batch."""
        nonlocal examples_processed
        got = fetcher.result(handle)
        predictions_val, labels_val = got["predictions"], got["labels"].astype(np.float32)
        loss_val = float(got["loss"].reshape(-1)[0])
        now = time.time()
        seconds_per_batch, last_time[0] = max(now - last_time[0], 1e-9), now
        example_per_second = labels_val.shape[0] / seconds_per_batch
        total_example_per_sec.append(example_per_second)
        examples_processed += labels_val.shape[0]
        iteration_info_dict = evl_metrics.accumulate(predictions_val, labels_val, loss_val)
        iteration_info_dict["examples_per_second"] = example_per_second
        iterinfo_pre = ""
        if "student_state_loss" in got:                                 # cs/validate.py:268-275
            student_loss_val = float(got["student_state_loss"].reshape(-1)[0])
            iteration_info_dict["student_loss"] = student_loss_val
            iterinfo_pre = "student_loss: %f | " % student_loss_val
        iterinfo = utils.AddGlobalStepSummary(summary_writer, global_step_val, iteration_info_dict, summary_scope="Eval")
        logging.info("examples_processed: %d | %s%s", examples_processed, iterinfo_pre, iterinfo)

    pending = None
    for ids, q, labels, n, n_host in _batches(reader, device):
        out = graph.step(q, labels, n, num_frames_host=n_host)
        loss_t = out["loss"] if fused_ce else label_loss_fn.calculate_loss(out["predictions"], labels)
        fetch = {"predictions": out["predictions"], "labels": labels, "loss": loss_t.reshape(1)}
        if "student_state_loss" in out:
            fetch["student_state_loss"] = out["student_state_loss"].reshape(1)
        handle = fetcher.fetch(fetch)                                    # the fetch: clones + D2H on a copy stream
        if pending is not None:
            account(pending)                                             # batch k's metrics, under batch k+1
        pending = handle
    if pending is not None:
        account(pending)
    if FLAGS.precision == "high":      # the fixed e4m3 scales of the "high" mode assume bounded operands: say so when the last batch broke them
        for tw, key in ((getattr(graph, "teacher", None), "teacher_state"), (getattr(graph, "student", None), "student_state")):
            if tw is not None and hasattr(tw, "fp8_saturation"):
                sat = {k: v for k, v in tw.fp8_saturation(out.get(key) if pending is not None else None).items() if v}
                if sat:
                    logging.warning("%s: operands beyond the e4m3 scales of --precision high (low-order corrections clamp, the 1e-3 contract "
                                    "degrades): %s - see flags.py / EVC_HIGH_MOE_FP8=0", tw.scope, sat)
    logging.info("Done with batched inference. Now calculating global performance metrics.")
    epoch_info_dict = evl_metrics.get()
    epoch_info_dict["epoch_id"] = global_step_val
    logging.info(utils.AddEpochSummary(summary_writer, global_step_val, epoch_info_dict, summary_scope="Eval"))
    if total_example_per_sec:
        logging.info("Average examples processed in one second %0.20f" % (np.sum(np.asarray(total_example_per_sec)) / len(total_example_per_sec)))
    evl_metrics.clear()
    return global_step_val, epoch_info_dict


def evaluate(student_only=False, max_evals=None):
    """cs/validate.py:306-397.  Returns the last epoch_info_dict (None if nothing was evaluated)."""
    start_time = time.time()
    device = "cuda:%d" % FLAGS.gpu
    torch.cuda.set_device(FLAGS.gpu)
    ops.check_device(FLAGS.gpu)
    reader = get_reader()
    model = find_class_by_name(FLAGS.model, [frame_level_models, video_level_models])()
    label_loss_fn = find_class_by_name(FLAGS.label_loss, [losses])()
    if FLAGS.eval_data_pattern == "":
        raise IOError("'eval_data_pattern' was not specified. Nothing to evaluate.")
    graph = build_graph(reader, model, FLAGS.batch_size, device, student_only)
    logging.info("built evaluation graph")
    for tw in (graph.teacher, graph.student):
        if tw is not None:
            logging.info("Names of %s Parameters ::", "Teacher" if tw is graph.teacher else "Student")
            logging.info(list(tw.state_dict().keys()))
    summary_writer = utils.JsonlSummaryWriter(FLAGS.train_dir)
    evl_metrics = eval_util.EvaluationMetrics(reader.num_classes, FLAGS.top_k)
    last_global_step_val, last, evals = -1, None, 0
    while True:
        last_global_step_val, info = evaluation_loop(graph, reader, label_loss_fn, summary_writer, evl_metrics,
                                                     last_global_step_val, device)
        last = info or last
        evals += 1
        if FLAGS.run_once or (max_evals and evals >= max_evals):
            break
        if info is None:
            time.sleep(10)                                              # wait for the trainer to write a new checkpoint
    summary_writer.close()
    print("Total time taken is " + str(time.time() - start_time))
    return last


def main(argv=None, student_only=False):
    FLAGS.parse(sys.argv[1:] if argv is None else argv)
    logging.basicConfig(level=logging.INFO, format="INFO:evc:%(message)s")
    return evaluate(student_only)


if __name__ == "__main__":
    main()
