"""cs/utils.py: Dequantize, the summary/log-line helpers and GetListOfFeatureNamesAndSizes.

TensorBoard event files are replaced by a JSON-lines log (``summary_writer`` is
any object with ``add_scalar(tag, value, step)``; ``JsonlSummaryWriter`` below
writes ``events.jsonl`` into the train dir).  The returned log strings are the
reference's, character for character."""
from __future__ import annotations

import json
import os
import time

import numpy

from .flags import GetListOfFeatureNamesAndSizes  # noqa: F401  (cs/utils.py:128-149)


def Dequantize(feat_vector, max_quantized_value=2, min_quantized_value=-2):
    """Dequantize the feature from the byte format to the float format (cs/utils.py:10-25).
    Works on numpy arrays and torch tensors; the training path does this inside evc_l2norm_chunk_fwd."""
    assert max_quantized_value > min_quantized_value
    quantized_range = max_quantized_value - min_quantized_value
    scalar = quantized_range / 255.0
    bias = (quantized_range / 512.0) + min_quantized_value
    return feat_vector * scalar + bias


class JsonlSummaryWriter(object):
    """Stand-in for tf.summary.FileWriter: one JSON object per scalar."""

    def __init__(self, logdir, filename="events.jsonl"):
        os.makedirs(logdir, exist_ok=True)
        self._f = open(os.path.join(logdir, filename), "a")

    def add_scalar(self, tag, value, step):
        self._f.write(json.dumps({"wall_time": time.time(), "step": int(step), "tag": str(tag), "value": float(value)}) + "\n")

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def _add(summary_writer, tag, value, step):
    if summary_writer is not None:
        summary_writer.add_scalar(tag, value, int(step))


def AddGlobalStepSummary(summary_writer, global_step_val, global_step_info_dict, summary_scope="Eval"):
    """cs/utils.py:35-77."""
    this_hit_at_one = global_step_info_dict["hit_at_one"]
    this_perr = global_step_info_dict["perr"]
    this_loss = global_step_info_dict["loss"]
    examples_per_second = global_step_info_dict.get("examples_per_second", -1)
    _add(summary_writer, "GlobalStep/" + summary_scope + "_Hit@1", this_hit_at_one, global_step_val)
    _add(summary_writer, "GlobalStep/" + summary_scope + "_Perr", this_perr, global_step_val)
    _add(summary_writer, "GlobalStep/" + summary_scope + "_Loss", this_loss, global_step_val)
    if examples_per_second != -1:
        _add(summary_writer, "GlobalStep/" + summary_scope + "_Example_Second", examples_per_second, global_step_val)
    if summary_writer is not None:
        summary_writer.flush()
    return ("global_step {0} | Batch Hit@1: {1:.3f} | Batch PERR: {2:.3f} | Batch Loss: {3:.3f} "
            "| Examples_per_sec: {4:.3f}").format(global_step_val, this_hit_at_one, this_perr, this_loss, examples_per_second)


def AddEpochSummary(summary_writer, global_step_val, epoch_info_dict, summary_scope="Eval"):
    """cs/utils.py:80-126."""
    epoch_id = epoch_info_dict["epoch_id"]
    avg_hit_at_one = epoch_info_dict["avg_hit_at_one"]
    avg_perr = epoch_info_dict["avg_perr"]
    avg_loss = epoch_info_dict["avg_loss"]
    aps = epoch_info_dict["aps"]
    gap = epoch_info_dict["gap"]
    mean_ap = numpy.mean(aps)
    _add(summary_writer, "Epoch/" + summary_scope + "_Avg_Hit@1", avg_hit_at_one, global_step_val)
    _add(summary_writer, "Epoch/" + summary_scope + "_Avg_Perr", avg_perr, global_step_val)
    _add(summary_writer, "Epoch/" + summary_scope + "_Avg_Loss", avg_loss, global_step_val)
    _add(summary_writer, "Epoch/" + summary_scope + "_MAP", mean_ap, global_step_val)
    _add(summary_writer, "Epoch/" + summary_scope + "_GAP", gap, global_step_val)
    if summary_writer is not None:
        summary_writer.flush()
    return ("epoch/eval number {0} | Avg_Hit@1: {1:.3f} | Avg_PERR: {2:.3f} "
            "| MAP: {3:.3f} | GAP: {4:.3f} | Avg_Loss: {5:3f}").format(epoch_id, avg_hit_at_one, avg_perr, mean_ap, gap, avg_loss)


class AsyncFetcher(object):
    """sess.run-style fetch of device tensors WITHOUT stopping the GPU: fetch() clones the tensors in stream order (the
    graph's output buffers are overwritten by its next step) and copies the clones to pinned host memory on a copy
    stream; result() waits for that copy only.  A loop that calls fetch() for batch k, enqueues batch k+1 and then takes
    result() of batch k computes its host metrics while the GPU runs the next batch."""

    def __init__(self, device, slots=2):
        import torch
        self._torch, self.device = torch, torch.device(device)
        self._stream = torch.cuda.Stream(device=self.device)
        self._bufs = [dict() for _ in range(slots)]
        self._n = 0

    def fetch(self, tensors):
        """tensors: dict name -> device tensor.  Returns a handle for result()."""
        torch = self._torch
        clones = {k: t.detach().clone() for k, t in tensors.items()}
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        slot = self._bufs[self._n % len(self._bufs)]
        self._n += 1
        with torch.cuda.stream(self._stream):
            for k, t in clones.items():
                if k not in slot or slot[k].shape != t.shape or slot[k].dtype != t.dtype:
                    slot[k] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                slot[k].copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        for t in clones.values():
            t.record_stream(self._stream)
        return (ev, slot, tuple(clones.keys()))

    def result(self, handle):
        """dict name -> numpy array (views of the pinned staging buffers: valid until `slots` further fetch() calls)."""
        ev, slot, keys = handle
        ev.synchronize()
        return {k: slot[k].numpy() for k in keys}
