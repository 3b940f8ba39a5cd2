"""cs/train_convert_model.py: turn a teacher+student checkpoint into the starting point of student finetuning.

The reference builds the student-only *training* graph, initialises everything (global_step = 0, fresh Adam
slots), restores the 11 ``model_student/*`` variables by name from the latest checkpoint of --train_dir and
saves the lot as ``<train_dir with 'train' removed>finetune/model.ckpt`` (cs/train_convert_model.py:360-401,
501-511).  Here that is a sub-state-dict extraction: no GPU, no graph.

    python -m efficientvideoclassification_youtube8m_amd.train_convert_model --train_dir ./model_HLSTM_TeaStud_every10_train/
"""
from __future__ import annotations

import logging
import os
import sys

import torch

from .flags import FLAGS
from .train import latest_checkpoint

STUDENT_SCOPE = "model_student/"


def finetune_dir(train_dir):
    """(FLAGS.train_dir[:-1]).replace('train', '') + 'finetune/'  (cs/train_convert_model.py:398)."""
    return train_dir[:-1].replace("train", "") + "finetune/"


def extract_student(state_dict):
    """The 11 student variables (TF names, TF layouts), global_step reset to 0, no optimizer slots
    (train.restore_checkpoint then keeps the freshly initialised zero moments)."""
    out = {k: v for k, v in state_dict.items() if k.startswith(STUDENT_SCOPE) and torch.is_tensor(v)}
    if not out:
        raise ValueError("checkpoint holds no '%s*' variables - was it written by teacher+student training?" % STUDENT_SCOPE)
    out["global_step"] = 0
    return out


def convert(train_dir, out_dir=None):
    ck = latest_checkpoint(train_dir)
    if ck is None:
        raise IOError("No checkpoint file found in " + train_dir)
    sd = extract_student(torch.load(ck, map_location="cpu"))
    logging.info("saver_student loaded successfully !")
    logging.info([k for k in sd if k != "global_step"])
    out_dir = out_dir or finetune_dir(train_dir)
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "model.ckpt.pt")                  # the reference's un-numbered 'model.ckpt'
    torch.save(sd, path)
    torch.load(path, map_location="cpu")                           # saver.restore(sess, checkpoint_path): read it back
    logging.info("New student-model saved and restored successfully for finetuning!")
    return path


def main(argv=None):
    FLAGS.parse(sys.argv[1:] if argv is None else argv)
    logging.basicConfig(level=logging.INFO, format="INFO:evc:%(message)s")
    return convert(FLAGS.train_dir)


if __name__ == "__main__":
    main()
