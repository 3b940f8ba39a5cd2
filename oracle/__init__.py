"""CPU oracle for the frame-level aggregation + distillation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker / the timed
CPU baseline - never as a fallback for the HIP path.

Parity status: **parity unpinned at the TensorFlow boundary.**  (`lowprec.py`: the operand formats of the "high" precision forward - IEEE f16, OCP e4m3 - pinned on torch.float16 / torch.float8_e4m3fn, tests/test_oracle_lowprec.py.)  The reference
(`/root/reference/code_student_uniform/*.py`) is Python-2.7 + TensorFlow-1.3
graph wiring; its arithmetic lives in TensorFlow, which is not vendored, not
installed and not installable here, and the reference ships no tests or golden
vectors for the model math.  The model-math oracle (`model_math.py`) therefore
restates the published TF-1.x op semantics at the
reference's own call sites and is anchored by the known answers the reference
does publish (README.md:116 initial losses, README.md:93-105 shapes/variable
lists).  The *metric* oracle (`metrics.py`) IS pinned: it is checked against
golden vectors minted by importing the reference's own numpy metric code
(`eval_util.py`, `average_precision_calculator.py`,
`mean_average_precision_calculator.py`) in the build container
(`tests/golden/make_metric_golden.py`).
"""
