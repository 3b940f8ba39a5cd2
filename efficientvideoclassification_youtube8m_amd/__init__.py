"""MI355X-native teacher/student frame-level aggregation + distillation hot path.

Python host (PyTorch-ROCm for device memory, streams and RCCL) over the C-ABI
library ``libevc_hip.so`` of hand-written gfx950 HIP kernels (``csrc/``,
``include/evc.h``).  The module names mirror the reference's
(``frame_level_models``, ``video_level_models``, ``losses``, ``model_utils``,
``eval_util``, ``train``) so its ``--model HierarchicalLstmModel ...`` flag
surface drops in.  There is NO CPU fallback: every compute op raises if the
HIP library is missing.
"""
__version__ = "0.1.0"
