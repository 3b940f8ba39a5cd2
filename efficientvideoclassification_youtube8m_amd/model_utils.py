"""Frame sampling / pooling helpers (cs/model_utils.py) over the HIP kernels."""
from __future__ import annotations

import torch

from . import ops


def SampleRandomFrames(model_input, num_frames, num_samples, uniform=None, normalize=False):
    """cs/model_utils.py:39-58.  frame_index = int32(U[0,1) * float32(num_frames));
    gathers [B, num_samples, F].  ``uniform`` ([B,num_samples] f32) may be supplied
    to pin the draw; otherwise torch's device RNG is used."""
    B, T, F = model_input.shape
    if uniform is None:
        uniform = torch.rand((B, num_samples), dtype=torch.float32, device=model_input.device)
    nf = num_frames.reshape(-1).to(torch.int32)
    out = torch.empty((B * num_samples, F), dtype=torch.float32, device=model_input.device)
    ops.sample_frames_gather(model_input, uniform, nf, out, None, normalize=normalize)
    return out.view(B, num_samples, F)


def SampleRandomSequence(model_input, num_frames, num_samples, uniform=None, normalize=False):
    """cs/model_utils.py:11-36 (reached with --sample_random_frames False): num_samples consecutive frames from a random start,
    start = int32(U[0,1) * float32(max(n - num_samples, 0) + 1)), indices clipped at n - 1.  ``uniform`` [B] pins the draw."""
    B, T, F = model_input.shape
    if uniform is None:
        uniform = torch.rand((B,), dtype=torch.float32, device=model_input.device)
    nf = num_frames.reshape(-1).to(torch.int32)
    out = torch.empty((B * num_samples, F), dtype=torch.float32, device=model_input.device)
    ops.sample_sequence_gather(model_input, uniform.reshape(-1).contiguous(), nf, num_samples, out, None, normalize=normalize)
    return out.view(B, num_samples, F)


def FramePooling(frames, method, **unused_params):
    """cs/model_utils.py:60-83: 'max' | 'average' | 'none'."""
    B, S, C = frames.shape
    if method == "max":
        pooled = torch.empty((B, C), dtype=torch.float32, device=frames.device)
        am = torch.empty((B, C), dtype=torch.int32, device=frames.device)
        ops.framepool_max_fwd(frames.contiguous(), B, S, C, pooled, None, am)
        return pooled
    elif method == "average":
        n = torch.full((B,), S, dtype=torch.int32, device=frames.device)
        avg = torch.empty((B, C), dtype=torch.float32, device=frames.device)
        if C % 4 == 0 and C <= 1280:
            ops.meanpool(frames.contiguous(), n, avg, None, normalize=False)
            return avg
        raise ValueError("average pooling supports feature sizes that are multiples of 4 and <= 1280")
    elif method == "none":
        return frames.reshape(-1, C)
    else:
        raise ValueError("Unrecognized pooling method: %s" % method)
