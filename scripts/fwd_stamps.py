#!/usr/bin/env python3
"""Where does the time of one L1 forward step go?  DIAGNOSTIC build of the library (-DEVC_STAMPS: every workgroup of
lstm_fwd_step_kernel writes s_memrealtime marks - entry, ring filled (first barrier), loop done, tail issued, stores
acknowledged) over one teacher L1 layer at the bench's dims; prints per-launch medians over the workgroups.

    EVC_OUT=$PWD/efficientvideoclassification_youtube8m_amd/libevc_stamps.so bash efficientvideoclassification_youtube8m_amd/csrc/build.sh -DEVC_STAMPS
    EVC_LIB=efficientvideoclassification_youtube8m_amd/libevc_stamps.so python scripts/fwd_stamps.py
"""
import ctypes, os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientvideoclassification_youtube8m_amd import _lib, ops      # noqa: E402

DEV = "cuda:0"
T, M, Kin, H = 15, int(os.environ.get("ROWS", 3700)), int(os.environ.get("KIN", 1152)), 1024
rng = np.random.default_rng(0)
x = torch.tensor(rng.standard_normal((T, M, Kin)).astype(np.float32) * 0.05, device=DEV).to(torch.bfloat16)
w = torch.tensor(rng.standard_normal((4 * H, Kin + H)).astype(np.float32) * 0.03, device=DEV).to(torch.bfloat16)
b = torch.zeros(4 * H, device=DEV)
lens = torch.full((M,), T, dtype=torch.int32, device=DEV)
hbuf = torch.empty((T + 1, M, H), dtype=torch.bfloat16, device=DEV)
c_state = torch.empty((M, H), device=DEV)
h_state = torch.empty((M, H), device=DEV)
gates = torch.empty((T, M, H, 4), dtype=torch.bfloat16, device=DEV)
c_all = torch.empty((T + 1, M, H), dtype=torch.bfloat16, device=DEV)
fn = ctypes.CDLL(_lib.LIB_PATH).evc_debug_read_stamps       # (the diagnostic entry point is not part of include/evc.h)
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
for it in range(5):
    ops.lstm_layer_fwd(x, w, b, lens, T, M, Kin, H, hbuf, c_state, h_state, H, gates=gates, c_all=c_all)
torch.cuda.synchronize()
buf = np.zeros((8, 512, 8), dtype=np.uint64)
assert fn(buf.ctypes.data) == 0
tick = 0.01     # us per s_memrealtime tick (100 MHz)
print("launch (t & 7) | workgroups | entry spread | entry->ring filled | loop | tail issue | store ack | kernel span | gap to next launch's first entry")
spans = {}
for slot in range(8):
    s = buf[slot]
    live = s[:, 0] > 0
    n = int(live.sum())
    if not n:
        continue
    s = s[live].astype(np.int64)
    e0 = s[:, 0].min()
    spans[slot] = (e0, s[:, 5].max())
    med = lambda a: float(np.median(a)) * tick
    print("%d | %d | %.2f us | %.2f | %.2f | %.2f | %.2f | %.2f" % (slot, n, (s[:, 0].max() - e0) * tick, med(s[:, 1] - s[:, 0]), med(s[:, 2] - s[:, 1]),
          med(s[:, 3] - s[:, 2]), med(s[:, 4] - s[:, 3]), (s[:, 5].max() - e0) * tick))
order = sorted(spans, key=lambda k: spans[k][0])
for a, c in zip(order, order[1:]):
    print("gap %d -> %d: %.2f us (first entry of the next launch - last exit of this one)" % (a, c, (spans[c][0] - spans[a][1]) * tick))
