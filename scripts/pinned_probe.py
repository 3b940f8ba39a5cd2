"""How fast does the CPU read torch's pinned host memory on this ROCm build?  (numpy metrics straight on a pinned D2H
buffer vs on a pageable copy of it.)"""
import time
import numpy as np
import torch

x = torch.rand(256, 4716, device="cuda:0")
pin = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
pin.copy_(x, non_blocking=True)
torch.cuda.synchronize()
for name, arr in (("pinned view", pin.numpy()), ("pageable copy", None)):
    t0 = time.perf_counter()
    if arr is None:
        arr = np.array(pin.numpy())          # one sequential pass over the pinned buffer
    t1 = time.perf_counter()
    for _ in range(3):
        s = np.argpartition(-arr, 20, axis=1)[:, :20].sum() + arr.sum()
    t2 = time.perf_counter()
    print("%-14s copy %.2f ms, 3 x (argpartition + sum) %.2f ms" % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
t0 = time.perf_counter(); y = x.cpu(); t1 = time.perf_counter()
print("x.cpu() (synchronous pageable D2H): %.2f ms" % ((t1 - t0) * 1e3))
