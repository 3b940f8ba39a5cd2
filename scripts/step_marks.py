"""Where does the time of one training step go?  Timing events on the teacher / student streams."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import synthetic_inputs  # noqa: E402
from efficientvideoclassification_youtube8m_amd.distill import DistillGraph  # noqa: E402

dev = "cuda:0"
g = DistillGraph(256, every_n=10, device=dev)
batches = [synthetic_inputs(256, 300, 1152, 4716, 100 + i, dev, False) for i in range(4)]
nhost = [b[1].cpu().numpy() for b in batches]


def run(label, K=8):
    for i in range(3):
        x, n, y = batches[i % 4]
        g.step(x, y, n, num_frames_host=nhost[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        x, n, y = batches[i % 4]
        if i == K - 2:
            g.debug_marks = []
        g.step(x, y, n, num_frames_host=nhost[i % 4])
        if i == K - 2:
            marks, g.debug_marks = g.debug_marks, None
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    base = marks[0][1]
    print("%s: %.2f ms/step | %s" % (label, ms, "  ".join("%s@%.2f" % (n, base.elapsed_time(e)) for n, e in marks)), flush=True)


run("default")
aux_t, aux_s = g._aux_t, g._aux_s
g._aux_t = g._aux_s = None
g_over = g.overlap_towers


class NoAux:
    pass


# no aux streams: weight-gradient GEMMs and Adam stay on the tower's own stream
import efficientvideoclassification_youtube8m_amd.distill as D  # noqa: E402
orig = D.DistillGraph._step


def patched(self, *a, **k):
    return orig(self, *a, **k)


g._aux_t, g._aux_s = None, None
try:
    run("no aux streams")
except Exception as e:  # noqa: BLE001
    print("no-aux variant failed:", e)
g._aux_t, g._aux_s = aux_t, aux_s
g.overlap_towers = False
run("single stream")
g.overlap_towers = True
run("default again")
