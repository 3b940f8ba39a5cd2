"""Throughput of the host input path and of training fed from TFRecord files (PCIe-inclusive rate).

    python scripts/reader_bench.py [--videos 2048] [--batch 256] [--readers 1,2,4,8] [--train]
Writes a synthetic YT8M-shaped data set under $TMPDIR, then reports
  * parse+batch rate of readers.InputPipeline (host only, no GPU) per reader-thread count,
  * with --train: DistillGraph steps/s when every batch comes from the files through pinned staging + H2D.
"""
import argparse
import os
import sys
import tempfile
import time

import torch  # noqa: F401  (before the timers: the first import takes seconds)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from efficientvideoclassification_youtube8m_amd import readers  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--readers", default="1,2,4,8")
    ap.add_argument("--train", action="store_true")
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="evc_reader_bench_")
    t0 = time.time()
    files = readers.write_synthetic_frame_dataset(d, 8, a.videos // 8, seed=0)
    size = sum(os.path.getsize(f) for f in files)
    print("wrote %d videos, %.1f MB in %.1fs" % (a.videos, size / 1e6, time.time() - t0), flush=True)
    rd = readers.YT8MFrameFeatureReader(feature_names=["rgb", "audio"], feature_sizes=[1024, 128], max_frames=300)
    pat = os.path.join(d, "train*.tfrecord")
    for nr in [int(x) for x in a.readers.split(",")]:
        pipe = readers.get_input_data_tensors(rd, pat, batch_size=a.batch, num_epochs=2, num_readers=nr, seed=0, prefetch=max(3, nr),
                                              reuse_host_buffers=True)
        t0, n = time.time(), 0
        for ids, x, y, nf in pipe:
            n += len(ids)
        dt = time.time() - t0
        print("host pipeline: readers=%d  %.0f videos/s  %.2f GB/s of records  (%.1f M frames/s)"
              % (nr, n / dt, 2 * size / dt / 1e9, n * 300 / dt / 1e6), flush=True)
    if a.train:
        import torch
        from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
        g = DistillGraph(a.batch, every_n=10, device="cuda:0")
        for nr in (4, 8):
            pipe = readers.get_input_data_tensors(rd, pat, batch_size=a.batch, num_epochs=6, num_readers=nr, seed=0, device="cuda:0",
                                                  prefetch=max(3, nr))
            it = iter(pipe)
            for _ in range(3):
                ids, x, y, nf = next(it)
                g.step(x, y, nf)
            torch.cuda.synchronize()
            t0, n = time.time(), 0
            for ids, x, y, nf in it:
                if x.shape[0] != a.batch:
                    continue
                g.step(x, y, nf)
                n += x.shape[0]
            torch.cuda.synchronize()
            dt = time.time() - t0
            print("train from TFRecords (PCIe-inclusive): readers=%d  %.0f videos/s = %.2f M frames/s, %.2f ms/step"
                  % (nr, n / dt, n * 300 / dt / 1e6, dt / (n / a.batch) * 1e3), flush=True)


if __name__ == "__main__":
    main()
