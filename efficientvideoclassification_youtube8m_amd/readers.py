"""Readers for the YouTube-8M TFRecord data sets (cs/readers.py) and the input
pipeline of cs/train.py:129-176 / cs/validate.py's evaluation input.

TensorFlow is not a dependency: libevc_io.so (csrc/evc_io.cpp, C ABI in
include/evc_io.h) walks the TFRecord framing and parses the SequenceExample /
Example protos natively; reader threads call it with the GIL released.

MI355X-first differences from the reference pipeline:
  * frame features stay **uint8** end to end ([B, max_frames, 1152] = 88 MB per
    256-video batch instead of 354 MB float32) and are dequantised
    (cs/utils.py:22-25), zero-padded (cs/readers.py:170-173) and l2-normalised
    by the input kernel ``evc_l2norm_chunk_fwd`` on the GPU;
  * batches are assembled directly into pinned host buffers and copied to the
    device asynchronously while the previous step runs;
  * tf.train.shuffle_batch_join(capacity=50*B, min_after_dequeue=B) is kept as
    a windowed shuffle over record *references* (file, offset), so the pool
    costs a few MB, not 50 batches of pixels.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import random
import struct
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
IO_LIB_PATH = os.environ.get("EVC_IO_LIB", os.path.join(_HERE, "libevc_io.so"))

_vp, _i32, _i64 = C.c_void_p, C.c_int, C.c_int64
IO_SIGNATURES = {
    "evc_crc32c": ([_vp, _i64], C.c_uint32),
    "evc_masked_crc32c": ([_vp, _i64], C.c_uint32),
    "evc_tfrecord_scan": ([C.c_char_p, _vp, _vp, _i64, _i32], _i64),
    "evc_parse_yt8m_frame_example": ([_vp, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32], _i32),
    "evc_read_yt8m_frame_records": ([C.c_char_p, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32], _i32),
    "evc_parse_yt8m_video_example": ([_vp, _i64, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32], _i32),
    "evc_read_yt8m_video_records": ([C.c_char_p, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32], _i32),
    "evc_io_last_error": ([], C.c_char_p),
}
IO_EXPORTS = tuple(IO_SIGNATURES)
ID_CAP = 32
_io = None


class EvcIoError(IOError):
    pass


def load_io():
    """Load libevc_io.so; raise (never fall back to a Python parser) if it is missing."""
    global _io
    if _io is None:
        if not os.path.exists(IO_LIB_PATH):
            raise EvcIoError("libevc_io.so not found at %s - build it with csrc/build.sh" % IO_LIB_PATH)
        lib = C.CDLL(IO_LIB_PATH)
        for name, (args, res) in IO_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, res
        _io = lib
    return _io


def _io_check(rc, what):
    if rc < 0:
        raise EvcIoError("%s failed (%d): %s" % (what, rc, load_io().evc_io_last_error().decode()))
    return rc


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def scan_tfrecord(path, verify_crc=False):
    """(payload offsets, lengths) of every record of one TFRecord file."""
    lib = load_io()
    n = _io_check(lib.evc_tfrecord_scan(path.encode(), None, None, 0, int(verify_crc)), "evc_tfrecord_scan(%s)" % path)
    off, ln = np.zeros(n, np.int64), np.zeros(n, np.int64)
    if n:
        _io_check(lib.evc_tfrecord_scan(path.encode(), _ptr(off), _ptr(ln), n, 0), "evc_tfrecord_scan(%s)" % path)
    return off, ln


class _Names:
    """char*[] + int32[] views of the feature names / sizes for the C ABI."""

    def __init__(self, names, sizes):
        self._keep = [n.encode() for n in names]
        self.names = (C.c_char_p * len(names))(*self._keep)
        self.sizes = np.asarray(sizes, np.int32)
        self.n = len(names)
        self.row = int(sum(sizes))


class BaseReader(object):
    """Inherit from this class when implementing new readers (cs/readers.py:45-50)."""

    def prepare_reader(self, unused_filename_queue):
        raise NotImplementedError()


class YT8MFrameFeatureReader(BaseReader):
    """Reads TFRecords of SequenceExamples (cs/readers.py:115-246): sparse int64
    'labels' context feature + one byte-quantised vector per frame for each of
    ``feature_names``.  The video matrix is returned **quantised** (uint8); the
    GPU input kernel applies Dequantize(max=2, min=-2) and the zero padding."""

    def __init__(self, num_classes=4716, feature_sizes=[1024], feature_names=["inc3"], max_frames=300):
        assert len(feature_names) == len(feature_sizes), \
            "length of feature_names (={}) != length of feature_sizes (={})".format(len(feature_names), len(feature_sizes))
        self.num_classes = num_classes
        self.feature_sizes = list(feature_sizes)
        self.feature_names = list(feature_names)
        self.max_frames = max_frames
        self._names = _Names(self.feature_names, self.feature_sizes)

    row_dtype = np.uint8

    @property
    def row_shape(self):
        return (self.max_frames, self._names.row)

    def read_into(self, path, offsets, lengths, frames, num_frames, labels, ids):
        """Parse ``len(offsets)`` records of ``path`` into the given (contiguous) batch slices."""
        nm = self._names
        _io_check(load_io().evc_read_yt8m_frame_records(
            path.encode(), _ptr(offsets), _ptr(lengths), len(offsets), nm.names, _ptr(nm.sizes), nm.n, self.max_frames,
            self.num_classes, _ptr(frames), _ptr(num_frames), _ptr(labels), _ptr(ids), ID_CAP), "reading %s" % path)

    def prepare_reader(self, filename_queue, max_quantized_value=2, min_quantized_value=-2):
        """Generator over single-example batches (batch of 1, as cs/readers.py:236-246
        returns): (video_ids, video_matrix uint8 [1,max_frames,F], labels bool [1,C], num_frames [1])."""
        assert (max_quantized_value, min_quantized_value) == (2, -2), "the GPU input kernel dequantises with (2, -2)"
        for path in ([filename_queue] if isinstance(filename_queue, str) else filename_queue):
            off, ln = scan_tfrecord(path)
            for i in range(len(off)):
                fr = np.empty((1,) + self.row_shape, np.uint8)
                nf, lb, ids = np.zeros(1, np.int32), np.zeros((1, self.num_classes), np.uint8), np.zeros((1, ID_CAP), np.uint8)
                self.read_into(path, off[i:i + 1], ln[i:i + 1], fr, nf, lb, ids)
                yield [_id_str(ids[0])], fr, lb.astype(bool), nf


class YT8MAggregatedFeatureReader(BaseReader):
    """Reads TFRecords of pre-aggregated Examples (cs/readers.py:53-113): float32
    features (already dequantised averages), num_frames = 1 for every video."""

    def __init__(self, num_classes=4716, feature_sizes=[1024], feature_names=["mean_inc3"]):
        assert len(feature_names) == len(feature_sizes), \
            "length of feature_names (={}) != length of feature_sizes (={})".format(len(feature_names), len(feature_sizes))
        self.num_classes = num_classes
        self.feature_sizes = list(feature_sizes)
        self.feature_names = list(feature_names)
        self._names = _Names(self.feature_names, self.feature_sizes)

    row_dtype = np.float32

    @property
    def row_shape(self):
        return (self._names.row,)

    def read_into(self, path, offsets, lengths, feats, num_frames, labels, ids):
        nm = self._names
        _io_check(load_io().evc_read_yt8m_video_records(
            path.encode(), _ptr(offsets), _ptr(lengths), len(offsets), nm.names, _ptr(nm.sizes), nm.n, self.num_classes,
            _ptr(feats), _ptr(labels), _ptr(ids), ID_CAP), "reading %s" % path)
        num_frames[:] = 1                                              # tf.ones([batch]) cs/readers.py:113

    def prepare_reader(self, filename_queue, batch_size=1024):
        for path in ([filename_queue] if isinstance(filename_queue, str) else filename_queue):
            off, ln = scan_tfrecord(path)
            for s in range(0, len(off), batch_size):                   # reader.read_up_to(queue, batch_size)
                b = min(batch_size, len(off) - s)
                ft = np.empty((b,) + self.row_shape, np.float32)
                nf, lb, ids = np.zeros(b, np.int32), np.zeros((b, self.num_classes), np.uint8), np.zeros((b, ID_CAP), np.uint8)
                self.read_into(path, off[s:s + b], ln[s:s + b], ft, nf, lb, ids)
                yield [_id_str(r) for r in ids], ft, lb.astype(bool), nf.astype(np.float32)


def _id_str(row):
    return bytes(row).split(b"\0", 1)[0].decode("latin-1")


# ---------------------------------------------------------------------------------------------------
# Input pipeline
# ---------------------------------------------------------------------------------------------------
def _record_refs(files, num_epochs, shuffle, num_readers, rng):
    """Stream of (file index, record index) in the order the reference's reader threads would enqueue
    them: the file list is reshuffled every epoch (tf.train.string_input_producer(shuffle=True),
    cs/train.py:161-162) and ``num_readers`` files are consumed round-robin."""
    epoch = 0
    while num_epochs is None or epoch < num_epochs:
        order = list(range(len(files)))
        if shuffle:
            rng.shuffle(order)
        for g in range(0, len(order), max(1, num_readers)):
            group = order[g:g + max(1, num_readers)]
            cursors = [0] * len(group)
            live = True
            while live:
                live = False
                for j, fi in enumerate(group):
                    if cursors[j] < files[fi][1]:
                        yield fi, cursors[j]
                        cursors[j] += 1
                        live = True
        epoch += 1


def _shuffle_window(refs, capacity, rng):
    """tf.train.shuffle_batch_join's RandomShuffleQueue: fill to ``capacity`` (= 50*batch_size,
    cs/train.py:171), then every dequeue removes a uniformly random element."""
    pool = []
    for r in refs:
        pool.append(r)
        if len(pool) >= capacity:
            i = rng.randrange(len(pool))
            pool[i], pool[-1] = pool[-1], pool[i]
            yield pool.pop()
    while pool:
        i = rng.randrange(len(pool))
        pool[i], pool[-1] = pool[-1], pool[i]
        yield pool.pop()


class InputPipeline(object):
    """Iterator over batches ``(video_ids, features, labels uint8 [b, C], num_frames int32 [b])``.

    ``features`` is uint8 [b, max_frames, F] for the frame reader and float32 [b, F] for the aggregated
    one.  With ``device`` set the tensors are on that device (pinned staging + async copy); otherwise
    they are CPU tensors.  The last batch of the last epoch may be smaller
    (allow_smaller_final_batch=True, cs/train.py:175)."""

    def __init__(self, reader, data_pattern, batch_size, num_epochs=None, num_readers=1, shuffle=True, seed=None,
                 device=None, rank=0, world_size=1, prefetch=3, what="training", reuse_host_buffers=False,
                 with_host_counts=False, drop_remainder=None):
        files = sorted(glob.glob(data_pattern)) if isinstance(data_pattern, str) else list(data_pattern)
        if not files:
            raise IOError("Unable to find " + what + " files. data_pattern='" + str(data_pattern) + "'.")   # cs/train.py:155-157
        # Data parallelism: each rank owns every world_size-th file (falls back to records when there are
        # fewer files than ranks), so ranks never read the same video in an epoch.
        self.shard_records = world_size > 1 and len(files) < world_size
        if world_size > 1 and not self.shard_records:
            files = files[rank::world_size]
        self.reader, self.batch_size, self.device = reader, batch_size, device
        self.rank, self.world_size = rank, world_size
        self.index = []
        for f in files:
            off, ln = scan_tfrecord(f)
            if self.shard_records:
                off, ln = off[rank::world_size], ln[rank::world_size]
            self.index.append((f, off, ln))
        self.num_records = sum(len(o) for _, o, _ in self.index)
        # Data parallelism never sees a ragged batch: ranks own different records, so a smaller final batch would
        # differ in size between ranks (unequal collective payloads, per-rank loss scales that no longer add up to the
        # global-batch mean).  The remainder is dropped; ranks then agree on MIN(num_batches) (train.py).
        self.drop_remainder = (world_size > 1) if drop_remainder is None else bool(drop_remainder)
        total = None if num_epochs is None else self.num_records * num_epochs
        self.num_batches = None if total is None else (total // batch_size if self.drop_remainder else -(-total // batch_size))
        rng = random.Random(seed)
        refs = _record_refs([(f, len(o)) for f, o, _ in self.index], num_epochs, shuffle, num_readers, rng)
        self._refs = _shuffle_window(refs, 50 * batch_size, rng) if shuffle else refs
        self._pool = ThreadPoolExecutor(max(1, num_readers))
        self._prefetch = max(1, prefetch)
        self._pending = []
        self._ring, self._ring_pos = None, 0
        # CPU consumers get fresh tensors by default; with reuse_host_buffers a batch is only valid until
        # prefetch + 2 further batches have been drawn (saves the page faults of a new 88 MB buffer per batch)
        self._reuse = reuse_host_buffers
        self._copy_stream, self._staged, self._started = None, None, False
        # with_host_counts: batches carry a 5th element, num_frames as a numpy array (the training graph derives
        # its launch geometry from it without reading the device copy back)
        self._with_host = with_host_counts

    # -- host buffers ---------------------------------------------------------------------------------
    def _buffers(self):
        import torch
        pin = self.device is not None and torch.cuda.is_available()
        B, r = self.batch_size, self.reader
        tdt = torch.uint8 if r.row_dtype is np.uint8 else torch.float32
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=pin)
        return {"x": mk((B,) + r.row_shape, tdt), "n": mk((B,), torch.int32), "y": mk((B, r.num_classes), torch.uint8),
                "ids": mk((B, ID_CAP), torch.uint8), "event": None}

    def _next_buffers(self):
        if self.device is None and not self._reuse:
            return self._buffers()                                   # fresh CPU tensors, handed to the caller
        if self._ring is None:
            self._ring = [self._buffers() for _ in range(self._prefetch + 2)]
        buf = self._ring[self._ring_pos]
        self._ring_pos = (self._ring_pos + 1) % len(self._ring)
        if buf["event"] is not None:
            buf["event"].synchronize()                               # its previous H2D copy has finished
        return buf

    def _fill(self, batch_refs, buf):
        batch_refs.sort()                                            # same-file records contiguous; order inside a batch is free
        x, n, y, ids = (buf[k].numpy() for k in ("x", "n", "y", "ids"))
        s = 0
        while s < len(batch_refs):
            fi = batch_refs[s][0]
            e = s
            while e < len(batch_refs) and batch_refs[e][0] == fi:
                e += 1
            path, off, ln = self.index[fi]
            rec = np.fromiter((r[1] for r in batch_refs[s:e]), np.int64, e - s)
            self.reader.read_into(path, np.ascontiguousarray(off[rec]), np.ascontiguousarray(ln[rec]), x[s:e], n[s:e], y[s:e], ids[s:e])
            s = e
        return len(batch_refs), buf

    def _submit(self):
        refs = []
        for r in self._refs:
            refs.append(r)
            if len(refs) == self.batch_size:
                break
        if not refs or (self.drop_remainder and len(refs) < self.batch_size):
            return False
        self._pending.append(self._pool.submit(self._fill, refs, self._next_buffers()))
        return True

    def __iter__(self):
        return self

    def _take(self):
        """Next filled host batch (blocks on the reader threads), or None at the end of the data."""
        while len(self._pending) < self._prefetch and self._submit():
            pass
        if not self._pending:
            return None
        return self._pending.pop(0).result()

    def _stage(self):
        """Host batch -> device on the copy stream (never on the compute stream: the 88 MB H2D transfer of
        batch k+1 runs under the kernels of batch k)."""
        import torch
        got = self._take()
        if got is None:
            return None
        b, buf = got
        ids = [_id_str(r) for r in buf["ids"].numpy()[:b]]
        n_host = buf["n"].numpy()[:b].copy()
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self._copy_stream):
            out = tuple(buf[k][:b].to(self.device, non_blocking=True) for k in ("x", "y", "n"))
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        buf["event"] = ev                                            # the pinned buffers are reusable once it has fired
        return ids, out + ((n_host,) if self._with_host else ()), ev

    def __next__(self):
        if self.device is None:
            got = self._take()
            if got is None:
                self._pool.shutdown(wait=False)
                raise StopIteration
            b, buf = got
            extra = (buf["n"].numpy()[:b].copy(),) if self._with_host else ()
            return ([_id_str(r) for r in buf["ids"].numpy()[:b]], buf["x"][:b], buf["y"][:b], buf["n"][:b]) + extra
        import torch
        if not self._started:
            self._started, self._staged = True, self._stage()
        cur = self._staged
        if cur is None:
            self._pool.shutdown(wait=False)
            raise StopIteration
        self._staged = self._stage()                                 # start the next batch's copy before handing this one out
        ids, out, ev = cur
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ev)
        for t in out[:3]:
            t.record_stream(main)                                    # allocated on the copy stream, consumed on the caller's
        return (ids,) + out


def get_input_data_tensors(reader, data_pattern, batch_size=1000, num_epochs=None, num_readers=1, **kw):
    """Creates the section of the graph which reads the training data (cs/train.py:129-176):
    shuffled files, shuffled window of 50*batch_size examples, smaller final batch allowed."""
    return InputPipeline(reader, data_pattern, batch_size, num_epochs, num_readers, shuffle=True, what="training", **kw)


def get_input_evaluation_tensors(reader, data_pattern, batch_size=1024, num_readers=1, **kw):
    """Evaluation input (cs/validate.py get_input_evaluation_tensors): every file once, in order,
    no shuffling (string_input_producer(shuffle=False, num_epochs=1) + batch_join)."""
    return InputPipeline(reader, data_pattern, batch_size, 1, num_readers, shuffle=False, what="evaluation", **kw)


# ---------------------------------------------------------------------------------------------------
# Writer (fixtures, synthetic data sets, format conversion) - protobuf wire format by hand
# ---------------------------------------------------------------------------------------------------
def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _bytes_feature(values):
    return _ld(1, b"".join(_ld(1, bytes(v)) for v in values))


def _int64_feature(values):
    return _ld(3, _ld(1, b"".join(_varint(int(v)) for v in values)) if len(values) else b"")


def _float_feature(values):
    return _ld(2, _ld(1, np.asarray(values, "<f4").tobytes()))


def _map_entry(key, value):
    return _ld(1, _ld(1, key.encode()) + _ld(2, value))


def encode_frame_example(video_id, labels, features):
    """Serialise one frame-level tf.train.SequenceExample.  ``features``: name -> uint8 [n_frames, size]."""
    ctx = _map_entry("id", _bytes_feature([video_id.encode()])) + _map_entry("labels", _int64_feature(labels))
    fl = b""
    for name, mat in features.items():
        mat = np.ascontiguousarray(mat, np.uint8)
        fl += _map_entry(name, b"".join(_ld(1, _bytes_feature([row.tobytes()])) for row in mat))
    return _ld(1, ctx) + _ld(2, fl)


def encode_video_example(video_id, labels, features):
    """Serialise one video-level tf.train.Example.  ``features``: name -> float32 [size]."""
    fs = _map_entry("id", _bytes_feature([video_id.encode()])) + _map_entry("labels", _int64_feature(labels))
    for name, vec in features.items():
        fs += _map_entry(name, _float_feature(vec))
    return _ld(1, fs)


def write_tfrecord(path, payloads):
    """TFRecord framing: u64 length | masked crc32c(length) | data | masked crc32c(data)."""
    lib = load_io()
    with open(path, "wb") as f:
        for p in payloads:
            hdr = struct.pack("<Q", len(p))
            f.write(hdr)
            f.write(struct.pack("<I", lib.evc_masked_crc32c(C.cast(C.c_char_p(hdr), C.c_void_p), 8)))
            f.write(p)
            f.write(struct.pack("<I", lib.evc_masked_crc32c(C.cast(C.c_char_p(p), C.c_void_p), len(p))))


def write_synthetic_frame_dataset(directory, num_files, videos_per_file, feature_names=("rgb", "audio"),
                                  feature_sizes=(1024, 128), num_classes=4716, min_frames=120, max_frames=300, seed=0,
                                  prefix="train", first_file_index=0):
    """Writes a YouTube-8M-shaped data set of random videos (there is no network for the real one).
    Returns the list of files; the videos are reproducible from ``seed``."""
    os.makedirs(directory, exist_ok=True)
    rng = np.random.default_rng(seed)
    files = []
    for fi in range(first_file_index, first_file_index + num_files):      # (file index is part of the video ids)
        payloads = []
        for vi in range(videos_per_file):
            n = int(rng.integers(min_frames, max_frames + 1))
            feats = {nm: rng.integers(0, 256, (n, sz), dtype=np.uint8) for nm, sz in zip(feature_names, feature_sizes)}
            labels = sorted(set(int(v) for v in rng.integers(0, num_classes, int(rng.integers(1, 6)))))
            payloads.append(encode_frame_example("v%02d%04d" % (fi, vi), labels, feats))
        path = os.path.join(directory, "%s%04d.tfrecord" % (prefix, fi))
        write_tfrecord(path, payloads)
        files.append(path)
    return files
