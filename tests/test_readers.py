"""Host input path: TFRecord framing + SequenceExample/Example parsing in
libevc_io.so and the shuffle/batch pipeline of cs/train.py:129-176, all on CPU.

The proto parser is pinned against google.protobuf (an independent implementation) with the
message definitions of tensorflow/core/example/{feature,example}.proto built at run time; CRC32C
against the RFC 3720 known answers."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from efficientvideoclassification_youtube8m_amd import readers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _crc(b):
    return readers.load_io().evc_crc32c(C.cast(C.c_char_p(b), C.c_void_p), len(b))


def test_crc32c_known_answers():
    assert _crc(b"123456789") == 0xE3069283
    assert _crc(b"\x00" * 32) == 0x8A9136AA                       # RFC 3720 B.4
    assert _crc(b"\xff" * 32) == 0x62A8AB43
    assert _crc(bytes(range(32))) == 0x46DD794E
    assert _crc(b"") == 0


def test_io_header_matches_exports():
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "evc_io.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(evc_\w+)\s*\(", hdr))
    assert declared == set(readers.IO_EXPORTS)
    lib = readers.load_io()
    for name in declared:
        assert hasattr(lib, name), name


def _tf_example_protos():
    """tf.train.{Feature, Features, FeatureList, FeatureLists, Example, SequenceExample} via google.protobuf."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="evc_test_example.proto", package="evctest", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def field(m, name, num, typ, label=T.LABEL_OPTIONAL, type_name=None, oneof=None):
        f = m.field.add()
        f.name, f.number, f.type, f.label = name, num, typ, label
        if type_name:
            f.type_name = type_name
        if oneof is not None:
            f.oneof_index = oneof
        return f

    field(msg("BytesList"), "value", 1, T.TYPE_BYTES, T.LABEL_REPEATED)
    field(msg("FloatList"), "value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED)
    field(msg("Int64List"), "value", 1, T.TYPE_INT64, T.LABEL_REPEATED)
    m = msg("Feature")
    m.oneof_decl.add().name = "kind"
    field(m, "bytes_list", 1, T.TYPE_MESSAGE, type_name=".evctest.BytesList", oneof=0)
    field(m, "float_list", 2, T.TYPE_MESSAGE, type_name=".evctest.FloatList", oneof=0)
    field(m, "int64_list", 3, T.TYPE_MESSAGE, type_name=".evctest.Int64List", oneof=0)

    def map_of(parent, fname, value_type):
        e = parent.nested_type.add()
        e.name = fname.title().replace("_", "") + "Entry"
        e.options.map_entry = True
        field(e, "key", 1, T.TYPE_STRING)
        field(e, "value", 2, T.TYPE_MESSAGE, type_name=value_type)
        field(parent, fname, 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".evctest.%s.%s" % (parent.name, e.name))

    map_of(msg("Features"), "feature", ".evctest.Feature")
    field(msg("FeatureList"), "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".evctest.Feature")
    map_of(msg("FeatureLists"), "feature_list", ".evctest.FeatureList")
    field(msg("Example"), "features", 1, T.TYPE_MESSAGE, type_name=".evctest.Features")
    m = msg("SequenceExample")
    field(m, "context", 1, T.TYPE_MESSAGE, type_name=".evctest.Features")
    field(m, "feature_lists", 2, T.TYPE_MESSAGE, type_name=".evctest.FeatureLists")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("evctest." + n))
    return get("Example"), get("SequenceExample")


def _random_video(rng, n, sizes=(1024, 128), names=("rgb", "audio")):
    return {nm: rng.integers(0, 256, (n, sz), dtype=np.uint8) for nm, sz in zip(names, sizes)}


def _parse_frame(payload, names=("rgb", "audio"), sizes=(1024, 128), max_frames=300):
    lib = readers.load_io()
    nm = readers._Names(list(names), list(sizes))
    fr = np.full((max_frames, nm.row), 7, np.uint8)
    nf, nl = np.zeros(1, np.int32), np.zeros(1, np.int32)
    lab, ids = np.zeros(64, np.int64), np.zeros(32, np.uint8)
    rc = lib.evc_parse_yt8m_frame_example(C.cast(C.c_char_p(payload), C.c_void_p), len(payload), nm.names, readers._ptr(nm.sizes),
                                          nm.n, max_frames, readers._ptr(fr), readers._ptr(nf), readers._ptr(lab), 64,
                                          readers._ptr(nl), readers._ptr(ids), 32)
    if rc != 0:
        raise readers.EvcIoError(lib.evc_io_last_error().decode())
    return fr, int(nf[0]), lab[:nl[0]].tolist(), readers._id_str(ids)


def test_parser_against_google_protobuf(tmp_path):
    """Records serialised by google.protobuf parse to the same content, and records written by our
    encoder parse back identically under google.protobuf."""
    Example, SequenceExample = _tf_example_protos()
    rng = np.random.default_rng(0)
    for n in (1, 37, 300, 345):
        feats = _random_video(rng, n)
        labels = [3, 4715, 128, 0]
        se = SequenceExample()
        se.context.feature["id"].bytes_list.value.append(b"abcdEFGH")
        se.context.feature["labels"].int64_list.value.extend(labels)
        se.context.feature["unrelated"].float_list.value.extend([1.5, 2.5])
        for nm in ("audio", "rgb"):                                   # map order must not matter
            for row in feats[nm]:
                se.feature_lists.feature_list[nm].feature.add().bytes_list.value.append(row.tobytes())
        fr, nf, lab, vid = _parse_frame(se.SerializeToString())
        assert nf == min(n, 300) and lab == labels and vid == "abcdEFGH"
        want = np.concatenate([feats["rgb"], feats["audio"]], 1)[:300]
        np.testing.assert_array_equal(fr[:nf], want)
        assert not fr[nf:].any()                                      # resize_axis zero padding
        # our writer -> google parser
        ours = readers.encode_frame_example("abcdEFGH", labels, feats)
        back = SequenceExample.FromString(ours)
        assert list(back.context.feature["labels"].int64_list.value) == labels
        assert back.context.feature["id"].bytes_list.value[0] == b"abcdEFGH"
        got = np.stack([np.frombuffer(f.bytes_list.value[0], np.uint8) for f in back.feature_lists.feature_list["rgb"].feature])
        np.testing.assert_array_equal(got, feats["rgb"])
    # video-level Example
    ex = Example()
    ex.features.feature["id"].bytes_list.value.append(b"vid0")
    ex.features.feature["labels"].int64_list.value.extend([7, 9])
    mr, ma = rng.standard_normal(1024).astype(np.float32), rng.standard_normal(128).astype(np.float32)
    ex.features.feature["mean_rgb"].float_list.value.extend(mr.tolist())
    ex.features.feature["mean_audio"].float_list.value.extend(ma.tolist())
    payload = ex.SerializeToString()
    assert Example.FromString(readers.encode_video_example("vid0", [7, 9], {"mean_rgb": mr, "mean_audio": ma})) == ex
    readers.write_tfrecord(str(tmp_path / "g.tfrecord"), [payload])
    rd = readers.YT8MAggregatedFeatureReader(feature_names=["mean_rgb", "mean_audio"], feature_sizes=[1024, 128])
    (ids, ft, lb, nf), = list(rd.prepare_reader(str(tmp_path / "g.tfrecord")))
    assert ids == ["vid0"] and np.flatnonzero(lb[0]).tolist() == [7, 9]
    np.testing.assert_array_equal(ft[0], np.concatenate([mr, ma]))


def test_frame_parser_errors():
    rng = np.random.default_rng(1)
    with pytest.raises(readers.EvcIoError, match="expected 1024"):
        _parse_frame(readers.encode_frame_example("x", [1], {"rgb": rng.integers(0, 256, (5, 1000), dtype=np.uint8),
                                                             "audio": rng.integers(0, 256, (5, 128), dtype=np.uint8)}))
    with pytest.raises(readers.EvcIoError, match="disagree"):        # tf.assert_equal cs/readers.py:225
        _parse_frame(readers.encode_frame_example("x", [1], {"rgb": rng.integers(0, 256, (5, 1024), dtype=np.uint8),
                                                             "audio": rng.integers(0, 256, (6, 128), dtype=np.uint8)}))
    with pytest.raises(readers.EvcIoError, match="is missing from the record"):
        _parse_frame(readers.encode_frame_example("x", [1], {"inc3": rng.integers(0, 256, (5, 1024), dtype=np.uint8)}))
    # ONE requested feature list absent (parse_single_sequence_example raises; zero audio columns would train silently)
    with pytest.raises(readers.EvcIoError, match="feature list 'audio' is missing"):
        _parse_frame(readers.encode_frame_example("x", [1], {"rgb": rng.integers(0, 256, (5, 1024), dtype=np.uint8)}))
    with pytest.raises(readers.EvcIoError):
        _parse_frame(readers.encode_frame_example("x", [1], _random_video(rng, 4))[:-3])      # truncated proto
    fr, nf, lab, vid = _parse_frame(readers.encode_frame_example("x", [], _random_video(rng, 2)))
    assert nf == 2 and lab == []


def test_corrupt_length_field_is_an_error_not_an_abort(tmp_path):
    """A record header that claims more bytes than the file holds (with a matching length CRC, so only the bound
    against the file size can catch it) must come back as EvcIoError - never std::bad_alloc through the C ABI."""
    import struct
    lib = readers.load_io()
    p = tmp_path / "huge.tfrecord"
    hdr = struct.pack("<Q", 1 << 46)
    crc = lib.evc_masked_crc32c(C.cast(C.c_char_p(hdr), C.c_void_p), 8)
    p.write_bytes(hdr + struct.pack("<I", crc) + b"abcdefgh")
    for verify in (True, False):
        with pytest.raises(readers.EvcIoError, match="claims"):
            readers.scan_tfrecord(str(p), verify_crc=verify)
    # a (offset, length) pair outside the file handed to the batch reader
    rng = np.random.default_rng(0)
    good = tmp_path / "one.tfrecord"
    readers.write_tfrecord(str(good), [readers.encode_frame_example("a", [1], _random_video(rng, 3))])
    rd = readers.YT8MFrameFeatureReader(feature_names=["rgb", "audio"], feature_sizes=[1024, 128], max_frames=4)
    off, ln = readers.scan_tfrecord(str(good))
    x = np.zeros((1, 4, 1152), np.uint8); n = np.zeros(1, np.int32); y = np.zeros((1, 4716), np.uint8)
    ids = np.zeros((1, readers.ID_CAP), np.uint8)
    rd.read_into(str(good), off, ln, x, n, y, ids)
    assert n[0] == 3
    with pytest.raises(readers.EvcIoError, match="outside the file"):
        rd.read_into(str(good), off, np.asarray([1 << 40], np.int64), x, n, y, ids)


def test_tfrecord_scan_and_crc(tmp_path):
    rng = np.random.default_rng(2)
    payloads = [readers.encode_frame_example("v%d" % i, [i], _random_video(rng, 3 + i)) for i in range(5)]
    p = str(tmp_path / "a.tfrecord")
    readers.write_tfrecord(p, payloads)
    off, ln = readers.scan_tfrecord(p, verify_crc=True)
    assert ln.tolist() == [len(x) for x in payloads]
    raw = open(p, "rb").read()
    for o, l, x in zip(off, ln, payloads):
        assert raw[o:o + l] == x
    bad = bytearray(raw)
    bad[off[2] + 10] ^= 1
    pb = str(tmp_path / "bad.tfrecord")
    open(pb, "wb").write(bytes(bad))
    with pytest.raises(readers.EvcIoError, match="bad data crc"):
        readers.scan_tfrecord(pb, verify_crc=True)
    open(pb, "wb").write(raw[:-7])
    with pytest.raises(readers.EvcIoError, match="truncated"):
        readers.scan_tfrecord(pb, verify_crc=True)
    with pytest.raises(readers.EvcIoError, match="cannot open"):
        readers.scan_tfrecord(str(tmp_path / "missing.tfrecord"))


def _dataset(tmp_path, files=3, per_file=7, **kw):
    return readers.write_synthetic_frame_dataset(str(tmp_path), files, per_file, min_frames=2, max_frames=40, seed=5, **kw)


def test_pipeline_epoch_coverage_and_batching(tmp_path):
    _dataset(tmp_path)
    rd = readers.YT8MFrameFeatureReader(feature_names=["rgb", "audio"], feature_sizes=[1024, 128], max_frames=30)
    pat = str(tmp_path / "train*.tfrecord")
    # ground truth via the one-example generator (batch of 1, cs/readers.py:236-246)
    truth = {}
    for ids, mat, lab, nf in rd.prepare_reader(sorted(str(p) for p in tmp_path.glob("train*.tfrecord"))):
        assert mat.shape == (1, 30, 1152) and lab.shape == (1, 4716) and lab.dtype == bool
        truth[ids[0]] = (mat[0].copy(), lab[0].copy(), int(nf[0]))
    assert len(truth) == 21 and max(v[2] for v in truth.values()) == 30      # truncation at max_frames happened
    seen, sizes = [], []
    for ids, x, y, n in readers.get_input_data_tensors(rd, pat, batch_size=4, num_epochs=2, num_readers=2, seed=0):
        sizes.append(len(ids))
        assert x.dtype.is_floating_point is False and tuple(x.shape[1:]) == (30, 1152)
        for i, vid in enumerate(ids):
            np.testing.assert_array_equal(x[i].numpy(), truth[vid][0])
            np.testing.assert_array_equal(y[i].numpy().astype(bool), truth[vid][1])
            assert int(n[i]) == truth[vid][2]
            assert not x[i, int(n[i]):].any()
        seen += ids
    assert sizes == [4] * 10 + [2]                                           # allow_smaller_final_batch
    assert sorted(seen) == sorted(list(truth) * 2)                           # every video exactly num_epochs times
    assert seen[:21] != sorted(seen[:21])                                    # shuffled
    # evaluation input: one pass, file order, deterministic
    ev = [i for ids, *_ in readers.get_input_evaluation_tensors(rd, pat, batch_size=5) for i in ids]
    assert ev == sorted(truth)
    with pytest.raises(IOError, match="Unable to find training files"):
        readers.get_input_data_tensors(rd, str(tmp_path / "nope*.tfrecord"), batch_size=4)


def test_pipeline_rank_sharding(tmp_path):
    _dataset(tmp_path, files=4, per_file=5)
    rd = readers.YT8MFrameFeatureReader(feature_names=["rgb", "audio"], feature_sizes=[1024, 128], max_frames=30)
    pat = str(tmp_path / "train*.tfrecord")
    parts = [[i for ids, *_ in readers.get_input_data_tensors(rd, pat, 3, num_epochs=1, seed=r, rank=r, world_size=2,
                                                              drop_remainder=False) for i in ids] for r in range(2)]
    assert len(parts[0]) == len(parts[1]) == 10 and not set(parts[0]) & set(parts[1])
    one = str(tmp_path / "train0000.tfrecord")                               # fewer files than ranks -> record sharding
    parts = [[i for ids, *_ in readers.get_input_data_tensors(rd, one, 3, num_epochs=1, seed=r, rank=r, world_size=2,
                                                              drop_remainder=False) for i in ids] for r in range(2)]
    assert sorted(parts[0] + parts[1]) == ["v00%04d" % i for i in range(5)] and not set(parts[0]) & set(parts[1])


def test_pipeline_never_ragged_under_data_parallelism(tmp_path):
    """world_size > 1: every batch a rank hands out has exactly batch_size videos (the remainder is dropped) and
    num_batches counts whole batches only, so MIN over the ranks is a step count every rank can run with equal
    payloads (train.py).  Ranks own different numbers of records here (3 files: 2 vs 1)."""
    _dataset(tmp_path, files=3, per_file=7)
    rd = readers.YT8MFrameFeatureReader(feature_names=["rgb", "audio"], feature_sizes=[1024, 128], max_frames=30)
    pat = str(tmp_path / "train*.tfrecord")
    counts = []
    for r in range(2):
        pipe = readers.get_input_data_tensors(rd, pat, 4, num_epochs=1, seed=r, rank=r, world_size=2)
        assert pipe.drop_remainder
        sizes = [len(ids) for ids, *_ in pipe]
        assert sizes == [4] * pipe.num_batches, (r, sizes, pipe.num_batches)
        counts.append(pipe.num_batches)
    assert counts == [14 // 4, 7 // 4]
    single = readers.get_input_data_tensors(rd, pat, 4, num_epochs=1, seed=0)          # one process: smaller final batch kept
    assert not single.drop_remainder and [len(ids) for ids, *_ in single] == [4] * 5 + [1] and single.num_batches == 6


def test_aggregated_reader(tmp_path):
    rng = np.random.default_rng(3)
    rows = [(("a%d" % i), [i, i + 10], rng.standard_normal(1024).astype(np.float32), rng.standard_normal(128).astype(np.float32))
            for i in range(6)]
    p = str(tmp_path / "video.tfrecord")
    readers.write_tfrecord(p, [readers.encode_video_example(i, l, {"mean_audio": a, "mean_rgb": r}) for i, l, r, a in rows])
    rd = readers.YT8MAggregatedFeatureReader(feature_names=["mean_rgb", "mean_audio"], feature_sizes=[1024, 128])
    (ids, ft, lb, nf), = list(rd.prepare_reader(p))
    assert ids == [r[0] for r in rows] and (nf == 1).all()
    for k, (_, l, r, a) in enumerate(rows):
        np.testing.assert_array_equal(ft[k], np.concatenate([r, a]))
        assert np.flatnonzero(lb[k]).tolist() == l
    out = list(readers.get_input_evaluation_tensors(rd, p, batch_size=4))
    assert [len(o[0]) for o in out] == [4, 2] and out[0][1].dtype.is_floating_point
    bad = readers.YT8MAggregatedFeatureReader(feature_names=["mean_rgb", "mean_inc3"], feature_sizes=[1024, 128])
    with pytest.raises(readers.EvcIoError, match="missing"):
        list(bad.prepare_reader(p))


def test_reader_edge_cases(tmp_path):
    """Empty files, records without frames, more labels than fit, long ids."""
    rd = readers.YT8MFrameFeatureReader(feature_names=["rgb", "audio"], feature_sizes=[1024, 128], max_frames=30)
    empty = str(tmp_path / "empty.tfrecord")
    open(empty, "wb").close()
    assert readers.scan_tfrecord(empty, verify_crc=True)[0].size == 0
    assert list(rd.prepare_reader(empty)) == []
    assert list(readers.get_input_evaluation_tensors(rd, empty, batch_size=4)) == []
    rng = np.random.default_rng(9)
    zero = {"rgb": np.zeros((0, 1024), np.uint8), "audio": np.zeros((0, 128), np.uint8)}
    p = str(tmp_path / "edge.tfrecord")
    readers.write_tfrecord(p, [readers.encode_frame_example("z" * 50, [5], zero),
                               readers.encode_frame_example("b", list(range(40)) + [4715, 99999, -1], _random_video(rng, 3))])
    (ids, x, y, n), = list(readers.get_input_evaluation_tensors(rd, p, batch_size=8))
    assert n.tolist() == [0, 3] and not x[0].any()
    assert ids[0] == "z" * (readers.ID_CAP - 1) and ids[1] == "b"            # ids are cut at ID_CAP-1 bytes
    assert np.flatnonzero(y[0].numpy()).tolist() == [5]
    assert np.flatnonzero(y[1].numpy()).tolist() == list(range(40)) + [4715]   # out-of-range classes are dropped
