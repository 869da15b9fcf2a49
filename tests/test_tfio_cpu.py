"""TensorFlow file formats without TensorFlow: Saver-V2 checkpoint bundles and GZIP
TFRecords of SequenceExamples.  Known answers pin the checksum (RFC 3720 CRC-32C vectors), the leveldb masking
constant and the SSTable magic; everything else is round-tripped and checked at the byte level where the format
fixes the bytes."""
import os
from collections import OrderedDict
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "acoustic-image-generation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from acimg import tfio  # noqa: E402


def test_crc32c_known_answers():
    # RFC 3720 B.4 test patterns + the classic check value
    assert tfio.crc32c(b"123456789") == 0xE3069283
    assert tfio.crc32c(bytes(32)) == 0x8A9136AA
    assert tfio.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tfio.crc32c(bytes(range(32))) == 0x46DD794E
    assert tfio.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    # incremental = one shot; unaligned starts
    data = bytes((i * 7 + 3) & 0xFF for i in range(1000))
    assert tfio.crc32c(data[500:], tfio.crc32c(data[:500])) == tfio.crc32c(data)
    assert tfio.crc32c_array(np.frombuffer(data, np.uint8)[3:]) == tfio.crc32c(data[3:])
    # leveldb masking: rotate right 15, add 0xa282ead8 (crc32c.h); self-inverse pair
    assert tfio.mask_crc(0) == 0xA282EAD8
    for v in (0, 1, 0xDEADBEEF, 0xFFFFFFFF):
        assert tfio.unmask_crc(tfio.mask_crc(v)) == v


def test_checkpoint_roundtrip_and_layout(tmp_path):
    rng = np.random.RandomState(0)
    tensors = {
        "UNetAcRes/layer1/conv_1/kernel": rng.randn(3, 3, 12, 128).astype(np.float32),
        "UNetAcRes/layer1/conv_1/bias": rng.randn(128).astype(np.float32),
        "resnet_v1_50/conv1/BatchNorm/moving_variance": rng.rand(64).astype(np.float32),
        "global_step": np.array(1234, dtype=np.int64),
        "UNetAcRes/mean/kernel": rng.randn(12, 16, 145, 150).astype(np.float32),   # 16.7 MB: many index entries
    }
    for i in range(300):   # enough keys for several SSTable data blocks (prefix compression + restarts exercised)
        tensors["resnet_v1_50/block%d/unit_%d/bottleneck_v1/conv%d/weights" % (i % 4, i // 4, i % 3)] = \
            rng.randn(2, 3).astype(np.float32)
    prefix = str(tmp_path / "epoch_7.ckpt")
    tfio.write_checkpoint(prefix, tensors)
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    raw = open(prefix + ".index", "rb").read()
    assert raw[-8:] == bytes.fromhex("57fb808b247547db")          # kTableMagicNumber, little endian
    back = tfio.read_checkpoint(prefix, verify=True)
    assert set(back) == set(tensors)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    # keys come back sorted (an SSTable is ordered) with the header under the empty key
    listing = tfio.list_checkpoint(prefix)
    keys = list(listing)
    assert keys[0] == "" and keys[1:] == sorted(keys[1:]) and listing[""]["num_shards"] == 1
    e = listing["UNetAcRes/layer1/conv_1/kernel"]
    assert e["dtype"] == 1 and e["shape"] == (3, 3, 12, 128) and e["size"] == 3 * 3 * 12 * 128 * 4
    # selective read
    only = tfio.read_checkpoint(prefix, names=lambda n: n.startswith("UNetAcRes/"))
    assert set(only) == {k for k in tensors if k.startswith("UNetAcRes/")}
    # corruption is detected: flip one byte of a data block / of a tensor
    bad = bytearray(raw)
    bad[10] ^= 0x40
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(IOError):
        tfio.read_checkpoint(prefix)
    open(prefix + ".index", "wb").write(raw)
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(5)
        f.write(b"\x99")
    with pytest.raises(IOError):
        tfio.read_checkpoint(prefix, verify=True)


def test_checkpoint_feeds_the_model_state(tmp_path):
    """a TF-named bundle initialises the HIP host model through the same entry the reference uses
    (models/unet_acresnet.py:33-41 init_model -> assign_from_checkpoint_fn)"""
    import torch
    from acimg.session import Session
    from acimg.unet_vae import UNetSound
    from oracle import unet_vae as ouv

    params = ouv.init_params("UNetSound", seed=3, bias_std=0.1, bn_jitter=0.1)
    prefix = str(tmp_path / "model.ckpt")
    tfio.write_checkpoint(prefix, {k: v.numpy() for k, v in params.items()})
    sess = Session(torch.device("cpu"))
    m = UNetSound()
    m._build_model(torch.zeros(1, 99, 257, 1), session=sess)
    sess.finalize()
    loaded = m.init_model(sess, prefix)
    assert len(loaded) == len(params)
    sd = sess.store.state_dict()
    for k, v in params.items():
        assert torch.equal(sd[k], v), k


def test_tfrecord_sequence_example_roundtrip(tmp_path):
    """the dataset record of convert_data.py:247-279 / outdoor_data_mfcc.py:263-343"""
    rng = np.random.RandomState(1)
    context = {"classes": np.array([3], np.int64), "location": np.array([1], np.int64),
               "audio_image/height": np.array([36], np.int64), "audio_image/width": np.array([48], np.int64),
               "audio_image/depth": np.array([12], np.int64), "audio_data/mics": np.array([128], np.int64),
               "audio_data/samples": np.array([1024], np.int64), "video/height": np.array([224], np.int64),
               "video/width": np.array([298], np.int64), "video/depth": np.array([3], np.int64)}
    steps = 2
    lists = {"audio/image": [rng.rand(36 * 48 * 12).astype(np.float32) for _ in range(steps)],
             "audio/data": [rng.randint(-2 ** 31, 2 ** 31 - 1, size=1024).astype(np.int64) for _ in range(steps)],
             "video/image": [bytes(rng.randint(0, 256, size=224 * 298 * 3).astype(np.uint8)) for _ in range(steps)]}
    rec = tfio.build_sequence_example(context, lists)
    for comp in (None, "GZIP"):
        path = str(tmp_path / ("data_%s.tfrecord" % comp))
        tfio.write_tfrecord(path, [rec, rec[:100] + rec[100:]], compression=comp)
        got = list(tfio.read_tfrecord(path))          # compression auto-detected from the gzip magic
        assert got == [rec, rec]
        ctx, fl = tfio.parse_sequence_example(got[0])
        assert set(ctx) == set(context) and all(np.array_equal(ctx[k], context[k]) for k in context)
        assert np.array_equal(fl["audio/image"][1], lists["audio/image"][1])
        assert np.array_equal(fl["audio/data"][0], lists["audio/data"][0])      # negative int64 varints
        assert fl["video/image"][1] == [lists["video/image"][1]]
    # framing bytes: [len u64][masked crc(len)][payload][masked crc(payload)]
    raw = open(str(tmp_path / "data_None.tfrecord"), "rb").read()
    n = struct.unpack("<Q", raw[:8])[0]
    assert n == len(rec)
    assert struct.unpack("<I", raw[8:12])[0] == tfio.mask_crc(tfio.crc32c(raw[:8]))
    assert struct.unpack("<I", raw[12 + n:16 + n])[0] == tfio.mask_crc(tfio.crc32c(rec))
    bad = bytearray(raw)
    bad[40] ^= 1
    open(str(tmp_path / "bad.tfrecord"), "wb").write(bytes(bad))
    with pytest.raises(IOError):
        list(tfio.read_tfrecord(str(tmp_path / "bad.tfrecord")))


def test_iou_curve_host_arithmetic():
    """accuracy(tau) and the trapezoid area (iouenergythreshold.py:226-236, areaundercurve.py:26-40)"""
    from acimg import evaluate
    from sklearn import metrics

    ious = np.array([0.05, 0.15, 0.55, 0.95, 1.0, 0.3])
    acc = evaluate.accuracy_curve(ious)
    assert acc[0] == 1.0 and acc[-1] == 0.0 and abs(acc[5] - 0.5) < 1e-12      # IoU > 0.5: 3 of 6
    want = metrics.auc(evaluate.THRESHOLDS[::-1], list(acc[::-1]))
    assert abs(evaluate.area_under_curve(acc) - want) < 1e-12


def _dataset_record(rng, steps=12, with_acoustic=True):
    """one SequenceExample in the layout of convert_data.py:247-279 / outdoor_data_mfcc.py:263-343"""
    ai = rng.rand(steps, 36, 48, 12).astype(np.float32)
    sa = (rng.randn(steps, 1024) * 1000).astype(np.int32)
    vi = rng.randint(0, 256, size=(steps, 8, 10, 3)).astype(np.uint8)        # small frames keep the test light
    ctx = OrderedDict([("classes", np.array([7])), ("location", np.array([2])),
                       ("audio_data/mics", np.array([1])), ("audio_data/samples", np.array([1024])),
                       ("video/height", np.array([8])), ("video/width", np.array([10])), ("video/depth", np.array([3]))])
    lists = OrderedDict([("audio/data", [s.tobytes() for s in sa]), ("video/image", [v.tobytes() for v in vi])])
    if with_acoustic:
        ctx.update([("audio_image/height", np.array([36])), ("audio_image/width", np.array([48])),
                    ("audio_image/depth", np.array([12]))])
        lists["audio/image"] = [a.tobytes() for a in ai]
    return tfio.build_sequence_example(ctx, lists), ai, sa, vi


@pytest.mark.parametrize("compression", ["GZIP", None])
def test_native_record_reader_equals_python_reader(tmp_path, compression):
    """the C++ reader behind the C ABI (acimg_gzip_inflate, acimg_tfrecord_index, acimg_sequence_example_decode)
    returns byte-for-byte what tfio.py's Python writer put in / its Python reader gets out, and decodes a record like
    `_parse_sequence` (dataloader/outdoor_data_mfcc.py:263-343) incl. the left-right + up-down flip of :314-315"""
    rng = np.random.RandomState(5)
    recs, truth = [], []
    for i in range(3):
        r, ai, sa, vi = _dataset_record(rng, steps=12 if i < 2 else 5, with_acoustic=i != 1)
        recs.append(r)
        truth.append((ai if i != 1 else None, sa, vi))
    recs.append(b"")                                            # an empty record is legal framing
    path = str(tmp_path / "data.tfrecord")
    tfio.write_tfrecord(path, recs, compression=compression)
    py = list(tfio.read_tfrecord(path))
    nat = tfio.read_tfrecord_native(path)
    assert len(nat) == len(py) == 4
    for a, b, c in zip(nat, py, recs):
        assert bytes(a) == b == c
    for i in range(3):
        d = tfio.decode_sequence_example_native(nat[i])
        ai, sa, vi = truth[i]
        assert (d["action"], d["location"]) == (7, 2)
        np.testing.assert_array_equal(d["audio_samples"], sa)
        np.testing.assert_array_equal(d["video_images"], vi)
        if ai is None:
            assert d["audio_images"].size == 0 and d["dims"].audio_image_steps == 0
        else:
            np.testing.assert_array_equal(d["audio_images"], ai[:, ::-1, ::-1, :])
        ctx, lists = tfio.parse_sequence_example(bytes(nat[i]))       # the Python parser sees the same values
        assert int(ctx["classes"][0]) == d["action"] and len(lists["audio/data"]) == d["dims"].audio_data_steps
    with pytest.raises(Exception):
        tfio.decode_sequence_example_native(nat[3])             # no 'classes' / 'location': FixedLenFeature missing


def test_native_record_reader_detects_corruption(tmp_path):
    rng = np.random.RandomState(6)
    rec, _, _, _ = _dataset_record(rng, steps=2)
    good = str(tmp_path / "g.tfrecord")
    tfio.write_tfrecord(good, [rec, rec])
    raw = bytearray(open(good, "rb").read())
    for pos, what in ((20 + 12, "payload"), (9, "length"), (len(raw) - 2, "payload")):
        bad = bytearray(raw)
        bad[pos] ^= 0x40
        p = str(tmp_path / "bad.tfrecord")
        open(p, "wb").write(bad)
        with pytest.raises(IOError) as ei:
            tfio.read_tfrecord_native(p)
        assert what in str(ei.value), str(ei.value)
        assert len(tfio.read_tfrecord_native(p, verify=False)) == 2 or what == "length"
    p = str(tmp_path / "trunc.tfrecord")
    open(p, "wb").write(raw[:len(raw) - 7])
    with pytest.raises(IOError) as ei:
        tfio.read_tfrecord_native(p)
    assert "truncated" in str(ei.value)
    # a truncated GZIP stream is an error too, not a short read
    gz = str(tmp_path / "g.gz.tfrecord")
    tfio.write_tfrecord(gz, [rec] * 3, compression="GZIP")
    blob = open(gz, "rb").read()
    open(gz, "wb").write(blob[:len(blob) // 2])
    with pytest.raises(Exception) as ei:
        tfio.read_tfrecord_native(gz)
    assert "GZIP" in str(ei.value)
    # malformed protobuf inside a well-framed record
    with pytest.raises(Exception):
        tfio.decode_sequence_example_native(bytes(rec[:len(rec) // 3]))
    empty = str(tmp_path / "empty.tfrecord")
    open(empty, "wb").close()
    assert tfio.read_tfrecord_native(empty) == []
    # context dimensions from the file are range-checked BEFORE any product is formed: H = W = -1 (whose product with a
    # depth of 12 * 4 would "match" a 48-byte step), a zero and a huge dimension are all refused with EINVAL
    for h, w, d_, nbytes in ((-1, -1, 12, 48), (1 << 40, 1, 1, 4), (-36, -48, 12, 36 * 48 * 12 * 4)):
        ctx = OrderedDict([("classes", np.array([1])), ("location", np.array([1])),
                           ("audio_image/height", np.array([h])), ("audio_image/width", np.array([w])),
                           ("audio_image/depth", np.array([d_]))])
        bad = tfio.build_sequence_example(ctx, OrderedDict([("audio/image", [b"\0" * nbytes])]))
        with pytest.raises(Exception) as ei:
            tfio.decode_sequence_example_native(bad)
        assert "out of range" in str(ei.value), str(ei.value)


def test_reader_on_hand_built_known_answer_bundle(tmp_path):
    """f1: a Saver-V2 checkpoint built byte by byte from the published SSTable / tensor-bundle formats by
    tests/golden/make_bundle_golden.py (own CRC-32C, no acimg code; index key of the block = LevelDB's short
    successor, prefix-compressed keys, BundleHeaderProto / BundleEntryProto packed field by field) is read back
    exactly.  Still NOT a TensorFlow-written file: unpinned against TF itself."""
    import importlib.util
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_bundle_golden", os.path.join(gold, "make_bundle_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    prefix = os.path.join(gold, "bundle_known_answer", "model.ckpt")
    # the committed fixture IS what the committed script writes
    mk.OUT = str(tmp_path)
    mk.main()
    for sfx in (".index", ".data-00000-of-00001"):
        assert open(prefix + sfx, "rb").read() == open(str(tmp_path / ("model.ckpt" + sfx)), "rb").read()
    idx = open(prefix + ".index", "rb").read()
    assert idx[-8:] == struct.pack("<Q", 0xdb4775248b80fb57) and len(idx) >= 48
    assert idx[:3] == b"\x00\x00\x06" and idx[3:9] == b"\x08\x01\x1a\x02\x08\x01"     # "" -> BundleHeaderProto
    entries = tfio.list_checkpoint(prefix, verify=True)
    assert entries[""]["num_shards"] == 1
    want = mk.tensors()
    assert list(entries)[1:] == list(want)
    off = 0
    for name, a in want.items():
        e = entries[name]
        assert (e["offset"], e["size"], e["shard_id"], tuple(e["shape"])) == (off, a.nbytes, 0, a.shape), name
        assert tfio.unmask_crc(e["crc32c"]) == tfio.crc32c_array(a) == mk.crc32c(a.tobytes()) or a.nbytes > 4096
        off += a.nbytes
    got = tfio.read_checkpoint(prefix, verify=True)
    for name, a in want.items():
        assert got[name].dtype == a.dtype and np.array_equal(got[name], a), name
    assert int(got["global_step"]) == 42 and got["UNetAcRes/layer7/conv_2/kernel"].shape == (3, 3, 64, 64)
    only = tfio.read_checkpoint(prefix, names=lambda n: n.startswith("UNetAcRes/"))
    assert sorted(only) == sorted(n for n in want if n.startswith("UNetAcRes/"))
    # the model-side loader accepts the prefix (trainer/mfcctrainer.py:214-225: saver.restore(session, prefix))
    from acimg.vision import load_state_file
    assert set(load_state_file(prefix)) == set(want)
    # one flipped data byte fails the per-tensor checksum; one flipped index byte fails the block checksum
    bad = str(tmp_path / "bad.ckpt")
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    raw[5] ^= 1
    open(bad + ".data-00000-of-00001", "wb").write(raw)
    open(bad + ".index", "wb").write(idx)
    with pytest.raises(IOError):
        tfio.read_checkpoint(bad, verify=True)
    bi = bytearray(idx)
    bi[20] ^= 1
    open(bad + ".index", "wb").write(bi)
    with pytest.raises(IOError):
        tfio.list_checkpoint(bad, verify=True)


def test_record_readers_property_based(tmp_path):
    """random record sets (sizes 0 ... 70 kB, GZIP or plain): the native C++ reader, the Python reader and the
    writer's input agree byte for byte; random SequenceExamples decode to what was put in (hypothesis)"""
    from hypothesis import given, settings
    from hypothesis import strategies as hst

    path = str(tmp_path / "p.tfrecord")

    @settings(max_examples=20, deadline=None)
    @given(hst.lists(hst.binary(min_size=0, max_size=70000), min_size=0, max_size=6), hst.booleans())
    def roundtrip(recs, gz):
        tfio.write_tfrecord(path, recs, compression="GZIP" if gz else None)
        assert [bytes(r) for r in tfio.read_tfrecord_native(path)] == recs == list(tfio.read_tfrecord(path))

    roundtrip()

    @settings(max_examples=15, deadline=None)
    @given(hst.integers(1, 4), hst.integers(1, 5), hst.integers(1, 6), hst.integers(1, 3), hst.integers(0, 2 ** 31 - 1))
    def decode(steps, h, w, dch, seed):
        rng = np.random.RandomState(seed % (2 ** 31))
        ai = rng.rand(steps, h, w, dch).astype(np.float32)
        sa = rng.randint(-2 ** 31, 2 ** 31 - 1, size=(steps, 7)).astype(np.int32)
        ctx = OrderedDict([("classes", np.array([seed % 14])), ("location", np.array([seed % 61])),
                           ("audio_image/height", np.array([h])), ("audio_image/width", np.array([w])),
                           ("audio_image/depth", np.array([dch])), ("audio_data/mics", np.array([1])),
                           ("audio_data/samples", np.array([7]))])
        rec = tfio.build_sequence_example(ctx, OrderedDict([("audio/image", [a.tobytes() for a in ai]),
                                                            ("audio/data", [s.tobytes() for s in sa])]))
        d = tfio.decode_sequence_example_native(rec)
        assert (d["action"], d["location"]) == (seed % 14, seed % 61)
        np.testing.assert_array_equal(d["audio_images"], ai[:, ::-1, ::-1, :])
        np.testing.assert_array_equal(d["audio_samples"], sa)
        assert d["video_images"].size == 0

    decode()

    @settings(max_examples=50, deadline=None)
    @given(hst.integers(0, 2 ** 64 - 1))
    def varints(v):
        b = tfio._put_varint(v)
        assert tfio._varint(b + b"\x00", 0) == (v, len(b)) and 1 <= len(b) <= 10

    varints()
