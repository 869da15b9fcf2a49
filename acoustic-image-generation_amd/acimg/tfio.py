"""TensorFlow file formats without TensorFlow: the two data formats either side of the
train step.

  * Saver-V2 checkpoint bundles — what `tf.train.Saver.save/restore` and `slim.assign_from_checkpoint_fn` read
    and write in the reference (trainer/mfcctrainer.py:214-247,476-490; models/vision.py:26-43;
    models/unet_acresnet.py:33-41).  `read_checkpoint(prefix)` -> {variable name: ndarray},
    `write_checkpoint(prefix, {name: ndarray})` writes a bundle TensorFlow can restore.
      <prefix>.index                 an SSTable (LevelDB table format: prefix-compressed key/value blocks, per-block
                                     trailer = 1 byte compression + masked CRC-32C, 48-byte footer with the metaindex /
                                     index block handles and the magic 0xdb4775248b80fb57); key "" -> BundleHeaderProto,
                                     key <tensor name> -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}
      <prefix>.data-0000i-of-0000n   raw little-endian tensor bytes at [offset, offset + size)
  * TFRecord files of tf.train.SequenceExample — the dataset format of dataloader/outdoor_data_mfcc.py:263-343,
    558-575 written by convert_data.py:247-279 (GZIP-compressed TFRecords): `read_tfrecord(path)` yields record
    payloads, `parse_sequence_example(bytes)` -> (context dict, feature-list dict), and the matching writers.
      record = [uint64 length][uint32 masked crc32c(length)][payload][uint32 masked crc32c(payload)]

The protobuf wire format is decoded by hand (varints, length-delimited fields): only the handful of messages
above are needed.  CRC-32C comes from libacimg (`acimg_crc32c`, host code).  The formats are restated from
TensorFlow's public sources (tensor_bundle.proto, table/format.cc, record_writer.cc, example.proto,
feature.proto); no TensorFlow installation exists here to cross-check, so the known-answer tests pin the
checksum (RFC 3720 vectors), the masking constant and the table magic, and round-trip everything else.
"""
import ctypes
import gzip
import os
import struct
from collections import OrderedDict

import numpy as np

from . import _lib

TABLE_MAGIC = 0xdb4775248b80fb57
MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


# ---- checksums ------------------------------------------------------------------------------------------------
def crc32c(data, crc=0):
    b = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    buf = (ctypes.c_char * len(b)).from_buffer_copy(b) if len(b) else None
    return int(_lib.load().acimg_crc32c(buf, len(b), crc))


def crc32c_array(a):
    a = np.asarray(a)
    if not a.flags.c_contiguous:
        a = np.array(a, order="C")
    return int(_lib.load().acimg_crc32c(a.ctypes.data_as(ctypes.c_void_p), a.nbytes, 0))


def mask_crc(crc):
    """leveldb / TFRecord masking: rotate right by 15 and add a constant"""
    return (((crc >> 15) | (crc << 17)) + MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m):
    rot = (m - MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---- protobuf wire format -----------------------------------------------------------------------------------
def _varint(buf, pos):
    r, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, pos
        shift += 7


def _put_varint(v):
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


def _fields(buf):
    """yield (field number, wire type, value) — value is an int (varint / fixed) or a memoryview (bytes)"""
    buf = memoryview(buf)
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _key(fn, wt):
    return _put_varint((fn << 3) | wt)


def _ld(fn, payload):
    return _key(fn, 2) + _put_varint(len(payload)) + bytes(payload)


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


# ---- LevelDB table (SSTable) ----------------------------------------------------------------------------------
def _read_block(data, off, size, verify=True):
    body = data[off:off + size]
    ctype = data[off + size]
    stored = struct.unpack_from("<I", data, off + size + 1)[0]
    if verify and unmask_crc(stored) != crc32c(data[off:off + size + 1]):
        raise IOError("SSTable block checksum mismatch at offset %d" % off)
    if ctype != 0:
        raise IOError("compressed SSTable blocks (type %d) are not supported" % ctype)
    nrestart = struct.unpack_from("<I", body, len(body) - 4)[0]
    end = len(body) - 4 - 4 * nrestart
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(body, pos)
        non_shared, pos = _varint(body, pos)
        vlen, pos = _varint(body, pos)
        key = key[:shared] + bytes(body[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(body[pos:pos + vlen])))
        pos += vlen
    return out


def read_table(path, verify=True):
    """[(key bytes, value bytes)] of an SSTable, in key order"""
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise IOError("%s is not an SSTable (bad magic)" % path)
    foot = data[-48:]
    pos = 0
    _, pos = _varint(foot, pos)          # metaindex handle
    _, pos = _varint(foot, pos)
    ioff, pos = _varint(foot, pos)
    isize, pos = _varint(foot, pos)
    out = []
    for _, handle in _read_block(data, ioff, isize, verify):
        boff, p2 = _varint(handle, 0)
        bsize, _ = _varint(handle, p2)
        out.extend(_read_block(data, boff, bsize, verify))
    return out


def _build_block(entries, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(prev), len(k))
            while shared < m and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_table(path, entries, block_size=4096):
    """entries: [(key bytes, value bytes)] sorted by key"""
    f = bytearray()
    index = []

    def emit(block):
        off = len(f)
        f.extend(block)
        f.append(0)                                             # kNoCompression
        f.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return _put_varint(off) + _put_varint(len(block))

    cur, cur_bytes = [], 0
    for k, v in entries:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 8
        if cur_bytes >= block_size:
            index.append((cur[-1][0], emit(_build_block(cur))))
            cur, cur_bytes = [], 0
    if cur:
        index.append((cur[-1][0], emit(_build_block(cur))))
    meta = emit(_build_block([]))
    idx = emit(_build_block(index, restart_interval=1))
    foot = meta + idx
    foot += b"\x00" * (40 - len(foot)) + struct.pack("<Q", TABLE_MAGIC)
    f.extend(foot)
    with open(path, "wb") as fh:
        fh.write(bytes(f))


# ---- checkpoint bundle ------------------------------------------------------------------------------------------
def _parse_shape(buf):
    dims = []
    for fn, wt, v in _fields(buf):
        if fn == 2:
            size = 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    size = _signed64(v2)
            dims.append(size)
    return tuple(dims)


def _parse_entry(buf):
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, slices=0)
    for fn, wt, v in _fields(buf):
        if fn == 1:
            e["dtype"] = v
        elif fn == 2:
            e["shape"] = _parse_shape(v)
        elif fn == 3:
            e["shard_id"] = v
        elif fn == 4:
            e["offset"] = v
        elif fn == 5:
            e["size"] = v
        elif fn == 6:
            e["crc32c"] = v
        elif fn == 7:
            e["slices"] += 1
    return e


def list_checkpoint(prefix, verify=True):
    """OrderedDict name -> entry dict (dtype code, shape, shard_id, offset, size, crc32c), plus '' -> header"""
    out = OrderedDict()
    for k, v in read_table(prefix + ".index", verify):
        if k == b"":
            hdr = dict(num_shards=1, endianness=0)
            for fn, wt, val in _fields(v):
                if fn == 1:
                    hdr["num_shards"] = val
                elif fn == 2:
                    hdr["endianness"] = val
            out[""] = hdr
        else:
            out[k.decode("utf-8")] = _parse_entry(v)
    return out


def read_checkpoint(prefix, names=None, verify=False):
    """{variable name: ndarray} from a Saver-V2 bundle.  names: optional filter (iterable or predicate);
    verify: also check each tensor's CRC-32C."""
    entries = list_checkpoint(prefix)
    hdr = entries.pop("", dict(num_shards=1, endianness=0))
    if hdr.get("endianness", 0) != 0:
        raise IOError("big-endian bundles are not supported")
    pred = (lambda n: True) if names is None else (names if callable(names) else set(names).__contains__)
    shards = {}
    out = OrderedDict()
    for name, e in entries.items():
        if not pred(name):
            continue
        if e["slices"]:
            raise IOError("%s: partitioned (sliced) variables are not supported" % name)
        if e["dtype"] not in _DTYPES:
            continue   # strings / resources (e.g. the Saver's bookkeeping tensors)
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, hdr["num_shards"]), dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if verify and e["crc32c"] is not None and unmask_crc(e["crc32c"]) != crc32c_array(raw):
            raise IOError("%s: tensor checksum mismatch" % name)
        out[name] = np.frombuffer(raw.tobytes(), dtype=_DTYPES[e["dtype"]]).reshape(e["shape"])
    return out


def write_checkpoint(prefix, tensors):
    """Write {name: array} as a single-shard Saver-V2 bundle (<prefix>.index + <prefix>.data-00000-of-00001)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    def _c(v):
        a = np.asarray(v)
        return a if a.flags.c_contiguous else np.array(a, order="C")      # (ascontiguousarray would turn 0-d into 1-d)

    items = sorted(((k.encode("utf-8"), _c(v)) for k, v in tensors.items()), key=lambda kv: kv[0])
    entries = [(b"", _key(1, 0) + _put_varint(1) + _ld(3, _key(1, 0) + _put_varint(1)))]   # num_shards=1, version.producer=1
    off = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for k, a in items:
            if a.dtype not in _DTYPE_CODES:
                raise TypeError("%s: dtype %s has no TensorFlow code here" % (k, a.dtype))
            shape = b"".join(_ld(2, _key(1, 0) + _put_varint(int(d))) for d in a.shape)
            e = _key(1, 0) + _put_varint(_DTYPE_CODES[a.dtype]) + _ld(2, shape)
            if off:
                e += _key(4, 0) + _put_varint(off)
            e += _key(5, 0) + _put_varint(a.nbytes) + _key(6, 5) + struct.pack("<I", mask_crc(crc32c_array(a)))
            entries.append((k, e))
            f.write(a.tobytes())
            off += a.nbytes
    write_table(prefix + ".index", entries)


# ---- TFRecord + SequenceExample -------------------------------------------------------------------------------
def read_tfrecord(path, compression=None, verify=True):
    """yield the payload bytes of every record; compression: None | 'GZIP' (auto-detected from the magic if None)"""
    with open(path, "rb") as probe:
        gz = probe.read(2) == b"\x1f\x8b"
    if compression == "GZIP" or (compression is None and gz):
        f = gzip.open(path, "rb")
    else:
        f = open(path, "rb")
    with f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError("truncated TFRecord header")
            n, lcrc = struct.unpack("<QI", head)
            if verify and unmask_crc(lcrc) != crc32c(head[:8]):
                raise IOError("TFRecord length checksum mismatch")
            data = f.read(n)
            tail = f.read(4)
            if len(data) < n or len(tail) < 4:
                raise IOError("truncated TFRecord")
            if verify and unmask_crc(struct.unpack("<I", tail)[0]) != crc32c(data):
                raise IOError("TFRecord payload checksum mismatch")
            yield data


def write_tfrecord(path, records, compression=None):
    f = gzip.open(path, "wb") if compression == "GZIP" else open(path, "wb")
    with f:
        for r in records:
            head = struct.pack("<Q", len(r))
            f.write(head + struct.pack("<I", mask_crc(crc32c(head))) + r + struct.pack("<I", mask_crc(crc32c(r))))


def _parse_feature(buf):
    """tf.train.Feature -> list of bytes | float32 ndarray | int64 ndarray"""
    for fn, wt, v in _fields(buf):
        if fn == 1:      # BytesList
            return [bytes(x) for f2, _, x in _fields(v) if f2 == 1]
        if fn == 2:      # FloatList: packed (wire type 2) or repeated fixed32
            vals = []
            for f2, w2, x in _fields(v):
                if f2 == 1 and w2 == 2:
                    vals.append(np.frombuffer(bytes(x), dtype="<f4"))
                elif f2 == 1:
                    vals.append(np.array([struct.unpack("<f", struct.pack("<I", x))[0]], dtype=np.float32))
            return np.concatenate(vals) if vals else np.zeros(0, np.float32)
        if fn == 3:      # Int64List: packed varints or repeated
            vals = []
            for f2, w2, x in _fields(v):
                if f2 == 1 and w2 == 2:
                    pos, x = 0, bytes(x)
                    while pos < len(x):
                        val, pos = _varint(x, pos)
                        vals.append(_signed64(val))
                elif f2 == 1:
                    vals.append(_signed64(x))
            return np.array(vals, dtype=np.int64)
    return np.zeros(0, np.float32)


def _parse_feature_map(buf):
    out = OrderedDict()
    for fn, wt, v in _fields(buf):          # map<string, X> entries: field 1, each {key = 1, value = 2}
        if fn != 1:
            continue
        key, val = None, None
        for f2, _, x in _fields(v):
            if f2 == 1:
                key = bytes(x).decode("utf-8")
            elif f2 == 2:
                val = x
        out[key] = val
    return out


def parse_sequence_example(buf):
    """tf.train.SequenceExample -> (context {name: value}, feature_lists {name: [value per step]})"""
    context, lists = OrderedDict(), OrderedDict()
    for fn, wt, v in _fields(buf):
        if fn == 1:      # Features context
            for k, fv in _parse_feature_map(v).items():
                context[k] = _parse_feature(fv)
        elif fn == 2:    # FeatureLists
            for k, fl in _parse_feature_map(v).items():
                lists[k] = [_parse_feature(x) for f2, _, x in _fields(fl) if f2 == 1]
    return context, lists


def _build_feature(val):
    if isinstance(val, (bytes, bytearray)):
        val = [bytes(val)]
    if isinstance(val, list) and (not val or isinstance(val[0], (bytes, bytearray))):
        return _ld(1, b"".join(_ld(1, x) for x in val))
    a = np.asarray(val)
    if a.dtype.kind == "f":
        return _ld(2, _ld(1, a.astype("<f4").tobytes()))
    return _ld(3, _ld(1, b"".join(_put_varint(int(x)) for x in a.reshape(-1))))


def build_sequence_example(context, feature_lists):
    ctx = b"".join(_ld(1, _ld(1, k.encode("utf-8")) + _ld(2, _build_feature(v))) for k, v in context.items())
    fls = b"".join(_ld(1, _ld(1, k.encode("utf-8")) + _ld(2, b"".join(_ld(1, _build_feature(x)) for x in steps)))
                   for k, steps in feature_lists.items())
    return _ld(1, ctx) + _ld(2, fls)


# ---- native reader (libacimg.so: csrc/records.hip) -----------------------------------------------------------------
def read_tfrecord_native(path, verify=True):
    """[record payload bytes] of a (GZIP or plain) TFRecord file through the C++ reader behind the C ABI
    (acimg_gzip_inflate + acimg_tfrecord_index): the whole file is inflated into ONE caller-owned buffer and indexed;
    records are zero-copy memoryviews of it."""
    lib = _lib.load()
    raw = np.fromfile(path, dtype=np.uint8)
    produced = ctypes.c_size_t(0)
    rc = lib.acimg_gzip_inflate(raw.ctypes.data, raw.size, None, 0, ctypes.byref(produced))
    if rc not in (0, -2):
        _lib.check(rc, "gzip_inflate")
    buf = np.empty(max(produced.value, 1), dtype=np.uint8)
    _lib.check(lib.acimg_gzip_inflate(raw.ctypes.data, raw.size, buf.ctypes.data, buf.size, ctypes.byref(produced)),
               "gzip_inflate")
    n = produced.value
    count = lib.acimg_tfrecord_index(buf.ctypes.data, n, None, None, 0, int(bool(verify)))
    if count < 0:
        raise IOError(_lib.last_error())
    off = np.zeros(max(count, 1), dtype=np.uint64)
    ln = np.zeros(max(count, 1), dtype=np.uint64)
    lib.acimg_tfrecord_index(buf.ctypes.data, n, off.ctypes.data, ln.ctypes.data, count, 0)
    mv = memoryview(buf)
    return [mv[int(off[i]):int(off[i]) + int(ln[i])] for i in range(count)]


def decode_sequence_example_native(record):
    """One serialized SequenceExample -> what `_parse_sequence` (dataloader/outdoor_data_mfcc.py:263-343) returns:
    dict(audio_images float32 [steps,H,W,D] (flipped LR + UD), audio_samples int32 [-1, samples], video_images uint8
    [steps,H,W,D], action, location) via acimg_sequence_example_decode."""
    lib = _lib.load()
    rec = np.frombuffer(record, dtype=np.uint8)
    dims = _lib.SequenceDims()
    _lib.check(lib.acimg_sequence_example_decode(rec.ctypes.data, rec.size, ctypes.byref(dims), None, 0, None, 0,
                                                 None, 0), "sequence_example_decode")
    ai = np.empty((dims.audio_image_steps, dims.audio_height, dims.audio_width, dims.audio_depth), np.float32)
    sa = np.empty(dims.audio_data_values, np.int32)
    vi = np.empty((dims.video_steps, dims.video_height, dims.video_width, dims.video_depth), np.uint8)
    _lib.check(lib.acimg_sequence_example_decode(rec.ctypes.data, rec.size, ctypes.byref(dims),
                                                 ai.ctypes.data if ai.size else None, ai.size,
                                                 sa.ctypes.data if sa.size else None, sa.size,
                                                 vi.ctypes.data if vi.size else None, vi.size),
               "sequence_example_decode")
    if dims.samples > 0:
        sa = sa.reshape(-1, dims.samples)
    return dict(audio_images=ai, audio_samples=sa, video_images=vi, action=int(dims.classes),
                location=int(dims.location), dims=dims)
