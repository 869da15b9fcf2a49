"""Hand-builds a tiny TensorFlow Saver-V2 checkpoint (tests/golden/bundle_known_answer/model.ckpt.*) byte by byte from
the PUBLISHED formats, independently of acimg/tfio.py (no import of it, own CRC-32C, every field packed explicitly):

  * tensor bundle (tensorflow/core/util/tensor_bundle): `<prefix>.data-00000-of-00001` = the tensors' little-endian
    bytes back to back in key order; `<prefix>.index` = an SSTable whose key "" maps to BundleHeaderProto
    {num_shards = 1 (field 1), endianness = LITTLE (field 2, default 0, omitted), version {producer = 1} (field 3)}
    and whose key <variable name> maps to BundleEntryProto {dtype (1), shape (2: TensorShapeProto{dim (2){size (1)}}),
    shard_id (3, 0 omitted), offset (4, 0 omitted), size (5), crc32c (6, fixed32, MASKED crc of the tensor bytes)};
  * SSTable (LevelDB table format, table/table_builder.cc, table/format.cc): data blocks of prefix-compressed entries
    [varint shared][varint non_shared][varint value_len][key suffix][value] with a restart point every 16 entries,
    then uint32 restart offsets and uint32 restart count; every block followed by a 5-byte trailer [compression type
    0][uint32 masked crc32c(block + type)]; a (here empty) metaindex block; an index block (restart interval 1) with
    one entry per data block: key = a short separator >= the block's last key (LevelDB's FindShortSuccessor for the
    last block: first byte that can be incremented, incremented, rest dropped), value = BlockHandle [varint offset]
    [varint size]; 48-byte footer = metaindex handle, index handle, zero padding to 40 bytes, magic 0xdb4775248b80fb57;
  * masked CRC (lib/hash/crc32c.h): ((crc >> 15) | (crc << 17)) + 0xa282ead8, crc = CRC-32C (Castagnoli, reflected
    polynomial 0x82F63B78).

It stands in for a TF-written file, which cannot be produced here (no TensorFlow): the reader is checked against an
independent statement of the format, NOT against TensorFlow itself — f1 stays unpinned against a TF-written bundle.

    python tests/golden/make_bundle_golden.py
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "bundle_known_answer")


def crc32c(data):
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ 0x82F63B78 if crc & 1 else crc >> 1
    return crc ^ 0xFFFFFFFF


assert crc32c(b"123456789") == 0xE3069283          # the standard CRC-32C check value


def masked(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def tensors():
    """the variables of the known-answer checkpoint, values by formula (the test recomputes them)"""
    t = {
        "resnet_v1_50/conv1/BatchNorm/gamma": (1 + 0.01 * np.arange(64)).astype("<f4"),
        "resnet_v1_50/conv1/BatchNorm/moving_mean": (0.5 - 0.125 * np.arange(64)).astype("<f4"),
        "resnet_v1_50/conv_map/BatchNorm/beta": np.full(12, 7.0, "<f4"),          # init_model must SKIP conv_map
        "resnet_v1_50/logits/biases": np.full(5, -3.0, "<f4"),                     # ... and logits
        "UNetAcRes/final/bias": (np.arange(12) / 16.0).astype("<f4"),
        "UNetAcRes/layer7/conv_2/bias": (np.arange(64) * -0.25).astype("<f4"),
        "UNetAcRes/layer7/conv_2/kernel": (np.arange(3 * 3 * 64 * 64) % 251 / 256.0 - 0.5).astype("<f4").reshape(3, 3, 64, 64),
        "beta1_power": np.array(0.81, "<f4"),
        "global_step": np.array(42, "<i8"),
    }
    return dict(sorted(t.items(), key=lambda kv: kv[0].encode()))


DT = {np.dtype("<f4"): 1, np.dtype("<i8"): 9}


def main():
    os.makedirs(OUT, exist_ok=True)
    tens = tensors()
    # ---- data shard + one BundleEntryProto per tensor
    data = bytearray()
    entries = [(b"", b"\x08\x01" + b"\x1a\x02\x08\x01")]     # header: field 1 varint 1; field 3 len 2 {field 1 varint 1}
    for name, a in tens.items():
        raw = a.tobytes()
        shape = b"".join(b"\x12" + varint(len(b"\x08" + varint(d))) + b"\x08" + varint(d) for d in a.shape)
        e = b"\x08" + varint(DT[a.dtype]) + b"\x12" + varint(len(shape)) + shape
        if len(data):
            e += b"\x20" + varint(len(data))                  # offset (field 4), omitted when 0
        e += b"\x28" + varint(len(raw))                       # size (field 5)
        e += b"\x35" + struct.pack("<I", masked(crc32c(raw)))  # crc32c (field 6, fixed32)
        entries.append((name.encode(), e))
        data += raw
    open(os.path.join(OUT, "model.ckpt.data-00000-of-00001"), "wb").write(bytes(data))

    # ---- one data block: prefix compression, restart every 16 entries
    block, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % 16 == 0:
            restarts.append(len(block))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        block += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts:
        block += struct.pack("<I", r)
    block += struct.pack("<I", len(restarts))

    f = bytearray()

    def emit(b):
        off = len(f)
        f.extend(b)
        f.append(0)
        f.extend(struct.pack("<I", masked(crc32c(bytes(b) + b"\x00"))))
        return varint(off) + varint(len(b))

    h_data = emit(block)
    h_meta = emit(struct.pack("<I", 0) + struct.pack("<I", 1))           # empty block: one restart at 0
    last = entries[-1][0]                                                  # b"resnet_v1_50/logits/biases"
    succ = bytes([last[0] + 1])                                            # FindShortSuccessor: b"s"
    idx = varint(0) + varint(len(succ)) + varint(len(h_data)) + succ + h_data + struct.pack("<I", 0) + struct.pack("<I", 1)
    h_idx = emit(idx)
    foot = h_meta + h_idx
    foot += b"\x00" * (40 - len(foot)) + struct.pack("<Q", 0xdb4775248b80fb57)
    f.extend(foot)
    open(os.path.join(OUT, "model.ckpt.index"), "wb").write(bytes(f))
    print("wrote", OUT, len(f), "index bytes,", len(data), "data bytes")


if __name__ == "__main__":
    main()
