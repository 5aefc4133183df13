"""Reader for TensorFlow-1 "tensor bundle" checkpoints (`<prefix>.index` + `<prefix>.data-00000-of-00001`).

The reference saves and restores its policies with `tf.train.Saver` (src/trpo.py:268-270 save, :365 / :205
`U.load_state` restore); its one shipped artifact is `src/checkpoint_tmp/DeepMimic/trpo-walk-0/DeepMimic/trpo-walk-0.*`.
TensorFlow is not part of this stack, so the two files are read directly:

* `.index` is a LevelDB-format sorted string table (one or more prefix-compressed data blocks, an index block, a
  48-byte footer); the value of every key is a `BundleEntryProto` {1: dtype, 2: shape{2: dim{1: size}}, 3: shard,
  4: offset, 5: size, 6: crc32c (fixed32, masked)}.  The empty key holds the `BundleHeaderProto`.
* `.data-*` is the raw little-endian tensor bytes at [offset, offset + size).

Only what that format needs is implemented (no snappy blocks: TF writes the index uncompressed; single shard).
Host-side utility (numpy only); it is not on the device hot path.
"""
import os
import struct

import numpy as np

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 6: np.dtype("i1"),
           9: np.dtype("<i8"), 10: np.dtype("bool")}


def _varint(buf, i):
    r = 0
    s = 0
    while True:
        c = buf[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return r, i


def _block(buf, off, size):
    """Entries of one table block: [(key, value)].  Layout: entries | restart offsets (u32 each) | n_restarts (u32),
    followed in the file by a 1-byte compression type and a 4-byte crc."""
    if buf[off + size] != 0:
        raise ValueError("compressed table block (type %d) is not supported" % buf[off + size])
    blk = buf[off:off + size]
    n_restart = struct.unpack("<I", blk[-4:])[0]
    end = len(blk) - 4 - 4 * n_restart
    out = []
    key = b""
    i = 0
    while i < end:
        shared, i = _varint(blk, i)
        non_shared, i = _varint(blk, i)
        vlen, i = _varint(blk, i)
        key = key[:shared] + blk[i:i + non_shared]
        i += non_shared
        out.append((key, blk[i:i + vlen]))
        i += vlen
    return out


def _fields(buf):
    """Protobuf wire-format fields of a message: [(field_number, wire_type, value)]."""
    i = 0
    out = []
    while i < len(buf):
        tag, i = _varint(buf, i)
        f, w = tag >> 3, tag & 7
        if w == 0:
            v, i = _varint(buf, i)
        elif w == 1:
            v = buf[i:i + 8]; i += 8
        elif w == 2:
            n, i = _varint(buf, i)
            v = buf[i:i + n]; i += n
        elif w == 5:
            v = buf[i:i + 4]; i += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % w)
        out.append((f, w, v))
    return out


_CRC_TABLE = None


def crc32c(data):
    """CRC-32C (Castagnoli), table driven; used to verify tensor bytes against the bundle entry."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for n in range(256):
            c = n
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    t = _CRC_TABLE
    c = 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _unmask(m):
    rot = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


def read_index(prefix):
    """{name: (dtype, shape, shard, offset, size, crc32c or None)} of a checkpoint prefix."""
    buf = open(prefix + ".index", "rb").read()
    if len(buf) < 48 or buf[-8:] != bytes.fromhex("57fb808b247547db"):
        raise ValueError("%s.index is not a table file (bad magic)" % prefix)
    foot = buf[-48:]
    _mo, i = _varint(foot, 0)
    _ms, i = _varint(foot, i)
    io, i = _varint(foot, i)
    isz, i = _varint(foot, i)
    entries = {}
    for _k, handle in _block(buf, io, isz):
        o, j = _varint(handle, 0)
        s, j = _varint(handle, j)
        for key, val in _block(buf, o, s):
            if key == b"":
                continue                     # BundleHeaderProto
            dtype = shard = offset = size = 0
            shape = []
            crc = None
            for f, w, v in _fields(val):
                if f == 1:
                    dtype = v
                elif f == 2:
                    for f2, _w2, v2 in _fields(v):
                        if f2 == 2:
                            d = [x for (f3, _w3, x) in _fields(v2) if f3 == 1]
                            shape.append(d[0] if d else 0)
                elif f == 3:
                    shard = v
                elif f == 4:
                    offset = v
                elif f == 5:
                    size = v
                elif f == 6:
                    crc = _unmask(struct.unpack("<I", v)[0])
            if dtype not in _DTYPES:
                raise ValueError("tensor %r has unsupported dtype enum %d" % (key, dtype))
            entries[key.decode()] = (_DTYPES[dtype], tuple(shape), shard, offset, size, crc)
    return entries


def load_checkpoint(prefix, scope=None, verify=True):
    """Read every tensor of a TF1 checkpoint into {name: ndarray}.  `scope` keeps names under 'scope/' and strips it
    (the reference's variables live under 'pi/' and 'oldpi/', src/trpo.py:128-129)."""
    entries = read_index(prefix)
    data_path = prefix + ".data-00000-of-00001"
    if not os.path.exists(data_path):
        raise FileNotFoundError(data_path)
    data = open(data_path, "rb").read()
    out = {}
    for name, (dt, shape, shard, off, size, crc) in entries.items():
        if shard != 0:
            raise ValueError("multi-shard checkpoints are not supported (%s in shard %d)" % (name, shard))
        if scope is not None:
            if not name.startswith(scope + "/"):
                continue
            short = name[len(scope) + 1:]
        else:
            short = name
        raw = data[off:off + size]
        if len(raw) != size or size != int(np.prod(shape, dtype=np.int64)) * dt.itemsize:
            raise ValueError("tensor %s: bad extent (offset %d size %d shape %s)" % (name, off, size, shape))
        if verify and crc is not None and crc32c(raw) != crc:
            raise ValueError("tensor %s: crc32c mismatch" % name)
        out[short] = np.frombuffer(raw, dtype=dt).reshape(shape).copy()
    return out
