"""Reader and writer for TensorFlow-1 "tensor bundle" checkpoints (`<prefix>.index` + `<prefix>.data-00000-of-00001`).

The reference saves and restores its policies with `tf.train.Saver` (src/trpo.py:268-270 save, :365 / :205
`U.load_state` restore); its one shipped artifact is `src/checkpoint_tmp/DeepMimic/trpo-walk-0/DeepMimic/trpo-walk-0.*`.
TensorFlow is not part of this stack, so the two files are read directly:

* `.index` is a LevelDB-format sorted string table (one or more prefix-compressed data blocks, an index block, a
  48-byte footer); the value of every key is a `BundleEntryProto` {1: dtype, 2: shape{2: dim{1: size}}, 3: shard,
  4: offset, 5: size, 6: crc32c (fixed32, masked)}.  The empty key holds the `BundleHeaderProto`.
* `.data-*` is the raw little-endian tensor bytes at [offset, offset + size).

Only what that format needs is implemented (no snappy blocks: TF writes the index uncompressed; single shard).
`save_checkpoint` writes the same two files, so that a policy trained here loads through the reference's
`U.load_state` / `tf.train.Saver.restore` (src/utils/tf_util.py:314-319, src/trpo.py:367): re-saving the tensors of the shipped
checkpoint reproduces both of its files byte for byte (tests/test_policy.py).
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


# ---- writer ------------------------------------------------------------------------------------------------------------------
_DTYPE_ENUM = {np.dtype("<f4"): 1, np.dtype("<f8"): 2, np.dtype("<i4"): 3, np.dtype("u1"): 4, np.dtype("i1"): 6, np.dtype("<i8"): 9, np.dtype("bool"): 10}
_BLOCK_SIZE = 4096            # table::Options::block_size of the bundle writer's index table
_RESTART_INTERVAL = 16
_MAGIC = bytes.fromhex("57fb808b247547db")


def _put_varint(v):
    out = bytearray()
    v = int(v)
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80); v >>= 7
    out.append(v)
    return bytes(out)


def _mask(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


class _BlockBuilder(object):
    """leveldb/TF table block: prefix-compressed entries, a restart point every 16 entries."""

    def __init__(self):
        self.buf = bytearray(); self.restarts = [0]; self.count = 0; self.last = b""

    def add(self, key, value):
        shared = 0
        if self.count < _RESTART_INTERVAL:
            m = min(len(self.last), len(key))
            while shared < m and self.last[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf)); self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last = key; self.count += 1

    def size_estimate(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _shortest_separator(a, b):
    """leveldb BytewiseComparator::FindShortestSeparator(a, b): a short key k with a <= k < b."""
    n = min(len(a), len(b)); i = 0
    while i < n and a[i] == b[i]:
        i += 1
    if i < n and a[i] < 0xFF and a[i] + 1 < b[i]:
        return a[:i] + bytes([a[i] + 1])
    return a


def _short_successor(a):
    """leveldb BytewiseComparator::FindShortSuccessor(a): a short key >= a."""
    for i, c in enumerate(a):
        if c != 0xFF:
            return a[:i] + bytes([c + 1])
    return a


def _entry_proto(dtype_enum, shape, offset, size, crc):
    out = bytearray(b"\x08" + _put_varint(dtype_enum))
    dims = b"".join(b"\x12" + _put_varint(len(b"\x08" + _put_varint(d))) + b"\x08" + _put_varint(d) for d in shape)
    out += b"\x12" + _put_varint(len(dims)) + dims
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size)
    out += b"\x35" + struct.pack("<I", _mask(crc))
    return bytes(out)


def save_checkpoint(prefix, tensors):
    """Write {name: ndarray} as a single-shard TF1 tensor bundle: `<prefix>.data-00000-of-00001` (raw little-endian bytes in sorted
    key order) and `<prefix>.index` (uncompressed table: BundleHeaderProto under the empty key, one BundleEntryProto per tensor),
    plus the `checkpoint` state file `tf.train.latest_checkpoint` reads.  float32 / float64 / int32 / int64 / uint8 / int8 / bool."""
    names = sorted(tensors, key=lambda k: k.encode())
    data = bytearray()
    entries = [(b"", b"\x08\x01\x1a\x02\x08\x01")]                      # BundleHeaderProto{num_shards: 1, version{producer: 1}} (little endian = default)
    for name in names:
        a = np.asarray(tensors[name])
        dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
        dt = np.dtype(dt.str.replace("=", "<")) if dt.itemsize > 1 else dt
        if np.dtype(dt) not in _DTYPE_ENUM:
            raise ValueError("tensor %r: dtype %s cannot be stored" % (name, a.dtype))
        raw = np.ascontiguousarray(a, dtype=dt).tobytes()
        entries.append((name.encode(), _entry_proto(_DTYPE_ENUM[np.dtype(dt)], a.shape, len(data), len(raw), crc32c(raw))))
        data += raw
    out = bytearray()

    def emit(block_bytes):
        off = len(out)
        out.extend(block_bytes + b"\x00" + struct.pack("<I", _mask(crc32c(block_bytes + b"\x00"))))
        return _put_varint(off) + _put_varint(len(block_bytes))

    index = _BlockBuilder()
    blk = _BlockBuilder(); pending = None                                   # (last key of the finished block, its handle)
    for key, val in entries:
        if pending is not None:
            index.add(_shortest_separator(pending[0], key), pending[1]); pending = None
        blk.add(key, val)
        if blk.size_estimate() >= _BLOCK_SIZE:
            pending = (key, emit(blk.finish())); blk = _BlockBuilder()
    if blk.buf:
        pending = (blk.last, emit(blk.finish()))
    if pending is not None:
        index.add(_short_successor(pending[0]), pending[1])
    meta_handle = emit(_BlockBuilder().finish())
    index_handle = emit(index.finish())
    foot = meta_handle + index_handle
    out.extend(foot + bytes(40 - len(foot)) + _MAGIC)
    d = os.path.dirname(os.path.abspath(prefix))
    os.makedirs(d, exist_ok=True)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    base = os.path.basename(prefix)
    with open(os.path.join(d, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
    return prefix
