"""TensorBoard event files (`events.out.tfevents.*`) without TensorFlow / tensorboard: the scalar subset the reference's
trainer writes through `torch.utils.tensorboard.SummaryWriter.add_scalar` (scripts/train.py:22-23, :85-94: tags
'Train/avg_reward' and 'Train/loss') and that its plotting reads back from data/torch_logs/.

Format (TFRecord framing + two protobuf messages, hand-encoded):
    record  = uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data)
    Event   = { 1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary }
    Summary = { 1: repeated Value { 1: string tag, 2: float simple_value } }
The first record of a file is Event{wall_time, file_version = "brain.Event:2"}.  `read_scalars` parses the same subset
(the reference's shipped logs are the compatibility check: tests/test_formats.py)."""
import os
import socket
import struct
import time

_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf, off):
    n = shift = 0
    while True:
        b = buf[off]
        off += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return n, off


def _len_delim(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_scalar_event(tag, value, step, wall_time):
    val = _len_delim(1, tag.encode()) + bytes([0x15]) + struct.pack("<f", float(value))  # Value{tag, simple_value}
    summary = _len_delim(1, val)
    return bytes([0x09]) + struct.pack("<d", wall_time) + bytes([0x10]) + _varint(int(step)) + _len_delim(5, summary)


def encode_version_event(wall_time):
    return bytes([0x09]) + struct.pack("<d", wall_time) + _len_delim(3, b"brain.Event:2")


def _record(data):
    head = struct.pack("<Q", len(data))
    return head + struct.pack("<I", masked_crc(head)) + data + struct.pack("<I", masked_crc(data))


class SummaryWriter(object):
    """The slice of torch.utils.tensorboard.SummaryWriter the reference uses: SummaryWriter(log_dir=...), add_scalar,
    flush, close.  One file per writer, named like tensorboard names it."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        now = time.time()
        self.path = os.path.join(log_dir, "events.out.tfevents.%010d.%s.%d.0" % (int(now), socket.gethostname(), os.getpid()))
        self._f = open(self.path, "wb")
        self._f.write(_record(encode_version_event(now)))

    def add_scalar(self, tag, scalar_value, global_step=None, walltime=None):
        self._f.write(_record(encode_scalar_event(tag, scalar_value, 0 if global_step is None else int(global_step),
                                                  time.time() if walltime is None else walltime)))

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_records(path, check_crc=True):
    with open(path, "rb") as f:
        buf = f.read()
    off = 0
    while off + 12 <= len(buf):
        (n,) = struct.unpack_from("<Q", buf, off)
        (c1,) = struct.unpack_from("<I", buf, off + 8)
        data = buf[off + 12:off + 12 + n]
        (c2,) = struct.unpack_from("<I", buf, off + 12 + n)
        if check_crc and (c1 != masked_crc(buf[off:off + 8]) or c2 != masked_crc(data)):
            raise ValueError("corrupt record at byte %d of %s" % (off, path))
        yield data
        off += 16 + n


def _parse_fields(buf):
    off = 0
    while off < len(buf):
        key, off = _read_varint(buf, off)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, off = _read_varint(buf, off)
        elif wt == 1:
            v, off = buf[off:off + 8], off + 8
        elif wt == 2:
            n, off = _read_varint(buf, off)
            v, off = buf[off:off + n], off + n
        elif wt == 5:
            v, off = buf[off:off + 4], off + 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield field, wt, v


def read_scalars(path, check_crc=True):
    """[(wall_time, step, tag, value)] of every simple_value scalar in an event file (other records are skipped)."""
    out = []
    for data in read_records(path, check_crc):
        wall, step, summary = 0.0, 0, None
        for field, wt, v in _parse_fields(data):
            if field == 1 and wt == 1:
                (wall,) = struct.unpack("<d", v)
            elif field == 2 and wt == 0:
                step = v
            elif field == 5 and wt == 2:
                summary = v
        if summary is None:
            continue
        for field, wt, val in _parse_fields(summary):
            if field != 1 or wt != 2:
                continue
            tag, sv = None, None
            for f2, w2, v2 in _parse_fields(val):
                if f2 == 1 and w2 == 2:
                    tag = v2.decode()
                elif f2 == 2 and w2 == 5:
                    (sv,) = struct.unpack("<f", v2)
            if tag is not None and sv is not None:
                out.append((wall, step, tag, sv))
    return out
