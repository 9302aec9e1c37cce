"""TensorBoard event files for the training scalars (the summaries half): what
`tf.summary.create_file_writer(logdir / 'train')` + `tf.summary.scalar(name, value, step)` leave on disk in the
reference (`composer/models/transformer.py:903, 933-951`), written without TensorFlow.

Formats, from TensorFlow's public definitions (tensorflow/core/lib/io/record_writer.cc, core/util/event.proto,
core/framework/summary.proto, core/framework/tensor.proto); TensorFlow / TensorBoard are not in this image, so the
files are checked by this module's own reader, CRC-32C test vectors and hand-assembled records -- not by TensorBoard:

  file      events.out.tfevents.<unix time>.<hostname>[.<pid>.<n>].v2
  record    uint64 length | uint32 masked_crc32c(length) | bytes data | uint32 masked_crc32c(data)      (little endian)
            masked(c) = ((c >> 15 | c << 17) + 0xa282ead8) mod 2^32,  CRC-32C = Castagnoli, reflected 0x82F63B78
  Event     1: double wall_time   2: int64 step   3: string file_version ("brain.Event:2", first record)   5: Summary
  Summary   1: repeated Value { 1: string tag   9: SummaryMetadata { 1: PluginData { 1: string plugin_name = "scalars" } }
                                 8: TensorProto { 1: dtype = DT_FLOAT (1)   2: TensorShapeProto {} (scalar)   5: float_val } }
            (the TF2 form of tf.summary.scalar; the TF1 form `2: float simple_value` is accepted by the reader as well)
"""
import os
import socket
import struct
import time

_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire format (just what the two messages need) ---------------------------------------
def _varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _field(num: int, wire: int) -> bytes:
    return _varint((num << 3) | wire)


def _ld(num: int, payload: bytes) -> bytes:          # length-delimited
    return _field(num, 2) + _varint(len(payload)) + payload


def _scalar_summary(tag: str, value: float) -> bytes:
    tensor = _field(1, 0) + _varint(1) + _ld(2, b'') + _ld(5, struct.pack('<f', value))      # DT_FLOAT, scalar shape, packed float_val
    meta = _ld(1, _ld(1, b'scalars'))
    val = _ld(1, tag.encode()) + _ld(8, tensor) + _ld(9, meta)          # fields in number order, as protobuf serialisers emit them
    return _ld(1, val)


def _event(wall_time: float, step: int = None, file_version: str = None, summary: bytes = None) -> bytes:
    out = _field(1, 1) + struct.pack('<d', wall_time)
    if step is not None:
        out += _field(2, 0) + _varint(step)
    if file_version is not None:
        out += _ld(3, file_version.encode())
    if summary is not None:
        out += _ld(5, summary)
    return out


def _record(data: bytes) -> bytes:
    head = struct.pack('<Q', len(data))
    return head + struct.pack('<I', masked_crc32c(head)) + data + struct.pack('<I', masked_crc32c(data))


class EventFileWriter:
    """`with summary_log.as_default(): tf.summary.scalar(name, value, step=...)` -> writer.scalar(name, value, step)."""

    _count = 0

    def __init__(self, directory, filename_suffix='.v2'):
        os.makedirs(directory, exist_ok=True)
        now = time.time()
        EventFileWriter._count += 1               # TensorFlow's per-process writer uid: two writers opened within one second stay apart
        self.path = os.path.join(str(directory), 'events.out.tfevents.%010d.%s.%d.%d%s' % (
            int(now), socket.gethostname(), os.getpid(), EventFileWriter._count - 1, filename_suffix))
        self._f = open(self.path, 'ab')
        self._f.write(_record(_event(now, step=0, file_version='brain.Event:2')))
        self._f.flush()

    def scalar(self, name, value, step):
        self._f.write(_record(_event(time.time(), step=int(step), summary=_scalar_summary(name, float(value)))))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


# ---- reader (tests; also lets a run's curve be printed without TensorBoard) -----------------------
def _read_varint(buf, pos):
    v, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return v, pos


def _fields(buf):
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        num, wire = key >> 3, key & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wire == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        else:
            raise ValueError('unsupported wire type %d' % wire)
        yield num, wire, v


def read_records(path):
    """Yields the payload of every TFRecord of the file, checking both CRCs."""
    with open(path, 'rb') as f:
        blob = f.read()
    pos = 0
    while pos < len(blob):
        head = blob[pos:pos + 8]
        (n,) = struct.unpack('<Q', head)
        if struct.unpack('<I', blob[pos + 8:pos + 12])[0] != masked_crc32c(head):
            raise ValueError('corrupt record length at byte %d' % pos)
        data = blob[pos + 12:pos + 12 + n]
        if struct.unpack('<I', blob[pos + 12 + n:pos + 16 + n])[0] != masked_crc32c(data):
            raise ValueError('corrupt record data at byte %d' % pos)
        pos += 16 + n
        yield data


def read_scalars(path):
    """-> (file_version, [(tag, step, value, wall_time), ...]) for both the TF2 tensor form and the TF1 simple_value form."""
    version, out = None, []
    for rec in read_records(path):
        wall, step, summary = 0.0, 0, None
        for num, wire, v in _fields(rec):
            if num == 1 and wire == 1:
                (wall,) = struct.unpack('<d', v)
            elif num == 2 and wire == 0:
                step = v
            elif num == 3 and wire == 2:
                version = v.decode()
            elif num == 5 and wire == 2:
                summary = v
        if summary is None:
            continue
        for num, wire, val in _fields(summary):
            if num != 1:
                continue
            tag, value = None, None
            for n2, w2, v2 in _fields(val):
                if n2 == 1:
                    tag = v2.decode()
                elif n2 == 2 and w2 == 5:
                    (value,) = struct.unpack('<f', v2)
                elif n2 == 8:
                    for n3, w3, v3 in _fields(v2):
                        if n3 == 5:
                            (value,) = struct.unpack('<f', v3[:4])
            out.append((tag, step, value, wall))
    return version, out
