"""TFRecord datasets (the reader half): what `composer export-dataset` writes
(`composer/cli.py:346-380`) and `load_tfrecord_dataset` reads (`composer/models/__init__.py:315-374`), without TensorFlow.

File = TFRecord framing (see composer_amd/tbevents.py) around serialized `tf.train.Example` messages:
  record 0   {'model_type': bytes 'transformer', 'batch_size': int64, 'window_size': int64}
  record k   {'x': bytes, 'y': bytes}, each the `tf.io.serialize_tensor` of an int32 [batch_size, window_size] batch

Wire formats, from TensorFlow's public definitions (core/example/example.proto, feature.proto, core/framework/tensor.proto,
tensor_shape.proto, types.proto):
  Example   1: Features { 1: map<string, Feature> = repeated { 1: string key   2: Feature value } }
  Feature   oneof { 1: BytesList { 1: repeated bytes }   2: FloatList { 1: packed float }   3: Int64List { 1: packed int64 } }
  TensorProto  1: dtype (DT_INT32 = 3)   2: TensorShapeProto { 2: repeated Dim { 1: int64 size } }   4: bytes tensor_content
               (row-major little-endian; the reader also takes the `7: int_val` form that small tensors may use)
TensorFlow is not in this image: checked by hand-assembled messages and round trips (tests/test_tfrecord.py), not against
files TensorFlow wrote.
"""
import struct

import numpy as np

from .tbevents import _field, _fields, _ld, _read_varint, _record, _varint, read_records

DT_INT32 = 3


# ---- messages -------------------------------------------------------------------------------------
def bytes_feature(value: bytes) -> bytes:
    """io_utils.py:8-15"""
    return _ld(1, _ld(1, bytes(value)))


def int64_feature(value: int) -> bytes:
    """io_utils.py:26-33 (Int64List.value is packed in proto3)"""
    return _ld(3, _ld(1, _varint(int(value))))


def example(features: dict) -> bytes:
    """tf.train.Example(features=tf.train.Features(feature=...)).SerializeToString(); map entries in key order."""
    entries = b''.join(_ld(1, _ld(1, k.encode()) + _ld(2, features[k])) for k in sorted(features))
    return _ld(1, entries)


def serialize_tensor(a) -> bytes:
    """tf.io.serialize_tensor of an int32 array."""
    a = np.ascontiguousarray(a, dtype='<i4')
    shape = b''.join(_ld(2, _field(1, 0) + _varint(d)) for d in a.shape)
    return _field(1, 0) + _varint(DT_INT32) + _ld(2, shape) + _ld(4, a.tobytes())


def parse_example(buf: bytes) -> dict:
    """-> {key: list of bytes | list of int | list of float}"""
    out = {}
    for num, _, features in _fields(buf):
        if num != 1:
            continue
        for n1, _, entry in _fields(features):
            if n1 != 1:
                continue
            key, feat = None, b''
            for n2, _, v in _fields(entry):
                if n2 == 1:
                    key = v.decode()
                elif n2 == 2:
                    feat = v
            vals = []
            for kind, _, lst in _fields(feat):
                for n3, wire, v in _fields(lst):
                    if n3 != 1:
                        continue
                    if kind == 1:
                        vals.append(bytes(v))
                    elif kind == 3:
                        if wire == 2:                          # packed
                            pos = 0
                            while pos < len(v):
                                x, pos = _read_varint(v, pos)
                                vals.append(x - (1 << 64) if x >> 63 else x)
                        else:
                            vals.append(v - (1 << 64) if v >> 63 else v)
                    elif kind == 2:
                        if wire == 2:
                            vals.extend(struct.unpack('<%df' % (len(v) // 4), v))
                        else:
                            vals.append(struct.unpack('<f', v)[0])
            out[key] = vals
    return out


def parse_tensor(buf: bytes) -> np.ndarray:
    """tf.io.parse_tensor(buf, tf.int32)"""
    dtype, shape, content, int_val = None, [], None, []
    for num, wire, v in _fields(buf):
        if num == 1:
            dtype = v
        elif num == 2:
            for n1, _, dim in _fields(v):
                if n1 == 2:
                    size = 0
                    for n2, _, s in _fields(dim):
                        if n2 == 1:
                            size = s
                    shape.append(size)
        elif num == 4:
            content = v
        elif num == 7:
            if wire == 2:
                pos = 0
                while pos < len(v):
                    x, pos = _read_varint(v, pos)
                    int_val.append(x)
            else:
                int_val.append(v)
    if dtype != DT_INT32:
        raise ValueError('expected a DT_INT32 tensor, found dtype %r' % dtype)
    n = int(np.prod(shape)) if shape else 1
    if content is not None:
        a = np.frombuffer(content, '<i4')
    else:
        a = np.array([x - (1 << 64) if x >> 63 else x for x in int_val], np.int64).astype(np.int32)
        if len(a) == 1 and n > 1:
            a = np.full(n, a[0], np.int32)                      # a single int_val stands for a constant tensor
    if len(a) != n:
        raise ValueError('tensor has %d values for shape %s' % (len(a), shape))
    return a.reshape(shape).astype(np.int32)


# ---- export / load --------------------------------------------------------------------------------
def export_dataset(batches, output_path, model_type='transformer'):
    """cli.py:362-378: header record from the first batch's shape, then one record per (x, y) batch.  Returns the count."""
    n = 0
    with open(output_path, 'wb') as f:
        for x, y in batches:
            if n == 0:
                B, W = x.shape
                f.write(_record(example({'model_type': bytes_feature(model_type.encode('utf-8')),
                                         'batch_size': int64_feature(B), 'window_size': int64_feature(W)})))
            f.write(_record(example({'x': bytes_feature(serialize_tensor(x)), 'y': bytes_feature(serialize_tensor(y))})))
            n += 1
    if n == 0:
        raise ValueError('the dataset has no batches to export')
    return n


class RecordBatchDataset:
    """Re-iterable (x, y) batches read from a TFRecord file; already batched and windowed by the exporting config
    (models/__init__.py:319-321), cached in memory (:363), shuffled per pass with a 500*batch_size element buffer (:365-368).
    Data parallelism: rank r of `world_size` takes batches r, r+world_size, ... of the pass order (remainder dropped)."""

    def __init__(self, xs, ys, shuffle=True, seed=0, rank=0, world_size=1):
        self.xs, self.ys = xs, ys
        self.B, self.W = (xs.shape[1], xs.shape[2]) if len(xs) else (0, 0)
        self.shuffle, self.seed, self.rank, self.world = shuffle, int(seed), int(rank), int(world_size)
        self._epoch = 0

    def __len__(self):
        return len(self.xs) // self.world

    def _order(self):
        n = len(self.xs)
        if not self.shuffle:
            return np.arange(n)
        rng = np.random.default_rng([self.seed, self._epoch])
        cap = 500 * self.B
        if cap >= n:
            return rng.permutation(n)
        buf, out, nxt = list(range(cap)), np.empty(n, np.int64), cap
        for i in range(n):
            j = int(rng.integers(0, len(buf)))
            out[i] = buf[j]
            if nxt < n:
                buf[j] = nxt
                nxt += 1
            else:
                buf[j] = buf[-1]
                buf.pop()
        return out

    def __iter__(self):
        order = self._order()
        self._epoch += 1
        for b in range(len(order) // self.world):
            i = order[b * self.world + self.rank]
            yield self.xs[i].copy(), self.ys[i].copy()


def load_tfrecord_dataset(filepath, shuffle=True, seed=0, rank=0, world_size=1):
    """models/__init__.py:315-374 -> (dataset, header dict with model_type / batch_size / window_size)."""
    records = read_records(str(filepath))
    try:
        head = parse_example(next(records))
    except StopIteration:
        raise ValueError('%s is empty' % filepath)
    for k in ('model_type', 'batch_size', 'window_size'):
        if len(head.get(k, [])) != 1:
            raise ValueError('%s: the first record has no %r feature (not written by export-dataset?)' % (filepath, k))
    header = {'model_type': head['model_type'][0].decode('utf-8'), 'batch_size': int(head['batch_size'][0]),
              'window_size': int(head['window_size'][0])}
    B, W = header['batch_size'], header['window_size']
    xs, ys = [], []
    for rec in records:
        ex = parse_example(rec)
        xs.append(parse_tensor(ex['x'][0]).reshape(B, W))         # tf.reshape(..., target_shape), :360-361
        ys.append(parse_tensor(ex['y'][0]).reshape(B, W))
    xs = np.stack(xs) if xs else np.zeros((0, B, W), np.int32)
    ys = np.stack(ys) if ys else np.zeros((0, B, W), np.int32)
    return RecordBatchDataset(xs, ys, shuffle, seed, rank, world_size), header
