"""TensorBundle checkpoints: the `ckpt-N.index` + `ckpt-N.data-00000-of-00001` pair that
`tf.train.Checkpoint(step, epoch, optimizer, model)` / `CheckpointManager` write in the reference
(`composer/models/transformer.py:890-891, 941-943, 953-955`; restored at :896 and `models/__init__.py:75-80`), without
TensorFlow.

Formats, restated from TensorFlow's public sources (tensorflow/core/util/tensor_bundle/tensor_bundle.cc,
core/protobuf/tensor_bundle.proto, core/lib/io/{table_builder,block_builder,format}.cc = the LevelDB table format,
core/protobuf/trackable_object_graph.proto):

  <prefix>.data-00000-of-00001   the tensors' bytes back to back (row-major, little endian), in key order.
        DT_STRING tensors: [varint64 length of every element][uint32 masked crc32c over the lengths, each taken as a
        little-endian uint32 (uint64 when >= 2^32)][the elements' bytes]
  <prefix>.index                 a LevelDB-format table, sorted keys:
        ""            -> BundleHeaderProto { 1: num_shards = 1   2: endianness = LITTLE (0)   3: VersionDef { 1: producer = 1 } }
        <tensor key>  -> BundleEntryProto  { 1: dtype   2: TensorShapeProto   3: shard_id   4: offset   5: size
                                             6: fixed32 masked crc32c of the tensor's bytes }
     table = data blocks | metaindex block | index block | 48-byte footer
        block   = entries { varint shared, varint non_shared, varint value_len, key suffix, value } ... restart offsets
                  (uint32 each, one every 16 entries) | uint32 restart count;  then a 5-byte trailer { type 0 = raw,
                  uint32 masked crc32c(block | type) }
        index   = one entry per data block: key >= the block's last key, value = BlockHandle {varint offset, varint size}
        footer  = metaindex BlockHandle | index BlockHandle | zero padding to 40 bytes | magic 0xdb4775248b80fb57 (LE)
  object-based (TF2) checkpoints name a variable `<attribute path>/.ATTRIBUTES/VARIABLE_VALUE`, an optimizer slot
  `<variable path>/.OPTIMIZER_SLOT/<optimizer path>/<slot>/.ATTRIBUTES/VARIABLE_VALUE`, and store the serialized
  TrackableObjectGraph under `_CHECKPOINTABLE_OBJECT_GRAPH` (a scalar DT_STRING tensor).

**Parity unpinned**: neither TensorFlow nor a checkpoint it wrote exists in this image or in the reference repository,
so these files are checked by this module's own reader, by blocks / protos assembled by hand in
tests/test_tensorbundle.py and by the CRC-32C vectors -- not by restoring them in TensorFlow.  The key names follow
the attribute names of the reference's Keras model ("Checkpoint / resume").
"""
import struct

import numpy as np

from .tbevents import _field, _fields, _ld, _read_varint, _varint, crc32c, masked_crc32c

TABLE_MAGIC = 0xdb4775248b80fb57
BLOCK_SIZE = 262144                 # table::Options::block_size
RESTART_INTERVAL = 16               # table::Options::block_restart_interval
OBJECT_GRAPH_KEY = '_CHECKPOINTABLE_OBJECT_GRAPH'
VALUE_SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'
SLOT_MARK = '/.OPTIMIZER_SLOT/'

DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_INT16, DT_INT8, DT_STRING, DT_INT64, DT_BOOL = 1, 2, 3, 4, 5, 6, 7, 9, 10
DT_BFLOAT16, DT_UINT16, DT_HALF, DT_UINT32, DT_UINT64 = 14, 17, 19, 22, 23
_NP_OF = {DT_FLOAT: '<f4', DT_DOUBLE: '<f8', DT_INT32: '<i4', DT_UINT8: 'u1', DT_INT16: '<i2', DT_INT8: 'i1', DT_INT64: '<i8',
          DT_BOOL: '?', DT_UINT16: '<u2', DT_HALF: '<f2', DT_UINT32: '<u4', DT_UINT64: '<u8', DT_BFLOAT16: '<u2'}
_DT_OF = {np.dtype(v).str: k for k, v in _NP_OF.items() if k != DT_BFLOAT16}


def _mask(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------- table (index file)
class _BlockBuilder:
    def __init__(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b''

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count % RESTART_INTERVAL == 0:
            if self.count:
                self.restarts.append(len(self.buf))
        else:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self) -> bytes:
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _handle(offset, size):
    return _varint(offset) + _varint(size)


def write_table(path, items):
    """items: (key bytes, value bytes) in strictly increasing key order."""
    out = bytearray()

    def emit(block: bytes):
        off = len(out)
        out.extend(block + b'\x00' + struct.pack('<I', _mask(crc32c(block + b'\x00'))))
        return off, len(block)

    index, blk, prev = _BlockBuilder(), _BlockBuilder(), None
    for key, value in items:
        if prev is not None and key <= prev:
            raise ValueError('table keys must be strictly increasing: %r after %r' % (key, prev))
        blk.add(key, value)
        prev = key
        if blk.size() >= BLOCK_SIZE:
            index.add(blk.last, _handle(*emit(blk.finish())))
            blk = _BlockBuilder()
    if blk.count or not index.count:
        index.add(blk.last, _handle(*emit(blk.finish())))
    meta = emit(_BlockBuilder().finish())
    idx = emit(index.finish())
    footer = _handle(*meta) + _handle(*idx)
    out.extend(footer + bytes(40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))
    with open(path, 'wb') as f:
        f.write(bytes(out))


def _read_block(blob, offset, size):
    body, trailer = blob[offset:offset + size], blob[offset + size:offset + size + 5]
    if len(trailer) != 5:
        raise ValueError('truncated table block at byte %d' % offset)
    if trailer[0] != 0:
        raise ValueError('compressed table blocks (type %d) are not supported' % trailer[0])
    if struct.unpack('<I', trailer[1:])[0] != _mask(crc32c(body + trailer[:1])):
        raise ValueError('table block at byte %d fails its checksum' % offset)
    (nrestart,) = struct.unpack('<I', body[-4:])
    end, pos, key, out = len(body) - 4 - 4 * nrestart, 0, b'', []
    while pos < end:
        shared, pos = _read_varint(body, pos)
        non_shared, pos = _read_varint(body, pos)
        vlen, pos = _read_varint(body, pos)
        key = key[:shared] + body[pos:pos + non_shared]
        pos += non_shared
        out.append((key, body[pos:pos + vlen]))
        pos += vlen
    return out


def read_table(path):
    """-> list of (key bytes, value bytes) in file order, every block checksum verified."""
    with open(path, 'rb') as f:
        blob = f.read()
    if len(blob) < 48 or struct.unpack('<Q', blob[-8:])[0] != TABLE_MAGIC:
        raise ValueError('%s is not a TensorBundle index (no table footer)' % path)
    footer = blob[-48:]
    pos = 0
    _, pos = _read_varint(footer, pos)
    _, pos = _read_varint(footer, pos)
    ioff, pos = _read_varint(footer, pos)
    isize, pos = _read_varint(footer, pos)
    out = []
    for _, h in _read_block(blob, ioff, isize):
        off, p = _read_varint(h, 0)
        size, _ = _read_varint(h, p)
        out.extend(_read_block(blob, off, size))
    return out


# ---------------------------------------------------------------------------------------------- bundle
def _shape_proto(shape):
    return b''.join(_ld(2, _field(1, 0) + _varint(int(d))) for d in shape)


def _encode_strings(elems):
    lengths, c = b'', 0
    for e in elems:
        lengths += _varint(len(e))
        c = _crc_extend(c, struct.pack('<I', len(e)) if len(e) < (1 << 32) else struct.pack('<Q', len(e)))
    cks = struct.pack('<I', _mask(c))
    c = _crc_extend(c, cks)
    for e in elems:
        c = _crc_extend(c, e)
    return lengths + cks + b''.join(elems), c


def _crc_extend(c, data):
    """crc32c::Extend: the CRC of (bytes so far | data), from the CRC of the bytes so far."""
    from .tbevents import _TABLE
    c ^= 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def write_bundle(prefix, tensors):
    """tensors: {key: numpy array | bytes | list of bytes (a DT_STRING tensor)}.  bfloat16 arrays are passed as
    (uint16 array, 'bfloat16') tuples.  Writes <prefix>.index and <prefix>.data-00000-of-00001."""
    keys = sorted(tensors, key=lambda k: k.encode())
    items = [(b'', _field(1, 0) + _varint(1) + _ld(3, _field(1, 0) + _varint(1)))]         # num_shards 1, LITTLE, producer 1
    offset = 0
    with open(str(prefix) + '.data-00000-of-00001', 'wb') as data:
        for k in keys:
            if not k:
                raise ValueError('the empty key is reserved for the bundle header')
            v = tensors[k]
            if isinstance(v, (bytes, bytearray)) or (isinstance(v, list) and all(isinstance(e, (bytes, bytearray)) for e in v)):
                elems = [bytes(v)] if isinstance(v, (bytes, bytearray)) else [bytes(e) for e in v]
                shape = [] if isinstance(v, (bytes, bytearray)) else [len(elems)]
                payload, c = _encode_strings(elems)
                dtype = DT_STRING
            else:
                if isinstance(v, tuple) and v[1] == 'bfloat16':
                    a, dtype = np.asarray(v[0], dtype='<u2', order='C'), DT_BFLOAT16
                else:
                    a = np.asarray(v)
                    a = np.asarray(a, dtype=a.dtype.newbyteorder('<') if a.dtype.byteorder == '>' else a.dtype, order='C')   # (ascontiguousarray would make scalars 1-d)
                    if a.dtype.str not in _DT_OF:
                        raise TypeError('%s: dtype %s has no TensorFlow counterpart here' % (k, a.dtype))
                    dtype = _DT_OF[a.dtype.str]
                shape, payload = a.shape, a.tobytes()
                c = crc32c(payload)
            entry = _field(1, 0) + _varint(dtype) + _ld(2, _shape_proto(shape))
            if offset:
                entry += _field(4, 0) + _varint(offset)
            entry += _field(5, 0) + _varint(len(payload)) + _field(6, 5) + struct.pack('<I', _mask(c))
            items.append((k.encode(), entry))
            data.write(payload)
            offset += len(payload)
    write_table(str(prefix) + '.index', items)


def read_bundle(prefix, verify=True):
    """-> {key: numpy array | bytes (scalar string) | list of bytes}; bfloat16 tensors come back as float32."""
    items = read_table(str(prefix) + '.index')
    if not items or items[0][0] != b'':
        raise ValueError('%s.index has no bundle header' % prefix)
    shards, endian = 0, 0
    for num, _, v in _fields(items[0][1]):
        if num == 1:
            shards = v
        elif num == 2:
            endian = v
    if shards != 1 or endian != 0:
        raise ValueError('only single-shard little-endian bundles are supported (num_shards=%d, endianness=%d)' % (shards, endian))
    with open(str(prefix) + '.data-00000-of-00001', 'rb') as f:
        blob = f.read()
    out = {}
    for key, entry in items[1:]:
        dtype, shape, offset, size, want = 0, [], 0, 0, None
        for num, wire, v in _fields(entry):
            if num == 1:
                dtype = v
            elif num == 2:
                for n1, _, dim in _fields(v):
                    if n1 == 2:
                        shape.append(next((s for n2, _, s in _fields(dim) if n2 == 1), 0))
            elif num == 3 and v != 0:
                raise ValueError('%s: shard %d of a single-shard bundle' % (key, v))
            elif num == 4:
                offset = v
            elif num == 5:
                size = v
            elif num == 6:
                (want,) = struct.unpack('<I', v)
            elif num == 7:
                raise ValueError('%s: sliced (partitioned) variables are not supported' % key.decode())
        raw = blob[offset:offset + size]
        if len(raw) != size:
            raise ValueError('%s: data file too short' % key.decode())
        name = key.decode()
        if dtype == DT_STRING:
            n = int(np.prod(shape)) if shape else 1
            pos, lens = 0, []
            for _ in range(n):
                ln, pos = _read_varint(raw, pos)
                lens.append(ln)
            pos += 4
            elems = []
            for ln in lens:
                elems.append(bytes(raw[pos:pos + ln]))
                pos += ln
            if verify and want is not None and _encode_strings(elems)[1] != _unmask(want):
                raise ValueError('%s fails its checksum' % name)
            out[name] = elems[0] if not shape else elems
            continue
        if dtype not in _NP_OF:
            raise ValueError('%s: unsupported dtype enum %d' % (name, dtype))
        if verify and want is not None and _mask(crc32c(raw)) != want:
            raise ValueError('%s fails its checksum' % name)
        a = np.frombuffer(raw, _NP_OF[dtype]).reshape(shape)
        if dtype == DT_BFLOAT16:
            a = (a.astype(np.uint32) << 16).view(np.float32)
        out[name] = a
    return out


def _unmask(m):
    r = (m - 0xA282EAD8) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------- object graph
def object_graph(keys, full_names=None):
    """Serialized TrackableObjectGraph for object-based checkpoint keys (the tree the key paths spell out; node 0 is the
    root; optimizer slots become slot_variables of the optimizer node).  full_names: {key: variable name} (optional)."""
    nodes = [{'children': {}, 'attrs': [], 'slots': []}]

    def walk(path):
        cur = 0
        for part in path:
            nxt = nodes[cur]['children'].get(part)
            if nxt is None:
                nodes.append({'children': {}, 'attrs': [], 'slots': []})
                nxt = len(nodes) - 1
                nodes[cur]['children'][part] = nxt
            cur = nxt
        return cur

    plain = sorted(k for k in keys if k.endswith(VALUE_SUFFIX) and SLOT_MARK not in k)
    slots = sorted(k for k in keys if k.endswith(VALUE_SUFFIX) and SLOT_MARK in k)
    for k in plain:
        node = walk(k[:-len(VALUE_SUFFIX)].split('/'))
        nodes[node]['attrs'].append((k, (full_names or {}).get(k, '')))
    for k in slots:
        var_path, rest = k[:-len(VALUE_SUFFIX)].split(SLOT_MARK)
        opt_path, slot_name = rest.rsplit('/', 1)
        var, opt = walk(var_path.split('/')), walk(opt_path.split('/'))
        nodes.append({'children': {}, 'attrs': [(k, (full_names or {}).get(k, ''))], 'slots': []})
        nodes[opt]['slots'].append((var, slot_name, len(nodes) - 1))
    out = b''
    for n in nodes:
        body = b''
        for name, child in n['children'].items():
            body += _ld(1, _field(1, 0) + _varint(child) + _ld(2, name.encode()))
        for key, full in n['attrs']:
            body += _ld(2, _ld(1, b'VARIABLE_VALUE') + (_ld(2, full.encode()) if full else b'') + _ld(3, key.encode()))
        for var, slot_name, slot_node in n['slots']:
            body += _ld(3, _field(1, 0) + _varint(var) + _ld(2, slot_name.encode()) + _field(3, 0) + _varint(slot_node))
        out += _ld(1, body)
    return out


def checkpoint_keys_of_graph(graph: bytes):
    """The checkpoint_key of every SerializedTensor in a serialized TrackableObjectGraph."""
    keys = []
    for num, _, node in _fields(graph):
        if num != 1:
            continue
        for n1, _, attr in _fields(node):
            if n1 == 2:
                keys.extend(v.decode() for n2, _, v in _fields(attr) if n2 == 3)
    return keys


# ---------------------------------------------------------------------------------------------- state dict <-> bundle keys
def bundle_from_state(tensors, meta):
    """{'model/<p>', 'optimizer/m/<p>', 'optimizer/v/<p>', 'optimizer/iter'} + {'step','epoch','save_counter'} -> the keys
    of tf.train.Checkpoint(step, epoch, optimizer, model) (transformer.py:890)."""
    out = {}
    for k, v in tensors.items():
        if k.startswith('model/'):
            out[k + VALUE_SUFFIX] = np.asarray(v)
        elif k.startswith('optimizer/m/') or k.startswith('optimizer/v/'):
            slot, p = k.split('/', 2)[1:]
            out['model/' + p + SLOT_MARK + 'optimizer/' + slot + VALUE_SUFFIX] = np.asarray(v)
        elif k == 'optimizer/iter':
            out['optimizer/iter' + VALUE_SUFFIX] = np.asarray(v, np.int64)
        else:
            raise KeyError('no TensorBundle name for %r' % k)
    for name in ('step', 'epoch', 'save_counter'):
        if name in meta:
            out[name + VALUE_SUFFIX] = np.asarray(meta[name], np.int64)
    out[OBJECT_GRAPH_KEY] = object_graph(out.keys())
    return out


def state_from_bundle(bundle):
    """Inverse of bundle_from_state; keys that are not part of the contract (other optimizer hyper-parameters, Keras
    bookkeeping) are ignored."""
    tensors, meta = {}, {}
    for k, v in bundle.items():
        if not k.endswith(VALUE_SUFFIX):
            continue
        name = k[:-len(VALUE_SUFFIX)]
        if SLOT_MARK in name:
            var, rest = name.split(SLOT_MARK)
            slot = rest.rsplit('/', 1)[1]
            if var.startswith('model/') and slot in ('m', 'v'):
                tensors['optimizer/%s/%s' % (slot, var[len('model/'):])] = v
        elif name.startswith('model/'):
            tensors[name] = v
        elif name == 'optimizer/iter':
            tensors[name] = np.int64(v)
        elif name in ('step', 'epoch', 'save_counter'):
            meta[name] = int(v)
    return tensors, meta
