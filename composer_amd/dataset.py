"""`.data` event-sequence files -> integer id stream -> (x, y) minibatches, without TensorFlow.

Restates the neighbours of the hot path:
  * integer vocabulary: reference composer/dataset/sequence.py:740-766 (value ranges), :792-805 (dimensions),
    :826-844 (ranges), :1590-1640 (event_to_id / id_to_event);
  * `.data` layout: struct 'Q' + 'hhh' + 'hh'*N, native little-endian, 14-byte header (:1441-1442, 1500-1552, 1866);
  * pipeline: reference composer/models/__init__.py:160-313 -- ids of all files concatenated in file order, chopped
    into non-overlapping windows of W+1 (remainder dropped), (x, y) = (w[:-1], w[1:]), shuffle buffer of 500*B
    windows, batches of B windows (remainder dropped); `<root>/{train,test}/**/*.data` (cli.py:221-230).
Pure host code (numpy); the batches feed Transformer.train / evaluate.
"""
import collections
import struct
from pathlib import Path

import numpy as np

MAGIC_INTEGER = 9223372036854775805          # IntegerEncodedEventSequence.get_encoding_type(), sequence.py:1866
HEADER = struct.Struct('=Qhhh')              # 14 bytes
assert HEADER.size == 14

# EventType, sequence.py:87-92
NOTE_ON, NOTE_OFF, TIME_SHIFT, VELOCITY, SUSTAIN_ON, SUSTAIN_OFF = 1, 2, 3, 4, 5, 6
NONE_VALUE = -1                              # Event.NONE_VALUE, sequence.py:125
_ORDER = (NOTE_ON, NOTE_OFF, VELOCITY, TIME_SHIFT, SUSTAIN_ON, SUSTAIN_OFF)   # insertion order at sequence.py:740-766


def event_value_ranges(time_step_increment, max_time_steps, velocity_bins):
    """sequence.py:740-766.  None = the event type takes no value."""
    r = collections.OrderedDict()
    r[NOTE_ON] = range(0, 128)
    r[NOTE_OFF] = range(0, 128)
    r[VELOCITY] = range(0, velocity_bins)
    r[TIME_SHIFT] = range(1, max_time_steps + 1)
    r[SUSTAIN_ON] = None
    r[SUSTAIN_OFF] = None
    return r


def event_ranges(value_ranges):
    """sequence.py:792-805 + 826-844: position of each event type in the id space."""
    out, off = collections.OrderedDict(), 0
    for t, vr in value_ranges.items():
        dim = 0 if vr is None else vr.stop - vr.start
        if dim == 0:
            dim = 1
        out[t] = range(off, off + dim)
        off += dim
    return out


def vocab_size(time_step_increment, max_time_steps, velocity_bins):
    """cli.py:400-412 (_get_event_vocab_size): 390 at the default dataset config."""
    return event_ranges(event_value_ranges(time_step_increment, max_time_steps, velocity_bins))[SUSTAIN_OFF].stop


def event_to_id(event_type, value, ranges, value_ranges):
    """sequence.py:1590-1612"""
    off = 0
    if value_ranges[event_type] is not None:
        off = value - value_ranges[event_type].start
    return ranges[event_type].start + off


def id_to_event(event_id, ranges, value_ranges):
    """sequence.py:1615-1640 -> (event_type, value or None)"""
    for t, interval in ranges.items():
        if event_id in interval:
            value = None
            if value_ranges[t] is not None:
                value = event_id - interval.start + value_ranges[t].start
            return t, value
    raise ValueError('event id %d outside the vocabulary' % event_id)


def read_data_file(path):
    """-> (ids uint16 [N], (time_step_increment, max_time_steps, velocity_bins)); sequence.py:1643-1695 vectorised."""
    raw = Path(path).read_bytes()
    if len(raw) < HEADER.size:
        raise ValueError('%s: shorter than the 14-byte header' % path)
    magic, tsi, mts, vb = HEADER.unpack_from(raw)
    if magic != MAGIC_INTEGER:
        raise ValueError('%s: encoding type id %d is not an IntegerEncodedEventSequence' % (path, magic))
    n = (len(raw) - HEADER.size) // 4
    ev = np.frombuffer(raw, dtype='<i2', count=2 * n, offset=HEADER.size).reshape(n, 2)
    vr = event_value_ranges(tsi, mts, vb)
    rg = event_ranges(vr)
    start = np.zeros(8, np.int64)
    vstart = np.zeros(8, np.int64)
    has_value = np.zeros(8, bool)
    for t in _ORDER:
        start[t] = rg[t].start
        if vr[t] is not None:
            vstart[t] = vr[t].start
            has_value[t] = True
    types = ev[:, 0].astype(np.int64)
    if n and (types.min() < 1 or types.max() > 6):
        raise ValueError('%s: unknown event type id' % path)
    ids = start[types] + np.where(has_value[types], ev[:, 1].astype(np.int64) - vstart[types], 0)
    return ids.astype(np.uint16), (tsi, mts, vb)


def write_data_file(path, events, time_step_increment=10, max_time_steps=100, velocity_bins=32):
    """events: iterable of (event_type, value or None); sequence.py:1500-1526."""
    flat = []
    for t, v in events:
        flat += [int(t), NONE_VALUE if v is None else int(v)]
    with open(path, 'wb') as f:
        f.write(HEADER.pack(MAGIC_INTEGER, time_step_increment, max_time_steps, velocity_bins))
        f.write(np.asarray(flat, dtype='<i2').tobytes())


def get_processed_files(dataset_path):
    """preprocess.py:16-33: every *.data under the directory."""
    p = Path(dataset_path)
    if not p.is_dir():
        raise ValueError('\'{}\' is an invalid dataset path!'.format(p))
    return sorted(p.glob('**/*.data'))


class WindowDataset:
    """Re-iterable dataset of (x, y) int32 batches [B, W]; one pass = one epoch (reshuffled each iteration).

    rank/world_size: data parallelism -- every global batch has world_size*B windows and rank r takes
    rows [r*B, (r+1)*B); all ranks walk the same shuffled order (same seed)."""

    def __init__(self, ids, batch_size, window_size, shuffle=True, seed=0, rank=0, world_size=1):
        ids = np.asarray(ids)
        n = len(ids) // (window_size + 1)                       # drop_remainder, models/__init__.py:304
        self.windows = ids[:n * (window_size + 1)].reshape(n, window_size + 1).astype(np.int32)
        self.B, self.W = int(batch_size), int(window_size)
        self.shuffle, self.seed = shuffle, int(seed)
        self.rank, self.world = int(rank), int(world_size)
        self._epoch = 0

    def __len__(self):
        return len(self.windows) // (self.B * self.world)

    def _order(self):
        n = len(self.windows)
        if not self.shuffle:
            return np.arange(n)
        # tf.data shuffle(buffer) semantics (models/__init__.py:306-309): keep a buffer of `cap` elements, emit a
        # uniformly random one, refill from the stream
        rng = np.random.default_rng([self.seed, self._epoch])
        cap = 500 * self.B
        if cap >= n:
            return rng.permutation(n)
        buf = list(range(cap))
        out = np.empty(n, np.int64)
        nxt = cap
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
        G = self.B * self.world
        for b in range(len(order) // G):                        # batch(B, drop_remainder=True)
            rows = order[b * G + self.rank * self.B: b * G + (self.rank + 1) * self.B]
            w = self.windows[rows]
            yield w[:, :-1].copy(), w[:, 1:].copy()


def load_dataset(filepaths, batch_size, window_size, shuffle=True, seed=0, rank=0, world_size=1, expect_settings=None):
    """models/__init__.py:238-313 without tf.data."""
    parts = []
    for p in filepaths:
        ids, settings = read_data_file(p)
        if expect_settings is not None and tuple(settings) != tuple(expect_settings):
            raise ValueError('%s was preprocessed with %s but the config says %s' % (p, settings, tuple(expect_settings)))
        parts.append(ids)
    ids = np.concatenate(parts) if parts else np.zeros(0, np.uint16)
    return WindowDataset(ids, batch_size, window_size, shuffle, seed, rank, world_size)


def write_synthetic_data_file(path, n_events, seed=0, time_step_increment=10, max_time_steps=100, velocity_bins=32):
    """A valid `.data` file with events drawn uniformly from the valid (type, value) pairs."""
    rng = np.random.default_rng(seed)
    vr = event_value_ranges(time_step_increment, max_time_steps, velocity_bins)
    events = []
    for _ in range(n_events):
        t = _ORDER[int(rng.integers(0, 6))]
        v = None if vr[t] is None else int(rng.integers(vr[t].start, vr[t].stop))
        events.append((t, v))
    write_data_file(path, events, time_step_increment, max_time_steps, velocity_bins)
    return events
