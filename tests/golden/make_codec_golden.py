"""Generates tests/golden/codec.npz by IMPORTING the reference's own composer/dataset/sequence.py (possible in the
build container only, with the shims of SURVEY section 8c: a stub `pretty_midi`, np.int/np.float aliases, a bare
`composer` package object that skips composer/__init__.py -> cli -> TensorFlow).

    python tests/golden/make_codec_golden.py

Vectors: the known-answer ids of the reference's tests/test_sequences.py:310-397 (velocity_bins=4), the full
id <-> (type, value) table at the default dataset config (vocab 390), and the bytes of two `.data` files written by the
reference's IntegerEncodedEventSequence.to_file together with the ids its event_ids_from_file returns."""
import os
import sys
import types
import tempfile
import numpy as np

np.int, np.float = int, float
pm = types.ModuleType('pretty_midi')
for n in ('PrettyMIDI', 'Instrument', 'Note', 'ControlChange'):
    setattr(pm, n, type(n, (), {}))
sys.modules['pretty_midi'] = pm
pkg = types.ModuleType('composer'); pkg.__path__ = ['/root/reference/composer']
sys.modules['composer'] = pkg
import composer.dataset.sequence as S      # noqa: E402  (the reference module itself)

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}

# 1. known-answer vector of the reference's own test (tests/test_sequences.py:310-351)
E, T = S.Event, S.EventType
ev = [E(T.VELOCITY, 1), E(T.NOTE_ON, 1), E(T.NOTE_ON, 4)] + [E(T.TIME_SHIFT, 100)] * 4 + [E(T.SUSTAIN_ON, None), E(T.NOTE_OFF, 1),
     E(T.NOTE_OFF, 4), E(T.TIME_SHIFT, 100), E(T.SUSTAIN_OFF, None), E(T.VELOCITY, 3), E(T.NOTE_ON, 3)] + [E(T.TIME_SHIFT, 100)] * 6 + [E(T.NOTE_OFF, 3)]
seq = S.EventSequence(ev, 10, 100, 4)
ids = [S.IntegerEncodedEventSequence.event_to_id(e.type, e.value, seq.event_ranges, seq.event_value_ranges) for e in seq.events]
assert ids == [257, 1, 4, 359, 359, 359, 359, 360, 129, 132, 359, 361, 259, 3, 359, 359, 359, 359, 359, 359, 131]
out['kat_types'] = np.array([int(e.type) for e in ev]); out['kat_values'] = np.array([-1 if e.value is None else e.value for e in ev])
out['kat_ids'] = np.array(ids); out['kat_settings'] = np.array([10, 100, 4])

# 2. full table at the default config (default_config.yml:3-6)
vr = S.EventSequence._compute_event_value_ranges(10, 100, 32)
rg = S.EventSequence._compute_event_ranges(S.EventSequence._compute_event_dimensions(vr))
V = S.OneHotEncodedEventSequence.get_one_hot_size(rg)
tab = []
for i in range(V):
    e = S.IntegerEncodedEventSequence.id_to_event(i, rg, vr)
    assert S.IntegerEncodedEventSequence.event_to_id(e.type, e.value, rg, vr) == i
    tab.append((int(e.type), -1 if e.value is None else e.value))
out['vocab'] = np.int64(V); out['table'] = np.array(tab)
out['range_starts'] = np.array([rg[t].start for t in rg]); out['range_types'] = np.array([int(t) for t in rg])

# 3. `.data` bytes written by the reference and the ids it reads back
rng = np.random.default_rng(7)
for k, (settings, n) in enumerate((((10, 100, 32), 257), ((10, 100, 4), 21))):
    vr = S.EventSequence._compute_event_value_ranges(*settings)
    types_ = list(vr.keys())
    events = []
    for _ in range(n):
        t = types_[int(rng.integers(0, len(types_)))]
        v = S.Event.NONE_VALUE if vr[t] is None else int(rng.integers(vr[t].start, vr[t].stop))
        events.append((int(t), v))
    enc = S.IntegerEncodedEventSequence(settings[0], settings[1], settings[2], events)
    with tempfile.NamedTemporaryFile(suffix='.data', delete=False) as f:
        path = f.name
    enc.to_file(path)
    raw = open(path, 'rb').read()
    ids, _, _, st = S.IntegerEncodedEventSequence.event_ids_from_file(path)
    os.remove(path)
    out['file%d_bytes' % k] = np.frombuffer(raw, dtype=np.uint8)
    out['file%d_ids' % k] = np.array(list(ids)); out['file%d_settings' % k] = np.array(st)
    out['file%d_events' % k] = np.array(events)
np.savez_compressed(os.path.join(HERE, 'codec.npz'), **out)
print('vocab', V, 'header bytes', bytes(out['file0_bytes'][:14]).hex(), 'ok')
