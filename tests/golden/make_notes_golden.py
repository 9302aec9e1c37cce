"""Generates tests/golden/notes.npz by IMPORTING the reference's own composer/dataset/sequence.py (build container only;
same shims as make_codec_golden.py: stub `pretty_midi`, np.int/np.float aliases, bare `composer` package).

    python tests/golden/make_notes_golden.py

Vectors: for seeded random note sequences (overlapping notes, repeated pitches, zero-length notes, fractional
millisecond times, pedal periods) and several codec settings, the reference's
  NoteSequence.to_event_sequence(...)            -> event (type, value) list, for sustain modes EVENTS / NONE / EXTEND, clean on/off
  NoteSequence.trim_start()                      -> shifted times
  EventSequence.to_note_sequence()               -> notes / pedal periods rebuilt from the events
  IntegerEncodedEventSequence.event_to_id        -> ids of the events (what `composer generate` feeds the model)
Stored flat (ragged lists concatenated with offsets)."""
import os
import sys
import types
import copy
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
rng = np.random.default_rng(2024)
MODES = {'events': S.NoteSequence.SustainPeriodEncodeMode.EVENTS, 'none': S.NoteSequence.SustainPeriodEncodeMode.NONE,
         'extend': S.NoteSequence.SustainPeriodEncodeMode.EXTEND}
cases = []
for ci in range(24):
    settings = [(10, 100, 32), (10, 100, 4), (20, 50, 16), (5, 100, 8)][ci % 4]
    nn = int(rng.integers(0 if ci % 6 == 5 else 1, 40))
    frac = ci % 3 == 0                                   # fractional-millisecond times (MIDI seconds * 1000)
    notes = []
    for _ in range(nn):
        st = float(rng.uniform(0, 20000)) if frac else float(int(rng.integers(0, 2000)) * 10)
        du = float(rng.uniform(0, 3000)) if frac else float(int(rng.integers(0, 300)) * 10)
        if rng.random() < 0.1:
            du = 0.0
        notes.append((st, st + du, int(rng.integers(40, 52 if ci % 2 else 100)), int(rng.integers(1, 128))))
    ns_p = int(rng.integers(0, 4)) if ci % 6 != 5 else 3
    periods, t = [], float(rng.uniform(0, 1500))
    for _ in range(ns_p):
        d = float(rng.uniform(100, 4000)) if frac else float(int(rng.integers(10, 400)) * 10)
        periods.append((t, t + d))
        t += d + (float(rng.uniform(50, 3000)) if frac else float(int(rng.integers(5, 300)) * 10))
    for mode in ('events', 'none', 'extend'):
        for clean in (True, False):
            if mode != 'events' and not clean:
                continue
            mk = lambda: S.NoteSequence([S.Note(*n) for n in notes], [S.SustainPeriod(*p) for p in periods])
            seq = mk()
            trimmed = None
            if nn > 0 and ci % 2 == 0:
                seq.trim_start()
                trimmed = [(n.start, n.end) for n in seq.notes]
            es = seq.to_event_sequence(settings[0], settings[1], settings[2], MODES[mode], clean)
            ev = [(int(e.type), -1 if e.value is None else int(e.value)) for e in es.events]
            ids = [S.IntegerEncodedEventSequence.event_to_id(e.type, e.value, es.event_ranges, es.event_value_ranges) for e in es.events]
            back = es.to_note_sequence()
            cases.append(dict(settings=settings, notes=notes, periods=periods, mode=mode, clean=clean, trim=trimmed is not None,
                              events=ev, ids=ids, back_notes=[(n.start, n.end, n.pitch, n.velocity) for n in back.notes],
                              back_periods=[(p.start, p.end) for p in back.sustain_periods]))

out = {'n_cases': np.int64(len(cases))}
for i, c in enumerate(cases):
    p = 'c%03d_' % i
    out[p + 'settings'] = np.array(c['settings'])
    out[p + 'notes'] = np.array(c['notes'], dtype=np.float64).reshape(-1, 4)
    out[p + 'periods'] = np.array(c['periods'], dtype=np.float64).reshape(-1, 2)
    out[p + 'mode'] = np.array(c['mode'])
    out[p + 'flags'] = np.array([int(c['clean']), int(c['trim'])])
    out[p + 'events'] = np.array(c['events'], dtype=np.int64).reshape(-1, 2)
    out[p + 'ids'] = np.array(c['ids'], dtype=np.int64)
    out[p + 'back_notes'] = np.array(c['back_notes'], dtype=np.float64).reshape(-1, 4)
    out[p + 'back_periods'] = np.array(c['back_periods'], dtype=np.float64).reshape(-1, 2)
np.savez_compressed(os.path.join(HERE, 'notes.npz'), **out)
print('wrote', len(cases), 'cases;', sum(len(c['events']) for c in cases), 'events')
