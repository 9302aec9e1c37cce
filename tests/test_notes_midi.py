"""SURVEY section 8f rank 3 (MIDI in/out for `composer generate`), CPU only.

* `composer_amd.notes` against `tests/golden/notes.npz`: 96 cases produced by the REFERENCE's own
  composer/dataset/sequence.py (to_event_sequence in all three sustain modes with and without cleaning, trim_start,
  to_note_sequence, event ids) -- `tests/golden/make_notes_golden.py`;
* the known-answer sequences of the reference's tests/test_sequences.py:106-230 (data transcribed, velocity_bins = 4);
* `composer_amd.midi` against files assembled by hand from the SMF 1.0 specification, and writer -> reader round trips."""
import os
import struct

import numpy as np
import pytest

from composer_amd import dataset as ds
from composer_amd import midi
from composer_amd.notes import Note, NoteSequence, SustainPeriod, ids_to_midi, prompt_ids_from_midi

HERE = os.path.dirname(os.path.abspath(__file__))
ON, OFF, TS, VEL, SON, SOFF = ds.NOTE_ON, ds.NOTE_OFF, ds.TIME_SHIFT, ds.VELOCITY, ds.SUSTAIN_ON, ds.SUSTAIN_OFF


def test_note_event_conversion_matches_the_reference_module():
    g = np.load(os.path.join(HERE, "golden", "notes.npz"))
    n_cases = int(g["n_cases"])
    assert n_cases == 96
    checked_events = 0
    for i in range(n_cases):
        p = "c%03d_" % i
        tsi, mts, vb = (int(v) for v in g[p + "settings"])
        mode = str(g[p + "mode"])
        clean, trim = (bool(v) for v in g[p + "flags"])
        seq = NoteSequence([Note(float(a), float(b), int(c), int(d)) for a, b, c, d in g[p + "notes"]],
                           [SustainPeriod(float(a), float(b)) for a, b in g[p + "periods"]])
        if trim:
            seq.trim_start()
        events = seq.to_events(tsi, mts, vb, sustain=mode, clean=clean)
        want = [(int(t), None if v < 0 else int(v)) for t, v in g[p + "events"]]
        assert events == want, "case %d (%s, clean=%s)" % (i, mode, clean)
        vr = ds.event_value_ranges(tsi, mts, vb)
        rg = ds.event_ranges(vr)
        assert [ds.event_to_id(t, v, rg, vr) for t, v in events] == [int(x) for x in g[p + "ids"]]
        back = NoteSequence.from_events(events, tsi, vb)
        assert [(n.start, n.end, n.pitch, n.velocity) for n in back.notes] == [tuple(r) for r in g[p + "back_notes"].tolist()]
        assert [(s.start, s.end) for s in back.sustain_periods] == [tuple(r) for r in g[p + "back_periods"].tolist()]
        checked_events += len(events)
    assert checked_events > 10000


def test_reference_known_answer_sequences():
    """tests/test_sequences.py:106-230 of the reference (time step 10 ms, 100 steps max, 4 velocity bins)."""
    a = NoteSequence([Note(0, 2000, 2, 64), Note(3000, 4000, 1, 9)]).to_events(10, 100, 4)
    assert a == [(VEL, 2), (ON, 2), (TS, 100), (TS, 100), (OFF, 2), (TS, 100), (VEL, 0), (ON, 1), (TS, 100), (OFF, 1)]
    b = NoteSequence([Note(0, 4000, 1, 37), Note(0, 4000, 4, 37), Note(5000, 11000, 3, 96)], [SustainPeriod(4000, 5000)]).to_events(10, 100, 4)
    assert b == ([(VEL, 1), (ON, 1), (ON, 4)] + [(TS, 100)] * 4 + [(SON, None), (OFF, 1), (OFF, 4), (TS, 100), (SOFF, None), (VEL, 3), (ON, 3)]
                 + [(TS, 100)] * 6 + [(OFF, 3)])
    c = NoteSequence(None, [SustainPeriod(0, 1000), SustainPeriod(2500, 5670), SustainPeriod(8000, 10000)]).to_events(10, 100, 4)
    assert c == [(SON, None), (TS, 100), (SOFF, None), (TS, 100), (TS, 50), (SON, None), (TS, 100), (TS, 100), (TS, 100), (TS, 17),
                 (SOFF, None), (TS, 100), (TS, 100), (TS, 33), (SON, None), (TS, 100), (TS, 100), (SOFF, None)]
    # the ids `composer generate` would feed the model for sequence b (tests/test_sequences.py:310-351)
    vr = ds.event_value_ranges(10, 100, 4)
    rg = ds.event_ranges(vr)
    assert [ds.event_to_id(t, v, rg, vr) for t, v in b] == [257, 1, 4, 359, 359, 359, 359, 360, 129, 132, 359, 361, 259, 3,
                                                             359, 359, 359, 359, 359, 359, 131]


def _smf(division, tracks, fmt=1):
    out = b"MThd" + struct.pack(">IHHH", 6, fmt, len(tracks), division)
    for t in tracks:
        out += b"MTrk" + struct.pack(">I", len(t)) + t
    return out


def test_reader_on_hand_assembled_files(tmp_path):
    # format 1, 480 ticks per quarter.  Track 0: tempo 500000 us (120 bpm) at 0, 250000 us (240 bpm) at tick 960.
    t0 = bytes([0x00, 0xFF, 0x51, 0x03, 0x07, 0xA1, 0x20,            # tempo 500000
                0x87, 0x40, 0xFF, 0x51, 0x03, 0x03, 0xD0, 0x90,      # delta 960 (0x87 0x40): tempo 250000
                0x00, 0xFF, 0x2F, 0x00])
    # Track 1, channel 0: program 5; note 60 on (vel 100) at 0, RUNNING STATUS note 64 on at 480, note 60 off (vel-0 on) at 960,
    # pedal down (cc64=127) at 960, note 64 off (0x80) at 1440, pedal up at 1920, a text meta event in between
    t1 = bytes([0x00, 0xC0, 0x05,
                0x00, 0x90, 60, 100,
                0x83, 0x60, 64, 90,                                   # delta 480, running status
                0x83, 0x60, 60, 0,                                    # delta 480: note-on velocity 0 == off
                0x00, 0xB0, 64, 127,
                0x00, 0xFF, 0x01, 0x02, 0x68, 0x69,                   # text "hi"
                0x83, 0x60, 0x80, 64, 0,                              # delta 480: real note-off
                0x83, 0x60, 0xB0, 64, 0,                              # delta 480: pedal up
                0x00, 0xFF, 0x2F, 0x00])
    # Track 2, channel 9 (drums): a hit at tick 0..240
    t2 = bytes([0x00, 0x99, 36, 127, 0x81, 0x70, 0x89, 36, 0, 0x00, 0xFF, 0x2F, 0x00])
    path = tmp_path / "hand.mid"
    path.write_bytes(_smf(480, [t0, t1, t2]))
    insts = midi.read(path)
    assert [(i.program, i.is_drum) for i in insts] == [(5, False), (0, True)]
    piano = insts[0]
    # 480 ticks = 0.5 s at 120 bpm; after tick 960 (1.0 s) a tick lasts 250000 us / 480
    assert [(n.pitch, n.velocity) for n in piano.notes] == [(60, 100), (64, 90)]
    np.testing.assert_allclose([(n.start, n.end) for n in piano.notes], [(0.0, 1.0), (0.5, 1.25)], atol=1e-12)
    assert [(c.number, c.value) for c in piano.control_changes] == [(64, 127), (64, 0)]
    np.testing.assert_allclose([c.time for c in piano.control_changes], [1.0, 1.5], atol=1e-12)
    np.testing.assert_allclose([(n.start, n.end) for n in insts[1].notes], [(0.0, 0.25)], atol=1e-12)
    # the reference's from_midi on top: milliseconds, pedal period 1000..1500, drums ignored
    seq = NoteSequence.from_midi(path)
    assert [(n.start, n.end, n.pitch, n.velocity) for n in seq.notes] == [(0.0, 1000.0, 60, 100), (500.0, 1250.0, 64, 90)]
    assert [(s.start, s.end) for s in seq.sustain_periods] == [(1000.0, 1500.0)]
    assert len(NoteSequence.from_midi(path, ignore_drums=False).notes) == 3
    assert len(NoteSequence.from_midi(path, programs=[7]).notes) == 0

    # format 0 (single track, all channels), a repeated note-on of a sounding pitch: one note-off closes both
    t = bytes([0x00, 0x91, 70, 50, 0x60, 0x91, 70, 60, 0x60, 0x81, 70, 0, 0x00, 0xFF, 0x2F, 0x00])
    p0 = tmp_path / "fmt0.mid"
    p0.write_bytes(_smf(96, [t], fmt=0))
    (inst,) = midi.read(p0)
    np.testing.assert_allclose([(n.start, n.end, n.velocity) for n in inst.notes], [(0.0, 1.0, 50), (0.5, 1.0, 60)], atol=1e-12)

    for bad in (b"RIFF" + bytes(20), _smf(0xE728, [t]), b"MThd" + struct.pack(">IHHH", 6, 1, 1, 0)):
        pb = tmp_path / "bad.mid"
        pb.write_bytes(bad)
        with pytest.raises(midi.MidiFormatError):
            midi.read(pb)


def test_writer_layout_and_round_trip(tmp_path):
    seq = NoteSequence([Note(0, 500, 60, 100), Note(250, 1000, 64, 33), Note(1000, 1010, 67, 127)], [SustainPeriod(100, 900)])
    path = tmp_path / "out.mid"
    seq.to_midi(path, program=1)
    blob = path.read_bytes()
    assert blob[:14] == b"MThd" + struct.pack(">IHHH", 6, 1, 2, 220)          # format 1, conductor + 1 instrument, 220 tpq
    assert b"\xFF\x51\x03\x07\xA1\x20" in blob and b"\xFF\x58\x04\x04\x02\x18\x08" in blob   # 120 bpm, 4/4
    back = NoteSequence.from_midi(path)
    # tick = round(seconds * 440): every time is within half a tick (1.14 ms) of the original
    assert [(n.pitch, n.velocity) for n in back.notes] == [(60, 100), (64, 33), (67, 127)]
    np.testing.assert_allclose([(n.start, n.end) for n in back.notes], [(0, 500), (250, 1000), (1000, 1010)], atol=1.2)
    np.testing.assert_allclose([(s.start, s.end) for s in back.sustain_periods], [(100, 900)], atol=1.2)
    (inst,) = midi.read(path)
    assert inst.program == 1 and not inst.is_drum


def test_generate_side_helpers(tmp_path):
    """cli.py:645-660 / 676-680: prompt ids from a MIDI file, and event ids back to a MIDI file."""
    rng = np.random.default_rng(5)
    notes, t = [], 300.0
    for _ in range(40):
        d = float(rng.integers(5, 80)) * 10
        notes.append(Note(t, t + d, int(rng.integers(48, 84)), int(rng.integers(16, 128))))   # bin >= 4: a bin-0 note decodes to
        # MIDI velocity 0, which a MIDI file cannot hold as a note-on (same in the reference)
        t += float(rng.integers(0, 40)) * 10
    src = tmp_path / "prompt.mid"
    NoteSequence(notes, [SustainPeriod(400.0, 2000.0)]).to_midi(src)
    ids = prompt_ids_from_midi(src, 25)
    assert len(ids) == 25 and all(0 <= i < 390 for i in ids)
    # same ids as converting the re-read note sequence by hand
    seq = NoteSequence.from_midi(src).trim_start()
    vr = ds.event_value_ranges(10, 100, 32)
    rg = ds.event_ranges(vr)
    assert ids == [ds.event_to_id(t_, v, rg, vr) for t_, v in seq.to_events()[:25]]
    # ids -> MIDI -> notes: identical to decoding the events directly (times are multiples of 10 ms: exact in ticks? no --
    # 10 ms = 4.4 ticks, so compare within half a tick)
    all_ids = [ds.event_to_id(t_, v, rg, vr) for t_, v in seq.to_events()]
    out = tmp_path / "gen" / "out.mid"
    out.parent.mkdir()
    ids_to_midi(all_ids, out)
    direct = NoteSequence.from_events(seq.to_events())
    reread = NoteSequence.from_midi(out)
    assert sorted((n.pitch, n.velocity) for n in reread.notes) == sorted((n.pitch, n.velocity) for n in direct.notes)
    np.testing.assert_allclose(sorted(n.start for n in reread.notes), sorted(n.start for n in direct.notes), atol=1.2)
