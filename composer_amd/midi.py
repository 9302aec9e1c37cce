"""A small Standard MIDI File (SMF) reader / writer: what `pretty_midi` does for the reference's
`NoteSequence.from_midi / to_midi` (`composer/dataset/sequence.py:594-680`).  `pretty_midi` (and `mido` under it) is not
in this image and is a third-party dependency of the reference (unpinned, `environment.yml`), so this module follows the
SMF 1.0 specification and restates the parts of pretty_midi's published behaviour that the reference relies on:

reading (pretty_midi.PrettyMIDI(file), `_load_tempo_changes`, `_load_instruments`)
  * formats 0 and 1, metrical time division (ticks per quarter note; SMPTE division is rejected), running status, meta
    and sysex events skipped except set-tempo (0x51);
  * ticks -> seconds through the tempo map (default 120 bpm until the first set-tempo).  pretty_midi reads tempo events
    from track 0 only; here the set-tempo events of ALL tracks are merged (identical for conforming format-1 files);
  * one Instrument per (program, channel, track): a note belongs to the program that is current on its channel when it
    ENDS; channel 9 is the drum channel; a note-off (or note-on with velocity 0) closes every open note of that
    (channel, pitch) that started on an earlier tick; notes still open at the end of the track are dropped;
  * control changes go to the instrument of (current program, channel, track); those seen before that instrument has any
    note are attached to the first instrument of the same (channel, track) that gets one, otherwise dropped.

writing (pretty_midi.PrettyMIDI().write): format 1, 220 ticks per quarter note, track 0 = {4/4 time signature, 120 bpm},
  one track per instrument on channels 0..15 skipping 9 (drums on 9) = {program change, control changes, notes}; a
  note-off is a note-on with velocity 0; tick = round(seconds * 440); at equal ticks: program change, control change,
  note-off, note-on (so a note that rounds to zero ticks puts its off before its on and is lost on re-reading, as with
  pretty_midi).

Pinning: the reader is checked against hand-assembled files (bytes written out in tests/test_midi.py from the SMF
specification) and the writer by round trips through the reader -- **not** against pretty_midi itself (absent here).
"""
import dataclasses
import struct
from typing import List


@dataclasses.dataclass
class MidiNote:
    velocity: int
    pitch: int
    start: float            # seconds
    end: float


@dataclasses.dataclass
class ControlChange:
    number: int
    value: int
    time: float             # seconds


@dataclasses.dataclass
class Instrument:
    program: int = 0
    is_drum: bool = False
    notes: List[MidiNote] = dataclasses.field(default_factory=list)
    control_changes: List[ControlChange] = dataclasses.field(default_factory=list)


class MidiFormatError(ValueError):
    pass


# ----------------------------------------------------------------------------------------------- reading
def _vlq(data, pos):
    value = 0
    while True:
        if pos >= len(data):
            raise MidiFormatError('truncated variable-length quantity')
        b = data[pos]
        pos += 1
        value = (value << 7) | (b & 0x7F)
        if not b & 0x80:
            return value, pos


def _parse_track(data):
    """-> list of (abs_tick, kind, channel, a, b); kinds: 'on', 'off', 'cc', 'pc', 'tempo'."""
    out, pos, tick, status = [], 0, 0, None
    while pos < len(data):
        delta, pos = _vlq(data, pos)
        tick += delta
        if pos >= len(data):
            raise MidiFormatError('event without a status byte')
        b = data[pos]
        if b == 0xFF:                                   # meta
            if pos + 1 >= len(data):
                raise MidiFormatError('truncated meta event')
            mtype = data[pos + 1]
            length, p2 = _vlq(data, pos + 2)
            payload = data[p2:p2 + length]
            pos = p2 + length
            if mtype == 0x51 and length == 3:
                out.append((tick, 'tempo', 0, (payload[0] << 16) | (payload[1] << 8) | payload[2], 0))
            elif mtype == 0x2F:
                break
            continue                                    # (meta events do not cancel running status in practice)
        if b in (0xF0, 0xF7):                           # sysex
            length, p2 = _vlq(data, pos + 1)
            pos = p2 + length
            status = None
            continue
        if b & 0x80:
            status = b
            pos += 1
        elif status is None:
            raise MidiFormatError('data byte without running status')
        hi, ch = status & 0xF0, status & 0x0F
        need = 1 if hi in (0xC0, 0xD0) else 2
        if pos + need > len(data):
            raise MidiFormatError('truncated channel event')
        a = data[pos]
        bb = data[pos + 1] if need == 2 else 0
        pos += need
        if hi == 0x90:
            out.append((tick, 'on' if bb > 0 else 'off', ch, a, bb))
        elif hi == 0x80:
            out.append((tick, 'off', ch, a, bb))
        elif hi == 0xB0:
            out.append((tick, 'cc', ch, a, bb))
        elif hi == 0xC0:
            out.append((tick, 'pc', ch, a, 0))
        # 0xA0 aftertouch, 0xD0 channel pressure, 0xE0 pitch bend: not used by the reference
    return out


def read(filepath) -> List[Instrument]:
    """Parses an SMF file into instruments with times in seconds (see the module docstring for the rules)."""
    with open(filepath, 'rb') as f:
        blob = f.read()
    if len(blob) < 14 or blob[:4] != b'MThd':
        raise MidiFormatError('%s is not a Standard MIDI File (no MThd chunk)' % filepath)
    hlen, fmt, ntrks, division = struct.unpack('>IHHH', blob[4:14])
    if hlen < 6 or fmt not in (0, 1, 2):
        raise MidiFormatError('unsupported SMF header (length %d, format %d)' % (hlen, fmt))
    if division & 0x8000:
        raise MidiFormatError('SMPTE time division is not supported')
    if division == 0:
        raise MidiFormatError('zero ticks per quarter note')
    pos, tracks = 8 + hlen, []
    while pos + 8 <= len(blob) and len(tracks) < ntrks:
        tag, length = blob[pos:pos + 4], struct.unpack('>I', blob[pos + 4:pos + 8])[0]
        body = blob[pos + 8:pos + 8 + length]
        pos += 8 + length
        if tag == b'MTrk':
            tracks.append(_parse_track(body))

    # tempo map: (tick, seconds at that tick, seconds per tick from there on)
    tempos = sorted((t, us) for tr in tracks for (t, kind, _, us, _) in tr if kind == 'tempo')
    segs = [(0, 0.0, 0.5 / division)]
    for t, us in tempos:
        t0, s0, scale = segs[-1]
        s = s0 + (t - t0) * scale
        if t == t0:
            segs[-1] = (t, s, us * 1e-6 / division)
        else:
            segs.append((t, s, us * 1e-6 / division))

    def seconds(tick):
        lo, hi = 0, len(segs) - 1
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if segs[mid][0] <= tick:
                lo = mid
            else:
                hi = mid - 1
        t0, s0, scale = segs[lo]
        return s0 + (tick - t0) * scale

    instruments, order = {}, []
    stragglers = {}                                       # (channel, track) -> control changes without an instrument yet

    def instrument(program, ch, ti):
        key = (program, ch, ti)
        if key not in instruments:
            inst = Instrument(program=program, is_drum=(ch == 9))
            inst.control_changes.extend(stragglers.pop((ch, ti), []))
            instruments[key] = inst
            order.append(key)
        return instruments[key]

    for ti, tr in enumerate(tracks):
        program = [0] * 16
        open_notes = {}
        for tick, kind, ch, a, b in tr:
            if kind == 'pc':
                program[ch] = a
            elif kind == 'on':
                open_notes.setdefault((ch, a), []).append((tick, b))
            elif kind == 'off':
                key = (ch, a)
                if key in open_notes:
                    started = open_notes[key]
                    close = [(t0, v) for (t0, v) in started if t0 != tick]
                    keep = [(t0, v) for (t0, v) in started if t0 == tick]
                    for t0, v in close:
                        instrument(program[ch], ch, ti).notes.append(MidiNote(v, a, seconds(t0), seconds(tick)))
                    if close and keep:
                        open_notes[key] = keep
                    else:
                        del open_notes[key]
            elif kind == 'cc':
                cc = ControlChange(a, b, seconds(tick))
                key = (program[ch], ch, ti)
                if key in instruments:
                    instruments[key].control_changes.append(cc)
                else:
                    stragglers.setdefault((ch, ti), []).append(cc)
    return [instruments[k] for k in order]


# ----------------------------------------------------------------------------------------------- writing
RESOLUTION = 220                      # ticks per quarter note (pretty_midi's default)
TICKS_PER_SECOND = RESOLUTION * 2     # at the 120 bpm this writer declares


def _enc_vlq(v):
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    return bytes(reversed(out))


def _track(events):
    """events: (tick, priority, bytes) -> MTrk chunk with deltas, end of track one tick after the last event."""
    events = sorted(events, key=lambda e: (e[0], e[1]))
    body, last = bytearray(), 0
    for tick, _, payload in events:
        body += _enc_vlq(tick - last) + payload
        last = tick
    body += _enc_vlq(1) + b'\xFF\x2F\x00'
    return b'MTrk' + struct.pack('>I', len(body)) + bytes(body)


def write(filepath, instruments: List[Instrument]):
    """Writes a format-1 file: conductor track + one track per instrument (module docstring for the layout)."""
    tick = lambda s: max(0, int(round(s * TICKS_PER_SECOND)))
    chunks = [_track([(0, 0, b'\xFF\x58\x04\x04\x02\x18\x08'), (0, 1, b'\xFF\x51\x03\x07\xA1\x20')])]
    free = [c for c in range(16) if c != 9]
    for k, inst in enumerate(instruments):
        ch = 9 if inst.is_drum else free[k % len(free)]
        ev = [(0, 5, bytes([0xC0 | ch, inst.program & 0x7F]))]
        for c in inst.control_changes:
            ev.append((tick(c.time), 7, bytes([0xB0 | ch, c.number & 0x7F, c.value & 0x7F])))
        for n in inst.notes:
            ev.append((tick(n.start), 9, bytes([0x90 | ch, n.pitch & 0x7F, n.velocity & 0x7F])))
            ev.append((tick(n.end), 8, bytes([0x90 | ch, n.pitch & 0x7F, 0])))
        chunks.append(_track(ev))
    with open(filepath, 'wb') as f:
        f.write(b'MThd' + struct.pack('>IHHH', 6, 1, len(chunks), RESOLUTION))
        for c in chunks:
            f.write(c)
