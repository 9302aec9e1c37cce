"""TensorBoard event files written by composer_amd.tbevents (reference: tf.summary scalars, transformer.py:903,933-951).
TensorFlow is not in this image: the checks are CRC-32C known answers (RFC 3720 B.4), a record assembled by hand from
the TFRecord / Event / Summary definitions, and write -> read round trips."""
import glob
import struct

import pytest

from composer_amd import checkpoint as ckpt
from composer_amd import tbevents as tb


def test_crc32c_known_answers():
    assert tb.crc32c(b"123456789") == 0xE3069283
    assert tb.crc32c(bytes(32)) == 0x8A9136AA                       # RFC 3720 B.4
    assert tb.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert tb.crc32c(bytes(range(32))) == 0x46DD794E
    assert tb.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert tb.crc32c(b"") == 0
    c = tb.crc32c(b"123456789")
    assert tb.masked_crc32c(b"123456789") == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def test_hand_assembled_record_is_read(tmp_path):
    # Event{wall_time=2.0, step=300, summary{value{tag="loss", simple_value=0.5}}} in the TF1 simple_value form, byte by byte
    value = b"\x0a\x04loss" + b"\x15" + struct.pack("<f", 0.5)
    summary = b"\x0a" + bytes([len(value)]) + value
    event = b"\x09" + struct.pack("<d", 2.0) + b"\x10\xac\x02" + b"\x2a" + bytes([len(summary)]) + summary
    head = struct.pack("<Q", len(event))
    blob = head + struct.pack("<I", tb.masked_crc32c(head)) + event + struct.pack("<I", tb.masked_crc32c(event))
    p = tmp_path / "events.out.tfevents.0.x"
    p.write_bytes(blob)
    version, scalars = tb.read_scalars(p)
    assert version is None and scalars == [("loss", 300, 0.5, 2.0)]
    bad = bytearray(blob)
    bad[20] ^= 1
    p.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        tb.read_scalars(p)


def test_writer_layout_and_round_trip(tmp_path):
    w = tb.EventFileWriter(tmp_path)
    vals = [("loss", 3.25, 1), ("accuracy", 0.125, 1), ("loss", 1e-3, 2 ** 40), ("epoch_loss", -7.5, 0)]
    for name, v, step in vals:
        w.scalar(name, v, step)
    w.close()
    (path,) = glob.glob(str(tmp_path / "events.out.tfevents.*.v2"))
    recs = list(tb.read_records(path))
    assert len(recs) == 5
    # first record: wall_time, step 0, file_version "brain.Event:2"
    assert recs[0][0] == 0x09 and recs[0][9:] == b"\x10\x00\x1a\x0dbrain.Event:2"
    # a scalar record: Event.summary(5) > Summary.value(1) > {tag(1), tensor(8){dtype(1)=DT_FLOAT, shape(2)={}, float_val(5) packed}, metadata(9)}
    body = recs[1]
    assert body[9:11] == b"\x10\x01" and body[11] == 0x2a
    assert b"\x0a\x04loss" in body and b"\x0a\x07scalars" in body
    assert b"\x08\x01\x12\x00\x2a\x04" + struct.pack("<f", 3.25) in body
    version, scalars = tb.read_scalars(path)
    assert version == "brain.Event:2"
    assert [(t, s) for t, s, _, _ in scalars] == [(n, s) for n, _, s in vals]
    for (_, v, _), (_, _, got, wall) in zip(vals, scalars):
        assert got == struct.unpack("<f", struct.pack("<f", v))[0] and wall > 1e9


def test_scalar_log_writes_both_files(tmp_path):
    s = ckpt.ScalarLog(tmp_path / "train")
    s.scalar("loss", 1.5, 1)
    s.scalar("epoch_accuracy", 0.25, 2)
    s.close()
    (path,) = glob.glob(str(tmp_path / "train" / "events.out.tfevents.*"))
    _, scalars = tb.read_scalars(path)
    assert [(t, st, v) for t, st, v, _ in scalars] == [("loss", 1, 1.5), ("epoch_accuracy", 2, 0.25)]
    assert (tmp_path / "train" / "scalars.jsonl").read_text().count("\n") == 2
