"""TFRecord datasets (composer_amd.tfrecord; reference: `composer export-dataset`, cli.py:346-380, and load_tfrecord_dataset,
models/__init__.py:315-374).  TensorFlow is not in this image, so the reader is checked on messages assembled byte by byte
from the proto definitions, and the writer by reading its output back."""
import struct

import numpy as np
import pytest
from click.testing import CliRunner

from composer_amd import cli, dataset as ds, tbevents, tfrecord as tr


def test_hand_assembled_example_and_tensor():
    # Features{feature{"batch_size": Int64List[2] packed}, feature{"n": Int64List[-1] unpacked}, feature{"model_type": BytesList["rnn"]}}
    f_b = b"\x1a\x03" + b"\x0a\x01\x02"
    f_n = b"\x1a\x0b" + b"\x08" + b"\xff" * 9 + b"\x01"
    f_m = b"\x0a\x05" + b"\x0a\x03rnn"
    entry = lambda k, f: b"\x0a" + bytes([2 + len(k) + 2 + len(f)]) + b"\x0a" + bytes([len(k)]) + k + b"\x12" + bytes([len(f)]) + f
    feats = entry(b"model_type", f_m) + entry(b"batch_size", f_b) + entry(b"n", f_n)
    ex = b"\x0a" + bytes([len(feats)]) + feats
    assert tr.parse_example(ex) == {"model_type": [b"rnn"], "batch_size": [2], "n": [-1]}
    # TensorProto{dtype=DT_INT32, shape{dim{size:2} dim{size:3}}, tensor_content = 6 little-endian int32}
    vals = [0, 1, -2, 389, 65536, -2147483648]
    t = b"\x08\x03" + b"\x12\x08" + b"\x12\x02\x08\x02" + b"\x12\x02\x08\x03" + b"\x22\x18" + struct.pack("<6i", *vals)
    assert tr.parse_tensor(t).tolist() == [vals[:3], vals[3:]]
    assert tr.serialize_tensor(np.array(vals, np.int32).reshape(2, 3)) == t
    # the int_val form (field 7, packed) and a wrong dtype
    t7 = b"\x08\x03" + b"\x12\x04" + b"\x12\x02\x08\x02" + b"\x3a\x03\x05\xac\x02"
    assert tr.parse_tensor(t7).tolist() == [5, 300]
    with pytest.raises(ValueError):
        tr.parse_tensor(b"\x08\x01" + t[2:])
    # writer side of the Example: exactly the bytes above for the same features (map entries in key order)
    got = tr.example({"model_type": tr.bytes_feature(b"rnn"), "batch_size": tr.int64_feature(2)})
    feats2 = entry(b"batch_size", f_b) + entry(b"model_type", f_m)
    assert got == b"\x0a" + bytes([len(feats2)]) + feats2


def test_export_then_load_is_the_same_pipeline(tmp_path):
    ds.write_synthetic_data_file(tmp_path / "a.data", 5000, seed=3)
    ids, _ = ds.read_data_file(tmp_path / "a.data")
    src = ds.WindowDataset(ids, batch_size=4, window_size=16, shuffle=False)
    out = tmp_path / "train.tfrecord"
    n = tr.export_dataset(src, out)
    assert n == len(src) == len(ids) // 17 // 4
    recs = list(tbevents.read_records(out))
    assert len(recs) == n + 1
    assert tr.parse_example(recs[0]) == {"model_type": [b"transformer"], "batch_size": [4], "window_size": [16]}
    loaded, header = tr.load_tfrecord_dataset(out, shuffle=False)
    assert header == {"model_type": "transformer", "batch_size": 4, "window_size": 16} and len(loaded) == n
    for (x0, y0), (x1, y1) in zip(src, loaded):
        assert x1.dtype == np.int32 and np.array_equal(x0, x1) and np.array_equal(y0, y1)
    # shuffled passes: every batch exactly once, a different order each pass, the same order for the same seed
    a, _ = tr.load_tfrecord_dataset(out, shuffle=True, seed=7)
    b, _ = tr.load_tfrecord_dataset(out, shuffle=True, seed=7)
    p1 = [x.tobytes() for x, _ in a]
    p2 = [x.tobytes() for x, _ in a]
    assert sorted(p1) == sorted(x.tobytes() for x, _ in src) and p1 != p2 and p1 == [x.tobytes() for x, _ in b]
    # two ranks split a pass without overlap
    r0, _ = tr.load_tfrecord_dataset(out, shuffle=True, seed=7, rank=0, world_size=2)
    r1, _ = tr.load_tfrecord_dataset(out, shuffle=True, seed=7, rank=1, world_size=2)
    q0, q1 = [x.tobytes() for x, _ in r0], [x.tobytes() for x, _ in r1]
    assert len(q0) == len(q1) == n // 2 and not set(q0) & set(q1)
    assert [v for pair in zip(q0, q1) for v in pair] == p1[:2 * (n // 2)]


def test_corrupt_and_foreign_files(tmp_path):
    p = tmp_path / "x.tfrecord"
    p.write_bytes(b"")
    with pytest.raises(ValueError):
        tr.load_tfrecord_dataset(p)
    tr.export_dataset([(np.zeros((2, 3), np.int32), np.ones((2, 3), np.int32))], p)
    blob = bytearray(p.read_bytes())
    blob[-6] ^= 0x40
    p.write_bytes(bytes(blob))
    with pytest.raises(ValueError):
        tr.load_tfrecord_dataset(p)
    with pytest.raises(ValueError):
        tr.export_dataset([], p)


def test_cli_export_dataset_and_get_dataset(tmp_path):
    (tmp_path / "set" / "train").mkdir(parents=True)
    ds.write_synthetic_data_file(tmp_path / "set" / "train" / "a.data", 40000, seed=1)
    cfg = tmp_path / "c.yml"
    text = open(cli.get_default_config()).read().replace("batch_size: 1 ", "batch_size: 2 ").replace("window_size: 1024", "window_size: 64")
    cfg.write_text(text)
    config = cli.cfgmod.get(str(cfg))
    assert (config.transformer.train.batch_size, config.transformer.model.window_size) == (2, 64)
    out = tmp_path / "train.tfrecord"
    r = CliRunner().invoke(cli.cli, ["export-dataset", "transformer", str(tmp_path / "set" / "train"), str(out), "-c", str(cfg)])
    assert r.exit_code == 0, r.output
    got = cli.get_dataset(cli.ModelType.TRANSFORMER, out, config, "train", shuffle_dataset=False)
    want = cli.get_dataset(cli.ModelType.TRANSFORMER, tmp_path / "set", config, "train", shuffle_dataset=False)
    assert len(got) == len(want) > 100
    for (x0, y0), (x1, y1) in zip(want, got):
        assert np.array_equal(x0, x1) and np.array_equal(y0, y1)
    # a file exported under another config is refused (cli.py:258-268)
    cfg.write_text(text.replace("batch_size: 2 ", "batch_size: 4 "))
    with pytest.raises(SystemExit):
        cli.get_dataset(cli.ModelType.TRANSFORMER, out, cli.cfgmod.get(str(cfg)), "train")
    with pytest.raises(SystemExit):
        cli.get_dataset(cli.ModelType.TRANSFORMER, tmp_path / "nope.bin", config, "train")
