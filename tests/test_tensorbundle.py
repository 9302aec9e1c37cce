"""TensorBundle checkpoints (composer_amd.tensorbundle; reference: tf.train.Checkpoint / CheckpointManager, transformer.py:890-896,
941-955).  TensorFlow is not in this image and the reference repository holds no checkpoint, so the checks are: a table
block, footer and bundle entry assembled byte by byte from the format definitions and read by the module; files written by
the module re-read; the directory contract through CheckpointManager.  (Parity with TensorFlow itself: unpinned.)"""
import os
import struct

import numpy as np
import pytest

from composer_amd import checkpoint as ckpt
from composer_amd import tensorbundle as tb
from composer_amd.tbevents import crc32c


def _trailer(block):
    return b"\x00" + struct.pack("<I", tb._mask(crc32c(block + b"\x00")))


def test_hand_assembled_index_and_data(tmp_path):
    # data file: one float32 [2] tensor then one int64 scalar
    w = struct.pack("<2f", 1.5, -2.0)
    it = struct.pack("<q", 7)
    (tmp_path / "ckpt-1.data-00000-of-00001").write_bytes(w + it)
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"                                       # num_shards=1, version{producer=1}
    e_w = b"\x08\x01" + b"\x12\x04\x12\x02\x08\x02" + b"\x28\x08" + b"\x35" + struct.pack("<I", tb._mask(crc32c(w)))
    e_it = b"\x08\x09" + b"\x12\x00" + b"\x20\x08" + b"\x28\x08" + b"\x35" + struct.pack("<I", tb._mask(crc32c(it)))
    k1, k2 = b"model/w/.ATTRIBUTES/VARIABLE_VALUE", b"model/w2/.ATTRIBUTES/VARIABLE_VALUE"
    # one data block, prefix compression on the third key (shares "model/w" with the second)
    ent = lambda shared, key, val: bytes([shared, len(key) - shared, len(val)]) + key[shared:] + val
    block = ent(0, b"", header) + ent(0, k1, e_w) + ent(7, k2, e_it) + struct.pack("<I", 0) + struct.pack("<I", 1)
    meta = struct.pack("<II", 0, 1)
    off_meta = len(block) + 5
    off_index = off_meta + len(meta) + 5
    handle = lambda o, n: tb._varint(o) + tb._varint(n)
    iv = handle(0, len(block))
    index = bytes([0, len(k2), len(iv)]) + k2 + iv + struct.pack("<II", 0, 1)
    footer = handle(off_meta, len(meta)) + handle(off_index, len(index))
    footer += bytes(40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    (tmp_path / "ckpt-1.index").write_bytes(block + _trailer(block) + meta + _trailer(meta) + index + _trailer(index) + footer)
    got = tb.read_bundle(tmp_path / "ckpt-1")
    assert sorted(got) == [k1.decode(), k2.decode()]
    assert got[k1.decode()].tolist() == [1.5, -2.0] and got[k1.decode()].dtype == np.float32
    assert got[k2.decode()].shape == () and int(got[k2.decode()]) == 7
    # the writer produces the same bytes for the same content
    tb.write_bundle(tmp_path / "mine", {k1.decode(): np.array([1.5, -2.0], np.float32), k2.decode(): np.int64(7)})
    assert (tmp_path / "mine.index").read_bytes() == (tmp_path / "ckpt-1.index").read_bytes()
    assert (tmp_path / "mine.data-00000-of-00001").read_bytes() == w + it
    # corruption is caught: one bit in the data file, one bit in the index
    blob = bytearray(w + it)
    blob[1] ^= 4
    (tmp_path / "ckpt-1.data-00000-of-00001").write_bytes(bytes(blob))
    with pytest.raises(ValueError, match="checksum"):
        tb.read_bundle(tmp_path / "ckpt-1")
    idx = bytearray((tmp_path / "mine.index").read_bytes())
    idx[10] ^= 1
    (tmp_path / "mine.index").write_bytes(bytes(idx))
    with pytest.raises(ValueError, match="checksum"):
        tb.read_bundle(tmp_path / "mine")


def test_round_trip_dtypes_strings_and_many_blocks(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {"a/%04d" % i: rng.standard_normal((3, 5)).astype(np.float32) for i in range(40)}
    tensors["big"] = rng.standard_normal((300, 257)).astype(np.float32)
    tensors["i32"] = np.arange(-3, 4, dtype=np.int32)
    tensors["flag"] = np.array([True, False])
    tensors["half"] = np.array([0.5, -1.25], np.float16)
    tensors["bf"] = (np.array([0x3F80, 0xC000], np.uint16), "bfloat16")              # 1.0, -2.0
    tensors["empty"] = np.zeros((0, 4), np.float32)
    tensors["s0"] = b"scalar string"
    tensors["s3"] = [b"", b"x" * 300, b"\x00\xff"]
    old = tb.BLOCK_SIZE
    tb.BLOCK_SIZE = 512                                                              # force several data blocks
    try:
        tb.write_bundle(tmp_path / "c", tensors)
    finally:
        tb.BLOCK_SIZE = old
    keys = [k for k, _ in tb.read_table(tmp_path / "c.index")]
    assert keys == sorted(keys) and keys[0] == b"" and len(keys) == len(tensors) + 1
    got = tb.read_bundle(tmp_path / "c")
    for k, v in tensors.items():
        if k == "bf":
            assert got[k].dtype == np.float32 and got[k].tolist() == [1.0, -2.0]
        elif isinstance(v, (bytes, list)):
            assert got[k] == v
        else:
            assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v)
    assert os.path.getsize(tmp_path / "c.data-00000-of-00001") > 300 * 257 * 4
    with pytest.raises(ValueError):
        tb.write_table(tmp_path / "bad", [(b"b", b""), (b"a", b"")])
    with pytest.raises(ValueError):
        (tmp_path / "junk.index").write_bytes(b"not a table" * 10)
        tb.read_table(tmp_path / "junk.index")


def test_string_tensor_layout():
    # [varint lengths][masked crc32c of the lengths as little-endian uint32][bytes]
    payload, _ = tb._encode_strings([b"ab", b"", b"xyz"])
    lens = struct.pack("<III", 2, 0, 3)
    assert payload == b"\x02\x00\x03" + struct.pack("<I", tb._mask(crc32c(lens))) + b"abxyz"
    assert tb._crc_extend(crc32c(b"1234"), b"56789") == crc32c(b"123456789") == 0xE3069283
    assert tb._unmask(tb._mask(0xDEADBEEF)) == 0xDEADBEEF


def test_checkpoint_names_and_object_graph(tmp_path):
    params = ["wte/weight", "decoder_blocks/0/attn/c_attn/bias", "ln_f/gamma"]
    sd = {}
    for i, p in enumerate(params):
        sd["model/" + p] = np.full((2, 2), i, np.float32)
        sd["optimizer/m/" + p] = np.full((2, 2), 10 + i, np.float32)
        sd["optimizer/v/" + p] = np.full((2, 2), 20 + i, np.float32)
    sd["optimizer/iter"] = np.int64(41)
    bundle = tb.bundle_from_state(sd, {"step": 42, "epoch": 3, "save_counter": 5})
    want = {"model/wte/weight/.ATTRIBUTES/VARIABLE_VALUE", "model/wte/weight/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE",
            "model/decoder_blocks/0/attn/c_attn/bias/.OPTIMIZER_SLOT/optimizer/v/.ATTRIBUTES/VARIABLE_VALUE",
            "optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE", "step/.ATTRIBUTES/VARIABLE_VALUE", "epoch/.ATTRIBUTES/VARIABLE_VALUE",
            "save_counter/.ATTRIBUTES/VARIABLE_VALUE", "_CHECKPOINTABLE_OBJECT_GRAPH"}
    assert want <= set(bundle) and len(bundle) == 3 * 3 + 1 + 3 + 1
    # every tensor key is named by exactly one SerializedTensor of the graph
    named = tb.checkpoint_keys_of_graph(bundle["_CHECKPOINTABLE_OBJECT_GRAPH"])
    assert sorted(named) == sorted(k for k in bundle if k != "_CHECKPOINTABLE_OBJECT_GRAPH")
    back, meta = tb.state_from_bundle(bundle)
    assert meta == {"step": 42, "epoch": 3, "save_counter": 5} and sorted(back) == sorted(sd)
    assert all(np.array_equal(back[k], sd[k]) for k in sd)

    # through the manager: file names, rotation, pointer file, restore of either format
    m = ckpt.CheckpointManager(tmp_path, max_to_keep=2, format="tensorbundle")
    for step in (1, 2, 3):
        m.save(sd, {"step": step, "epoch": 1})
    assert sorted(os.listdir(tmp_path)) == ["checkpoint", "ckpt-2.data-00000-of-00001", "ckpt-2.index",
                                            "ckpt-3.data-00000-of-00001", "ckpt-3.index"]
    assert 'model_checkpoint_path: "ckpt-3"' in (tmp_path / "checkpoint").read_text()
    m2 = ckpt.CheckpointManager(tmp_path, max_to_keep=2)                              # a resumed run: finds them, switches to npz
    assert m2.latest_checkpoint.endswith("ckpt-3")
    t, meta = ckpt.load(m2.latest_checkpoint)
    assert meta == {"step": 3, "epoch": 1, "save_counter": 3} and int(t["optimizer/iter"]) == 41
    assert np.array_equal(t["optimizer/v/ln_f/gamma"], sd["optimizer/v/ln_f/gamma"])
    assert os.path.basename(m2.save(sd, {"step": 4, "epoch": 1})) == "ckpt-4"
    assert sorted(os.listdir(tmp_path)) == ["checkpoint", "ckpt-3.data-00000-of-00001", "ckpt-3.index", "ckpt-4.npz"]
    with pytest.raises(ValueError):
        ckpt.CheckpointManager(tmp_path, format="hdf5")
