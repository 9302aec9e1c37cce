"""CPU: the hand-written protobuf encoders / parsers of the TFRecord, TensorBoard-event and TensorBundle code against an
INDEPENDENT implementation of the wire format -- Google's `protobuf` runtime (in the image) -- with message types declared here
from TensorFlow's published .proto definitions (field names, numbers and types of tensorflow/core/example/{example,feature}.proto,
core/util/event.proto, core/framework/{summary,tensor,tensor_shape,versions}.proto, core/protobuf/tensor_bundle.proto).
TensorFlow itself is not installable here, so this pins the ENCODING of those messages (what bytes a given message is), not
TensorFlow's choice of which messages to write."""
import struct
import numpy as np
import pytest
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

from composer_amd import tbevents, tfrecord, tensorbundle

F = descriptor_pb2.FieldDescriptorProto


def _pool():
    fd = descriptor_pb2.FileDescriptorProto(name="tf_subset.proto", package="tensorflow", syntax="proto3")

    def msg(name, fields, nested=()):
        m = descriptor_pb2.DescriptorProto(name=name)
        for fname, num, typ, label, tname, oneof in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if tname:
                f.type_name = tname
            if oneof is not None:
                f.oneof_index = oneof
        for n in nested:
            m.nested_type.add().CopyFrom(n)
        return m

    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    # feature.proto / example.proto
    fd.message_type.add().CopyFrom(msg("BytesList", [("value", 1, F.TYPE_BYTES, REP, None, None)]))
    fd.message_type.add().CopyFrom(msg("FloatList", [("value", 1, F.TYPE_FLOAT, REP, None, None)]))
    fd.message_type.add().CopyFrom(msg("Int64List", [("value", 1, F.TYPE_INT64, REP, None, None)]))
    feat = msg("Feature", [("bytes_list", 1, F.TYPE_MESSAGE, OPT, ".tensorflow.BytesList", 0),
                           ("float_list", 2, F.TYPE_MESSAGE, OPT, ".tensorflow.FloatList", 0),
                           ("int64_list", 3, F.TYPE_MESSAGE, OPT, ".tensorflow.Int64List", 0)])
    feat.oneof_decl.add(name="kind")
    fd.message_type.add().CopyFrom(feat)
    entry = msg("FeatureEntry", [("key", 1, F.TYPE_STRING, OPT, None, None), ("value", 2, F.TYPE_MESSAGE, OPT, ".tensorflow.Feature", None)])
    entry.options.map_entry = True
    fd.message_type.add().CopyFrom(msg("Features", [("feature", 1, F.TYPE_MESSAGE, REP, ".tensorflow.Features.FeatureEntry", None)], nested=[entry]))
    fd.message_type.add().CopyFrom(msg("Example", [("features", 1, F.TYPE_MESSAGE, OPT, ".tensorflow.Features", None)]))
    # tensor_shape.proto / tensor.proto (the fields used)
    dim = msg("Dim", [("size", 1, F.TYPE_INT64, OPT, None, None), ("name", 2, F.TYPE_STRING, OPT, None, None)])
    fd.message_type.add().CopyFrom(msg("TensorShapeProto", [("dim", 2, F.TYPE_MESSAGE, REP, ".tensorflow.TensorShapeProto.Dim", None),
                                                            ("unknown_rank", 3, F.TYPE_BOOL, OPT, None, None)], nested=[dim]))
    fd.message_type.add().CopyFrom(msg("TensorProto", [("dtype", 1, F.TYPE_INT32, OPT, None, None),
                                                       ("tensor_shape", 2, F.TYPE_MESSAGE, OPT, ".tensorflow.TensorShapeProto", None),
                                                       ("version_number", 3, F.TYPE_INT32, OPT, None, None),
                                                       ("tensor_content", 4, F.TYPE_BYTES, OPT, None, None),
                                                       ("float_val", 5, F.TYPE_FLOAT, REP, None, None)]))
    # summary.proto / event.proto
    plug = msg("PluginData", [("plugin_name", 1, F.TYPE_STRING, OPT, None, None), ("content", 2, F.TYPE_BYTES, OPT, None, None)])
    fd.message_type.add().CopyFrom(msg("SummaryMetadata", [("plugin_data", 1, F.TYPE_MESSAGE, OPT, ".tensorflow.SummaryMetadata.PluginData", None),
                                                           ("display_name", 2, F.TYPE_STRING, OPT, None, None)], nested=[plug]))
    val = msg("Value", [("tag", 1, F.TYPE_STRING, OPT, None, None), ("simple_value", 2, F.TYPE_FLOAT, OPT, None, None),
                        ("tensor", 8, F.TYPE_MESSAGE, OPT, ".tensorflow.TensorProto", None),
                        ("metadata", 9, F.TYPE_MESSAGE, OPT, ".tensorflow.SummaryMetadata", None)])
    fd.message_type.add().CopyFrom(msg("Summary", [("value", 1, F.TYPE_MESSAGE, REP, ".tensorflow.Summary.Value", None)], nested=[val]))
    fd.message_type.add().CopyFrom(msg("Event", [("wall_time", 1, F.TYPE_DOUBLE, OPT, None, None), ("step", 2, F.TYPE_INT64, OPT, None, None),
                                                 ("file_version", 3, F.TYPE_STRING, OPT, None, None),
                                                 ("summary", 5, F.TYPE_MESSAGE, OPT, ".tensorflow.Summary", None)]))
    # versions.proto / tensor_bundle.proto
    fd.message_type.add().CopyFrom(msg("VersionDef", [("producer", 1, F.TYPE_INT32, OPT, None, None), ("min_consumer", 2, F.TYPE_INT32, OPT, None, None)]))
    fd.message_type.add().CopyFrom(msg("BundleHeaderProto", [("num_shards", 1, F.TYPE_INT32, OPT, None, None), ("endianness", 2, F.TYPE_INT32, OPT, None, None),
                                                             ("version", 3, F.TYPE_MESSAGE, OPT, ".tensorflow.VersionDef", None)]))
    fd.message_type.add().CopyFrom(msg("BundleEntryProto", [("dtype", 1, F.TYPE_INT32, OPT, None, None),
                                                            ("shape", 2, F.TYPE_MESSAGE, OPT, ".tensorflow.TensorShapeProto", None),
                                                            ("shard_id", 3, F.TYPE_INT32, OPT, None, None), ("offset", 4, F.TYPE_INT64, OPT, None, None),
                                                            ("size", 5, F.TYPE_INT64, OPT, None, None), ("crc32c", 6, F.TYPE_FIXED32, OPT, None, None)]))
    # trackable_object_graph.proto
    ref = msg("ObjectReference", [("node_id", 1, F.TYPE_INT32, OPT, None, None), ("local_name", 2, F.TYPE_STRING, OPT, None, None)])
    ser = msg("SerializedTensor", [("name", 1, F.TYPE_STRING, OPT, None, None), ("full_name", 2, F.TYPE_STRING, OPT, None, None),
                                   ("checkpoint_key", 3, F.TYPE_STRING, OPT, None, None)])
    slot = msg("SlotVariableReference", [("original_variable_node_id", 1, F.TYPE_INT32, OPT, None, None), ("slot_name", 2, F.TYPE_STRING, OPT, None, None),
                                         ("slot_variable_node_id", 3, F.TYPE_INT32, OPT, None, None)])
    obj = msg("TrackableObject", [("children", 1, F.TYPE_MESSAGE, REP, ".tensorflow.TrackableObjectGraph.TrackableObject.ObjectReference", None),
                                  ("attributes", 2, F.TYPE_MESSAGE, REP, ".tensorflow.TrackableObjectGraph.TrackableObject.SerializedTensor", None),
                                  ("slot_variables", 3, F.TYPE_MESSAGE, REP, ".tensorflow.TrackableObjectGraph.TrackableObject.SlotVariableReference", None)],
              nested=[ref, ser, slot])
    fd.message_type.add().CopyFrom(msg("TrackableObjectGraph", [("nodes", 1, F.TYPE_MESSAGE, REP, ".tensorflow.TrackableObjectGraph.TrackableObject", None)],
                                       nested=[obj]))
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return pool


@pytest.fixture(scope="module")
def M():
    pool = _pool()
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("tensorflow." + n))
    return {n: get(n) for n in ("Example", "Event", "Summary", "TensorProto", "BundleHeaderProto", "BundleEntryProto", "TrackableObjectGraph")}


def test_example_bytes_parse_with_the_protobuf_runtime_and_back(M):
    x = np.arange(12, dtype=np.int32).reshape(3, 4)
    mine = tfrecord.example({"x": tfrecord.bytes_feature(tfrecord.serialize_tensor(x)), "batch_size": tfrecord.int64_feature(7),
                             "model_type": tfrecord.bytes_feature(b"transformer")})
    ex = M["Example"]()
    ex.ParseFromString(mine)
    assert set(ex.features.feature) == {"x", "batch_size", "model_type"}
    assert list(ex.features.feature["batch_size"].int64_list.value) == [7]
    assert ex.features.feature["model_type"].bytes_list.value[0] == b"transformer"
    t = M["TensorProto"]()
    t.ParseFromString(ex.features.feature["x"].bytes_list.value[0])
    assert t.dtype == 3 and [d.size for d in t.tensor_shape.dim] == [3, 4]          # DT_INT32
    assert np.array_equal(np.frombuffer(t.tensor_content, "<i4").reshape(3, 4), x)
    # the other direction: a message serialised by the protobuf runtime goes through the hand-written parsers
    ex2 = M["Example"]()
    ex2.features.feature["window_size"].int64_list.value.append(1024)
    t2 = M["TensorProto"](dtype=3, tensor_content=x.tobytes())
    for d in x.shape:
        t2.tensor_shape.dim.add(size=d)
    ex2.features.feature["y"].bytes_list.value.append(t2.SerializeToString())
    got = tfrecord.parse_example(ex2.SerializeToString())
    assert got["window_size"] == [1024]
    assert np.array_equal(tfrecord.parse_tensor(got["y"][0]), x)
    # deterministic (key-sorted) serialisation of the same content is byte-identical
    ex3 = M["Example"]()
    ex3.ParseFromString(mine)
    assert ex3.SerializeToString(deterministic=True) == mine


def test_event_file_records_parse_with_the_protobuf_runtime(M, tmp_path):
    w = tbevents.EventFileWriter(tmp_path)
    w.scalar("loss", 5.25, 3)
    w.scalar("accuracy", 0.125, 3)
    w.close()
    recs = list(tbevents.read_records(w.path))
    assert len(recs) == 3
    ev0 = M["Event"](); ev0.ParseFromString(recs[0])
    assert ev0.file_version == "brain.Event:2" and ev0.step == 0 and ev0.wall_time > 1e9
    for rec, tag, v in ((recs[1], "loss", 5.25), (recs[2], "accuracy", 0.125)):
        ev = M["Event"](); ev.ParseFromString(rec)
        assert ev.step == 3 and len(ev.summary.value) == 1
        val = ev.summary.value[0]
        assert val.tag == tag and val.metadata.plugin_data.plugin_name == "scalars"
        assert val.tensor.dtype == 1 and len(val.tensor.tensor_shape.dim) == 0 and list(val.tensor.float_val) == [v]
        assert ev.SerializeToString(deterministic=True) == rec          # canonical field order: byte-identical
    # a TF1-style record (simple_value) written by the protobuf runtime is read by the hand-written reader
    ev = M["Event"](wall_time=12.5, step=9)
    ev.summary.value.add(tag="epoch_loss", simple_value=0.75)
    p = tmp_path / "events.out.tfevents.0000000001.host.v2"
    with open(p, "wb") as f:
        f.write(tbevents._record(M["Event"](wall_time=1.0, file_version="brain.Event:2").SerializeToString()))
        f.write(tbevents._record(ev.SerializeToString()))
    version, scalars = tbevents.read_scalars(str(p))
    assert version == "brain.Event:2" and [(t, s, v) for t, s, v, _ in scalars] == [("epoch_loss", 9, 0.75)]


def test_bundle_index_entries_parse_with_the_protobuf_runtime(M, tmp_path):
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    b = np.int64(41)
    tensorbundle.write_bundle(tmp_path / "ckpt-1", {"model/w": a, "step": b})
    items = dict(tensorbundle.read_table(str(tmp_path / "ckpt-1.index")))
    hdr = M["BundleHeaderProto"](); hdr.ParseFromString(items[b""])
    assert hdr.num_shards == 1 and hdr.endianness == 0 and hdr.version.producer == 1
    e = M["BundleEntryProto"](); e.ParseFromString(items[b"model/w"])
    assert e.dtype == 1 and [d.size for d in e.shape.dim] == [2, 3] and e.shard_id == 0 and e.size == 24
    blob = open(tmp_path / "ckpt-1.data-00000-of-00001", "rb").read()
    assert np.array_equal(np.frombuffer(blob[e.offset:e.offset + e.size], "<f4").reshape(2, 3), a)
    assert e.crc32c == tensorbundle._mask(tbevents.crc32c(blob[e.offset:e.offset + e.size]))
    s = M["BundleEntryProto"](); s.ParseFromString(items[b"step"])
    assert s.dtype == 9 and len(s.shape.dim) == 0 and s.size == 8                    # DT_INT64 scalar
    assert struct.unpack("<q", blob[s.offset:s.offset + 8])[0] == 41
    # an index whose entries were serialised by the protobuf runtime is read back by read_bundle
    hdr2 = M["BundleHeaderProto"](num_shards=1); hdr2.version.producer = 1
    e2 = M["BundleEntryProto"](dtype=1, size=24, crc32c=tensorbundle._mask(tbevents.crc32c(a.tobytes())))
    for d in a.shape:
        e2.shape.dim.add(size=d)
    tensorbundle.write_table(str(tmp_path / "ckpt-2.index"), [(b"", hdr2.SerializeToString()), (b"model/w", e2.SerializeToString())])
    open(tmp_path / "ckpt-2.data-00000-of-00001", "wb").write(a.tobytes())
    got = tensorbundle.read_bundle(tmp_path / "ckpt-2")
    assert np.array_equal(got["model/w"], a)


def test_object_graph_parses_with_the_protobuf_runtime_and_is_a_consistent_tree(M):
    """The `_CHECKPOINTABLE_OBJECT_GRAPH` entry of a checkpoint written by `bundle_from_state`: a TrackableObjectGraph whose tree
    spells every checkpoint key from the root, with the Adam slots as slot_variables of the optimizer node."""
    state = {"model/wte/weight": np.zeros((3, 2), np.float32), "model/decoder_blocks/0/attn/c_attn/bias": np.zeros((1, 6), np.float32),
             "optimizer/m/wte/weight": np.zeros((3, 2), np.float32), "optimizer/v/wte/weight": np.zeros((3, 2), np.float32),
             "optimizer/m/decoder_blocks/0/attn/c_attn/bias": np.zeros((1, 6), np.float32),
             "optimizer/v/decoder_blocks/0/attn/c_attn/bias": np.zeros((1, 6), np.float32), "optimizer/iter": np.int64(5)}
    bundle = tensorbundle.bundle_from_state(state, {"step": 6, "epoch": 2, "save_counter": 1})
    g = M["TrackableObjectGraph"]()
    g.ParseFromString(bundle[tensorbundle.OBJECT_GRAPH_KEY])
    assert g.SerializeToString(deterministic=True) == bundle[tensorbundle.OBJECT_GRAPH_KEY]
    # every key reachable by walking local_names from node 0
    paths = {}

    def walk(i, prefix):
        for a in g.nodes[i].attributes:
            paths[a.checkpoint_key] = (prefix, a.name)
        for c in g.nodes[i].children:
            walk(c.node_id, prefix + [c.local_name])
    walk(0, [])
    plain = [k for k in bundle if k.endswith(tensorbundle.VALUE_SUFFIX) and tensorbundle.SLOT_MARK not in k]
    for k in plain:
        assert paths[k] == (k[:-len(tensorbundle.VALUE_SUFFIX)].split("/"), "VARIABLE_VALUE"), k
    # slots: optimizer node -> (variable node, slot name, slot node holding the key)
    opt = next(c.node_id for c in g.nodes[0].children if c.local_name == "optimizer")
    seen = set()
    for sv in g.nodes[opt].slot_variables:
        (attr,) = g.nodes[sv.slot_variable_node_id].attributes
        var_key = next(k for k, (pth, _) in paths.items() if any(a.checkpoint_key == k for a in g.nodes[sv.original_variable_node_id].attributes))
        assert attr.checkpoint_key == var_key[:-len(tensorbundle.VALUE_SUFFIX)] + tensorbundle.SLOT_MARK + "optimizer/" + sv.slot_name + tensorbundle.VALUE_SUFFIX
        seen.add(attr.checkpoint_key)
    assert seen == {k for k in bundle if tensorbundle.SLOT_MARK in k}
    assert sorted(tensorbundle.checkpoint_keys_of_graph(bundle[tensorbundle.OBJECT_GRAPH_KEY])) == sorted(k for k in bundle if k.endswith(tensorbundle.VALUE_SUFFIX))
