"""TensorFlow object-graph checkpoint reader / writer (row f3): table format, tensor bundle, object graph and the
FACT variable-path mapping.  No TensorFlow exists in this image, so the reader is checked against this repo's own
writer and against hand-assembled graphs that follow the Keras tracking rules (mint_amd/tf_checkpoint.py header)."""
import os
import struct

import numpy as np
import pytest

from mint_amd import tf_checkpoint as T
from mint_amd.tfrecord import _enc_varint, _masked
from oracle import fact_oracle as O


def test_table_roundtrip_many_blocks_and_prefix_compression(tmp_path):
    rng = np.random.RandomState(0)
    items = [(b"", b"hdr")]
    for i in range(400):
        items.append((("model/layer_with_weights-%03d/fn/kernel/.ATTRIBUTES/VARIABLE_VALUE" % i).encode(),
                      bytes(rng.randint(0, 256, size=rng.randint(1, 60)).astype(np.uint8))))
    path = str(tmp_path / "t.index")
    T.write_table(path, items, block_bytes=512, restart_interval=4)
    assert T.read_table(path) == sorted(items)
    raw = open(path, "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == T.TABLE_MAGIC
    with pytest.raises(ValueError):
        open(str(tmp_path / "bad"), "wb").write(raw[:-1] + b"\x00")
        T.read_table(str(tmp_path / "bad"))


def test_snappy_compressed_block_is_decoded(tmp_path):
    pa = pytest.importorskip("pyarrow")
    # one data block, snappy-compressed by an independent encoder, wrapped with a hand-built index block + footer
    entries = [(b"a/key", b"v1" * 40), (b"a/key2", b"v2" * 40)]
    path = str(tmp_path / "plain.index")
    T.write_table(path, entries)
    raw = open(path, "rb").read()
    plain = T.read_table(path)
    # re-assemble: [snappy(data block)][type 1][crc] [meta][index][footer]
    def block(entries_):
        blk, prev = bytearray(), b""
        for k, v in entries_:
            blk += _enc_varint(0) + _enc_varint(len(k)) + _enc_varint(len(v)) + k + v
        blk += struct.pack("<I", 0) + struct.pack("<I", 1)
        return bytes(blk)
    data = block(entries)
    comp = pa.compress(data, codec="snappy", asbytes=True)
    out = bytearray(comp + b"\x01" + struct.pack("<I", _masked(comp + b"\x01")))
    meta = block([])
    moff = len(out)
    out += meta + b"\x00" + struct.pack("<I", _masked(meta + b"\x00"))
    idx = block([(b"a/key2", _enc_varint(0) + _enc_varint(len(comp)))])
    ioff = len(out)
    out += idx + b"\x00" + struct.pack("<I", _masked(idx + b"\x00"))
    footer = _enc_varint(moff) + _enc_varint(len(meta)) + _enc_varint(ioff) + _enc_varint(len(idx))
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", T.TABLE_MAGIC)
    p2 = str(tmp_path / "snappy.index")
    open(p2, "wb").write(bytes(out))
    assert T.read_table(p2) == plain == entries
    assert raw  # (plain file kept for reference)


def test_bundle_roundtrip_dtypes_and_crc(tmp_path):
    rng = np.random.RandomState(1)
    tensors = {"a/f32": rng.randn(7, 5).astype(np.float32), "b/i64": np.asarray(123456789012, dtype=np.int64),
               "c/vec": rng.randn(3000).astype(np.float32), "d/i32": np.arange(6, dtype=np.int32).reshape(2, 3)}
    prefix = str(tmp_path / "ckpt-1")
    T.write_bundle(prefix, tensors, {"_S": b"hello \x00 world"})
    rd = T.TensorBundleReader(prefix)
    assert rd.keys() == sorted(list(tensors) + ["_S"])
    for k, v in tensors.items():
        got = rd.get(k, verify_crc=True)
        assert got.dtype == v.dtype and got.shape == v.shape
        np.testing.assert_array_equal(got, v)
    assert rd.get("_S") == b"hello \x00 world"
    # flip one data byte: the crc check must notice
    dpath = prefix + ".data-00000-of-00001"
    raw = bytearray(open(dpath, "rb").read())
    raw[rd.entries["c/vec"]["offset"] + 17] ^= 0x40
    open(dpath, "wb").write(bytes(raw))
    with pytest.raises(IOError):
        T.TensorBundleReader(prefix).get("c/vec", verify_crc=True)


def test_variable_paths_follow_the_reference_classes():
    names = [n for n, _ in O.param_shapes(O.FACT_V5_CFG)]
    assert len(names) == 184
    paths = ["/".join(T.tf_variable_path(n)) for n in names]
    assert len(set(paths)) == len(paths)
    P = dict(zip(names, paths))
    # Transformer.net is a Sequential of [Residual(Norm(Attention)), Residual(Norm(MLP))] x L (base_models.py:94-107)
    assert P["cross_modal_layer/transformer/layer_0/attn/to_qkv/kernel"] == \
        "cross_modal_layer/transformer_layer/net/layer_with_weights-0/fn/fn/to_qkv/kernel"
    assert P["cross_modal_layer/transformer/layer_11/mlp/dense_2/bias"] == \
        "cross_modal_layer/transformer_layer/net/layer_with_weights-23/fn/fn/net/layer_with_weights-1/bias"
    assert P["motion_transformer/layer_1/attn_norm/gamma"] == "motion_transformer/net/layer_with_weights-2/fn/norm/gamma"
    assert P["audio_transformer/layer_0/mlp_norm/beta"] == "audio_transformer/net/layer_with_weights-1/fn/norm/beta"
    assert P["cross_modal_layer/output/kernel"] == "cross_modal_layer/cross_output_layer/kernel"
    assert P["motion_pos_embedding/position_embedding"] == "motion_pos_embedding/pos_embedding"
    assert P["audio_linear_embedding/bias"] == "audio_linear_embedding/net/bias"
    with pytest.raises(KeyError):
        T.tf_variable_path("cross_modal_layer/transformer/layer_0/attn/to_k/kernel")


def _tiny_state(seed=0):
    rng = np.random.RandomState(seed)
    shapes = dict(O.param_shapes(O.TINY_CFG))
    mk = lambda: {n: rng.randn(*s).astype(np.float32) for n, s in shapes.items()}
    return shapes, mk(), mk(), {n: np.abs(v) for n, v in mk().items()}


def test_fact_checkpoint_roundtrip_with_adam_slots(tmp_path):
    shapes, params, m, v = _tiny_state()
    prefix = str(tmp_path / "ckpt-1000")
    T.write_fact_checkpoint(prefix, params, m, v, iterations=1000)
    assert T.latest_checkpoint(str(tmp_path)) == prefix
    ck = T.read_fact_checkpoint(prefix, list(shapes), shapes, verify_crc=True)
    assert ck["iterations"] == 1000
    for n in shapes:
        np.testing.assert_array_equal(ck["params"][n], params[n])
        np.testing.assert_array_equal(ck["adam_m"][n], m[n])
        np.testing.assert_array_equal(ck["adam_v"][n], v[n])
    # keys carry the attribute path the reference's objects produce
    rd = T.TensorBundleReader(prefix)
    assert "model/cross_modal_layer/transformer_layer/net/layer_with_weights-0/fn/fn/to_qkv/kernel" + T.VAR_SUFFIX in rd.entries
    assert "optimizer/iter" + T.VAR_SUFFIX in rd.entries
    assert any(k.endswith("/.OPTIMIZER_SLOT/optimizer/m" + T.VAR_SUFFIX) for k in rd.entries)
    # weights-only checkpoint (evaluator): no slots, no counter
    p2 = str(tmp_path / "w" / "ckpt-5")
    T.write_fact_checkpoint(p2, params)
    ck2 = T.read_fact_checkpoint(p2, list(shapes))
    assert ck2["adam_m"] is None and ck2["iterations"] is None
    # a missing variable is reported by name, a wrong shape too
    bad = dict(params)
    del bad["audio_linear_embedding/bias"]
    T.write_fact_checkpoint(str(tmp_path / "bad"), bad, update_state_file=False)
    with pytest.raises(KeyError, match="audio_linear_embedding/bias"):
        T.read_fact_checkpoint(str(tmp_path / "bad"), list(shapes))
    wrong = dict(shapes)
    wrong["audio_linear_embedding/bias"] = (7,)
    with pytest.raises(ValueError, match="audio_linear_embedding/bias"):
        T.read_fact_checkpoint(prefix, list(shapes), wrong)


def test_reader_follows_graph_edges_not_key_strings(tmp_path):
    """Keras names a variable's checkpoint key after the FIRST path that reached it; a Sequential exposes every layer
    under both `layer_with_weights-k` and `layer-j`.  The reader must resolve variables through the child edges of
    the object graph, whatever string the key happens to be."""
    shapes, params, _, _ = _tiny_state(3)
    g = T.ObjectGraph()
    g.add_path([])
    tensors = {}
    for i, (n, a) in enumerate(params.items()):
        key = "model/some/other/first-discovery/path-%d%s" % (i, T.VAR_SUFFIX)  # NOT the path we walk
        nid = g.add_path(["model"] + T.tf_variable_path(n), key)
        tensors[key] = a
    # alias edges like Keras' `layer-j`: same nodes reachable under a second name
    net = g.walk(["model", "cross_modal_layer", "transformer_layer", "net"])
    for j, (name, nid) in enumerate(list(g.nodes[net]["children"].items())):
        g.nodes[net]["children"]["layer-%d" % j] = nid
    gs = g.add_path(["global_step"], "global_step" + T.VAR_SUFFIX)
    tensors["global_step" + T.VAR_SUFFIX] = np.asarray(777, dtype=np.int64)
    prefix = str(tmp_path / "ckpt-777")
    T.write_bundle(prefix, tensors, {T.OBJECT_GRAPH_KEY: g.serialize()})
    ck = T.read_fact_checkpoint(prefix, list(shapes), shapes)
    assert ck["global_step"] == 777 and ck["iterations"] is None
    for n in shapes:
        np.testing.assert_array_equal(ck["params"][n], params[n])


def _official_bundle_messages():
    """BundleHeaderProto / BundleEntryProto / TrackableObjectGraph built with the OFFICIAL protobuf runtime from the field
    numbers and types of tensorflow/core/protobuf/{tensor_bundle,trackable_object_graph}.proto and
    framework/{tensor_shape,versions}.proto (restated here; TensorFlow itself is not installed)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="mint_amd_test_tf_bundle.proto", package="tfb", syntax="proto3")

    def msg(parent, name, fields):
        m = parent.message_type.add() if isinstance(parent, descriptor_pb2.FileDescriptorProto) else parent.nested_type.add()
        m.name = name
        for fname, num, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = tname
        return m
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    shape = msg(fd, "TensorShapeProto", [("dim", 2, F.TYPE_MESSAGE, REP, ".tfb.TensorShapeProto.Dim"),
                                         ("unknown_rank", 3, F.TYPE_BOOL, OPT, None)])
    msg(shape, "Dim", [("size", 1, F.TYPE_INT64, OPT, None), ("name", 2, F.TYPE_STRING, OPT, None)])
    msg(fd, "VersionDef", [("producer", 1, F.TYPE_INT32, OPT, None), ("min_consumer", 2, F.TYPE_INT32, OPT, None),
                           ("bad_consumers", 3, F.TYPE_INT32, REP, None)])
    msg(fd, "BundleHeaderProto", [("num_shards", 1, F.TYPE_INT32, OPT, None), ("endianness", 2, F.TYPE_INT32, OPT, None),
                                  ("version", 3, F.TYPE_MESSAGE, OPT, ".tfb.VersionDef")])
    msg(fd, "BundleEntryProto", [("dtype", 1, F.TYPE_INT32, OPT, None),
                                 ("shape", 2, F.TYPE_MESSAGE, OPT, ".tfb.TensorShapeProto"),
                                 ("shard_id", 3, F.TYPE_INT32, OPT, None), ("offset", 4, F.TYPE_INT64, OPT, None),
                                 ("size", 5, F.TYPE_INT64, OPT, None), ("crc32c", 6, F.TYPE_FIXED32, OPT, None)])
    graph = msg(fd, "TrackableObjectGraph", [("nodes", 1, F.TYPE_MESSAGE, REP, ".tfb.TrackableObjectGraph.TrackableObject")])
    obj = msg(graph, "TrackableObject", [
        ("children", 1, F.TYPE_MESSAGE, REP, ".tfb.TrackableObjectGraph.TrackableObject.ObjectReference"),
        ("attributes", 2, F.TYPE_MESSAGE, REP, ".tfb.TrackableObjectGraph.TrackableObject.SerializedTensor"),
        ("slot_variables", 3, F.TYPE_MESSAGE, REP, ".tfb.TrackableObjectGraph.TrackableObject.SlotVariableReference")])
    msg(obj, "ObjectReference", [("node_id", 1, F.TYPE_INT32, OPT, None), ("local_name", 2, F.TYPE_STRING, OPT, None)])
    msg(obj, "SerializedTensor", [("name", 1, F.TYPE_STRING, OPT, None), ("full_name", 2, F.TYPE_STRING, OPT, None),
                                  ("checkpoint_key", 3, F.TYPE_STRING, OPT, None)])
    msg(obj, "SlotVariableReference", [("original_variable_node_id", 1, F.TYPE_INT32, OPT, None),
                                       ("slot_name", 2, F.TYPE_STRING, OPT, None),
                                       ("slot_variable_node_id", 3, F.TYPE_INT32, OPT, None)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("tfb." + n))
    return get("BundleHeaderProto"), get("BundleEntryProto"), get("TrackableObjectGraph")


def test_written_bundle_parses_with_the_official_protobuf_runtime(tmp_path):
    """The hand-rolled encoders of mint_amd/tf_checkpoint.py against google.protobuf: every index record of a written
    checkpoint parses as BundleHeaderProto / BundleEntryProto, the object graph parses as TrackableObjectGraph, and walking
    the OFFICIAL message's edges reaches the same checkpoint keys, slot variables and data-file extents our reader uses."""
    pytest.importorskip("google.protobuf")
    Header, Entry, Graph = _official_bundle_messages()
    shapes, params, m, v = _tiny_state(3)
    prefix = str(tmp_path / "ckpt-7")
    T.write_fact_checkpoint(prefix, params, m, v, iterations=7)
    table = dict(T.read_table(prefix + ".index"))
    hdr = Header()
    hdr.ParseFromString(table[b""])
    assert hdr.num_shards == 1 and hdr.endianness == 0 and hdr.version.producer >= 1
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    rd = T.TensorBundleReader(prefix)
    n_entries = 0
    for key, val in table.items():
        if key == b"":
            continue
        e = Entry()
        e.ParseFromString(val)
        assert e.SerializeToString() == val or Entry.FromString(e.SerializeToString()) == e   # canonical or equivalent
        assert e.shard_id == 0 and 0 <= e.offset and e.offset + e.size <= len(data)
        dims = tuple(d.size for d in e.shape.dim)
        if e.dtype == T.DT_FLOAT:
            assert e.size == 4 * int(np.prod(dims, dtype=np.int64))
            arr = np.frombuffer(data, "<f4", count=e.size // 4, offset=e.offset).reshape(dims)
            np.testing.assert_array_equal(arr, rd.get(key.decode(), verify_crc=True))
        # crc32c field: masked crc of the tensor bytes, as our reader checks it
        from mint_amd.tfrecord import _masked
        if e.dtype != T.DT_STRING:
            assert e.crc32c == _masked(data[e.offset:e.offset + e.size])
        n_entries += 1
    assert n_entries == 3 * len(shapes) + 1 + 1   # params + m + v, optimizer/iter, the object graph
    g = Graph()
    g.ParseFromString(rd.get(T.OBJECT_GRAPH_KEY))
    assert len(g.nodes) > len(shapes)

    def walk(path):
        nid = 0
        for part in path:
            nxt = [c.node_id for c in g.nodes[nid].children if c.local_name == part]
            assert len(nxt) == 1, (path, part)
            nid = nxt[0]
        return nid
    ours = T.ObjectGraph(rd.get(T.OBJECT_GRAPH_KEY))
    for name in shapes:
        path = ["model"] + T.tf_variable_path(name)
        nid = walk(path)
        keys = [a.checkpoint_key for a in g.nodes[nid].attributes if a.name == "VARIABLE_VALUE"]
        assert len(keys) == 1 and keys[0].endswith(T.VAR_SUFFIX)
        assert ours.walk(path) == nid and ours.variable_key(nid) == keys[0]
        np.testing.assert_array_equal(rd.get(keys[0]), params[name])
    # Adam slots hang off the optimizer node and point at (variable node, slot node) pairs
    opt = g.nodes[walk(["optimizer"])]
    slots = {(s.original_variable_node_id, s.slot_name): s.slot_variable_node_id for s in opt.slot_variables}
    assert len(slots) == 2 * len(shapes)
    name = list(shapes)[5]
    vid = walk(["model"] + T.tf_variable_path(name))
    for slot_name, ref in (("m", m), ("v", v)):
        sid = slots[(vid, slot_name)]
        key = [a.checkpoint_key for a in g.nodes[sid].attributes if a.name == "VARIABLE_VALUE"][0]
        np.testing.assert_array_equal(rd.get(key), ref[name])
    it = [a.checkpoint_key for a in g.nodes[walk(["optimizer", "iter"])].attributes][0]
    assert int(rd.get(it).reshape(-1)[0]) == 7


def test_string_tensor_checksums_follow_bundle_writer_rule(tmp_path):
    """TensorFlow's BundleWriter (tensor_bundle.cc WriteStringTensor) checksums the element sizes of a DT_STRING tensor
    as FIXED-WIDTH uint32, not as the varints it stores, then extends the entry crc over the 4 bytes of the masked length
    checksum and over the string bytes; BundleReader rejects anything else ("length checksum does not match").  The
    expectation below is a bit-by-bit CRC-32C written out independently of mint_amd's table-driven routine."""
    def crc_bitwise(data, crc=0):
        crc ^= 0xFFFFFFFF
        for b in data:
            crc ^= b
            for _ in range(8):
                crc = (crc >> 1) ^ 0x82F63B78 if crc & 1 else crc >> 1
        return crc ^ 0xFFFFFFFF

    def mask(c):
        return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF

    payload = bytes(range(256)) * 3 + b"object graph"   # 780 bytes: the varint length is 2 bytes (0x8c 0x06)
    prefix = str(tmp_path / "ckpt-1")
    T.write_bundle(prefix, {"x": np.arange(4, dtype=np.float32)}, {"_CHECKPOINTABLE_OBJECT_GRAPH": payload})
    rd = T.TensorBundleReader(prefix)
    e = rd.entries["_CHECKPOINTABLE_OBJECT_GRAPH"]
    raw = rd._raw(e)
    assert raw[:2] == b"\x8c\x06"
    c = crc_bitwise(struct.pack("<I", len(payload)))
    assert struct.unpack("<I", raw[2:6])[0] == mask(c)
    c = crc_bitwise(struct.pack("<I", mask(c)), c)   # Extend(crc, &length_checksum, 4)
    c = crc_bitwise(payload, c)                     # Extend(crc, string bytes)
    assert e["crc32c"] == mask(c)
    assert rd.get("_CHECKPOINTABLE_OBJECT_GRAPH", verify_crc=True) == payload
    # a corrupted length checksum / payload byte is detected on read
    data_path = prefix + ".data-00000-of-00001"
    blob = bytearray(open(data_path, "rb").read())
    blob[e["offset"] + 10] ^= 1
    open(data_path, "wb").write(bytes(blob))
    with pytest.raises(IOError):
        T.TensorBundleReader(prefix).get("_CHECKPOINTABLE_OBJECT_GRAPH", verify_crc=True)


def test_exported_graph_carries_the_step_counter_edges(tmp_path):
    """trainer.py:151 `model_.global_step = optimizer.iterations` and evaluator.py:64-67 Checkpoint(model=, global_step=):
    the exported graph has `model/global_step` and a root `global_step` edge onto the optimizer/iter variable."""
    shapes, params, _, _ = _tiny_state()
    names = list(shapes)
    prefix = str(tmp_path / "ckpt-7")
    T.write_fact_checkpoint(prefix, params, iterations=7)
    rd = T.TensorBundleReader(prefix)
    graph = T.ObjectGraph(rd.get(T.OBJECT_GRAPH_KEY, verify_crc=True))
    it = graph.walk(["optimizer", "iter"])
    assert it is not None and graph.walk(["global_step"]) == it and graph.walk(["model", "global_step"]) == it
    got = T.read_fact_checkpoint(prefix, names)
    assert got["iterations"] == 7 and got["global_step"] == 7
