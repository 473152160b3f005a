"""TensorFlow object-graph checkpoints without TensorFlow.

The reference saves `tf.train.Checkpoint(optimizer=optimizer, model=model_)` through a CheckpointManager
(trainer.py:168-173) and the evaluator restores `Checkpoint(model=..., global_step=...)` (evaluator.py:64-67): a
*tensor bundle* (`<prefix>.index` = a LevelDB-format sorted string table of BundleEntryProto records,
`<prefix>.data-00000-of-00001` = raw little-endian tensor bytes) whose entry `_CHECKPOINTABLE_OBJECT_GRAPH` holds a
serialized TrackableObjectGraph: nodes, named child edges, per-variable checkpoint keys and the optimizer's slot
variables.  This module

  * reads such a bundle (`TensorBundleReader`) and its object graph (`ObjectGraph`),
  * maps this repo's variable names onto the Keras attribute paths the reference's classes produce
    (`tf_variable_path`; FACTModel / CrossModalLayer / Transformer / Residual / Norm / Attention / MLP / LinearEmbedding /
    PositionEmbedding of mint/core/fact_model.py:28-70 and base_models.py:22-202; a Sequential tracks its layers as
    `layer_with_weights-k`), following the graph's edges rather than guessing key strings,
  * imports weights, Adam slots (`m`, `v`) and `optimizer/iter` (`read_fact_checkpoint`), and
  * writes the same format (`write_fact_checkpoint`) so a model trained here can be handed back to the reference's
    evaluator.

Status: the container and the GPU boxes have no TensorFlow and the reference ships no checkpoint file, so the reader
is verified against this module's own writer and against the published format descriptions only (table format:
leveldb `doc/table_format.md`; bundle protos: tensorflow/core/protobuf/tensor_bundle.proto and
trackable_object_graph.proto) - NOT yet against a TensorFlow-written file.  The wire encoding of every record this
module writes is cross-checked against the official protobuf runtime (dynamic descriptors with those .proto files' field
numbers: tests/test_tf_checkpoint.py::test_written_bundle_parses_with_the_official_protobuf_runtime).  Snappy-compressed table blocks (TensorFlow
writes bundle indices uncompressed) are decoded through pyarrow when it is importable.
"""
import os
import re
import struct

import numpy as np

from mint_amd.tfrecord import _enc_varint, _fields, _ld, _masked, _varint, crc32c

TABLE_MAGIC = 0xDB4775248B80FB57
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
VAR_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
DT_FLOAT, DT_STRING, DT_INT64 = 1, 7, 9
_NP_OF = {DT_FLOAT: np.dtype("<f4"), DT_INT64: np.dtype("<i8"), 3: np.dtype("<i4"), 2: np.dtype("<f8")}
_DT_OF = {np.dtype("float32"): DT_FLOAT, np.dtype("int64"): DT_INT64, np.dtype("int32"): 3, np.dtype("float64"): 2}


# ---- LevelDB table (sorted string table) ---------------------------------------------------------------------
def _block_entries(block):
    """(key, value) pairs of one table block: prefix-compressed entries followed by the restart array."""
    (nrestart,) = struct.unpack("<I", block[-4:])
    end = len(block) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        unshared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + unshared])
        pos += unshared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _read_block(buf, offset, size):
    raw, ctype = buf[offset:offset + size], buf[offset + size]
    if ctype == 0:
        return raw
    if ctype == 1:  # snappy
        import pyarrow as pa
        n, _ = _varint(raw, 0)  # snappy preamble: uncompressed length
        return pa.decompress(raw, decompressed_size=n, codec="snappy").to_pybytes()
    raise ValueError("unsupported table block compression %d" % ctype)


def read_table(path):
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    buf = open(path, "rb").read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s is not a LevelDB-format table (bad magic)" % path)
    footer = buf[-48:]
    _, p = _varint(footer, 0)       # metaindex handle
    _, p = _varint(footer, p)
    ioff, p = _varint(footer, p)    # index handle
    isize, p = _varint(footer, p)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize)):
        off, q = _varint(handle, 0)
        size, q = _varint(handle, q)
        out.extend(_block_entries(_read_block(buf, off, size)))
    return out


def write_table(path, items, block_bytes=4096, restart_interval=16):
    """Write sorted (key, value) pairs as an uncompressed LevelDB-format table (what BundleWriter produces)."""
    items = sorted(items)
    out = bytearray()

    def emit_block(entries):
        blk, restarts, prev = bytearray(), [], b""
        for i, (k, v) in enumerate(entries):
            shared = 0
            if i % restart_interval == 0:
                restarts.append(len(blk))
            else:
                while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                    shared += 1
            blk += _enc_varint(shared) + _enc_varint(len(k) - shared) + _enc_varint(len(v)) + k[shared:] + v
            prev = k
        if not restarts:
            restarts = [0]
        for r in restarts:
            blk += struct.pack("<I", r)
        blk += struct.pack("<I", len(restarts))
        off = len(out)
        out.extend(blk)
        out.extend(b"\x00" + struct.pack("<I", _masked(bytes(blk) + b"\x00")))
        return off, len(blk)

    index, cur, cur_bytes = [], [], 0
    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v)
        if cur_bytes >= block_bytes:
            off, size = emit_block(cur)
            index.append((cur[-1][0], _enc_varint(off) + _enc_varint(size)))
            cur, cur_bytes = [], 0
    if cur:
        off, size = emit_block(cur)
        index.append((cur[-1][0], _enc_varint(off) + _enc_varint(size)))
    moff, msize = emit_block([])
    ioff, isize = emit_block(index)
    footer = _enc_varint(moff) + _enc_varint(msize) + _enc_varint(ioff) + _enc_varint(isize)
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    with open(path, "wb") as f:
        f.write(bytes(out))


# ---- tensor bundle ---------------------------------------------------------------------------------------
def _u(v):
    return v if isinstance(v, int) else int.from_bytes(bytes(v), "little")


class TensorBundleReader:
    """`<prefix>.index` + `<prefix>.data-XXXXX-of-YYYYY` (tensorflow/core/util/tensor_bundle)."""

    def __init__(self, prefix):
        self.prefix = prefix
        self.entries, self.num_shards = {}, 1
        for key, val in read_table(prefix + ".index"):
            if key == b"":  # BundleHeaderProto: num_shards = 1, endianness = 2, version = 3
                for num, wt, v in _fields(val):
                    if num == 1:
                        self.num_shards = _u(v)
                    elif num == 2 and _u(v) != 0:
                        raise ValueError("big-endian tensor bundles are not supported")
                continue
            e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None}
            for num, wt, v in _fields(val):  # BundleEntryProto
                if num == 1:
                    e["dtype"] = _u(v)
                elif num == 2:  # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
                    for dn, dw, dv in _fields(v):
                        if dn == 2:
                            size = 0
                            for sn, sw, sv in _fields(dv):
                                if sn == 1:
                                    size = _u(sv)
                            e["shape"].append(size)
                elif num == 3:
                    e["shard_id"] = _u(v)
                elif num == 4:
                    e["offset"] = _u(v)
                elif num == 5:
                    e["size"] = _u(v)
                elif num == 6:
                    e["crc32c"] = _u(v)
            self.entries[key.decode("utf-8")] = e

    def keys(self):
        return sorted(self.entries)

    def _raw(self, e):
        path = "%s.data-%05d-of-%05d" % (self.prefix, e["shard_id"], self.num_shards)
        with open(path, "rb") as f:
            f.seek(e["offset"])
            data = f.read(e["size"])
        if len(data) != e["size"]:
            raise IOError("truncated tensor data in %s" % path)
        return data

    def get(self, key, verify_crc=False):
        """ndarray of a numeric entry, or bytes of a scalar DT_STRING entry."""
        if key not in self.entries:
            raise KeyError(key)
        e = self.entries[key]
        raw = self._raw(e)
        if verify_crc and e["crc32c"] is not None and e["dtype"] != DT_STRING and _masked(raw) != e["crc32c"]:
            raise IOError("crc32c mismatch for %s" % key)
        if e["dtype"] == DT_STRING:
            # scalar string tensor (tensor_bundle.cc WriteStringTensor): varint64 length, 4-byte masked crc32c of the
            # length taken as a FIXED-WIDTH integer, then the bytes; the entry crc runs over fixed length | checksum | bytes
            n, p = _varint(raw, 0)
            body = bytes(raw[p + 4:p + 4 + n])
            if verify_crc:
                len_ck, entry_ck = _string_checksums(body)
                if struct.unpack("<I", raw[p:p + 4])[0] != len_ck:
                    raise IOError("length checksum mismatch for %s" % key)
                if e["crc32c"] is not None and e["crc32c"] != entry_ck:
                    raise IOError("crc32c mismatch for %s" % key)
            return body
        if e["dtype"] not in _NP_OF:
            raise ValueError("unsupported dtype %d for %s" % (e["dtype"], key))
        return np.frombuffer(raw, dtype=_NP_OF[e["dtype"]]).reshape(e["shape"]).copy()


def _mask(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _string_checksums(s):
    """(length checksum, entry crc32c) of a scalar DT_STRING tensor the way TensorFlow's BundleWriter computes them:
    element sizes are checksummed as fixed-width little-endian uint32 (uint64 above 4 GiB), NOT as the varints that are
    stored; the running crc then extends over the 4 bytes of the masked length checksum and over the string bytes."""
    fixed = struct.pack("<I", len(s)) if len(s) <= 0xFFFFFFFF else struct.pack("<Q", len(s))
    len_ck = _mask(crc32c(fixed))
    return len_ck, _mask(crc32c(fixed + struct.pack("<I", len_ck) + bytes(s)))


def write_bundle(prefix, tensors, strings=None):
    """tensors: {key: ndarray}, strings: {key: bytes} (scalar DT_STRING) -> `<prefix>.index` / `.data-00000-of-00001`."""
    items, data = [(b"", _enc_varint(1 << 3) + _enc_varint(1) + _ld(3, _enc_varint(1 << 3) + _enc_varint(1)))], bytearray()

    def entry(dtype, shape, payload, crc=None):
        shp = b"".join(_ld(2, _enc_varint(1 << 3) + _enc_varint(int(s))) for s in shape)
        e = _enc_varint(1 << 3) + _enc_varint(dtype) + _ld(2, shp)
        e += _enc_varint(4 << 3) + _enc_varint(len(data)) + _enc_varint(5 << 3) + _enc_varint(len(payload))
        e += _enc_varint((6 << 3) | 5) + struct.pack("<I", _masked(payload) if crc is None else crc)
        data.extend(payload)
        return e
    for key in sorted(set(tensors) | set(strings or {})):
        if strings and key in strings:
            s = strings[key]
            len_ck, entry_ck = _string_checksums(s)
            items.append((key.encode("utf-8"),
                          entry(DT_STRING, [], _enc_varint(len(s)) + struct.pack("<I", len_ck) + s, crc=entry_ck)))
        else:
            a = np.asarray(tensors[key])
            if a.ndim:  # (ascontiguousarray would turn a scalar into shape (1,))
                a = np.ascontiguousarray(a)
            a = a.astype(a.dtype.newbyteorder("<"), copy=False)
            items.append((key.encode("utf-8"), entry(_DT_OF[np.dtype(a.dtype.name)], a.shape, a.tobytes())))
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    write_table(prefix + ".index", items)


# ---- object graph --------------------------------------------------------------------------------------------
class ObjectGraph:
    """TrackableObjectGraph: nodes[i] = {children: {local_name: node_id}, attributes: {name: checkpoint_key},
    slots: [(original_variable_node_id, slot_name, slot_variable_node_id)]}; node 0 is the root Checkpoint."""

    def __init__(self, data=None):
        self.nodes = []
        if data is not None:
            self._parse(data)

    def _parse(self, data):
        for num, wt, nv in _fields(data):
            if num != 1:
                continue
            node = {"children": {}, "attributes": {}, "slots": []}
            for fn, fw, fv in _fields(nv):
                if fn == 1:  # ObjectReference { node_id = 1, local_name = 2 }
                    nid, name = 0, ""
                    for cn, cw, cv in _fields(fv):
                        if cn == 1:
                            nid = _u(cv)
                        elif cn == 2:
                            name = bytes(cv).decode("utf-8")
                    node["children"][name] = nid
                elif fn == 2:  # SerializedTensor { name = 1, full_name = 2, checkpoint_key = 3 }
                    name, key = "", ""
                    for an, aw, av in _fields(fv):
                        if an == 1:
                            name = bytes(av).decode("utf-8")
                        elif an == 3:
                            key = bytes(av).decode("utf-8")
                    node["attributes"][name] = key
                elif fn == 3:  # SlotVariableReference { original_variable_node_id = 1, slot_name = 2, slot_variable_node_id = 3 }
                    orig, sname, snode = 0, "", 0
                    for sn, sw, sv in _fields(fv):
                        if sn == 1:
                            orig = _u(sv)
                        elif sn == 2:
                            sname = bytes(sv).decode("utf-8")
                        elif sn == 3:
                            snode = _u(sv)
                    node["slots"].append((orig, sname, snode))
            self.nodes.append(node)

    def walk(self, path):
        """node id reached from the root along the named child edges, or None."""
        nid = 0
        for name in path:
            if nid >= len(self.nodes) or name not in self.nodes[nid]["children"]:
                return None
            nid = self.nodes[nid]["children"][name]
        return nid

    def variable_key(self, nid):
        return self.nodes[nid]["attributes"].get("VARIABLE_VALUE")

    # -- construction (export) --
    def add_path(self, path, checkpoint_key=None):
        while not self.nodes:
            self.nodes.append({"children": {}, "attributes": {}, "slots": []})
        nid = 0
        for name in path:
            ch = self.nodes[nid]["children"]
            if name not in ch:
                ch[name] = len(self.nodes)
                self.nodes.append({"children": {}, "attributes": {}, "slots": []})
            nid = ch[name]
        if checkpoint_key is not None:
            self.nodes[nid]["attributes"]["VARIABLE_VALUE"] = checkpoint_key
        return nid

    def serialize(self):
        out = b""
        for node in self.nodes:
            body = b""
            for name, nid in node["children"].items():
                body += _ld(1, _enc_varint(1 << 3) + _enc_varint(nid) + _ld(2, name.encode("utf-8")))
            for name, key in node["attributes"].items():
                body += _ld(2, _ld(1, name.encode("utf-8")) + _ld(3, key.encode("utf-8")))
            for orig, sname, snode in node["slots"]:
                body += _ld(3, _enc_varint(1 << 3) + _enc_varint(orig) + _ld(2, sname.encode("utf-8")) +
                            _enc_varint(3 << 3) + _enc_varint(snode))
            out += _ld(1, body)
        return out


# ---- FACT variable names <-> Keras attribute paths -----------------------------------------------------------
_STACKS = {"cross_modal_layer/transformer": ["cross_modal_layer", "transformer_layer", "net"],
           "motion_transformer": ["motion_transformer", "net"],
           "audio_transformer": ["audio_transformer", "net"]}
_LAYER = re.compile(r"^(cross_modal_layer/transformer|motion_transformer|audio_transformer)/layer_(\d+)/(.+)$")


def tf_variable_path(name):
    """Attribute path below the checkpoint's `model` edge for one of this repo's variable names."""
    m = _LAYER.match(name)
    if m:
        stack, l, rest = m.group(1), int(m.group(2)), m.group(3)
        base = _STACKS[stack]
        attn, mlp = ["layer_with_weights-%d" % (2 * l)], ["layer_with_weights-%d" % (2 * l + 1)]
        table = {
            "attn_norm/gamma": attn + ["fn", "norm", "gamma"], "attn_norm/beta": attn + ["fn", "norm", "beta"],
            "attn/to_qkv/kernel": attn + ["fn", "fn", "to_qkv", "kernel"],
            "attn/to_out/kernel": attn + ["fn", "fn", "to_out", "kernel"],
            "attn/to_out/bias": attn + ["fn", "fn", "to_out", "bias"],
            "mlp_norm/gamma": mlp + ["fn", "norm", "gamma"], "mlp_norm/beta": mlp + ["fn", "norm", "beta"],
            "mlp/dense_1/kernel": mlp + ["fn", "fn", "net", "layer_with_weights-0", "kernel"],
            "mlp/dense_1/bias": mlp + ["fn", "fn", "net", "layer_with_weights-0", "bias"],
            "mlp/dense_2/kernel": mlp + ["fn", "fn", "net", "layer_with_weights-1", "kernel"],
            "mlp/dense_2/bias": mlp + ["fn", "fn", "net", "layer_with_weights-1", "bias"],
        }
        if rest not in table:
            raise KeyError(name)
        return base + table[rest]
    flat = {"cross_modal_layer/output/kernel": ["cross_modal_layer", "cross_output_layer", "kernel"],
            "cross_modal_layer/output/bias": ["cross_modal_layer", "cross_output_layer", "bias"]}
    for mod in ("motion", "audio"):
        flat["%s_pos_embedding/position_embedding" % mod] = ["%s_pos_embedding" % mod, "pos_embedding"]
        flat["%s_linear_embedding/kernel" % mod] = ["%s_linear_embedding" % mod, "net", "kernel"]
        flat["%s_linear_embedding/bias" % mod] = ["%s_linear_embedding" % mod, "net", "bias"]
    if name not in flat:
        raise KeyError(name)
    return flat[name]


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by the `checkpoint` state file (text-format CheckpointState)."""
    state = os.path.join(directory, "checkpoint")
    if not os.path.exists(state):
        return None
    m = re.search(r'^model_checkpoint_path:\s*"([^"]+)"', open(state).read(), re.M)
    if not m:
        return None
    p = m.group(1)
    return p if os.path.isabs(p) else os.path.join(directory, p)


def read_fact_checkpoint(prefix, names, shapes=None, verify_crc=False):
    """Read a reference checkpoint: returns {"params": {name: ndarray}, "adam_m": {...} | None, "adam_v": {...} | None,
    "iterations": int | None, "global_step": int | None}.  `names` are this repo's variable names (FACTModel.variable_names);
    `shapes` ({name: shape}) is checked when given.  Raises KeyError naming every variable the graph does not hold."""
    rd = TensorBundleReader(prefix)
    graph = ObjectGraph(rd.get(OBJECT_GRAPH_KEY))
    model_root = ["model"]
    params, node_of, missing = {}, {}, []
    for n in names:
        nid = graph.walk(model_root + tf_variable_path(n))
        key = graph.variable_key(nid) if nid is not None else None
        if key is None or key not in rd.entries:
            missing.append(n)
            continue
        a = rd.get(key, verify_crc)
        if shapes is not None and tuple(a.shape) != tuple(shapes[n]):
            raise ValueError("%s: checkpoint shape %s, model shape %s" % (n, tuple(a.shape), tuple(shapes[n])))
        params[n], node_of[nid] = a.astype(np.float32), n
    if missing:
        raise KeyError("variables not found in %s: %s" % (prefix, ", ".join(missing[:8]) + (" ..." if len(missing) > 8 else "")))
    out = {"params": params, "adam_m": None, "adam_v": None, "iterations": None, "global_step": None}
    opt = graph.walk(["optimizer"])
    if opt is not None:
        it = graph.walk(["optimizer", "iter"])
        if it is not None and graph.variable_key(it) in rd.entries:
            out["iterations"] = int(np.asarray(rd.get(graph.variable_key(it))).reshape(-1)[0])
        slots = {"m": {}, "v": {}}
        for orig, sname, snode in graph.nodes[opt]["slots"]:
            if sname in slots and orig in node_of:
                key = graph.variable_key(snode)
                if key in rd.entries:
                    slots[sname][node_of[orig]] = rd.get(key, verify_crc).astype(np.float32)
        if len(slots["m"]) == len(names) and len(slots["v"]) == len(names):
            out["adam_m"], out["adam_v"] = slots["m"], slots["v"]
    gs = graph.walk(["global_step"])  # evaluator.py:64-67 Checkpoint(model=..., global_step=...)
    if gs is None:
        gs = graph.walk(["model", "global_step"])  # trainer.py:151 model_.global_step = optimizer.iterations
    if gs is not None and graph.variable_key(gs) in rd.entries:
        out["global_step"] = int(np.asarray(rd.get(graph.variable_key(gs))).reshape(-1)[0])
    return out


def write_fact_checkpoint(prefix, params, adam_m=None, adam_v=None, iterations=None, update_state_file=True):
    """Write {name: ndarray} (this repo's names) as an object-graph checkpoint the reference's Checkpoint(optimizer=,
    model=) can restore: variables under `model/...`, Adam slots `m` / `v` and `optimizer/iter` when given."""
    graph = ObjectGraph()
    graph.add_path([])
    tensors, var_node = {}, {}
    for n, a in params.items():
        path = ["model"] + tf_variable_path(n)
        key = "/".join(path) + VAR_SUFFIX
        var_node[n] = graph.add_path(path, key)
        tensors[key] = np.asarray(a, dtype=np.float32)
    if iterations is not None or adam_m is not None:
        opt = graph.add_path(["optimizer"])
        key = "optimizer/iter" + VAR_SUFFIX
        it_node = graph.add_path(["optimizer", "iter"], key)
        tensors[key] = np.asarray(int(iterations or 0), dtype=np.int64)
        # the reference sets model_.global_step = optimizer.iterations (trainer.py:151) and the evaluator restores
        # Checkpoint(model=, global_step=) (evaluator.py:64-67): both edges point at the same variable node
        graph.nodes[graph.walk(["model"])]["children"]["global_step"] = it_node
        graph.nodes[0]["children"]["global_step"] = it_node
        for sname, slot in (("m", adam_m), ("v", adam_v)):
            if slot is None:
                continue
            for n, a in slot.items():
                key = "/".join(["model"] + tf_variable_path(n)) + "/.OPTIMIZER_SLOT/optimizer/%s" % sname + VAR_SUFFIX
                graph.nodes.append({"children": {}, "attributes": {"VARIABLE_VALUE": key}, "slots": []})
                graph.nodes[opt]["slots"].append((var_node[n], sname, len(graph.nodes) - 1))
                tensors[key] = np.asarray(a, dtype=np.float32)
    write_bundle(prefix, tensors, {OBJECT_GRAPH_KEY: graph.serialize()})
    if update_state_file:
        d, base = os.path.dirname(os.path.abspath(prefix)), os.path.basename(prefix)
        with open(os.path.join(d, "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
