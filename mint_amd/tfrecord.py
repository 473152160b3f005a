"""TFRecord + tf.train.Example reading/writing without TensorFlow.

Record framing: uint64 length | uint32 masked-crc32c(length) | payload | uint32 masked-crc32c(payload).
Payload: tf.train.Example = Features{ map<string, Feature{bytes_list|float_list|int64_list}> } as the
reference's offline tool writes it (tools/preprocessing.py:54-69) and `create_input` parses it
(mint/core/inputs.py:41-96).
"""
import struct

import numpy as np

# ---- crc32c (Castagnoli), table driven -------------------------------------------------------
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _TABLE.append(_c)


def _crc32c_py(data):
    crc = 0xFFFFFFFF
    for b in data:
        crc = _TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


_native = None


def crc32c(data):
    """CRC-32C of a bytes-like object: the library's host routine (fact_crc32c, slicing-by-8) for anything large,
    the table loop above for small inputs or when the library is not built."""
    global _native
    if len(data) < 4096:
        return _crc32c_py(data)
    if _native is None:
        try:
            from mint_amd import _lib
            _native = _lib.lib().fact_crc32c
        except Exception:
            _native = False
    if not _native:
        return _crc32c_py(data)
    import ctypes
    buf = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    arr = (ctypes.c_char * len(buf)).from_buffer_copy(buf) if isinstance(buf, bytearray) else buf
    return int(_native(arr, len(buf), 0)) & 0xFFFFFFFF


def _masked(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire helpers ---------------------------------------------------------------------
def _varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf):
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, val


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def parse_example(payload):
    """tf.train.Example bytes -> {name: list[bytes] | np.float32 array | np.int64 array}."""
    out = {}
    payload = memoryview(payload)  # nested length-delimited fields are sliced without copying (a record is megabytes)
    for num, wt, features in _fields(payload):
        if num != 1 or wt != 2:
            continue
        for fnum, fwt, entry in _fields(features):
            if fnum != 1 or fwt != 2:
                continue
            key, feat = None, None
            for enum, ewt, ev in _fields(entry):
                if enum == 1:
                    key = bytes(ev).decode("utf-8")
                elif enum == 2:
                    feat = ev
            value = None
            for knum, kwt, kv in _fields(feat if feat is not None else b""):
                if knum == 1:  # BytesList
                    value = [bytes(v) for n, w, v in _fields(kv) if n == 1]
                elif knum == 2:  # FloatList (packed or not)
                    parts = []
                    for n, w, v in _fields(kv):
                        if n == 1:
                            parts.append(np.frombuffer(v, dtype="<f4"))  # packed: a view into the record
                    value = (parts[0] if len(parts) == 1 else np.concatenate(parts)) if parts else np.zeros(0, np.float32)
                elif knum == 3:  # Int64List (packed or not)
                    vals = []
                    for n, w, v in _fields(kv):
                        if n != 1:
                            continue
                        if w == 0:
                            vals.append(v)
                        else:
                            p = 0
                            while p < len(v):
                                x, p = _varint(v, p)
                                vals.append(x)
                    value = np.array([x - (1 << 64) if x >= (1 << 63) else x for x in vals], dtype=np.int64)
            out[key] = value
    return out


def make_example(features):
    """{name: bytes/str | float ndarray | int ndarray/list} -> serialized tf.train.Example."""
    entries = b""
    for key in sorted(features):
        v = features[key]
        if isinstance(v, (bytes, str)):
            v = v.encode("utf-8") if isinstance(v, str) else v
            feat = _ld(1, _ld(1, v))
        else:
            arr = np.asarray(v)
            if arr.dtype.kind == "f":
                feat = _ld(2, _ld(1, arr.astype("<f4").tobytes()))
            else:
                feat = _ld(3, _ld(1, b"".join(_enc_varint(int(x)) for x in arr.flatten())))
        entries += _ld(1, _ld(1, key.encode("utf-8")) + _ld(2, feat))
    return _ld(1, entries)


def write_records(path, payloads):
    with open(path, "wb") as f:
        for p in payloads:
            hdr = struct.pack("<Q", len(p))
            f.write(hdr + struct.pack("<I", _masked(hdr)) + p + struct.pack("<I", _masked(p)))


def read_records(path, verify=False):
    with open(path, "rb") as f:
        while True:
            hdr = f.read(8)
            if not hdr:
                return
            if len(hdr) < 8:
                raise IOError("truncated TFRecord header in %s" % path)
            (n,) = struct.unpack("<Q", hdr)
            (hcrc,) = struct.unpack("<I", f.read(4))
            data = f.read(n)
            (dcrc,) = struct.unpack("<I", f.read(4))
            if len(data) < n:
                raise IOError("truncated TFRecord payload in %s" % path)
            if verify and (hcrc != _masked(hdr) or dcrc != _masked(data)):
                raise IOError("TFRecord CRC mismatch in %s" % path)
            yield data
