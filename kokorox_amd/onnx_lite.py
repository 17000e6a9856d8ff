"""A small reader (and, for tests, writer) of the ONNX protobuf subset a weight importer needs.

No `onnx` / `protobuf` package is required: ONNX files are plain protocol-buffer wire format, and the importer only
has to walk ModelProto -> GraphProto -> {initializer TensorProto, node NodeProto -> AttributeProto}.
Field numbers follow onnx/onnx.proto3 (public schema):

  ModelProto      7 graph
  GraphProto      1 node*, 2 name, 5 initializer*
  NodeProto       1 input*, 2 output*, 3 name, 4 op_type, 5 attribute*, 7 domain
  AttributeProto  1 name, 2 f, 3 i, 4 s, 5 t, 7 floats*, 8 ints*, 20 type
  TensorProto     1 dims*, 2 data_type, 4 float_data*, 5 int32_data*, 7 int64_data*, 8 name, 9 raw_data,
                  10 double_data*, 11 uint64_data*, 13 external_data*, 14 data_location

Reference context: the one weight file byteowlz/kokorox loads is `onnx/model.onnx` of onnx-community/Kokoro-82M-v1.0-ONNX
(/root/reference/kokorox/src/utils/hf_cache.rs:8-10), with `model_fp16 / model_quantized / model_uint8 / model_q8f16 /
model_q4 / model_q4f16` selectable next to it (hf_cache.rs:135-144).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

# TensorProto.DataType
FLOAT, UINT8, INT8, UINT16, INT16, INT32, INT64, STRING, BOOL, FLOAT16, DOUBLE, UINT32, UINT64 = range(1, 14)
BFLOAT16 = 16
_NP = {FLOAT: "<f4", UINT8: "u1", INT8: "i1", UINT16: "<u2", INT16: "<i2", INT32: "<i4", INT64: "<i8", BOOL: "u1",
       FLOAT16: "<f2", DOUBLE: "<f8", UINT32: "<u4", UINT64: "<u8"}


# ---- wire format --------------------------------------------------------------------------------------------
def _varint(buf: memoryview, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if b < 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf: memoryview):
    """Yield (field number, wire type, value) over one message; value is an int, or a memoryview for wire type 2."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise ValueError("truncated length-delimited field")
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v) -> List[int]:
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed64(x))
    return out


# ---- messages -----------------------------------------------------------------------------------------------
@dataclass
class Tensor:
    name: str = ""
    dims: Tuple[int, ...] = ()
    data_type: int = 0
    array: Optional[np.ndarray] = None  # decoded (native dtype of the tensor)


@dataclass
class Node:
    op_type: str = ""
    name: str = ""
    domain: str = ""
    inputs: List[str] = field(default_factory=list)
    outputs: List[str] = field(default_factory=list)
    attrs: Dict[str, object] = field(default_factory=dict)  # ints, floats, bytes, lists, Tensor


@dataclass
class Graph:
    name: str = ""
    nodes: List[Node] = field(default_factory=list)
    initializers: Dict[str, Tensor] = field(default_factory=dict)


def _parse_tensor(buf: memoryview) -> Tensor:
    t = Tensor()
    dims: List[int] = []
    raw = None
    floats: List[float] = []
    ints: List[int] = []
    doubles: List[float] = []
    external = False
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_varints(v) if wt == 2 else [_signed64(v)]
        elif fno == 2:
            t.data_type = v
        elif fno == 4:
            floats += list(np.frombuffer(v, dtype="<f4")) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif fno in (5, 7, 11):
            ints += _packed_varints(v) if wt == 2 else [_signed64(v)]
        elif fno == 8:
            t.name = bytes(v).decode("utf-8")
        elif fno == 9:
            raw = v
        elif fno == 10:
            doubles += list(np.frombuffer(v, dtype="<f8")) if wt == 2 else [struct.unpack("<d", v)[0]]
        elif fno == 13 or (fno == 14 and v == 1):
            external = True
    t.dims = tuple(dims)
    if external:
        raise ValueError(f"tensor {t.name}: external data files are not supported (export with embedded weights)")
    if t.data_type not in _NP and t.data_type != BFLOAT16:
        raise ValueError(f"tensor {t.name}: unsupported ONNX data type {t.data_type}")
    n = int(np.prod(dims)) if dims else 1
    if raw is not None:
        if t.data_type == BFLOAT16:
            a = (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)
        else:
            a = np.frombuffer(raw, dtype=_NP[t.data_type])
    elif t.data_type == FLOAT:
        a = np.asarray(floats, dtype=np.float32)
    elif t.data_type == DOUBLE:
        a = np.asarray(doubles, dtype=np.float64)
    elif t.data_type == FLOAT16:  # stored as uint16 bit patterns in int32_data
        a = np.asarray(ints, dtype=np.uint16).view(np.float16)
    elif t.data_type == BFLOAT16:
        a = (np.asarray(ints, dtype=np.uint32) << 16).view(np.float32)
    else:
        a = np.asarray(ints, dtype=np.dtype(_NP[t.data_type]))
    if a.size != n:
        raise ValueError(f"tensor {t.name}: {a.size} values for dims {t.dims}")
    t.array = a.reshape(t.dims)
    return t


def _parse_attr(buf: memoryview):
    name, val, floats, ints = "", None, [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode("utf-8")
        elif fno == 2:
            val = struct.unpack("<f", v)[0]
        elif fno == 3:
            val = _signed64(v)
        elif fno == 4:
            val = bytes(v)
        elif fno == 5:
            val = _parse_tensor(v)
        elif fno == 7:
            floats += list(np.frombuffer(v, dtype="<f4")) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif fno == 8:
            ints += _packed_varints(v) if wt == 2 else [_signed64(v)]
    if val is None:
        val = ints if ints else (floats if floats else None)
    return name, val


def _parse_node(buf: memoryview) -> Node:
    n = Node()
    for fno, wt, v in _fields(buf):
        if fno == 1:
            n.inputs.append(bytes(v).decode("utf-8"))
        elif fno == 2:
            n.outputs.append(bytes(v).decode("utf-8"))
        elif fno == 3:
            n.name = bytes(v).decode("utf-8")
        elif fno == 4:
            n.op_type = bytes(v).decode("utf-8")
        elif fno == 5:
            k, val = _parse_attr(v)
            n.attrs[k] = val
        elif fno == 7:
            n.domain = bytes(v).decode("utf-8")
    return n


def read_graph(path: str) -> Graph:
    """Parse an .onnx file far enough for a weight importer: nodes (names, op types, inputs, attributes) and every
    initializer decoded to numpy."""
    with open(path, "rb") as f:
        data = memoryview(f.read())
    g = Graph()
    found = False
    for fno, wt, v in _fields(data):
        if fno == 7 and wt == 2:  # ModelProto.graph
            found = True
            for gf, gw, gv in _fields(v):
                if gf == 1:
                    g.nodes.append(_parse_node(gv))
                elif gf == 2:
                    g.name = bytes(gv).decode("utf-8")
                elif gf == 5:
                    t = _parse_tensor(gv)
                    g.initializers[t.name] = t
    if not found:
        raise ValueError(f"{path}: no GraphProto (not an ONNX model?)")
    # Constant nodes carry tensors too (weights folded at export time often end up there)
    for n in g.nodes:
        if n.op_type == "Constant" and isinstance(n.attrs.get("value"), Tensor) and n.outputs:
            t = n.attrs["value"]
            t.name = n.outputs[0]
            g.initializers.setdefault(t.name, t)
    return g


# ---- writer (tests only: serialise synthetic models that look like an exporter's output) --------------------------
def _wv(x: int) -> bytes:
    if x < 0:
        x += 1 << 64
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(fno: int, payload: bytes) -> bytes:
    return _wv((fno << 3) | 2) + _wv(len(payload)) + payload


def _vi(fno: int, x: int) -> bytes:
    return _wv(fno << 3) + _wv(x)


def tensor_bytes(name: str, a: np.ndarray, data_type: Optional[int] = None, raw: bool = True) -> bytes:
    a = np.asarray(a)
    if data_type is None:
        data_type = {np.dtype("float32"): FLOAT, np.dtype("float16"): FLOAT16, np.dtype("uint8"): UINT8,
                     np.dtype("int8"): INT8, np.dtype("int64"): INT64, np.dtype("int32"): INT32}[a.dtype]
    out = b"".join(_vi(1, int(d)) for d in a.shape) + _vi(2, data_type) + _ld(8, name.encode())
    if raw or data_type != FLOAT:
        out += _ld(9, np.ascontiguousarray(a).tobytes())
    else:
        out += _ld(4, np.ascontiguousarray(a, dtype="<f4").tobytes())  # packed float_data
    return out


def node_bytes(op_type: str, inputs, outputs, name: str = "", attrs: Optional[dict] = None, domain: str = "") -> bytes:
    out = b"".join(_ld(1, s.encode()) for s in inputs) + b"".join(_ld(2, s.encode()) for s in outputs)
    out += _ld(3, name.encode()) + _ld(4, op_type.encode())
    for k, v in (attrs or {}).items():
        ab = _ld(1, k.encode())
        if isinstance(v, bool) or isinstance(v, int):
            ab += _vi(3, int(v)) + _vi(20, 2)
        elif isinstance(v, float):
            ab += _wv((2 << 3) | 5) + struct.pack("<f", v) + _vi(20, 1)
        elif isinstance(v, (bytes, str)):
            ab += _ld(4, v if isinstance(v, bytes) else v.encode()) + _vi(20, 3)
        elif isinstance(v, (list, tuple)):
            ab += b"".join(_vi(8, int(x)) for x in v) + _vi(20, 7)
        else:
            raise TypeError(k)
        out += _ld(5, ab)
    if domain:
        out += _ld(7, domain.encode())
    return out


def model_bytes(nodes: List[bytes], initializers: List[bytes], graph_name: str = "main_graph") -> bytes:
    g = b"".join(_ld(1, n) for n in nodes) + _ld(2, graph_name.encode()) + b"".join(_ld(5, t) for t in initializers)
    return _vi(1, 8) + _ld(2, b"kokorox_amd.onnx_lite") + _ld(7, g)  # ir_version 8, producer_name, graph
