#!/usr/bin/env python3
"""Minimal ONNX (protobuf wire format) reader: enough to list a graph's nodes and pull its initializers as numpy arrays.
No onnx / protobuf-schema dependency (neither is installed here); field numbers are from the public onnx.proto3."""
import struct
import sys
import numpy as np


def _varint(b, i):
    r = 0; s = 0
    while True:
        c = b[i]; i += 1
        r |= (c & 0x7F) << s; s += 7
        if c < 0x80:
            return r, i


def fields(b):
    """yield (field_number, wire_type, value) of one message; length-delimited values are memoryviews"""
    i = 0; n = len(b)
    while i < n:
        key, i = _varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(b, i)
        elif w == 1:
            v = bytes(b[i:i + 8]); i += 8
        elif w == 2:
            ln, i = _varint(b, i); v = b[i:i + ln]; i += ln
        elif w == 5:
            v = bytes(b[i:i + 4]); i += 4
        else:
            raise ValueError("wire type %d" % w)
        yield f, w, v


_DT = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 9: np.bool_, 11: np.float64}


def tensor(b):
    dims = []; dt = 1; name = ""; raw = None; f32 = []; i64 = []; i32 = []
    for f, w, v in fields(b):
        if f == 1:
            if w == 0: dims.append(v)
            else:
                j = 0
                while j < len(v):
                    d, j = _varint(v, j); dims.append(d)
        elif f == 2: dt = v
        elif f == 8: name = bytes(v).decode()
        elif f == 9: raw = bytes(v)
        elif f == 4:
            f32 += list(struct.unpack("<%df" % (len(v) // 4), bytes(v))) if w == 2 else [struct.unpack("<f", v)[0]]
        elif f == 7:
            if w == 0: i64.append(v)
            else:
                j = 0
                while j < len(v):
                    d, j = _varint(v, j); i64.append(d if d < (1 << 63) else d - (1 << 64))
        elif f == 5:
            if w == 0: i32.append(v)
            else:
                j = 0
                while j < len(v):
                    d, j = _varint(v, j); i32.append(d)
    t = _DT[dt]
    if raw is not None: a = np.frombuffer(raw, dtype=t).copy()
    elif f32: a = np.array(f32, dtype=t)
    elif i64: a = np.array(i64, dtype=t)
    elif i32: a = np.array(i32, dtype=t)
    else: a = np.zeros(0, dtype=t)
    return name, a.reshape(dims) if dims else a.reshape(())


def attribute(b):
    name = ""; val = None; ints = []; floats = []
    for f, w, v in fields(b):
        if f == 1: name = bytes(v).decode()
        elif f == 2: val = struct.unpack("<f", v)[0]
        elif f == 3: val = v if v < (1 << 63) else v - (1 << 64)
        elif f == 4: val = bytes(v)
        elif f == 5: val = tensor(v)[1]
        elif f == 8:
            if w == 0: ints.append(v)
            else:
                j = 0
                while j < len(v):
                    d, j = _varint(v, j); ints.append(d if d < (1 << 63) else d - (1 << 64))
        elif f == 7:
            floats += list(struct.unpack("<%df" % (len(v) // 4), bytes(v))) if w == 2 else [struct.unpack("<f", v)[0]]
    if val is None: val = ints if ints else floats
    return name, val


def node(b):
    ins = []; outs = []; op = ""; name = ""; attrs = {}
    for f, w, v in fields(b):
        if f == 1: ins.append(bytes(v).decode())
        elif f == 2: outs.append(bytes(v).decode())
        elif f == 3: name = bytes(v).decode()
        elif f == 4: op = bytes(v).decode()
        elif f == 5:
            k, a = attribute(v); attrs[k] = a
    return {"op": op, "name": name, "in": ins, "out": outs, "attr": attrs}


def value_info(b):
    name = ""; shape = []
    for f, w, v in fields(b):
        if f == 1: name = bytes(v).decode()
        elif f == 2:
            for f2, _, v2 in fields(v):
                if f2 == 1:      # tensor_type
                    for f3, _, v3 in fields(v2):
                        if f3 == 2:      # shape
                            for f4, _, v4 in fields(v3):
                                if f4 == 1:
                                    d = None
                                    for f5, w5, v5 in fields(v4):
                                        if f5 == 1: d = v5
                                        elif f5 == 2: d = bytes(v5).decode()
                                    shape.append(d)
    return name, shape


def load(path):
    b = memoryview(open(path, "rb").read())
    g = None
    for f, w, v in fields(b):
        if f == 7: g = v
    nodes = []; init = {}; inputs = []; outputs = []
    for f, w, v in fields(g):
        if f == 1: nodes.append(node(v))
        elif f == 5:
            k, a = tensor(v); init[k] = a
        elif f == 11: inputs.append(value_info(v))
        elif f == 12: outputs.append(value_info(v))
    return {"nodes": nodes, "init": init, "inputs": inputs, "outputs": outputs}


if __name__ == "__main__":
    m = load(sys.argv[1])
    print("inputs", m["inputs"]); print("outputs", m["outputs"])
    for k, a in m["init"].items():
        print("init %-60s %-14s %s" % (k, a.shape, a.dtype), (a.ravel()[:4] if a.size < 8 else ""))
    for n in m["nodes"]:
        print("%-14s %s -> %s %s" % (n["op"], n["in"], n["out"], {k: (v if not isinstance(v, np.ndarray) or v.size < 6 else v.shape) for k, v in n["attr"].items()}))
