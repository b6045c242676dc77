"""Test-side PCD v0.7 reader (ascii / binary / binary_compressed), numpy only.

Independent of the product's C++ reader (haf_grasping_amd/csrc/pcd_io.cpp) so the two can be
checked against each other.  Format notes: SURVEY.md Appendix B.5.
"""
import ctypes
import struct

import numpy as np

_libc = ctypes.CDLL(None)
_libc.strtof.restype = ctypes.c_float
_libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]


def _strtof(tok):
    """Correctly rounded decimal -> float32 (what an istream >> float does); avoids double rounding."""
    return _libc.strtof(tok.encode(), None)


def lzf_decompress(data, out_len):
    out = bytearray(out_len)
    ip, op, n = 0, 0, len(data)
    while ip < n:
        ctrl = data[ip]
        ip += 1
        if ctrl < 32:
            ln = ctrl + 1
            out[op:op + ln] = data[ip:ip + ln]
            ip += ln
            op += ln
        else:
            ln = ctrl >> 5
            ref = op - ((ctrl & 0x1F) << 8) - 1
            if ln == 7:
                ln += data[ip]
                ip += 1
            ref -= data[ip]
            ip += 1
            ln += 2
            for _ in range(ln):       # may overlap: byte-wise copy
                out[op] = out[ref]
                op += 1
                ref += 1
    assert op == out_len, (op, out_len)
    return bytes(out)


def load_pcd(path):
    """Returns float32 [POINTS, 3] (x, y, z).  Honours POINTS from the header (like PCL), not the row count."""
    with open(path, "rb") as f:
        raw = f.read()
    pos = 0
    hdr = {}
    while True:
        nl = raw.index(b"\n", pos)
        line = raw[pos:nl].decode("ascii", "replace").strip()
        pos = nl + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        hdr[key] = val.split()
        if key == "DATA":
            break
    fields = hdr["FIELDS"]
    sizes = [int(s) for s in hdr["SIZE"]]
    types = hdr["TYPE"]
    counts = [int(c) for c in hdr.get("COUNT", ["1"] * len(fields))]
    npts = int(hdr["POINTS"][0]) if "POINTS" in hdr else int(hdr["WIDTH"][0]) * int(hdr["HEIGHT"][0])
    ix = [fields.index(k) for k in ("x", "y", "z")]
    for i in ix:
        assert sizes[i] == 4 and types[i] == "F" and counts[i] == 1
    mode = hdr["DATA"][0]
    if mode == "ascii":
        rows = raw[pos:].decode("ascii").split("\n")
        out = np.empty((npts, 3), np.float32)
        col = np.cumsum([0] + counts)                 # token position of a field (COUNT > 1 fields take several tokens)
        ix = [int(col[i]) for i in ix]
        k = 0
        for r in rows:
            if k == npts:
                break
            t = r.split()
            if not t:
                continue
            out[k] = [_strtof(t[ix[0]]), _strtof(t[ix[1]]), _strtof(t[ix[2]])]
            k += 1
        assert k == npts, (k, npts)
        return out
    offs = np.cumsum([0] + [s * c for s, c in zip(sizes, counts)])
    rec = int(offs[-1])
    if mode == "binary":
        buf = np.frombuffer(raw, np.uint8, npts * rec, pos).reshape(npts, rec)
        return np.stack([buf[:, offs[i]:offs[i] + 4].copy().view(np.float32)[:, 0] for i in ix], 1)
    if mode == "binary_compressed":
        csize, usize = struct.unpack_from("<II", raw, pos)
        data = lzf_decompress(raw[pos + 8:pos + 8 + csize], usize)
        cols = []
        for i in ix:   # SoA: all values of field 0, then field 1, ...
            cols.append(np.frombuffer(data, np.float32, npts, int(offs[i]) * npts))
        return np.stack(cols, 1).astype(np.float32)
    raise ValueError("unsupported DATA " + mode)
