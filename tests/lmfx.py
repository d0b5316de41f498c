"""LMFX1: the flat little-endian container the Go dump tool (tools/go_dump) writes and
tests/test_lattigo_fixtures.py reads.

    file   := magic "LMFX1\\0\\0\\0" record*
    record := u32 name_len | name (UTF-8) | u8 kind | u32 ndim | u64 dims[ndim] | payload
    kind   := 1 (bytes) | 8 (u64, little-endian)

Test infrastructure; the writer exists so that the ingestion harness can be exercised without Go
(the oracle writes files of the same layout into a temporary directory).
"""
import struct

import numpy as np

MAGIC = b"LMFX1\0\0\0"


def read(path):
    out = {}
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != MAGIC:
        raise ValueError(f"{path}: not an LMFX1 file")
    off = 8
    while off < len(data):
        (nlen,) = struct.unpack_from("<I", data, off)
        off += 4
        name = data[off:off + nlen].decode()
        off += nlen
        kind, ndim = struct.unpack_from("<BI", data, off)
        off += 5
        dims = struct.unpack_from(f"<{ndim}Q", data, off)
        off += 8 * ndim
        count = int(np.prod(dims)) if ndim else 1
        if kind == 8:
            arr = np.frombuffer(data, dtype="<u8", count=count, offset=off).reshape(dims).astype(np.uint64)
            off += 8 * count
        elif kind == 1:
            arr = np.frombuffer(data, dtype=np.uint8, count=count, offset=off).reshape(dims).copy()
            off += count
        else:
            raise ValueError(f"{path}: record {name!r} has unknown kind {kind}")
        out[name] = arr
    return out


def write(path, records):
    """records: dict name -> uint64 array / uint8 array / bytes / int"""
    with open(path, "wb") as f:
        f.write(MAGIC)
        for name, v in records.items():
            if isinstance(v, (bytes, bytearray)):
                v = np.frombuffer(bytes(v), dtype=np.uint8)
            v = np.asarray(v)
            if v.dtype == np.uint8:
                kind = 1
            else:
                v = v.astype("<u8")
                kind = 8
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<BI", kind, v.ndim))
            f.write(struct.pack(f"<{v.ndim}Q", *v.shape))
            f.write(np.ascontiguousarray(v).tobytes())
