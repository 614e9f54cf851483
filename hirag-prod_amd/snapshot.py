"""Snapshot files for an embedded table, so that an HBM-resident index survives a restart.

One file per table: a JSON header (dimension, storage type, row count, section offsets) followed by
four raw little-endian sections -- the rows exactly as stored (float32, float16, or bfloat16 bit
patterns; no padding), the ids, the group codes and one deleted flag per row -- each starting on a
4 KiB boundary.  Tables are streamed in chunks, so one larger than host memory can be written and
read.  Loading feeds the stored bits back unchanged; only the inverse norms are recomputed.

The reference keeps these columns in PostgreSQL (``vector`` / ``halfvec`` column,
``src/hirag_prod/schema/vector_config.py:11``) and has no file format of its own to match.
"""

from __future__ import annotations

import json
from typing import Callable

import numpy as np

MAGIC = b"HVSIDX01"
CHUNK_BYTES = 256 << 20
_NP = {"f32": np.float32, "f16": np.float16, "bf16": np.uint16}


def write_index(index, path: str) -> None:
    """``index`` needs ``dim``, ``dtype``, ``len()`` and ``export(first, n)``."""
    n = len(index)
    esz = np.dtype(_NP[index.dtype]).itemsize
    widths = [index.dim * esz, 8, 4, 1]
    offs, cur = [], 4096
    for w in widths:
        offs.append(cur)
        cur += (n * w + 4095) // 4096 * 4096
    header = json.dumps({"format": 1, "dim": index.dim, "dtype": index.dtype, "rows": n,
                         "sections": {"rows": offs[0], "ids": offs[1], "group_codes": offs[2], "deleted": offs[3]}}).encode()
    if len(header) > 4096 - 16:
        raise ValueError("snapshot header too large")
    per = max(1, CHUNK_BYTES // widths[0])
    with open(path, "wb") as fh:
        fh.write(MAGIC + len(header).to_bytes(8, "little") + header)
        fh.truncate(cur)
        for first in range(0, n, per):
            m = min(per, n - first)
            rows, ids, groups, deleted = index.export(first, m)
            parts = (np.ascontiguousarray(rows, dtype=_NP[index.dtype]), np.ascontiguousarray(ids, dtype=np.int64),
                     np.ascontiguousarray(groups, dtype=np.int32), np.ascontiguousarray(deleted, dtype=np.uint8))
            for off, arr, w in zip(offs, parts, widths):
                fh.seek(off + first * w)
                fh.write(arr.tobytes())


def read_header(path: str) -> dict:
    with open(path, "rb") as fh:
        head = fh.read(16)
        if len(head) != 16 or head[:8] != MAGIC:
            raise ValueError(f"{path}: not an index snapshot")
        meta = json.loads(fh.read(int.from_bytes(head[8:16], "little")).decode())
    if meta.get("format") != 1:
        raise ValueError(f"{path}: unknown snapshot format {meta.get('format')!r}")
    return meta


def read_index(path: str, make_index: Callable[[int, str], object]):
    """``make_index(dim, dtype)`` returns an empty index offering ``add``, ``set_deleted`` and
    (optionally) ``reserve``."""
    meta = read_header(path)
    dim, dtype, n = int(meta["dim"]), meta["dtype"], int(meta["rows"])
    if dtype not in _NP:
        raise ValueError(f"{path}: unknown storage type {dtype!r}")
    index = make_index(dim, dtype)
    esz = np.dtype(_NP[dtype]).itemsize
    sec = meta["sections"]
    if hasattr(index, "reserve"):
        index.reserve(n)
    per = max(1, CHUNK_BYTES // (dim * esz))
    with open(path, "rb") as fh:

        def read(off, count, np_dtype):
            fh.seek(off)
            a = np.frombuffer(fh.read(count * np.dtype(np_dtype).itemsize), dtype=np_dtype)
            if a.size != count:
                raise ValueError(f"{path}: truncated snapshot")
            return a

        for first in range(0, n, per):
            m = min(per, n - first)
            rows = read(sec["rows"] + first * dim * esz, m * dim, _NP[dtype]).reshape(m, dim)
            index.add(rows, ids=read(sec["ids"] + first * 8, m, np.int64), group_codes=read(sec["group_codes"] + first * 4, m, np.int32))
        dead = np.nonzero(read(sec["deleted"], n, np.uint8))[0] if n else np.zeros(0, dtype=np.int64)
        if dead.size:
            index.set_deleted(dead.astype(np.int64), True)
    return index


__all__ = ["write_index", "read_index", "read_header"]
