"""Table ingest: delimited text (dbgen `.tbl`) and a binary columnar format.

`read_text` is what `sdql_lib.read_csv` runs.  Contract (reference src/sdqlpy/sdql_lib.py:69-129):
one numpy array per schema column in schema order — int -> int64, float -> float64, date
"yyyy-mm-dd" -> yyyymmdd int64, string(n) -> '<U n' — field i of a line fills column i, and the empty
field after a line-terminating delimiter fills the schema's trailing *_NA column.

Plain files go through the native parser (csrc/tblload.cpp: mmap, parallel line index, parallel
parse straight into the column buffers).  A file it declines (quoted fields, values only Python's
int()/float() understand, ragged or blank lines) is re-read by `_read_text_general`, a csv-module
path with the reference's exact cell semantics.

`write_columns` / `read_columns` store a table as one `.npy` per column plus a manifest; reading
memory-maps the arrays, so an SF=100 table is "loaded" in milliseconds and paged in by the upload.
"""
import csv
import ctypes as C
import json
import os

import numpy as np

from . import build as _build

_T_INT, _T_FLOAT, _T_DATE, _T_STR = 0, 1, 2, 3
_OK, _ERR_IO, _ERR_UNSUPPORTED, _ERR_RAGGED, _ERR_ARG = range(5)
_lib = None


def _native():
    global _lib
    if _lib is None:
        lib = C.CDLL(_build.build_tblload())
        lib.sdql_tbl_open.argtypes = [C.c_char_p, C.c_char, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        lib.sdql_tbl_parse.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int]
        lib.sdql_tbl_error.argtypes = [C.c_void_p]
        lib.sdql_tbl_error.restype = C.c_char_p
        lib.sdql_tbl_close.argtypes = [C.c_void_p]
        lib.sdql_tbl_close.restype = None
        lib.sdql_dict_encode.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.sdql_dict_encode.restype = C.c_int64
        _lib = lib
    return _lib


def _type_code(t):
    """(native type code, width, numpy dtype) of a schema type marker (int, float, date, string(n))."""
    name = getattr(t, "__name__", None)
    if t is int:
        return _T_INT, 0, np.dtype(np.int64)
    if t is float:
        return _T_FLOAT, 0, np.dtype(np.float64)
    if name == "date":
        return _T_DATE, 0, np.dtype(np.int64)
    if type(t).__name__ == "string":
        return _T_STR, int(t.max_size), np.dtype("<U%d" % int(t.max_size))
    return None, 0, None


class Declined(Exception):
    """The native parser does not reproduce Python's reading of this file; use the general path."""


def read_text_native(path, types, delimiter="|", threads=None):
    codes = [_type_code(t) for t in types]
    if any(c[0] is None or (c[0] == _T_STR and c[1] < 1) for c in codes) or len(delimiter.encode()) != 1:
        raise Declined("schema or delimiter outside the native parser's vocabulary")
    lib = _native()
    threads = threads or min(32, os.cpu_count() or 1)
    handle, nrows = C.c_void_p(), C.c_int64()
    try:
        rc = lib.sdql_tbl_open(os.fsencode(path), delimiter.encode(), threads, C.byref(handle), C.byref(nrows))
        if rc == _ERR_IO:
            raise OSError(lib.sdql_tbl_error(handle).decode())
        if rc != _OK:
            raise Declined(lib.sdql_tbl_error(handle).decode())
        n = nrows.value
        cols = [np.empty(n, dt) for _, _, dt in codes]
        k = len(codes)
        tarr = (C.c_int * k)(*[c[0] for c in codes])
        warr = (C.c_int * k)(*[c[1] for c in codes])
        outs = (C.c_void_p * k)(*[a.ctypes.data for a in cols])
        rc = lib.sdql_tbl_parse(handle, k, tarr, warr, outs, threads)
        if rc != _OK:
            raise Declined(lib.sdql_tbl_error(handle).decode())
        return cols
    finally:
        lib.sdql_tbl_close(handle)


def _cell_value(t, cell):
    code, _, _ = _type_code(t)
    if code == _T_DATE:
        return int(cell.replace("-", ""))
    if code == _T_STR:
        return cell
    return t(cell)


def _read_text_general(path, types, delimiter="|"):
    """csv-module path: quoting, blank lines and every literal Python's int()/float() accept."""
    with open(path, newline="\n") as fh:
        rows = [row for row in csv.reader(fh, delimiter=delimiter) if row]
    if any(len(r) != len(types) for r in rows):
        bad = next(i for i, r in enumerate(rows) if len(r) != len(types))
        raise ValueError("%s: line %d has %d fields, the schema has %d columns" % (path, bad + 1, len(rows[bad]), len(types)))
    cols = []
    for j, t in enumerate(types):
        code, _, dt = _type_code(t)
        values = [_cell_value(t, r[j]) for r in rows]
        cols.append(np.array(values, dt) if dt is not None else np.array(values))
    return cols


def read_text(path, types, delimiter="|"):
    """list of column arrays for the schema `types` (int / float / date / string(n) markers)."""
    try:
        return read_text_native(path, types, delimiter)
    except Declined:
        return _read_text_general(path, types, delimiter)


def dict_encode(arr, max_distinct=4096, threads=None):
    """(codes int64, distinct values in sorted order) of a '<U n' column with at most max_distinct
    distinct values, else None.  Same result as np.unique(arr, return_inverse=True), in one hash pass
    plus a parallel encode instead of a sort of every row."""
    arr = np.ascontiguousarray(arr)
    if arr.dtype.kind != "U" or arr.ndim != 1:
        raise TypeError("dict_encode needs a 1-d '<U n' array")
    width = arr.dtype.itemsize // 4
    n = len(arr)
    if width < 1 or n == 0:
        vals, codes = np.unique(arr, return_inverse=True)
        return codes.astype(np.int64), vals
    codes = np.empty(n, np.int64)
    dic = np.zeros(max_distinct * width, np.uint32)
    nd = _native().sdql_dict_encode(arr.ctypes.data, n, width, max_distinct, threads or min(32, os.cpu_count() or 1), codes.ctypes.data, dic.ctypes.data)
    if nd == -1:
        return None
    if nd < 0:
        raise ValueError("dict_encode: bad arguments")
    return codes, dic[: nd * width].view(arr.dtype).copy()


# ---- binary columnar format ------------------------------------------------------------------------
MANIFEST = "columns.json"


def write_columns(directory, headers, columns):
    """One `<header>.npy` per column + a manifest with order, dtypes and row count."""
    os.makedirs(directory, exist_ok=True)
    meta = {"format": "sdqlpy_amd.columns/1", "rows": int(len(columns[0])) if columns else 0, "columns": []}
    for h, a in zip(headers, columns):
        a = np.ascontiguousarray(a)
        if a.dtype.kind not in "ifU":
            raise TypeError("column %s: unsupported dtype %s" % (h, a.dtype))
        np.save(os.path.join(directory, h + ".npy"), a, allow_pickle=False)
        meta["columns"].append({"name": h, "dtype": a.dtype.str})
    with open(os.path.join(directory, MANIFEST), "w") as fh:
        json.dump(meta, fh, indent=1)
    return directory


def read_columns(directory, want=None, mmap=True):
    """(headers, columns) of a directory written by write_columns; `want` restricts and orders the
    columns.  Arrays are memory-mapped read-only unless mmap=False."""
    with open(os.path.join(directory, MANIFEST)) as fh:
        meta = json.load(fh)
    if meta.get("format") != "sdqlpy_amd.columns/1":
        raise ValueError("%s: not a sdqlpy_amd column directory" % directory)
    names = [c["name"] for c in meta["columns"]]
    dtypes = {c["name"]: np.dtype(c["dtype"]) for c in meta["columns"]}
    pick = list(want) if want is not None else names
    cols = []
    for h in pick:
        if h not in dtypes:
            raise KeyError("%s: no column %r" % (directory, h))
        a = np.load(os.path.join(directory, h + ".npy"), mmap_mode="r" if mmap else None, allow_pickle=False)
        if a.dtype != dtypes[h] or a.shape != (meta["rows"],):
            raise ValueError("%s: column %s does not match its manifest entry" % (directory, h))
        cols.append(a)
    return pick, cols
