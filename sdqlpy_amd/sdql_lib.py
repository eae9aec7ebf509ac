"""Host-side mirror of the reference's user-facing module (reference: src/sdqlpy/sdql_lib.py).

A query file written for the reference keeps working after changing its import to
``from sdqlpy_amd.sdql_lib import *``: the same names with the same signatures and argument
meaning —

    string, date                     column type markers             (ref sdql_lib.py:14-21)
    read_csv                         '|'-separated text -> columnar   (ref sdql_lib.py:69-129)
    sr_dict, record, vector          semiring containers             (ref sdql_lib.py:132-338)
    extractYear ... unique, dense    DSL helper functions            (ref sdql_lib.py:341-368)
    sdqlpy_init                      execution mode + worker count   (ref sdql_lib.py:372-387)
    sdql_compile                     query decorator                 (ref sdql_lib.py:389-435)
    benchmark                        1 warm-up + N timed runs        (ref sdql_lib.py:437-475)

What differs is what runs the decorated function.  The reference re-parses the file, prints TBB +
phmap C++ and imports the built extension; here the function's AST is lowered (frontend.py) onto
the pattern calls of the C-ABI library libsdqlhip.so (include/sdqh.h), whose hot loops are
hand-written HIP kernels for gfx950.  There is no CPU execution path in this package: if the HIP
library cannot be loaded, or a query uses a loop shape outside the backend's vocabulary, the call
raises — it never falls back to interpreting the query in Python.
"""
import functools
import inspect
import os
import statistics
import time

import numpy as np

__all__ = [
    "string", "date", "read_csv", "write_columns", "read_columns", "shard_rows", "sr_dict", "record", "vector",
    "extractYear", "firstIndex", "startsWith", "endsWith", "dictSize", "substr", "unique", "dense",
    "sdqlpy_init", "sdql_compile", "benchmark", "invalidate",
]

# execution modes: 0 python (reference only), 1 compile + run, 2 run previously compiled, 3 HIP
MODE_PYTHON, MODE_COMPILE, MODE_PRECOMPILED, MODE_HIP = 0, 1, 2, 3


class string:
    """string(n): fixed-width text column, stored as numpy '<U n' (n UCS4 code units per row)."""
    max_size = None

    def __init__(self, max_size=25):
        self.max_size = max_size


class date:
    """date: yyyy-mm-dd text stored as the integer yyyymmdd (ref sdql_lib.py:83-84)."""

    def __init__(self):
        pass


# ------------------------------------------------------------------------------------------------
# containers
# ------------------------------------------------------------------------------------------------
class sr_dict:
    """Semiring dictionary.  ``sr_dict(d)`` wraps a dict, ``sr_dict(k, v)`` is the singleton
    {k: v}, ``sr_dict({"headers": [...], "data": [...]}, None, True)`` is a columnar table — the
    three constructor forms of the reference (ref sdql_lib.py:137-146)."""

    def __init__(self, initializer_dict=None, value=None, columnar_layout=False):
        if value is None:
            self._container = {} if initializer_dict is None else initializer_dict
        else:
            self._container = {initializer_dict: value}
        self._columnar = bool(columnar_layout)

    def getContainer(self):
        return self._container

    def get(self, key):
        return self._container.get(key)

    def __getitem__(self, key):
        return self.get(key)

    def getColumnarLayoutStatus(self):
        return self._columnar

    def __len__(self):
        return len(self._container)

    def __iter__(self):
        return iter(self._container)

    def items(self):
        return self._container.items()

    @staticmethod
    def _quote(x):
        return '"' + str(x) + '"' if isinstance(x, (str, np.str_, string)) else str(x)

    def __str__(self):
        body = ", ".join(self._quote(k) + ": " + self._quote(v) for k, v in self._container.items())
        return "{ " + body + " }" if body else "{  }"

    __repr__ = __str__

    def __hash__(self):
        return hash(str(self))

    def __eq__(self, other):
        if other is None:
            return self._container is None
        return self._container == other._container

    def __add__(self, other):
        # semiring addition: union of keys, values added on a clash (ref sdql_lib.py:186-198)
        if len(self._container) == 0:
            return other
        if len(other._container) == 0:
            return self
        for k, v in other._container.items():
            if k in self._container:
                self._container[k] += v
            else:
                self._container[k] = v
        return self

    # -- the three query combinators -----------------------------------------------------------
    # They are only meaningful inside a function decorated with @sdql_compile, where frontend.py
    # lowers them from the AST; calling them eagerly would mean interpreting the query on the CPU,
    # which this package deliberately does not do.
    def _no_interpreter(self, what):
        raise NotImplementedError(
            "sr_dict.%s is lowered from the AST of an @sdql_compile function onto the HIP backend; "
            "this package has no Python interpretation mode (use the reference's mode 0 for that)" % what)

    def sum(self, func, is_an_update_sum=True):
        self._no_interpreter("sum")

    def joinBuild(self, col, filter, outCols):
        self._no_interpreter("joinBuild")

    def joinProbe(self, indexedDict, col, filter, outputFunc, is_an_update_sum=True):
        self._no_interpreter("joinProbe")


class record(sr_dict):
    """Named tuple with field-wise semiring addition; equality and hash look at values only
    (ref sdql_lib.py:268-296)."""

    def __init__(self, initializer_dict=None):
        sr_dict.__init__(self, {} if initializer_dict is None else initializer_dict)

    def __getattr__(self, attr):
        if attr.startswith("_"):
            raise AttributeError(attr)
        return self._container.get(attr)

    def __eq__(self, other):
        if other is None:
            return self._container is None
        return list(self._container.values()) == list(other._container.values())

    def __hash__(self):
        return hash("".join(str(v) for v in self._container.values()))

    def concat(self, other):
        merged = dict(self._container)
        merged.update(other._container)
        return record(merged)


class vector:
    def __init__(self, initializer_list=None):
        self._items = [] if initializer_list is None else initializer_list

    def getContainer(self):
        return self._items

    def __len__(self):
        return len(self._items)

    def __str__(self):
        return "[ " + ", ".join(str(v) for v in self._items) + " ]" if self._items else "[  ]"


# ------------------------------------------------------------------------------------------------
# DSL helpers (ref sdql_lib.py:341-368)
# ------------------------------------------------------------------------------------------------
def extractYear(full_date_in_integer_format):
    return full_date_in_integer_format // 10000


def firstIndex(string, keyword):
    return string.find(keyword)


def startsWith(string, keyword):
    return string.startswith(keyword)


def endsWith(string, keyword):
    return string.endswith(keyword)


def dictSize(dict):
    if isinstance(dict, sr_dict):
        return len(dict.getContainer())
    return len(dict)


def substr(source, start, end):
    return source[start:end + 1]


def unique(arg):
    return arg


def dense(arg1, arg2):
    return arg2


# ------------------------------------------------------------------------------------------------
# loader
# ------------------------------------------------------------------------------------------------
def _column_types(header_type_dict):
    rec = next(iter(header_type_dict.keys()))
    fields = rec.getContainer()
    return list(fields.keys()), list(fields.values())


def table_from_columns(headers, columns, shard=None):
    """Columnar table from ready numpy arrays (int64 / float64 / '<U n'), the layout read_csv yields.
    `shard` = (rank, world) marks the result as one rank's row shard of a larger table (see shard_rows):
    a table derived from a shard's columns must carry its mark on, or a multi-GPU run takes it for whole."""
    cols = []
    n = None
    for h, c in zip(headers, columns):
        c = np.ascontiguousarray(c)
        if c.dtype.kind not in "ifU":
            raise TypeError("column %s: unsupported dtype %s" % (h, c.dtype))
        if c.dtype.kind == "i" and c.dtype != np.int64:
            c = c.astype(np.int64)
        if c.dtype.kind == "f" and c.dtype != np.float64:
            c = c.astype(np.float64)
        if n is None:
            n = len(c)
        elif len(c) != n:
            raise ValueError("column %s has %d rows, expected %d" % (h, len(c), n))
        cols.append(c)
    out = sr_dict({"headers": list(headers), "data": cols}, None, True)
    if shard is not None:
        out.shard = (int(shard[0]), int(shard[1]))
    return out


def shard_rows(table, rank, world):
    """Rows [n*rank/world, n*(rank+1)/world) of a columnar table, marked as this rank's shard
    (`.shard = (rank, world)`): what a rank of a multi-GPU run passes for a table it does not hold
    whole.  A table without the mark is taken to be complete on every rank."""
    c = table.getContainer()
    n = len(c["data"][0]) if c["data"] else 0
    lo, hi = n * rank // world, n * (rank + 1) // world
    out = sr_dict({"headers": list(c["headers"]), "data": [np.ascontiguousarray(a[lo:hi]) for a in c["data"]]}, None, True)
    out.shard = (rank, world)
    return out


def read_csv(file_path, header_type_dict, dataset_name, delimiter='|'):
    """Load a dbgen-style text table into one numpy array per column: int -> int64, float ->
    float64, date -> yyyymmdd int64, string(n) -> '<U n' (ref sdql_lib.py:69-129).  A trailing
    empty field produced by the line-terminating delimiter lands in the schema's *_NA column.
    Parsing is native and parallel (loader.py / csrc/tblload.cpp)."""
    from . import loader
    headers, types = _column_types(header_type_dict)
    cols = loader.read_text(file_path, types, delimiter)
    print("Reading " + file_path + " Finished.")
    return sr_dict({"headers": headers, "data": cols}, None, True)


def write_columns(directory, table):
    """Store a columnar table (read_csv / table_from_columns) in the binary column format."""
    from . import loader
    c = table.getContainer()
    return loader.write_columns(directory, c["headers"], c["data"])


def read_columns(directory, header_type_dict=None, mmap=True):
    """Load a table stored by write_columns; with a schema, its columns in schema order."""
    from . import loader
    want = _column_types(header_type_dict)[0] if header_type_dict is not None else None
    headers, cols = loader.read_columns(directory, want, mmap)
    return sr_dict({"headers": headers, "data": cols}, None, True)


# ------------------------------------------------------------------------------------------------
# init / compile / benchmark
# ------------------------------------------------------------------------------------------------
_state = {"mode": None, "threads": 1, "device": None, "runner": None}


def use_runner(runner):
    """Route decorated queries through a multi-GPU runner (dist.DistributedRunner), or back to the
    single-GPU engine with None.  sdqlpy_init(3, devices=N) does this itself; tests install a runner
    built over a gloo group and the CPU implementation of the ABI."""
    _state["runner"] = runner


def sdqlpy_init(execution_mode=0, threads_count=1, device=None, devices=None):
    """Select how decorated queries run (ref sdql_lib.py:372-387).

    1 / 2 (the reference's "compile" / "reuse compiled") and 3 all select the HIP backend here:
    there is nothing to compile at init time, plans are lowered on first call and cached.
    0 (interpret in Python) is the reference's own oracle mode and is not provided.
    ``threads_count`` is accepted for signature compatibility; the GPU grid is not capped by it.
    ``device`` picks the GPU (default: LOCAL_RANK or 0).
    ``devices=N`` (N > 1) runs every decorated query on N GPUs: one process per GPU (launch with
    ``python -m torch.distributed.run --nproc-per-node N``), each calling sdqlpy_init(3, devices=N)
    and passing its own row shards (tables marked by `shard_rows` / `tpch.generate(shard=...)`; an
    unmarked table is taken to be whole on every rank).  Scalars and small group-bys come back
    complete on every rank; a partitioned join returns this rank's key partition (dist.py)."""
    _state["runner"] = None
    if execution_mode == MODE_PYTHON:
        _state["mode"] = MODE_PYTHON
        return
    if execution_mode not in (MODE_COMPILE, MODE_PRECOMPILED, MODE_HIP):
        print("Execution mode is not supported. Failed.")
        return
    _state["mode"] = MODE_HIP
    _state["threads"] = int(threads_count)
    _state["device"] = device
    from . import engine
    eng = engine.default_engine(device=device, threads=int(threads_count))   # fails loudly if libsdqlhip.so is missing
    if devices is not None and int(devices) > 1:
        from . import dist as sdist
        _state["runner"] = sdist.default_runner(eng, int(devices))


def invalidate(table):
    """Tell the backend that a table's host arrays are about to change: its resident copy and every
    fact derived from it are dropped and the arrays become writable again.  (Arrays handed to a query
    are read-only afterwards — the device holds a copy, so a silent in-place edit would give answers
    for the old data; the reference re-reads the caller's buffers on every call.)"""
    from . import engine
    if engine._engine is not None:
        engine._engine.invalidate(table)


def sdql_compile(in_type):
    """Decorator.  ``in_type`` maps each parameter name to its table type in call order, exactly as
    the reference (ref sdql_lib.py:389-435; order contract sdql_compiler.py:544-551)."""
    def actual_decorator(func):
        cache = {}

        def run(args, top=None):
            if _state["mode"] is None:
                raise RuntimeError("call sdqlpy_init(...) before running a compiled query")
            if _state["mode"] == MODE_PYTHON:
                raise NotImplementedError(
                    "execution_mode 0 (interpret in Python) is not part of the MI355X backend; "
                    "it exists only in the reference, where it serves as the parity oracle")
            if _state.get("runner") is not None:                 # sdqlpy_init(3, devices=N): this rank's part of a multi-GPU run
                return _state["runner"].run(wrapper, list(args), top=top)
            from . import engine, frontend
            if "plan" not in cache:
                cache["plan"] = frontend.lower_function(func, in_type)
            eng = engine.default_engine(device=_state["device"], threads=_state["threads"])
            return engine.execute_plan(eng, cache["plan"], args, top)

        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            return run(args)

        def top(k, order):
            """The same query finished with ORDER BY ... LIMIT k: order = [(column, "asc" | "desc")],
            e.g. ``q3.top(10, [("revenue", "desc"), ("o_orderdate", "asc")])(li, cu, ord)``.  Not in
            the reference (its Q3 returns the unordered set); ties keep build-row order."""
            return lambda *args: run(args, (k, list(order)))

        wrapper.top = top
        wrapper.__sdql_in_type__ = in_type
        wrapper.__sdql_func__ = func
        return wrapper
    return actual_decorator


def benchmark(title, iterations, func, args, show_results=True, verbose=True):
    """One warm-up call, ``iterations`` timed calls (ms, wall clock around the whole call, result
    materialisation included), one more call whose result is shown (ref sdql_lib.py:437-475).
    Unlike the reference this does not touch /sys/devices/system/cpu/smt/control.

    What "materialisation" covers: the result's numeric columns and the row references / dictionary codes
    of its text columns are on the host when a call returns; the TEXT of a result with 4096 rows or more is
    gathered when it is first read (result.TextRefs: `str(res)`, `.to_dict()`, `.arrays`), as the reference's
    result object defers its conversion to `to_dict()` (src/sdqlpy/fastd.py:31-51).  With show_results the
    shown call pays for that gather; the timed calls do not (Q10's 389 K rows: 1.5 ms timed, 70 ms to decode)."""
    def complete(r):
        # (a query may return with its last device call queued: result.DeferredResultSet — the timed call includes finishing it)
        return r.wait() if hasattr(r, "wait") else r
    times = []
    complete(func(*args))
    for _ in range(iterations):
        t0 = time.time() * 1000
        complete(func(*args))
        t1 = time.time() * 1000
        times.append(t1 - t0)
    res = complete(func(*args))
    mean = sum(times) / max(1, len(times))
    if verbose:
        stdev = statistics.stdev(times) if len(times) > 1 else 0.0
        print(title + ": Mean: " + "{0:0.2f}".format(mean) + " | StDev: " + "{0:0.2f}".format(stdev))
        if show_results:
            print(str(res))
        try:
            size = res.size() if hasattr(res, "size") and callable(res.size) else len(res)
            print("Result Size: " + str(size))
        except TypeError:
            print("Scalar Result")
        print("============================================================================")
    else:
        print(title + "\t" + "{0:0.2f}".format(mean))
    return mean
