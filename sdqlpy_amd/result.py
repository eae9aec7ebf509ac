"""Result containers handed back by compiled queries.

The reference returns `sdqlpy.fastd.fastd` around a generated `FastDict_<abbr>_b` object — a
phmap set of result records with size() / to_dict() / print (reference src/sdqlpy/fastd.py:31-51,
src/sdqlpy/lib/fast_dict_generator.py:203-356).  Here a result is struct-of-arrays: one numpy
array per record field, rows in unspecified order (the reference's set is unordered too), with the
same size() / to_dict() surface.
"""
import numpy as np

from .sdql_lib import record, sr_dict


class Dictionary(np.ndarray):
    """The distinct values of a dictionary-coded text column (engine.dict_column): a decoder whose entries are
    pairwise different, so equal codes <=> equal text and grouping can stay on the integer codes."""


class TextRefs:
    """A text column of a large result, not decoded yet: row references (or dictionary codes) as they
    came from the device, and the host array they index.  The reference's result object defers its
    Python conversion to `to_dict()` in the same way (src/sdqlpy/fastd.py:31-51); here K-F's transfer is
    done when the query returns and only the gather of the (wide, UCS-4) strings waits for the first
    read — a 389 K-row Q10 result spends 70 of its 72 ms in that gather."""

    def __init__(self, refs, decoder):
        # own copy of the references: the array handed in may view the pinned result block of the call
        # that produced it, and a lazy column must not keep that block out of the pool
        self.refs, self.decoder = np.array(refs, copy=True), decoder

    dtype = property(lambda self: self.decoder.dtype)
    shape = property(lambda self: self.refs.shape)
    distinct = property(lambda self: isinstance(self.decoder, Dictionary))     # equal references <=> equal text

    def __len__(self):
        return len(self.refs)

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self.decoder)[self.refs]
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return np.asarray(self.decoder)[self.refs[idx]]
        out = TextRefs.__new__(TextRefs)                          # slices / index arrays stay references
        out.refs, out.decoder = self.refs[idx], self.decoder
        return out

    def tolist(self):
        return np.asarray(self.decoder)[self.refs].tolist()

    # comparisons are element-wise on the decoded text, as for an ndarray (identity comparison of two
    # lazy columns would silently yield one bool)
    def __eq__(self, other):
        return np.asarray(self) == (np.asarray(other) if isinstance(other, TextRefs) else other)

    def __ne__(self, other):
        return np.asarray(self) != (np.asarray(other) if isinstance(other, TextRefs) else other)

    __hash__ = None

    def astype(self, dtype):
        return np.asarray(self).astype(dtype)


LAZY_TEXT_ROWS = 4096        # results with at least this many rows keep their text columns as TextRefs


def decode_text(refs, decoder):
    """decoder[refs], deferred for large results."""
    return TextRefs(refs, decoder) if len(refs) >= LAZY_TEXT_ROWS else np.asarray(decoder)[refs]


class ResultSet:
    """Set of records: {record(field...): True}."""

    def __init__(self, columns, arrays, ready=None):
        """ready: a callable to run before the arrays are first read — the rows of a K-F result reach the host by a copy
        queued behind the query's kernels (sdqh_table_compact_async); the row count is known, the rows are waited for when
        something looks at them, as the reference's result object converts on `to_dict()` (src/sdqlpy/fastd.py:31-51)."""
        self.columns = list(columns)
        self._cols = [a if isinstance(a, TextRefs) else np.asarray(a) for a in arrays]
        n = len(self._cols[0]) if self._cols else 0
        for a in self._cols:
            if len(a) != n:
                raise ValueError("ragged result")
        self._n = n
        self._ready = ready

    def _wait(self):
        if self._ready is not None:
            ready, self._ready = self._ready, None
            ready()

    def wait(self):
        """Block until every row is on the host (a no-op for results that are complete already)."""
        self._wait()
        return self

    @property
    def arrays(self):
        """One numpy array per column (text columns of a large result are decoded on first use)."""
        self._wait()
        for i, a in enumerate(self._cols):
            if isinstance(a, TextRefs):
                self._cols[i] = np.asarray(a)
        return self._cols

    @arrays.setter
    def arrays(self, value):
        self._cols = list(value)

    def size(self):
        return self._n

    def __len__(self):
        return self._n

    def column(self, name):
        self._wait()
        i = self.columns.index(name)
        if isinstance(self._cols[i], TextRefs):
            self._cols[i] = np.asarray(self._cols[i])
        return self._cols[i]

    def rows(self):
        """Sorted list of plain-Python tuples (ints, floats, strs) — for comparisons."""
        cols = [a.tolist() for a in self.arrays]
        return sorted(zip(*cols)) if cols else []

    def ordered_rows(self):
        """Rows as plain-Python tuples in the order they are stored (the order of a top-k result)."""
        return list(zip(*[a.tolist() for a in self.arrays])) if self.arrays else []

    def top(self, k, order):
        self._wait()
        idx = self.top_index(k, order)
        return ResultSet(self.columns, [a[idx] for a in self._cols])     # undecoded text stays undecoded

    def top_index(self, k, order):
        """ORDER BY ... LIMIT k on the host: order = [(column, "asc" | "desc")], ties keep the stored
        row order (for K-F results that is build-row order, the same total order the device
        operator sdqh_table_topk uses).  Used for results that are small or not device tables."""
        keys = []
        for name, direction in reversed(list(order)):
            if name not in self.columns:
                raise KeyError("top: the result has no column %r" % name)
            a = self.column(name)
            if direction == "desc":
                if a.dtype.kind in "iuf":
                    a = -a if a.dtype.kind == "f" else ~a
                else:                                        # strings: rank them, then negate the rank
                    a = -np.unique(a, return_inverse=True)[1]
            keys.append(a)
        idx = np.lexsort(keys) if keys else np.arange(self._n)         # lexsort is stable
        return idx[:max(0, int(k))]

    def to_dict(self):
        out = {}
        cols = [a.tolist() for a in self.arrays]
        for row in zip(*cols):
            out[record(dict(zip(self.columns, row)))] = True
        return sr_dict(out)

    def __str__(self):
        return str(self.to_dict())

    # -- the rest of the reference container's surface (reference src/sdqlpy/fastd.py:31-51) --------
    def _row_of(self, key):
        fields = key.getContainer() if hasattr(key, "getContainer") else dict(key)
        if set(fields) != set(self.columns):
            raise KeyError("record fields %s do not match the result columns %s" % (sorted(fields), self.columns))
        return tuple(fields[c] for c in self.columns)

    def get(self, key):
        """True if the record is in the set, else None (`fastd.get`)."""
        if getattr(self, "_index", None) is None:
            self._index = set(zip(*[a.tolist() for a in self.arrays])) if self.arrays else set()
        return True if self._row_of(key) in self._index else None

    def set(self, key, value=True):
        """Add a record to the set (`fastd.set`; the value of a result set entry is always True)."""
        if value is not True:
            raise ValueError("a result set maps records to True")
        if self.get(key) is None:
            row = self._row_of(key)
            self.arrays = [np.append(a, np.array([v], dtype=a.dtype)) for a, v in zip(self.arrays, row)]
            self._n += 1
            self._index.add(row)

    def from_dict(self, data):
        """Replace the contents with {record: True, ...} (`fastd.from_dict`)."""
        recs = list((data.getContainer() if hasattr(data, "getContainer") else data).keys())
        rows = [self._row_of(r) for r in recs]
        self.arrays = [np.array([r[j] for r in rows], dtype=self.arrays[j].dtype) for j in range(len(self.columns))]
        self._n, self._index = len(rows), None
        return self

    def print(self):
        print(str(self.to_dict()))


class DictResult:
    """Dictionary from key record (or scalar) to value record (or scalar), struct-of-arrays."""

    def __init__(self, key_fields, val_fields, key_is_record=True, val_is_record=True):
        self.key_fields = list(key_fields)      # [(name, array)]
        self.val_fields = list(val_fields)
        self.key_is_record, self.val_is_record = key_is_record, val_is_record

    def size(self):
        fields = self.key_fields or self.val_fields
        return len(fields[0][1]) if fields else 0

    def __len__(self):
        return self.size()

    def to_dict(self):
        out = {}
        kcols = [a.tolist() for _, a in self.key_fields]
        vcols = [a.tolist() for _, a in self.val_fields]
        knames = [n for n, _ in self.key_fields]
        vnames = [n for n, _ in self.val_fields]
        for i in range(self.size()):
            k = record({n: c[i] for n, c in zip(knames, kcols)}) if self.key_is_record else kcols[0][i]
            v = record({n: c[i] for n, c in zip(vnames, vcols)}) if self.val_is_record else (vcols[0][i] if vcols else True)
            out[k] = v
        return sr_dict(out)

    def __str__(self):
        return str(self.to_dict())


class Pending:
    """The outcome of a plan step whose device call was launched and not waited for: resolve() — after the context has been
    synchronised — collects it and returns what the step would have returned (engine.PreparedPlan.run)."""
    __slots__ = ("resolve",)

    def __init__(self, resolve):
        self.resolve = resolve


class DeferredResultSet(ResultSet):
    """A result set whose query has been LAUNCHED but not waited for: the plan's last device call was queued and the call returned
    (engine.PreparedPlan.run with Engine.deferred_results) — the host goes on to queue the next query while this one runs, as a CUDA /
    HIP program overlaps launches with execution.  Everything that looks at the result — its size, its columns, its rows —
    first runs `thunk`, which synchronises the context, collects the device's output and finishes the plan's host-side steps;
    errors the data decides (more groups than the kernel holds ...) surface there, where the plan is re-run synchronously on its
    other path.  The reference's result object defers its conversion in the same spirit (src/sdqlpy/fastd.py:31-51)."""

    _launched = 0

    def __init__(self, thunk):
        self._thunk = thunk
        DeferredResultSet._launched += 1
        self._seq = DeferredResultSet._launched                 # launch order (engine.finish_outstanding finishes results in it)

    def _force(self):
        err = self.__dict__.get("_error")
        if err is not None:                                      # the thunk failed once: every later look at the result says so again
            raise err
        thunk = self.__dict__.pop("_thunk", None)
        if thunk is not None:
            try:
                rs = thunk()
                if not isinstance(rs, ResultSet):
                    raise TypeError("a deferred query finished with %r, not a result set" % type(rs).__name__)
                if isinstance(rs, DeferredResultSet):
                    rs._force()
            except BaseException as exc:
                self.__dict__["_error"] = exc
                raise
            self.__dict__.update(rs.__dict__)

    def __getattr__(self, name):                                 # only reached for attributes not set yet: columns, _cols, _n, _ready
        if name in ("_thunk", "_error"):
            raise AttributeError(name)
        if "_error" in self.__dict__:
            raise self.__dict__["_error"]
        if "_thunk" not in self.__dict__:
            raise AttributeError(name)
        self._force()
        return getattr(self, name)

    def wait(self):
        """Finish the query now (the result is complete and on the host when this returns)."""
        self._force()
        self._wait()
        return self

