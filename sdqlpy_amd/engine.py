"""Planner + executor: scan-operator IR (frontend.py) -> pattern calls of the C ABI (include/sdqh.h).

An `Engine` owns one ABI context (one GPU, one stream) and a cache of resident columns, so that —
like the reference, which times queries with the tables already loaded in RAM
(reference src/sdqlpy/sdql_lib.py:445-451) — repeated runs of a query find their columns in HBM.

Mapping of the reference's emitted loop shapes (src/sdqlpy/lib/sdql_ir_cpp_generator_par.py) to calls:

    K-A  scalar reduce            258-291  -> sdqh_scan_filter_sum
    K-C  aggregating dict, small  402-440  -> sdqh_groupby_small
    K-B  unique dict build        331-369  -> sdqh_hash_build_unique  (semi-join probes fused in)
    K-C  probe + aggregate        402-440  -> sdqh_hash_probe_aggregate (group = matched entry)
    K-F  finalise                 520-568  -> sdqh_table_compact / host reshape of <=64 groups

Anything else raises frontend.UnsupportedQuery.  There is no CPU execution path here.
"""
import os
import weakref

import numpy as np

from . import abi, build
from .frontend import (And, Bin, Call, Cmp, Col, Const, Contains, FinalizeOp, HostDictOp, IfElse, Lookup, Not, Or, PayloadField, RecordCons,
                       ScalarExprOp, ScalarField, ScanOp, SelectKeysOp, StrIn, UnsupportedQuery, WrapScalarOp)
from .result import DeferredResultSet, Pending, DictResult, ResultSet, TextRefs, decode_text

# value-tuple vocabulary: canonical shape of the whole value record -> (ABI shape, index of the COUNT field or None)
TUPLE_SHAPES = {
    "c0": (abi.TUPLE_A, None),
    "(c0*c1)": (abi.TUPLE_AB, None),
    "(c0*(1.0-c1))": (abi.TUPLE_A_1MB, None),
    "c0;c1;(c1*(1.0-c2));((c1*(1.0-c2))*(1.0+c3));1": (abi.TUPLE_PRICING, 4),
    "((c0*(1.0-c1))-(c2*c3))": (abi.TUPLE_A_1MB_M_CD, None),
    "1": (abi.TUPLE_COUNT, 0),
}

_engine = None


def load_hip_library():
    """Load libsdqlhip.so, building it in-tree if it is missing.  Raises if that is impossible."""
    path = build.HIP_LIB
    if not os.path.exists(path) or os.environ.get("SDQLPY_AMD_REBUILD") == "1":
        build.build_hip(force=True)
    # PyTorch-ROCm bundles its own HIP runtime.  If this library pulled in the system runtime first,
    # torch would later find "no GPUs" (two runtimes in one process); importing torch first makes
    # both share the runtime torch loads.  torch is only plumbing here (collectives, exchange buffers).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = abi.Library(path)
    if lib.backend_name() != "hip-gfx950":
        raise RuntimeError("%s is not the HIP backend (reports '%s')" % (path, lib.backend_name()))
    return lib


def default_engine(device=None, threads=1):
    global _engine
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if _engine is None or _engine.ctx.device != device:
        _engine = Engine(load_hip_library().context(device=device, threads=threads))
    return _engine


def use_engine(eng):
    """Route the decorator API (sdql_compile) to an explicitly constructed Engine.  Test plumbing:
    the suite drives the same front end over the CPU implementation of the ABI this way."""
    global _engine
    from . import sdql_lib
    _engine = eng
    sdql_lib._state.update(mode=sdql_lib.MODE_HIP, device=getattr(eng.ctx, "device", 0))
    return eng


def reset_default_engine():
    global _engine
    if _engine is not None:
        _engine.close()
    _engine = None


class _Lane:
    """An Engine as one of its lanes sees it: the engine's columns, caches and switches, and a context of its own for the calls."""

    def __init__(self, eng, ctx, index):
        object.__setattr__(self, "_eng", eng)
        object.__setattr__(self, "ctx", ctx)
        object.__setattr__(self, "lane_index", index)

    def __getattr__(self, name):
        return getattr(self._eng, name)

    def __setattr__(self, name, value):
        setattr(self._eng, name, value)


class Engine:
    lane_index = 0

    def __init__(self, ctx):
        self.ctx = ctx
        self._columns = {}         # id(ndarray) -> (ndarray, abi.Column)
        self.resident_bytes = 0
        self.generation = 0        # bumped by clear(): prepared plans bound to freed columns are rebuilt
        self.compact_hints = {}    # plan step -> rows its last K-F produced (sizes the next run's result block)
        self._rowids = {}
        self._dicts = {}           # id(string ndarray) -> (ndarray, codes int64 ndarray, distinct values)
        self._range_cache = {}     # id(int64 ndarray) -> (ndarray, (min, span))
        self._distinct_cache = {}  # id(ndarray) -> (ndarray, has no repeated value)
        self._frozen = {}          # id(ndarray) -> ndarray made read-only on adoption (see column())
        self._prepared = weakref.WeakSet()      # prepared plans bound to this engine (their recordings — PlanGraph — are dropped with the columns they read)
        self._outstanding = weakref.WeakSet()   # results launched and not looked at yet (DeferredResultSet): finished before the data they were computed from is dropped
        self.force_programs = os.environ.get("SDQLPY_AMD_FORCE_PROGRAMS") == "1"   # every loop as a row program (xplan.py), none through the fixed-shape calls
        # pure streaming loops (a sum / small group-by over one table, no lookups: Q1, Q6) go to their row program first: the
        # specialised kernel streams every column at its tightest exact encoding (dictionary codes of 1 / 2 bytes, csrc/sdqh_xkernels.hpp
        # x_tight), which the fixed-shape kernels' 4-byte twins cannot.  The HIP library only: the CPU implementation interprets programs.
        # K-F rows of a query's RESULT reach the host behind the call (sdqh_table_compact_async); the ResultSet waits on first read
        self.lazy_results = os.environ.get("SDQLPY_AMD_LAZY_RESULTS", "1") != "0"
        # a plan's last device call launched without being waited for; the result finishes the plan when first looked at (PreparedPlan.run)
        self.deferred_results = os.environ.get("SDQLPY_AMD_DEFERRED_RESULTS", "1") != "0" and ctx.library.backend_name() == "hip-gfx950"
        self.dict_programs = os.environ.get("SDQLPY_AMD_DICT_PROGRAMS", "1") != "0"      # sums over result dictionaries as device loops (xplan.prepare_dict_scan)
        self.stream_programs = os.environ.get("SDQLPY_AMD_STREAM_PROGRAMS", "1") != "0" and ctx.library.backend_name() == "hip-gfx950"
        # ... and so do the loops that aggregate into the entry a probe matches ("probe": Q3's lineitem loop — the specialised kernel
        # streams the key through its 4-byte twin and the date as a 2-byte code, tests the key bitmap in 32-bit arithmetic) and, on
        # request, unique builds ("build"); "values": the builds whose key and payload are plain columns of the scanned row and whose
        # conditions need no payload of a looked-up entry (Q3's orders build) — the value-queue stage kernel streams everything and
        # gathers nothing.  Measured per kind in profiles/r03_ab_tuned_vs_programs.txt.
        routes = os.environ.get("SDQLPY_AMD_PROGRAM_ROUTES", "probe,values,lookups")
        self.program_routes = {r for r in routes.split(",") if r} if ctx.library.backend_name() == "hip-gfx950" else set()
        # LANES: contexts of one family (abi.Context.fork — own stream, pool and result blocks, the engine's columns shared).  Every
        # prepared plan is bound to one, round robin, so that queries launched without being waited for (deferred results) share the
        # chip: one query's small dependent launches run under another's streaming kernel instead of queueing behind it on one stream
        # (bench step q1 + q3 + q5: 0.83 -> 0.57 ms, tools/step_lanes.py).  A query that is waited for before the next is launched
        # gains and loses nothing.  SDQLPY_AMD_LANES=1: one stream, as before.
        # PLAN GRAPHS: a prepared plan whose calls wait for nothing (its last call deferred) is recorded once into a graph (abi.Graph:
        # sdqh_graph_*) and from then on launched by ONE call — the host issues a query in microseconds instead of a dozen calls.
        # The value = recordings a plan may hold (each owns its tables' memory and a result block; a recording is reused when its last
        # result has been collected and dropped); 0 = off.
        self.plan_graphs = int(os.environ.get("SDQLPY_AMD_PLAN_GRAPHS", "2")) if ctx.library.backend_name() == "hip-gfx950" else 0
        # sums over result dictionaries that ran on the HOST (xplan.run_host_dict): {(source line, result name): {"runs", "why"}}; with
        # strict_device such a loop raises UnsupportedQuery instead (tests run the shipped queries that way to say which ones take it)
        self.host_loops = {}
        self.strict_device = os.environ.get("SDQLPY_AMD_STRICT_DEVICE", "0") == "1"
        self.plan_graphs_always = os.environ.get("SDQLPY_AMD_PLAN_GRAPHS_ALWAYS", "0") == "1"      # (default: only while no other result is in flight, see PreparedPlan.run)
        self.graph_stats = {"recorded": 0, "launched": 0, "refused": 0, "dropped": 0}
        lanes = os.environ.get("SDQLPY_AMD_LANES", "")
        self.nlanes = max(1, int(lanes)) if lanes else (3 if ctx.library.backend_name() == "hip-gfx950" else 1)
        self._lane_views = {}
        self._lane_next = 0

    def lane(self, k):
        """The engine as lane k sees it (k = 0: the engine itself)."""
        if k <= 0 or self.nlanes <= 1:
            return self
        view = self._lane_views.get(k)
        if view is None:
            while len(self.ctx.forks) < k:
                self.ctx.fork()
            view = self._lane_views[k] = _Lane(self, self.ctx.forks[k - 1], k)
        return view

    def next_lane(self):
        k = self._lane_next % max(1, self.nlanes)
        self._lane_next += 1
        return k

    def stats(self):
        """What the engine did besides launching kernels: loops that ran on the host (and why), plan graphs recorded / launched /
        refused, resident bytes."""
        return {"host_loops": [{"line": k[0], "result": k[1], "runs": v["runs"], "why": v["why"]} for k, v in sorted(self.host_loops.items(), key=lambda kv: str(kv[0]))],
                "plan_graphs": dict(self.graph_stats), "resident_bytes": int(self.resident_bytes), "lanes": int(self.nlanes)}

    def synchronize(self):
        """Everything launched through this engine, on whichever lane, has run."""
        self.ctx.synchronize()

    def close(self):
        self.clear()
        self.ctx.close()

    def finish_outstanding(self):
        """Finish every result that was launched and not looked at yet, so that it holds the rows of the data it was launched on: a
        result whose collection overflows re-runs its plan (PreparedPlan._deferred), and must do that before the tables change."""
        # in the order they were launched: under the multi-GPU runner finishing a result can issue collectives (a join repeated with exact
        # sizes), and every rank must issue them in the same order — a WeakSet iterates in address order, different on every rank
        for rs in sorted(self._outstanding, key=lambda r: r.__dict__.get("_seq", 0)):
            try:
                rs.wait()
            except Exception:                                   # kept by the result: raised again where the caller looks at it
                pass
        self._outstanding.clear()

    def drop_recordings(self):
        """Recorded plans (and the device memory they keep) go: the columns they name are about to change."""
        for pp in list(self._prepared):
            pp.drop_graphs()

    def trim(self):
        """Give the free blocks of the lanes' device-memory pools back to the runtime (option "pool_trim"): after clear(), before a
        workload whose memory another context — or another engine — will need."""
        root = self.ctx
        for c in [root] + list(getattr(root, "forks", [])):
            if c.handle is not None:
                c.set_option("pool_trim", 1)

    def clear(self):
        if self.ctx.handle is not None:
            self.finish_outstanding()
            self.drop_recordings()
        if self._columns and self.ctx.handle is not None:
            self.ctx.synchronize()                              # queries launched and not waited for may still read the columns
        for _, col in self._columns.values():
            col.free()
        self._columns.clear()
        self._thaw(list(self._frozen.values()))
        self._frozen.clear()
        self._range_cache.clear()
        self._distinct_cache.clear()
        self.__dict__.get("_dict_futures", {}).clear()        # (a pass still running finishes into nothing: its arrays stay alive in the task)
        self.resident_bytes = 0
        self.generation += 1

    @staticmethod
    def _thaw(arrays):
        """Make adopted arrays writable again.  A view can only be made writable while its base is: owners first, then views; a view
        whose base somebody else keeps read-only stays read-only (numpy refuses, and that is the right answer)."""
        for arr in sorted(arrays, key=lambda a: 0 if a.base is None else 1):
            try:
                arr.flags.writeable = True
            except ValueError:
                pass

    def column(self, arr):
        """Resident column for a host array (uploaded on first use, then cached by identity).

        The reference reads the caller's live buffers on every call (sdql_compiler.py:644-668); here the
        first call copies them to HBM, so a later in-place edit of the host array would silently leave the
        device (and the cached min/max, dictionaries, distinctness facts) describing the old data.  An
        adopted array is therefore made read-only: writing to it raises at the writer.  To change data,
        call `invalidate(table_or_array)` first (or pass new arrays)."""
        hit = self._columns.get(id(arr))
        if hit is not None and hit[0] is arr:
            return hit[1]
        col = self.ctx.upload(arr)
        self._columns[id(arr)] = (arr, col)
        self.resident_bytes += arr.nbytes
        if isinstance(arr, np.ndarray) and arr.flags.writeable:
            arr.flags.writeable = False
            self._frozen[id(arr)] = arr
        return col

    def iota_column(self, lo, span):
        """Resident column lo, lo+1, ..., lo+span-1: the key column of a dense group domain (xplan: large group-bys over a small key range)."""
        cache = self.__dict__.setdefault("_iota", {})
        arr = cache.get((lo, span))
        if arr is None:
            if len(cache) > 8:
                cache.clear()
            arr = cache[(lo, span)] = np.arange(lo, lo + span, dtype=np.int64)
        return self.column(arr)

    def invalidate(self, what):
        """Forget everything derived from a table (columnar sr_dict) or a single host array — resident
        column, min/max, dictionary codes, distinctness — and make the arrays writable again.  Plans
        prepared against them are rebuilt on their next run."""
        if hasattr(what, "getContainer"):
            arrays = list(what.getContainer().get("data", []))
        else:
            arrays = [what]
        if self.ctx.handle is not None:
            self.finish_outstanding()
            self.drop_recordings()
        if any(id(arr) in self._columns for arr in arrays) and self.ctx.handle is not None:
            self.ctx.synchronize()                              # (queries launched and not waited for may still read them)
        for arr in arrays:
            key = id(arr)
            hit = self._columns.pop(key, None)
            if hit is not None and hit[0] is arr:
                hit[1].free()
                self.resident_bytes -= arr.nbytes
            self.__dict__.get("_dict_futures", {}).pop(key, None)
            d = self._dicts.pop(key, None)
            if d is not None and d[1] is not None:
                self.invalidate(d[1])                        # the code column built from it
            self._range_cache.pop(key, None)
            self._distinct_cache.pop(key, None)
            frozen = self._frozen.pop(key, None)
            if frozen is not None:
                self._thaw([frozen])
        self.generation += 1

    def rowid_column(self, nrows):
        """Resident int64 column 0..nrows-1: how a string column travels as a payload or a group key
        (the device carries the row reference; the text is looked up when the result is built)."""
        arr = self._rowids.get(nrows)
        if arr is None:
            arr = self._rowids[nrows] = np.arange(nrows, dtype=np.int64)
        return self.column(arr)

    def dict_column(self, arr, max_distinct=4096):
        """(resident int64 code column, distinct values in sorted order) for a low-cardinality string
        column, or None: how a string column of the scanned table serves as a group key (Q4's
        o_orderpriority) and how `==` / startsWith on it become an integer range on the codes.  The
        dictionary is built once per host array (loader.dict_encode) and cached like an upload."""
        hit = self._dicts.get(id(arr))
        if hit is None or hit[0] is not arr:
            from . import loader
            pending = self.__dict__.setdefault("_dict_futures", {}).pop(id(arr), None)
            if pending is not None and pending[0] is arr and pending[2] == max_distinct:
                enc = pending[1].result()                        # started when the plan was bound (prefetch_dicts), beside the uploads
            else:
                enc = loader.dict_encode(arr, max_distinct)      # native hash pass; None = too many distinct values
            if enc is not None:
                from .result import Dictionary
                enc = (enc[0], np.asarray(enc[1]).view(Dictionary))          # tagged: its entries are pairwise distinct
            hit = self._dicts[id(arr)] = (arr,) + (enc if enc is not None else (None, None))
        if hit[1] is None:
            return None
        return self.column(hit[1]), hit[2]

    def prefetch_dicts(self, plan, tables, max_distinct=4096):
        """Start the host-side dictionary pass (loader.dict_encode: a native hash pass over every row) of the text columns a plan NAMES,
        in the background: binding the plan uploads its numeric columns next — 70 of Q1's first 145 ms at SF=10, during which the two
        flag columns' 29 ms of encoding now run too (ctypes releases the GIL on both).  dict_column picks the result up; a column the
        plan names but never needs coded costs idle cores a pass and nothing else."""
        min_rows = self.__dict__.get("prefetch_dict_rows")
        if min_rows is None:
            min_rows = self.__dict__["prefetch_dict_rows"] = int(os.environ.get("SDQLPY_AMD_PREFETCH_DICT_ROWS", 1 << 20))      # (< 0: off)
        if min_rows < 0:
            return
        from . import loader
        futures = self.__dict__.setdefault("_dict_futures", {})
        seen, stack = set(), list(getattr(plan, "ops", []))
        names = {}                                               # column names mentioned anywhere in the plan's loops
        while stack:
            x = stack.pop()
            if id(x) in seen or isinstance(x, (str, bytes, int, float, bool, type(None), np.ndarray)):
                continue
            seen.add(id(x))
            if isinstance(x, (list, tuple, set, frozenset)):
                stack.extend(x)
                continue
            if isinstance(x, dict):
                stack.extend(x.values())
                continue
            if isinstance(x, Col):
                names.setdefault(None, set()).add(x.name)
            if hasattr(x, "__dict__"):
                stack.extend(vars(x).values())
        wanted = names.get(None, set())
        for t in tables.values():
            for cname, arr in t.cols.items():
                if cname in wanted and isinstance(arr, np.ndarray) and arr.dtype.kind == "U" and arr.ndim == 1 and len(arr) >= min_rows \
                        and id(arr) not in self._dicts and id(arr) not in futures:
                    pool = self.__dict__.get("_dict_pool")
                    if pool is None:
                        from concurrent.futures import ThreadPoolExecutor
                        pool = self.__dict__["_dict_pool"] = ThreadPoolExecutor(max_workers=2, thread_name_prefix="sdqlpy-dict")
                    futures[id(arr)] = (arr, pool.submit(loader.dict_encode, arr, max_distinct), max_distinct)

    def adopt(self, arr, col):
        """Register an already-resident column for a host array identity (multi-GPU exchange buffers)."""
        self._columns[id(arr)] = (arr, col)


# ---- tables as the planner sees them -----------------------------------------------------------
class HostTable:
    """A columnar table argument: header name -> numpy array."""
    def __init__(self, name, srdict):
        c = srdict.getContainer()
        if not (isinstance(c, dict) and "headers" in c and "data" in c):
            raise TypeError("argument '%s' is not a columnar table (read_csv / tpch.generate output)" % name)
        self.name = name
        self.cols = dict(zip(c["headers"], c["data"]))
        self.nrows = len(c["data"][0]) if c["data"] else 0

    def array(self, col, op):
        if col not in self.cols:
            raise UnsupportedQuery("line %d: table '%s' has no column '%s'" % (op.lineno, self.name, col))
        return self.cols[col]


class BuiltTable:
    """Result of a unique build, resident on the device."""
    def __init__(self, table, key_name, key_is_record, val_fields, val_is_record, row_arrays):
        self.table = table                  # abi.Table
        self.key_name, self.key_is_record = key_name, key_is_record
        self.val_fields = val_fields        # [(field name, "key" | payload index)]
        self.val_is_record = val_is_record
        self.payload_dtypes = row_arrays    # payload index -> numpy dtype
        self.key_radix = None               # (parts, radix): the key is several fields packed into one mixed-radix integer (xplan, large group-bys)
        self.int_values = ()                # value fields that are integer-valued sums (row programs): returned as int64
        self.agg = None                     # set by a fused probe-aggregate: (key_fields, val_names, count field index, key / value is a record, number of summed doubles)
        self.agg_spec = None
        self.decoders = {}                  # payload index -> string array (the payload holds row references into it)
        self.field_decoders = {}            # field name -> string array: several text fields of one source row share ONE
                                            # row-reference payload slot and differ only in what it indexes (late materialisation)
        self.key_parts = None               # composite key: [part names]; the stored key is (part0 << 32) | part1
        self.key_decoder = None             # single key that is a row reference / dictionary code: what it indexes
        self.agg_fields = {}                # fused probe-aggregate: output key field -> entry field (whose decoder applies)
        self.shared_groups = False          # sdqh_table_share_groups ran: one live entry per output key
        self.slot_roots = None              # payload index -> the scanned-row columns it is a function of (_share_spec)
        self.slot_plain = {}                # payload index -> (lo, span) when it holds a plain int column / row number

    def layout_sig(self):
        """What a loop that looks this table up compiles against: field -> slot, slot dtypes, key shape, and WHICH arrays
        decode its slots (identities).  Closures cache their marshalled call per signature of the tables they look up."""
        return (tuple(self.val_fields), self.dtype_sig(), None if self.key_parts is None else tuple(self.key_parts),
                tuple(sorted((k, id(v)) for k, v in self.decoders.items())), tuple(sorted((k, id(v)) for k, v in self.field_decoders.items())),
                id(self.key_decoder), self.table.npayload, self.table.accumulate)

    def dtype_sig(self):
        """The payload dtypes as strings, made once (str(np.dtype) costs a microsecond, and every run of every loop that looks the table up asks)."""
        sig = self.__dict__.get("_dtype_sig")
        if sig is None or sig[0] is not self.payload_dtypes:
            sig = self.__dict__["_dtype_sig"] = (self.payload_dtypes, tuple(str(np.dtype(d)) for d in self.payload_dtypes))
        return sig[1]

    def slot_of(self, field):
        """"key" | payload index of a value field, or None."""
        if field is None:
            return self.val_fields[0][1] if len(self.val_fields) == 1 else None
        return dict(self.val_fields).get(field)

    def decoder_of(self, field, slot):
        return self.field_decoders.get(field, self.decoders.get(slot))

    def field_index(self, field, op):
        """Payload index of a value field (None = the scalar value)."""
        if field is None:
            if len(self.val_fields) != 1:
                raise UnsupportedQuery("line %d: the looked-up value is a record; name a field" % op.lineno)
            src = self.val_fields[0][1]
        else:
            src = dict(self.val_fields).get(field)
        if src is None or src == "key":
            raise UnsupportedQuery("line %d: looked-up field '%s' is not a payload of the table" % (op.lineno, field))
        return src


# ---- predicate / tuple lowering ----------------------------------------------------------------
def _flip(op):
    return {"<": ">", "<=": ">=", ">": "<", ">=": "<=", "==": "==", "!=": "!="}[op]


def _code_range(eng, arr, name, passes, iranges):
    """A predicate on a low-cardinality text column as an integer range on its dictionary codes:
    evaluate `passes` on the (sorted) distinct values; if the passing codes are one contiguous range
    (equality: one code; a prefix: a run of the sorted dictionary; nothing passes: an empty range)
    record it in iranges under a name of its own and return True.  The kernels then read 8-byte codes
    instead of 4 bytes per code unit, and need no string instance."""
    coded = eng.dict_column(arr)
    if coded is None:
        return False
    hit = np.nonzero(passes(coded[1]))[0]
    if len(hit) == 0:
        lo, hi = 1, 0
    elif hit[-1] - hit[0] + 1 == len(hit):
        lo, hi = int(hit[0]), int(hit[-1])
    else:
        return False
    key = "\0codes:" + name
    plo, phi = iranges.get(key, (abi.INT64_MIN, abi.INT64_MAX))[:2]
    iranges[key] = (max(plo, lo), min(phi, hi), coded[0])
    return True


def _build_filter(eng, op, htab, conds):
    """conds -> (abi.Filter, [semi-join lookups])."""
    iranges, franges, spreds, cpreds, lookups = {}, {}, [], [], []
    for c in conds:
        if isinstance(c, Contains):
            lookups.append(c.lookup)
            continue
        if isinstance(c, StrIn):
            arr = htab.array(c.col.name, op)
            if arr.dtype.kind != "U":
                raise UnsupportedQuery("line %d: `in` needs a string column" % op.lineno)
            if c.how == "prefix" and _code_range(eng, arr, c.col.name, lambda values: np.char.startswith(values, c.needle), iranges):
                continue
            spreds.append((eng.column(arr), c.needle, {"in": abi.STR_CONTAINS, "prefix": abi.STR_PREFIX, "suffix": abi.STR_SUFFIX}[c.how]))
            continue
        if not isinstance(c, Cmp):
            raise UnsupportedQuery("line %d: unsupported condition %r" % (op.lineno, c))
        left, right, sym = c.left, c.right, c.op
        if isinstance(left, Const) and isinstance(right, Col):
            left, right, sym = right, left, _flip(sym)
        if isinstance(left, Col) and isinstance(right, Col):         # column vs column (Q4 `l_commitdate < l_receiptdate`)
            a, b = htab.array(left.name, op), htab.array(right.name, op)
            if a.dtype != b.dtype or a.dtype.kind not in "if":
                raise UnsupportedQuery("line %d: column comparison needs two int or two float columns (%r)" % (op.lineno, c))
            if sym in (">", ">="):
                a, b, sym = b, a, _flip(sym)
            cpreds.append((eng.column(a), eng.column(b), {"<": abi.CMP_LT, "<=": abi.CMP_LE, "==": abi.CMP_EQ, "!=": abi.CMP_NE}[sym]))
            continue
        if not (isinstance(left, Col) and isinstance(right, Const)):
            raise UnsupportedQuery("line %d: only <column> <op> <constant | column> comparisons are supported (%r)" % (op.lineno, c))
        arr = htab.array(left.name, op)
        v = right.value
        if arr.dtype.kind == "U":
            if sym not in ("==", "!=") or not isinstance(v, str):
                raise UnsupportedQuery("line %d: string columns support == / != against a literal" % op.lineno)
            if sym == "==" and _code_range(eng, arr, left.name, lambda values: values == v, iranges):
                continue
            spreds.append((eng.column(arr), v, sym == "!="))
        elif arr.dtype == np.int64:
            lo, hi = iranges.get(left.name, (abi.INT64_MIN, abi.INT64_MAX))
            if isinstance(v, float) and v != int(v):      # x < 24.5  <=>  x <= 24
                fl = int(np.floor(v))
                v_lt, v_le, v_gt, v_ge = fl, fl, fl + 1, fl + 1
            else:
                v = int(v)
                v_lt, v_le, v_gt, v_ge = v - 1, v, v + 1, v
            if sym == "<": hi = min(hi, v_lt)
            elif sym == "<=": hi = min(hi, v_le)
            elif sym == ">": lo = max(lo, v_gt)
            elif sym == ">=": lo = max(lo, v_ge)
            elif sym == "==":
                if isinstance(v, float):
                    lo, hi = 1, 0
                else:
                    lo, hi = max(lo, v), min(hi, v)
            else:
                raise UnsupportedQuery("line %d: != on numeric columns is not supported" % op.lineno)
            iranges[left.name] = (lo, hi)
        elif arr.dtype == np.float64:
            lo, hi = franges.get(left.name, (-np.inf, np.inf))
            v = float(v)
            if sym == "<": hi = min(hi, abi.lt_float(v))
            elif sym == "<=": hi = min(hi, v)
            elif sym == ">": lo = max(lo, abi.gt_float(v))
            elif sym == ">=": lo = max(lo, v)
            elif sym == "==": lo, hi = max(lo, v), min(hi, v)
            else:
                raise UnsupportedQuery("line %d: != on numeric columns is not supported" % op.lineno)
            franges[left.name] = (lo, hi)
        else:
            raise UnsupportedQuery("line %d: column '%s' has unsupported dtype %s" % (op.lineno, left.name, arr.dtype))
    ip = [(r[2], r[0], r[1]) if len(r) == 3 else (eng.column(htab.array(n, op)), r[0], r[1]) for n, r in iranges.items()]
    fp = [(eng.column(htab.array(n, op)), lo, hi) for n, (lo, hi) in franges.items()]
    try:
        return abi.make_filter(ip, fp, spreds, cpreds), lookups
    except abi.SdqhError as exc:
        raise UnsupportedQuery("line %d: %s" % (op.lineno, exc))


def _build_tuple(eng, op, htab, val):
    """value record / scalar -> (abi.Tuple, [field names], count field index or None)."""
    if isinstance(val, RecordCons):
        names = [n for n, _ in val.fields]
        exprs = [e for _, e in val.fields]
    else:
        names, exprs = [None], [val]
    slots = []
    try:
        shape = ";".join(e.shape(slots) for e in exprs)
    except NotImplementedError:
        raise UnsupportedQuery("line %d: value expression outside the backend's vocabulary: %r" % (op.lineno, val))
    if shape not in TUPLE_SHAPES:
        raise UnsupportedQuery("line %d: value tuple shape '%s' is not in the HIP backend's vocabulary %s"
                               % (op.lineno, shape, sorted(TUPLE_SHAPES)))
    abi_shape, count_idx = TUPLE_SHAPES[shape]
    operands = []
    for kind in slots:
        if kind[0] != "col":
            raise UnsupportedQuery("line %d: value operands must be columns of the scanned table" % op.lineno)
        arr = htab.array(kind[1], op)
        if arr.dtype != np.float64:
            raise UnsupportedQuery("line %d: value operand '%s' must be a float column" % (op.lineno, kind[1]))
        operands.append(eng.column(arr))
    return abi.make_tuple(abi_shape, operands), names, count_idx


def _value_arrays(names, count_idx, values, counts, ints=()):
    """Scatter the ABI's (doubles, count) outputs back to the value record's fields.  `ints`: fields whose
    summed expression is integer-valued (`1 if c else 0`): summed as doubles (exact below 2^53), returned as int64."""
    out, v = [], 0
    for i, n in enumerate(names):
        if count_idx is not None and i == count_idx:
            out.append((n, np.asarray(counts, np.int64)))
        else:
            a = np.asarray(values[v], np.float64)
            out.append((n, np.rint(a).astype(np.int64) if n in ints else a))
            v += 1
    return out


# ---- operator preparation -----------------------------------------------------------------------
# Lowering an operator (predicate ranges, tuple shape, column binding, ctypes structs) does not
# depend on the data, only on the plan and on which tables it is bound to.  It is done once per
# (plan, tables) and yields a closure; running a query is then just the sequence of C-ABI calls.
def _probe_specs(eng, op, htab, lookups):
    specs = []
    for lk in lookups:
        if not isinstance(lk.key, Col):
            raise UnsupportedQuery("line %d: composite / derived lookup keys are not in the HIP backend's vocabulary yet" % op.lineno)
        arr = htab.array(lk.key.name, op)
        if arr.dtype != np.int64:
            raise UnsupportedQuery("line %d: lookup key '%s' must be an int column" % (op.lineno, lk.key.name))
        specs.append((lk.dict_name, eng.column(arr)))
    return specs


def _resolve_probes(op, env, specs):
    probes = []
    for name, col in specs:
        bt = env.get(name)
        if not isinstance(bt, BuiltTable):
            raise UnsupportedQuery("line %d: '%s' is not a built table" % (op.lineno, name))
        probes.append((bt.table, col))
    return probes


def _walk_lookups(e, found):
    """Collect the Lookup nodes an expression depends on, in first-use order."""
    if isinstance(e, Lookup):
        _walk_lookups(e.key, found)
        if repr(e) not in [repr(x) for x in found]:
            found.append(e)
    elif isinstance(e, PayloadField):
        _walk_lookups(e.lookup, found)
    elif isinstance(e, RecordCons):
        for _, x in e.fields:
            _walk_lookups(x, found)
    elif isinstance(e, Bin):
        _walk_lookups(e.left, found); _walk_lookups(e.right, found)
    elif isinstance(e, Call):
        for x in e.args:
            _walk_lookups(x, found)
    elif isinstance(e, Cmp):
        _walk_lookups(e.left, found); _walk_lookups(e.right, found)
    elif isinstance(e, (And, Or)):
        for x in e.terms:
            _walk_lookups(x, found)
    elif isinstance(e, Not):
        _walk_lookups(e.term, found)
    elif isinstance(e, Contains):
        _walk_lookups(e.lookup, found)
    elif isinstance(e, IfElse):
        _walk_lookups(e.cond, found); _walk_lookups(e.then, found); _walk_lookups(e.other, found)


def _link_key_sources(srcs, lookup_keys):
    for sr in srcs:
        if sr.kind == "lookup" and sr.key_src is None and len(lookup_keys[sr.lookup]) == 1:
            sr.key_src = lookup_keys[sr.lookup][0]


class _Src:
    """A value source resolved at run time into an abi source spec (+ how to decode it for the result)."""
    def __init__(self, kind, col=None, lookup=None, field=None, year=False, decoder=None, dtype=np.int64):
        self.kind, self.col, self.lookup, self.field, self.year, self.decoder, self.dtype = kind, col, lookup, field, year, decoder, dtype

    key_src = None        # lookup kind: the source of the lookup's (single) key — what a field that IS the table's key reads
    host = None           # col kind, plain int column: the host array (its value range is known)
    is_rowid = False      # col kind: the row number of the scanned table

    def _is_key_field(self, bt):
        return self.kind == "lookup" and self.key_src is not None and bt.slot_of(self.field) == "key"

    def spec(self, op, env, lookups):
        if self.kind == "col":
            return abi.src_col(self.col)
        bt = env[lookups[self.lookup].dict_name]
        if self._is_key_field(bt):                               # the matched entry's key equals the key it was looked up with
            return self.key_src.spec(op, env, lookups)
        return abi.src_lookup(self.lookup, bt.field_index(self.field, op), self.year)

    def decode_info(self, op, env, lookups):
        """(decoder string array or None, numpy dtype) — a payload inherits them from its table."""
        if self.kind == "col":
            return self.decoder, self.dtype
        bt = env[lookups[self.lookup].dict_name]
        if self._is_key_field(bt):
            return self.key_src.decode_info(op, env, lookups)
        idx = bt.field_index(self.field, op)
        return (None, np.dtype(np.int64)) if self.year else (bt.decoder_of(self.field, idx), np.dtype(bt.payload_dtypes[idx]))

    def same_slot(self, other, op, env, lookups):
        """Do two payload sources read the same device values (one slot can serve both)?"""
        a, b = self.spec(op, env, lookups), other.spec(op, env, lookups)
        return a[0] == b[0] and all(x is y or x == y for x, y in zip(a[1:], b[1:]))


def _source_of(eng, op, htab, e, lookups, as_group_key=False):
    idx_of = lambda lk: [repr(x) for x in lookups].index(repr(lk))      # noqa: E731
    if isinstance(e, Col):
        arr = htab.array(e.name, op)
        if arr.dtype.kind == "U":
            coded = eng.dict_column(arr) if as_group_key else None
            if coded is not None:                                # group key: dictionary codes, one group per distinct text
                return _Src("col", col=coded[0], decoder=coded[1], dtype=np.dtype(np.int64))
            src = _Src("col", col=eng.rowid_column(htab.nrows), decoder=arr, dtype=np.dtype(np.int64))
            src.is_rowid, src.host = True, (0, htab.nrows)
            return src
        src = _Src("col", col=eng.column(arr), dtype=arr.dtype)
        src.host = arr if arr.dtype == np.int64 else None
        return src
    if isinstance(e, PayloadField):
        return _Src("lookup", lookup=idx_of(e.lookup), field=e.field)
    if isinstance(e, Lookup):
        return _Src("lookup", lookup=idx_of(e), field=None)
    if isinstance(e, Call) and e.fn == "extractYear" and isinstance(e.args[0], (PayloadField, Lookup)):
        inner = e.args[0]
        lk, field = (inner.lookup, inner.field) if isinstance(inner, PayloadField) else (inner, None)
        return _Src("lookup", lookup=idx_of(lk), field=field, year=True)
    raise UnsupportedQuery("line %d: unsupported value source %r" % (op.lineno, e))


def _decode_column(values, decoder, dtype):
    if decoder is not None:
        if decoder.dtype.kind == "U":
            return decode_text(values, decoder)                  # text of a large result stays references until it is read
        return np.asarray(decoder)[values]
    return values.view(dtype) if np.dtype(dtype) != values.dtype else values


def _prepare_general(eng, op, htab, flt, contains_lookups, accumulate_into=()):
    """Loops with derived keys / payloads / operands (multi-join chains): sdqh_build and
    sdqh_lookup_aggregate with explicit lookup steps."""
    ctx = eng.ctx
    n = htab.nrows
    lookups = []
    # pure membership conditions first: they are the selective ones (a HAVING set, a filtered build),
    # and the first lookup is the one the kernels test on the streamed key before queueing a row
    for lk in contains_lookups:
        _walk_lookups(lk, lookups)
    if op.probe is not None:
        _walk_lookups(op.probe, lookups)
    _walk_lookups(op.key, lookups)
    if not (isinstance(op.val, Const)):
        _walk_lookups(op.val, lookups)
    if len(lookups) > abi.MAX_LOOKUP:
        raise UnsupportedQuery("line %d: more than %d lookups in one loop" % (op.lineno, abi.MAX_LOOKUP))
    lookup_keys = []
    for lk in lookups:
        parts = [x for _, x in lk.key.fields] if isinstance(lk.key, RecordCons) else [lk.key]
        if len(parts) > 2:
            raise UnsupportedQuery("line %d: lookup keys of more than two fields are not supported" % op.lineno)
        lookup_keys.append([_source_of(eng, op, htab, x, lookups) for x in parts])

    def resolve_lookups(env):
        out = []
        for lk, keys in zip(lookups, lookup_keys):
            bt = env.get(lk.dict_name)
            if not isinstance(bt, BuiltTable):
                raise UnsupportedQuery("line %d: '%s' is not a built table" % (op.lineno, lk.dict_name))
            if (bt.key_parts is not None) != (len(keys) == 2):
                raise UnsupportedQuery("line %d: lookup into '%s' does not match its key shape" % (op.lineno, lk.dict_name))
            out.append((bt.table, [k.spec(op, env, lookups) for k in keys]))
        return out

    key_is_record = isinstance(op.key, RecordCons)
    key_fields = op.key.fields if key_is_record else [(None, op.key)]
    if len(key_fields) > 2:
        raise UnsupportedQuery("line %d: keys of more than two fields are not supported" % op.lineno)
    key_srcs = [_source_of(eng, op, htab, e, lookups, as_group_key=not op.unique) for _, e in key_fields]

    if op.unique:
        val_is_record = isinstance(op.val, RecordCons)
        vfields = op.val.fields if val_is_record else ([] if (isinstance(op.val, Const) and op.val.value is True) else [(None, op.val)])
        # a value field that repeats the (single-column) key needs no payload slot: it is read back from the key
        is_key = [len(key_fields) == 1 and isinstance(key_fields[0][1], Col) and isinstance(e, Col) and e.name == key_fields[0][1].name for _, e in vfields]
        stored = [(fname, e) for (fname, e), k in zip(vfields, is_key) if not k]
        pay_srcs = [_source_of(eng, op, htab, e, lookups) for _, e in stored]
        _link_key_sources(pay_srcs + key_srcs + [k for ks in lookup_keys for k in ks], lookup_keys)
        key_names = [fname or (e.name if isinstance(e, Col) else "key%d" % i) for i, (fname, e) in enumerate(key_fields)]
        accumulate = op.out in accumulate_into

        cached = {}

        def run_build(env):
            # The marshalled call (sources, payload slots, decoders) depends on the layouts of the looked-up tables only, and
            # those are the same run after run: computed once per layout signature; a run then refreshes the table handles.
            # (At SF 0.002, where the kernels do nothing, Q5 spent 52 of its 177 us per run in this Python.)
            bts = [env.get(lk.dict_name) for lk in lookups]
            sig = tuple(bt.layout_sig() if isinstance(bt, BuiltTable) else None for bt in bts)
            c = cached.get("c")
            if c is not None and c[0] == sig:
                _, larr, karr, nk, parr, npay, val_fields, row_dtypes, slot_decoder, field_decoders, slot_roots, slot_plain, kparts, kpd, kdec = c
                for i, bt in enumerate(bts):
                    larr[i].table = bt.table.handle
                table = ctx.build_marshalled(n, flt, larr, len(bts), karr, nk, parr, npay, accumulate, keep=[bt.table for bt in bts])
                out = BuiltTable(table, key_names[0], key_is_record, val_fields, val_is_record, row_dtypes)
                out.decoders, out.field_decoders, out.slot_roots, out.slot_plain = slot_decoder, field_decoders, slot_roots, slot_plain
                out.key_parts, out.key_part_decoders, out.key_decoder = kparts, kpd, kdec
                return out
            bt = _run_build_first(env)
            look, keyspecs, uniq = bt._marshalled
            cached["c"] = (sig, abi._lookups(look), abi._sources(keyspecs), len(keyspecs), abi._sources(uniq), len(uniq), bt.val_fields, bt.payload_dtypes,
                           bt.decoders, bt.field_decoders, bt.slot_roots, bt.slot_plain, bt.key_parts, getattr(bt, "key_part_decoders", None), bt.key_decoder)
            return bt

        def _run_build_first(env):
            # fields that read the same device values share one payload slot: the text fields of one
            # source row are one row reference (they differ in the host array it indexes), a field that
            # is the looked-up table's key is the lookup key itself
            specs = [p.spec(op, env, lookups) for p in pay_srcs]
            infos = [p.decode_info(op, env, lookups) for p in pay_srcs]
            uniq, slot_idx = [], []
            for sp in specs:
                j = next((j for j, u in enumerate(uniq) if len(u) == len(sp) and all(x is y or (not hasattr(x, "handle") and x == y) for x, y in zip(u, sp))), None)
                if j is None:
                    uniq.append(sp); j = len(uniq) - 1
                slot_idx.append(j)
            if len(uniq) > abi.MAX_PAYLOAD:
                raise UnsupportedQuery("line %d: more than %d distinct payload values per entry" % (op.lineno, abi.MAX_PAYLOAD))
            look, keyspecs = resolve_lookups(env), [k.spec(op, env, lookups) for k in key_srcs]
            table = ctx.build(n, flt, look, keyspecs, uniq, accumulate=accumulate)
            it = iter(slot_idx)
            val_fields = [(fname, "key" if k else next(it)) for (fname, _), k in zip(vfields, is_key)]
            slot_dtype, slot_decoder = {}, {}
            for j, info in zip(slot_idx, infos):
                slot_dtype.setdefault(j, info[1])
                if info[0] is not None:
                    slot_decoder.setdefault(j, info[0])
            bt = BuiltTable(table, key_names[0], key_is_record, val_fields, val_is_record, [slot_dtype[j] for j in range(len(uniq))])
            bt.decoders = slot_decoder
            bt.field_decoders = {fname: info[0] for (fname, _), info in zip(stored, infos) if info[0] is not None}
            bt.slot_roots, bt.slot_plain = {}, {}           # what each payload slot is a function of (see _share_spec)
            for j, src in zip(slot_idx, pay_srcs):
                if j not in bt.slot_roots:
                    bt.slot_roots[j] = _roots_of(src, env, lookups, lookup_keys)
                    rng = _plain_range(eng, src, env, lookups)
                    if rng is not None:
                        bt.slot_plain[j] = rng
            if len(key_srcs) == 2:
                bt.key_parts = key_names
                bt.key_part_decoders = [k.decode_info(op, env, lookups)[0] for k in key_srcs]
                # bounds of the packed key where both parts' ranges are known (xplan.DictTable unpacks by signed division: high part < 2^31)
                r = [_plain_range(eng, k, env, lookups) for k in key_srcs]
                if all(x is not None and x[0] >= 0 and x[0] + x[1] <= (1 << 31) for x in r):
                    bt.key_bounds = ((r[0][0] << 32) | r[1][0], ((r[0][0] + r[0][1] - 1) << 32) | (r[1][0] + r[1][1] - 1))
            bt.key_decoder = key_srcs[0].decode_info(op, env, lookups)[0] if len(key_srcs) == 1 else None
            bt._marshalled = (look, keyspecs, uniq)
            return bt
        return run_build

    if op.kind != "dict":
        raise UnsupportedQuery("line %d: scalar sums with lookups are not supported yet" % op.lineno)
    # aggregation over a small group domain with lookups
    val_is_record = isinstance(op.val, RecordCons)
    names = [nm for nm, _ in op.val.fields] if val_is_record else [None]
    exprs = [e for _, e in op.val.fields] if val_is_record else [op.val]
    slots = []
    try:
        shape = ";".join(e.shape(slots) for e in exprs)
    except NotImplementedError:
        raise UnsupportedQuery("line %d: value expression outside the backend's vocabulary: %r" % (op.lineno, op.val))
    if shape not in TUPLE_SHAPES:
        raise UnsupportedQuery("line %d: value tuple shape '%s' is not in the HIP backend's vocabulary %s" % (op.lineno, shape, sorted(TUPLE_SHAPES)))
    abi_shape, count_idx = TUPLE_SHAPES[shape]
    operand_srcs = []
    for slot in slots:
        if slot[0] == "col":
            arr = htab.array(slot[1], op)
            if arr.dtype != np.float64:
                raise UnsupportedQuery("line %d: value operand '%s' must be a float column" % (op.lineno, slot[1]))
            operand_srcs.append(_Src("col", col=eng.column(arr), dtype=arr.dtype))
        else:                                                   # ("payload", dict_name, key repr, field)
            j = [repr(x.key) + x.dict_name for x in lookups].index(slot[2] + slot[1])
            operand_srcs.append(_Src("lookup", lookup=j, field=slot[3]))
    _link_key_sources(operand_srcs + key_srcs + [k for ks in lookup_keys for k in ks], lookup_keys)

    cached = {}

    def run_lookup_aggregate(env):
        # (the marshalled call is cached per layout signature of the looked-up tables, as in run_build)
        bts = [env.get(lk.dict_name) for lk in lookups]
        sig = tuple(bt.layout_sig() if isinstance(bt, BuiltTable) else None for bt in bts)
        c = cached.get("c")
        if c is None or c[0] != sig:
            for o in operand_srcs:
                if o.kind == "lookup" and np.dtype(o.decode_info(op, env, lookups)[1]) != np.float64:
                    raise UnsupportedQuery("line %d: a looked-up value operand must be a float payload" % op.lineno)
            look = resolve_lookups(env)
            keyspecs, opspecs = [k.spec(op, env, lookups) for k in key_srcs], [o.spec(op, env, lookups) for o in operand_srcs]
            c = cached["c"] = (sig, abi._lookups(look), abi._sources(keyspecs), len(keyspecs), abi._sources(opspecs),
                               [key_srcs[i].decode_info(op, env, lookups) for i in range(len(key_fields))])
        _, larr, karr, nk, oarr, key_infos = c
        for i, bt in enumerate(bts):
            larr[i].table = bt.table.handle
        def finish(keys, vals, cnts):
            kf = []
            for i, (fname, e) in enumerate(key_fields):
                decoder, dtype = key_infos[i]
                kf.append((fname or "key%d" % i, _decode_column(keys[:, i].copy(), decoder, dtype)))
            vf = _value_arrays(names, count_idx, [vals[:, j] for j in range(vals.shape[1])], cnts)
            d = _merge_equal_keys(DictResult(kf, vf, key_is_record, val_is_record))
            d.encoding_fp = cached.get("fp")                     # (the multi-GPU runner: may the ranks' partial groups be folded by packed key?)
            return d

        def collected(collect):
            def resolve():
                try:
                    return finish(*collect())
                except abi.SdqhError as exc:
                    if exc.code == abi.ERR_OVERFLOW:
                        raise UnsupportedQuery("line %d: more than %d groups" % (op.lineno, abi.MAX_LOOKUP_GROUPS))
                    raise
            return Pending(resolve)
        if "fp" not in cached or cached.get("fp_sig") is not key_infos:
            cached["fp"], cached["fp_sig"] = _key_decoders_fingerprint(key_infos), key_infos
        try:
            fold = env.get("__group_fold__")
            if fold is not None:
                # a row shard of a multi-GPU run: this rank's partial groups into the collective's buffer, the ranks' blocks folded on the
                # device behind ONE all-gather (as xplan's group-by programs: dist.DistributedRunner gives the exchange where it has checked
                # that the packed keys mean the same on every rank).  No exchange: waited for, merged on the host by decoded values.
                exchange = fold(op.out, cached["fp"])
                if exchange is not None:
                    return collected(ctx.lookup_aggregate_folded_marshalled(n, flt, larr, len(bts), karr, nk, abi_shape, oarr, exchange))
            elif op.out in env.get("__defer__", ()):
                # the plan's last device call: launched, not waited for (engine.PreparedPlan.run finishes the plan when the result is first looked at)
                return collected(ctx.lookup_aggregate_async_marshalled(n, flt, larr, len(bts), karr, nk, abi_shape, oarr))
            keys, vals, cnts = ctx.lookup_aggregate_marshalled(n, flt, larr, len(bts), karr, nk, abi_shape, oarr)
        except abi.SdqhError as exc:
            if exc.code == abi.ERR_OVERFLOW:
                raise UnsupportedQuery("line %d: more than %d groups" % (op.lineno, abi.MAX_LOOKUP_GROUPS))
            raise
        return finish(keys, vals, cnts)
    return run_lookup_aggregate


def _key_decoders_fingerprint(key_infos):
    """What the packed group key of a lookup-aggregate loop MEANS, as a digest (xplan._encoding_fingerprint's counterpart for the
    fixed-shape loop): per key part its dtype and its decoder's contents.  Ranks whose digests are equal may fold their partial groups by
    packed key on the device; a decoder that is a table's own large column (row references, local to a rank's shard) gives None."""
    import hashlib
    h = hashlib.sha1(b"lookup_aggregate;")
    for decoder, dtype in key_infos:
        h.update((np.dtype(dtype).str if dtype is not None else "-").encode())
        if decoder is None:
            h.update(b"raw;")
            continue
        arr = decoder if isinstance(decoder, np.ndarray) else None
        if arr is None or len(arr) > (1 << 16):
            return None
        a = np.ascontiguousarray(arr)
        h.update(a.dtype.str.encode())
        h.update(repr(a.tolist()).encode() if a.dtype == object else a.view(np.uint8).tobytes())
    return h.hexdigest()


_DENSE_MERGE_CELLS = 1 << 26


def _all_distinct(a):
    if a.dtype.kind in "iu" and len(a):
        lo, hi = int(a.min()), int(a.max())
        if hi - lo < _DENSE_MERGE_CELLS:
            seen = np.zeros(hi - lo + 1, bool); seen[a - lo] = True
            return int(seen.sum()) == len(a)
    return len(np.unique(a)) == len(a)


def _merge_equal_keys(d):
    """Groups were formed on row references (two may decode to the same text) or on the matched entry
    of a probed table while the output key names only fields several entries share (Q10: orders of one
    customer): fold rows with equal keys, the reference's `AddMap` of the last mile (map_helper.h:1-23)."""
    n = d.size()
    if n < 2:
        return d
    if n > 4096:                                             # large results: sort-based grouping in numpy
        cols = [a for _, a in d.key_fields]
        # an integer stand-in per column: dictionary codes as they are (equal code <=> equal text), other text factorised
        # (equal code <=> equal text), text and floating-point columns factorised (np.unique: NaNs fold into one code)
        ints = [c.refs if isinstance(c, TextRefs) and c.distinct else
                (np.asarray(c) if np.asarray(c).dtype.kind in "iu" else np.unique(np.asarray(c), return_inverse=True)[1].reshape(-1)) for c in cols]
        lo = [int(c.min()) for c in ints]
        span = [int(c.max()) - l + 1 for c, l in zip(ints, lo)]
        cells = 1
        for w in span:
            cells *= w
        if cells <= _DENSE_MERGE_CELLS:                      # the stand-ins span a small box: bucket the rows, no sort (Q16: 1.2 M rows into 190 K cells)
            code = np.zeros(n, np.int64)
            for c, l, w in zip(ints, lo, span):
                code = code * w + (c - l)
            rep = np.full(cells, -1, np.int64)
            rep[code[::-1]] = np.arange(n - 1, -1, -1)       # the first row of every occupied cell
            occupied = np.nonzero(rep >= 0)[0]
            if len(occupied) == n:
                return d
            rows = rep[occupied]
            kf = [(nm, c[rows]) for (nm, _), c in zip(d.key_fields, cols)]
            vf = [(nm, np.bincount(code, weights=a, minlength=cells)[occupied].astype(a.dtype)) for nm, a in d.val_fields]
            return DictResult(kf, vf, d.key_is_record, d.val_is_record)
        order = np.lexsort(list(reversed(ints)))
        new_group = np.zeros(n, bool); new_group[0] = True
        for c in ints:
            cs = c[order]
            new_group[1:] |= cs[1:] != cs[:-1]
        if new_group.all():
            return d
        gid = np.cumsum(new_group) - 1
        first = np.nonzero(new_group)[0]
        kf = [(nm, c[order[first]]) for (nm, _), c in zip(d.key_fields, cols)]
        vf = []
        for nm, a in d.val_fields:
            acc = np.zeros(len(first), a.dtype)
            np.add.at(acc, gid, a[order])
            vf.append((nm, acc))
        return DictResult(kf, vf, d.key_is_record, d.val_is_record)
    rows = list(zip(*[a.tolist() for _, a in d.key_fields]))
    if len(set(rows)) == n:
        return d
    order, merged = [], {}
    for i, k in enumerate(rows):
        if k not in merged:
            merged[k] = [a[i] for _, a in d.val_fields]; order.append(k)
        else:
            merged[k] = [x + a[i] for x, (_, a) in zip(merged[k], d.val_fields)]
    kf = [(nm, np.array([k[j] for k in order], dtype=a.dtype)) for j, (nm, a) in enumerate(d.key_fields)]
    vf = [(nm, np.array([merged[k][j] for k in order], dtype=a.dtype)) for j, (nm, a) in enumerate(d.val_fields)]
    return DictResult(kf, vf, d.key_is_record, d.val_is_record)


def _is_simple(op, htab, lookups):
    """True when the specialised single-purpose calls cover the loop (all keys / payloads / operands are
    plain columns of the scanned row, lookups keyed by one int column)."""
    def plain(e):
        return isinstance(e, Col) and htab.cols.get(e.name) is not None and htab.cols[e.name].dtype.kind != "U"
    for lk in ([op.probe] if op.probe else []) + list(lookups):
        if not (isinstance(lk.key, Col) and plain(lk.key)):
            return False
    if op.kind == "scalar":
        return not lookups and op.probe is None
    key_fields = op.key.fields if isinstance(op.key, RecordCons) else [(None, op.key)]
    if op.unique:
        if len(key_fields) != 1 or not plain(key_fields[0][1]):
            return False
        vals = op.val.fields if isinstance(op.val, RecordCons) else ([] if isinstance(op.val, Const) else [(None, op.val)])
        return all(plain(e) for _, e in vals)
    found = []
    _walk_lookups(op.val, found)
    if found:
        return False
    if op.probe is None and not lookups:
        return all(isinstance(e, Col) for _, e in key_fields)
    if op.probe is not None and not lookups:           # fused probe-aggregate: the group is the matched entry
        pk = op.probe.key.name
        # the group is the matched entry: every key field is the probe key or a field of the entry
        return all((isinstance(e, Col) and e.name == pk) or (isinstance(e, PayloadField) and repr(e.lookup) == repr(op.probe)) for _, e in key_fields)
    return False


def _prepare_scalar(eng, op, htab, conds, val):
    """One scalar sum `Σ val over rows passing conds` -> closure(env) -> float.  Conditions of the
    form `tbl[col] != None` become semi-join probes (sdqh_scan_probe_sum)."""
    ctx, n = eng.ctx, htab.nrows
    flt, lookups = _build_filter(eng, op, htab, conds)
    if op.probe is not None:
        raise UnsupportedQuery("line %d: scalar sums inside joinProbe are not supported" % op.lineno)
    specs = _probe_specs(eng, op, htab, lookups)
    tup, _, count_idx = _build_tuple(eng, op, htab, val)

    def run_scalar(env):
        if specs:
            vals, cnt = ctx.scan_probe_sum(n, flt, _resolve_probes(op, env, specs), tup)
        else:
            vals, cnt = ctx.scan_filter_sum(n, flt, tup)
        return float(cnt) if count_idx is not None else float(vals[0])
    return run_scalar


def _eval_scalar_expr(e, env, lineno):
    if isinstance(e, Const):
        return float(e.value)
    if isinstance(e, ScalarField):
        v = env[e.name]
        if e.field is None:
            return float(v)
        return float(v[e.field])
    if isinstance(e, Bin):
        a, b = _eval_scalar_expr(e.left, env, lineno), _eval_scalar_expr(e.right, env, lineno)
        return {"+": a + b, "-": a - b, "*": a * b, "/": (a / b) if b != 0.0 else float("nan")}[e.op]
    raise UnsupportedQuery("line %d: unsupported scalar expression %r" % (lineno, e))


def _has_codable_text_payload(eng, op, htab):
    vals = [e for _, e in op.val.fields] if isinstance(op.val, RecordCons) else [op.val]
    return any(isinstance(e, Col) and htab.cols.get(e.name) is not None and htab.cols[e.name].dtype.kind == "U"
               and eng.dict_column(htab.cols[e.name]) is not None for e in vals)


def _is_stream_loop(op):
    """A loop that only streams its table: a scalar sum or an aggregation (not a unique build), no probe, no lookup anywhere."""
    if op.kind not in ("scalar", "dict") or (op.kind == "dict" and op.unique) or op.probe is not None:
        return False
    found = []
    for c in list(op.conds) + [c for _, _, fc in (op.fields or []) for c in fc]:
        if isinstance(c, Contains):
            return False
        _walk_lookups(c, found)
    if op.kind == "dict":
        _walk_lookups(op.key, found)
    if op.val is not None and not isinstance(op.val, Const):
        _walk_lookups(op.val, found)
    for _, e, _ in (op.fields or []):
        _walk_lookups(e, found)
    return not found


def _plain_values_build(op, htab):
    """A unique build the value-queue stage kernel can run (csrc/sdqh_xkernels.hpp: x_vstage8): keyed by one integer column, at most
    two payload fields that are numeric columns of the scanned row, conditions that compare columns with constants, membership
    tests keyed by a plain integer column."""
    def num_col(e, kinds="if"):
        return isinstance(e, Col) and htab.cols.get(e.name) is not None and htab.cols[e.name].dtype.kind in kinds
    if not (op.kind == "dict" and op.unique) or isinstance(op.key, RecordCons) or not num_col(op.key, "i"):
        return False
    vals = [e for _, e in op.val.fields] if isinstance(op.val, RecordCons) else [op.val]
    vals = [e for e in vals if not (isinstance(e, Const) and e.value is True)]
    if len({repr(e) for e in vals}) > 2 or not all(num_col(e) for e in vals):
        return False
    for c in op.conds:
        if isinstance(c, Contains):
            if not num_col(c.lookup.key, "i"):
                return False
        elif not (isinstance(c, Cmp) and all(num_col(x) or (isinstance(x, Const) and isinstance(x.value, (int, float)) and not isinstance(x.value, bool)) for x in (c.left, c.right))):
            return False
    return op.probe is None or num_col(op.probe.key, "i")


def _single_key_lookup_build(op, htab):
    """A unique build keyed by ONE integer column whose conditions / payload look other tables up (Q5's orders: the customer's nation
    as payload) and that touches numeric columns only: the tight-encoded program (2-byte date codes, 8 rows per lane) beats the fixed
    `k_build_lookup` on the big ones (15 M orders 0.083 -> 0.065 ms); composite keys stay with the fixed call (its linearised /
    prefiltered layouts), text payloads too."""
    def num_col(e, kinds="if"):
        return isinstance(e, Col) and htab.cols.get(e.name) is not None and htab.cols[e.name].dtype.kind in kinds
    if not (op.kind == "dict" and op.unique) or isinstance(op.key, RecordCons) or not num_col(op.key, "i"):
        return False
    found, cols = [], []

    def walk(e):
        if isinstance(e, Col):
            cols.append(e)
        elif isinstance(e, (PayloadField, Contains)):
            found.append(e.lookup); walk(e.lookup.key)
        elif isinstance(e, Lookup):
            found.append(e); walk(e.key)
        elif isinstance(e, (Bin, Cmp)):
            walk(e.left); walk(e.right)
        elif isinstance(e, (And, Or)):
            for t in e.terms:
                walk(t)
        elif isinstance(e, Not):
            walk(e.term)
        elif isinstance(e, RecordCons):
            for _, x in e.fields:
                walk(x)
        elif isinstance(e, Const):
            if isinstance(e.value, str):
                cols.append(None)
        else:
            cols.append(None)                                     # anything else (conditional values, calls, text tests): not this route
    for c in op.conds:
        walk(c)
    if op.probe is not None:
        found.append(op.probe); walk(op.probe.key)
    in_conds = len(found)
    if not (isinstance(op.val, Const) and op.val.value is True):
        walk(op.val)
    # (a payload that reads a looked-up field: a build that only TESTS membership — Q18's orders against 57 keys — keeps the fixed
    # kernel's filtered staging, 0.05 ms against 0.34 ms as a program)
    return len(found) > in_conds and all(c is not None and num_col(c) for c in cols) and all(not isinstance(lk.key, RecordCons) for lk in found)


def _prepare_scan(eng, op, htab, accumulate_into, member_only=False, as_table=False, coded_text=False, _no_stream=False):
    """closure(env) for one table loop: the tuned fixed-shape calls when the loop is one of their shapes,
    a row program (xplan.py: a kernel specialised on the loop's own conditions and values) otherwise.
    A fixed-shape closure can still refuse at run time (a group count beyond its kernels): the loop then
    moves to its row program for good."""
    from . import xplan
    if getattr(eng, "force_programs", False):
        # every loop through a row program (tests / A-B measurements: the specialised kernels against the tuned ones)
        try:
            return xplan.prepare_scan(eng, op, htab, accumulate_into, member_only, as_table)
        except UnsupportedQuery:
            pass
    routed, small_only = False, False
    if not _no_stream and getattr(eng, "program_routes", None):
        if op.kind == "dict" and not op.unique and op.probe is not None and op.probe.dict_name in accumulate_into:
            # (a group named by fields of the matched entry only is a program while its groups are a handful — Q5: the nation's name; when
            # they are many — Q10: customer fields of an order — the fixed call takes over, which folds the entries that share those
            # fields on the device: sdqh_table_share_groups)
            kfs = op.key.fields if isinstance(op.key, RecordCons) else [(None, op.key)]
            routed = "probe" in eng.program_routes
            small_only = not (isinstance(op.probe.key, Col) and any(isinstance(e, Col) and e.name == op.probe.key.name for _, e in kfs))
        elif op.kind == "dict" and not op.unique and op.probe is None and "groups" in eng.program_routes and not as_table:
            routed, small_only = True, True                        # (any small-domain aggregation with lookups: the fixed lookup-aggregate call otherwise)
        elif op.kind == "dict" and op.unique:
            big = not member_only and htab.nrows >= (1 << 20)
            routed = "build" in eng.program_routes or ("values" in eng.program_routes and big and _plain_values_build(op, htab)) \
                or ("lookups" in eng.program_routes and big and _single_key_lookup_build(op, htab))
    if routed or (getattr(eng, "stream_programs", False) and not _no_stream and not as_table and not member_only and _is_stream_loop(op)):
        try:
            x = xplan.prepare_scan(eng, op, htab, accumulate_into, member_only, as_table, small_groups_only=small_only or not routed)
        except UnsupportedQuery:
            x = None
        if x is not None:
            state0 = {"fixed": None}

            def run_stream(env):
                if state0["fixed"] is None:
                    try:
                        return x(env)
                    except UnsupportedQuery:                        # e.g. more groups than the program's sinks hold: the fixed shapes decide
                        state0["fixed"] = _prepare_scan(eng, op, htab, accumulate_into, member_only, as_table, coded_text, _no_stream=True)
                return state0["fixed"](env)
            return run_stream
    if coded_text and op.kind == "dict" and op.unique and op.probe is None and _has_codable_text_payload(eng, op, htab):
        # a later loop tests this build's text values: build them as dictionary codes (row program), not row references
        try:
            return xplan.prepare_scan(eng, op, htab, accumulate_into, member_only, as_table)
        except UnsupportedQuery:
            pass
    if as_table and op.kind == "dict" and not op.unique and not (op.probe is not None and xplan.groups_by_entry(op)):
        # an aggregated dictionary that later loops look up has to be a table whatever its size: the small-domain
        # group-by calls return their groups to the host
        kf = op.key.fields if isinstance(op.key, RecordCons) else [(None, op.key)]
        single_int = len(kf) == 1 and isinstance(kf[0][1], Col) and htab.cols.get(kf[0][1].name) is not None and htab.cols[kf[0][1].name].dtype == np.int64
        if not single_int or op.probe is not None or any(not isinstance(c, Cmp) for c in op.conds):
            return xplan.prepare_scan(eng, op, htab, accumulate_into, member_only, as_table=True)
    try:
        fixed = _prepare_scan_fixed(eng, op, htab, accumulate_into, member_only, as_table)
    except UnsupportedQuery as first:
        try:
            return xplan.prepare_scan(eng, op, htab, accumulate_into, member_only, as_table)
        except UnsupportedQuery as second:
            raise UnsupportedQuery("%s\n  (as a row program: %s)" % (first, second))
    state = {"x": None}

    def run(env):
        if state["x"] is None:
            try:
                try:
                    out = fixed(env)
                except abi.SdqhError as exc:
                    # the library has no fixed-shape kernel for this loop (e.g. a filter layout without an instance): a row program's job
                    if exc.code != abi.ERR_UNSUPPORTED:
                        raise
                    raise UnsupportedQuery("line %d: %s" % (op.lineno, exc))
                if as_table and isinstance(out, DictResult):
                    # a later loop looks this dictionary up, and the fixed-shape call returned its groups to the host (a small
                    # group domain: nothing on the device was changed): the row program aggregates into a table instead (Q2)
                    raise UnsupportedQuery("line %d: the groups came back to the host, a later loop needs them as a table" % op.lineno)
                return out
            except UnsupportedQuery as first:
                try:
                    x = xplan.prepare_scan(eng, op, htab, accumulate_into, member_only, as_table)
                    out = x(env)
                except UnsupportedQuery as second:
                    raise UnsupportedQuery("%s\n  (as a row program: %s)" % (first, second))
                state["x"] = x
                return out
        return state["x"](env)
    return run


def _prepare_scan_fixed(eng, op, htab, accumulate_into, member_only=False, as_table=False):
    ctx = eng.ctx
    if op.kind == "scalar":
        return _prepare_scalar(eng, op, htab, op.conds, op.val)
    if op.kind == "scalar_record":
        runs = [(name, _prepare_scalar(eng, op, htab, op.conds + fconds, expr)) for name, expr, fconds in op.fields]
        return lambda env: {name: run(env) for name, run in runs}
    flt, lookups = _build_filter(eng, op, htab, op.conds)
    n = htab.nrows
    if not _is_simple(op, htab, lookups):
        return _prepare_general(eng, op, htab, flt, lookups, accumulate_into)

    # ---- dictionary outputs ----
    key_is_record = isinstance(op.key, RecordCons)
    key_fields = op.key.fields if key_is_record else [(None, op.key)]

    if op.unique:
        # K-B: unique build keyed by one int column of the scanned row
        if len(key_fields) != 1 or not isinstance(key_fields[0][1], Col):
            raise UnsupportedQuery("line %d: unique builds need a single int column as key (composite keys: not yet)" % op.lineno)
        kname = key_fields[0][1].name
        karr = htab.array(kname, op)
        if karr.dtype != np.int64:
            raise UnsupportedQuery("line %d: build key '%s' must be an int column" % (op.lineno, kname))
        val_is_record = isinstance(op.val, RecordCons)
        vfields = op.val.fields if val_is_record else ([] if (isinstance(op.val, Const) and op.val.value is True) else [(None, op.val)])
        payload_cols, payload_dtypes, val_fields = [], [], []
        for fname, e in vfields:
            if not isinstance(e, Col):
                raise UnsupportedQuery("line %d: build payloads must be columns of the scanned row (derived payloads: not yet)" % op.lineno)
            if e.name == kname:
                val_fields.append((fname, "key"))
                continue
            arr = htab.array(e.name, op)
            if arr.dtype.kind == "U":
                raise UnsupportedQuery("line %d: string payloads are not supported yet" % op.lineno)
            val_fields.append((fname, len(payload_cols)))
            payload_cols.append(eng.column(arr))
            payload_dtypes.append(arr.dtype)
        specs = _probe_specs(eng, op, htab, ([op.probe] if op.probe else []) + lookups)
        kcol = eng.column(karr)
        accumulate = op.out in accumulate_into
        key_name = key_fields[0][0] or kname

        def run_build(env):
            table = None
            if member_only and not payload_cols and not accumulate:
                try:
                    table = ctx.build_key_set(n, flt, _resolve_probes(op, env, specs), kcol)
                except abi.SdqhError as exc:
                    if exc.code != abi.ERR_UNSUPPORTED:          # key range unsuitable for a bitmap: ordinary build
                        raise
            if table is None:
                table = ctx.hash_build_unique(n, flt, _resolve_probes(op, env, specs), kcol, payload_cols, accumulate=accumulate)
            bt = BuiltTable(table, key_name, key_is_record, val_fields, val_is_record, payload_dtypes)
            bt.slot_cols = payload_cols                     # where each payload slot's values come from (ranges on demand, xplan._slot_range)
            return bt
        return run_build

    # ---- aggregations ----
    tup, vnames, count_idx = _build_tuple(eng, op, htab, op.val)
    val_is_record = isinstance(op.val, RecordCons)

    if op.probe is None and not lookups and len(key_fields) == 1 and isinstance(key_fields[0][1], Col) \
            and htab.array(key_fields[0][1].name, op).dtype == np.int64:
        # K-C keyed by one int column: a small value range goes to the register / LDS group-by below,
        # anything else to the large-domain group-by (sdqh_groupby_key): the result is a table
        gname = key_fields[0][1].name
        gcol = eng.column(htab.array(gname, op))
        glo, ghi = gcol.minmax() if n else (0, 0)
        if ghi - glo + 1 > abi.MAX_SMALL_GROUPS or as_table:
            hidden = op.out + "$groups"
            key_name = key_fields[0][0] or gname

            def run_groupby_key(env):
                table = ctx.groupby_key(n, flt, gcol, tup)
                bt = BuiltTable(table, key_name, key_is_record, [], val_is_record, [])
                bt.agg = ([(key_name, "key")], vnames, count_idx, key_is_record, val_is_record, abi.TUPLE_NVALUES[tup.shape])
                env[hidden] = bt
                return ("aggregated", hidden)
            return run_groupby_key

    if op.probe is None and not lookups:
        # K-C small domain: every key field is a column of the scanned row
        kcols, decoders = [], []
        for fname, e in key_fields:
            if not isinstance(e, Col):
                raise UnsupportedQuery("line %d: group keys must be columns of the scanned row" % op.lineno)
            arr = htab.array(e.name, op)
            if arr.dtype.kind == "U" and arr.dtype.itemsize == 4:
                decoders.append("U1")
            elif arr.dtype == np.int64:
                decoders.append("i8")
            else:
                raise UnsupportedQuery("line %d: group key '%s' must be a string(1) or int column" % (op.lineno, e.name))
            kcols.append(eng.column(arr))

        def run_groupby(env):
            try:
                keys, vals, cnts = ctx.groupby_small(n, flt, kcols, tup)
            except abi.SdqhError as exc:
                if exc.code == abi.ERR_OVERFLOW:
                    raise UnsupportedQuery("line %d: more than %d groups: large-domain group-by on row columns is not in the "
                                           "HIP backend's vocabulary yet" % (op.lineno, abi.MAX_SMALL_GROUPS))
                raise
            kf = []
            for i, (fname, e) in enumerate(key_fields):
                col = keys[:, i]
                kf.append((fname or e.name, col.astype(np.uint32).view("<U1") if decoders[i] == "U1" else col.copy()))
            vf = _value_arrays(vnames, count_idx, [vals[:, j] for j in range(vals.shape[1])], cnts)
            return DictResult(kf, vf, key_is_record, val_is_record)
        return run_groupby

    # K-C large: the group is the matched entry of the probed table
    if op.probe is None or lookups:
        raise UnsupportedQuery("line %d: aggregations with extra lookups are not supported yet" % op.lineno)
    if not isinstance(op.probe.key, Col):
        raise UnsupportedQuery("line %d: joinProbe index must be probed by a column" % op.lineno)
    pk = op.probe.key.name
    karr = htab.array(pk, op)
    if karr.dtype != np.int64:
        raise UnsupportedQuery("line %d: probe key '%s' must be an int column" % (op.lineno, pk))
    kcol = eng.column(karr)
    probe_name = op.probe.dict_name

    def run_probe_aggregate(env, rows=None):
        """rows = (n, key Column, tuple): aggregate THOSE rows into the table instead of the scanned table's (the multi-GPU runner:
        rows that reached this rank through the exchange; n = 0: bookkeeping only)."""
        bt = env.get(probe_name)
        if not isinstance(bt, BuiltTable):
            raise UnsupportedQuery("line %d: joinProbe index must be a built table" % op.lineno)
        if bt.agg_spec is None:                      # resolve the group key against the table's fields once
            out_key_fields = []
            for fname, e in key_fields:
                if isinstance(e, Col) and e.name == pk:
                    out_key_fields.append((fname or pk, "key"))
                elif isinstance(e, PayloadField) and e.lookup.dict_name == probe_name and repr(e.lookup.key) == repr(op.probe.key):
                    src = dict(bt.val_fields).get(e.field) if e.field is not None else (bt.val_fields[0][1] if bt.val_fields else None)
                    if src is None:
                        raise UnsupportedQuery("line %d: '%s' has no field '%s'" % (op.lineno, probe_name, e.field))
                    out_key_fields.append((fname or e.field, src))
                    bt.agg_fields[fname or e.field] = e.field      # output name -> entry field (its decoder)
                else:
                    raise UnsupportedQuery("line %d: group keys of a probe-aggregate must be the probe key or fields of the matched "
                                           "entry (the group must be determined by the probe key)" % op.lineno)
            bt.agg_spec = out_key_fields                # the group is the matched entry whether or not its key is among the output fields
        if not bt.table.accumulate or bt.agg is not None:
            raise UnsupportedQuery("line %d: table '%s' cannot take this aggregation" % (op.lineno, probe_name))
        if not any(src == "key" for _, src in bt.agg_spec):
            # the output key names fields of the matched entry only (Q10: customer fields of an order):
            # entries whose row references agree form one group on the device
            share = _share_spec(eng, bt)
            if share is not None:
                ctx.table_share_groups(bt.table, *share)
                bt.shared_groups = True
        if rows is None:
            ctx.hash_probe_aggregate(n, flt, bt.table, kcol, tup)
        elif rows[0]:
            ctx.hash_probe_aggregate(rows[0], abi.make_filter(), bt.table, rows[1], rows[2])
        bt.agg = (bt.agg_spec, vnames, count_idx, key_is_record, val_is_record, abi.TUPLE_NVALUES[tup.shape])
        return ("aggregated", probe_name)
    run_probe_aggregate.tuple_shape = tup.shape
    return run_probe_aggregate


def _roots_of(src, env, lookups, lookup_keys):
    """The columns of the scanned row a payload source is a function of ("row": the row number)."""
    if src.kind == "col":
        return frozenset(["row" if src.is_rowid else id(src.col)])
    if src._is_key_field(env[lookups[src.lookup].dict_name]):
        return _roots_of(src.key_src, env, lookups, lookup_keys)
    out = frozenset()
    for k in lookup_keys[src.lookup]:
        out |= _roots_of(k, env, lookups, lookup_keys)
    return out


def _plain_range(eng, src, env, lookups):
    """(lo, span) when the source is a plain int column of the scanned row (possibly read as the key
    of a looked-up entry) or its row number — a value that identifies its root; None otherwise."""
    while src.kind == "lookup":
        if not src._is_key_field(env[lookups[src.lookup].dict_name]):
            return None
        src = src.key_src
    if src.host is None or src.decoder is not None and not src.is_rowid:
        return None
    if src.is_rowid:
        return 0, max(1, src.host[1])
    cache = eng._range_cache
    hit = cache.get(id(src.host))
    if hit is None or hit[0] is not src.host:
        arr = src.host
        hit = cache[id(arr)] = (arr, (int(arr.min()), int(arr.max()) - int(arr.min()) + 1) if len(arr) else (0, 1))
    return hit[1]


def _share_spec(eng, bt):
    """(fields, lo, span) for sdqh_table_share_groups: the payload slots that determine the output
    record of an aggregation keyed by entry fields only.  A slot holding a plain column of the
    build's scanned row (Q10: o_custkey, read as the key of the looked-up customer entry) identifies
    that column's value, and every slot computed from the same columns (fields of the looked-up
    entry, of entries looked up through them) follows from it.  Slots not covered that way must be
    row references / dictionary codes whose decoder has no repeated value.  None: fold on the host."""
    fields_of = bt.agg_fields
    roots, plain = bt.slot_roots, bt.slot_plain
    slots = sorted({src for _, src in bt.agg_spec})
    if roots is None:
        return None
    group = [s for s in slots if s in plain]
    covered = frozenset().union(*[roots[s] for s in group]) if group else frozenset()
    lo, span = [plain[s][0] for s in group], [plain[s][1] for s in group]
    for slot in slots:
        if slot in plain or "row" in covered or roots[slot] <= covered:
            continue
        decs = [bt.decoder_of(fields_of.get(fname), src) for fname, src in bt.agg_spec if src == slot]
        if any(d is None for d in decs) or not any(_distinct_cached(eng, d) for d in decs):
            return None
        group.append(slot); lo.append(0); span.append(max(1, max(len(d) for d in decs)))
    cells = 1
    for w in span:
        cells *= w
    if not group or cells > abi.MAX_SHARE_CELLS:
        return None
    return group, lo, span


def _distinct_cached(eng, arr):
    cache = eng._distinct_cache
    hit = cache.get(id(arr))
    if hit is None or hit[0] is not arr:
        hit = cache[id(arr)] = (arr, _all_distinct(arr))
    return hit[1]


def _compact(eng, table, min_hits, hint_key, **want):
    """K-F in one call: the result size of the previous run of the same plan step (+ slack) sizes
    the device-writable result block, so the count and the rows arrive with one synchronisation."""
    hint = eng.compact_hints.get(hint_key)
    cap = 4096 if hint is None else hint + hint // 8 + 1024
    keys, payload, values, hits, n = eng.ctx.table_compact_into_block(table, min_hits, cap, **want)
    eng.compact_hints[hint_key] = n
    return keys, payload, values, hits


def _device_sort_spec(bt, key_fields, vnames, count_idx, order):
    """[(kind, index, descending, is_f64)] for sdqh_table_topk, or None when a column of `order`
    cannot be ordered on the device (text travels as row references; composite keys are packed)."""
    spec = []
    for name, direction in order:
        desc = direction == "desc"
        hit = [src for fname, src in key_fields if fname == name]
        if hit:
            src = hit[0]
            if src == "key":
                if bt.key_parts is not None or bt.key_decoder is not None:
                    return None
                spec.append((abi.SORT_KEY, 0, desc, False))
            else:
                if bt.decoders.get(src) is not None:
                    return None
                spec.append((abi.SORT_PAYLOAD, src, desc, np.dtype(bt.payload_dtypes[src]).kind == "f"))
        elif name in vnames:
            i = vnames.index(name)
            if count_idx is not None and i == count_idx:
                spec.append((abi.SORT_HITS, 0, desc, False))
            else:
                spec.append((abi.SORT_VALUE, i - (1 if count_idx is not None and count_idx < i else 0), desc, True))
        else:
            raise KeyError("top: the result has no column %r" % name)
    return spec


def _fetch_entries(eng, bt_table, min_hits, hint_key, top, spec, **want):
    """K-F rows of a table: all of them, or with `top` = (k, order) only the first k in that order
    when the device can order them (spec) — otherwise all, ordered afterwards on the host."""
    if top is not None and spec is not None and 1 <= top[0] <= abi.MAX_TOPK and len(spec) <= abi.MAX_SORT_KEYS:
        keys, payload, values, hits = eng.ctx.table_topk(bt_table, min_hits, top[0], spec, want_hits=want.get("want_hits", True))
        return keys, payload, values, hits, True
    keys, payload, values, hits = _compact(eng, bt_table, min_hits, hint_key, **want)
    return keys, payload, values, hits, False


def _lazy_rows_ok(bt, out_key_fields, fields_of, top):
    """May K-F's rows arrive behind the call (ResultSet waits for them on first read)?  Only when nothing between here and
    the ResultSet READS them: no ordering on the host, no text to decode, no packed / shared keys to unfold."""
    if top is not None or getattr(bt, "key_radix", None) is not None or bt.key_parts is not None:
        return False
    if not (any(src == "key" for _, src in out_key_fields) or bt.shared_groups) or bt.int_values:
        return False
    return all(src == "key" or bt.decoder_of(fields_of.get(f), src) is None for f, src in out_key_fields)


def _materialize(eng, value, env, hint_key=None, top=None, lazy_ok=False, defer=False):
    """Device-resident intermediate -> DictResult on the host (K-F's input).  With `top`, the
    DictResult carries `.ordered = True` when the device already applied ORDER BY / LIMIT."""
    if isinstance(value, DictResult):
        return value
    if isinstance(value, tuple) and value and value[0] == "aggregated":
        bt = env[value[1]]
        out_key_fields, vnames, count_idx, key_is_record, val_is_record, nv = bt.agg
        entry_is_group = any(src == "key" for _, src in out_key_fields) or bt.shared_groups
        spec = _device_sort_spec(bt, out_key_fields, vnames, count_idx, top[1]) if top is not None and entry_is_group and bt.key_radix is None else None
        fields_of = bt.agg_fields
        lazy = lazy_ok and bool(getattr(eng, "lazy_results", False)) and _lazy_rows_ok(bt, out_key_fields, fields_of, top)
        want_hits = count_idx is not None or (spec is not None and any(s[0] == abi.SORT_HITS for s in spec))
        hint = eng.compact_hints.get(hint_key)
        if lazy and defer and hint is not None:
            # the plan's last device call, launched and not waited for — not even for the row count (PreparedPlan.run): the block is
            # sized from the previous run of this step (the first run waits and learns the size); a result that outgrew it is noticed
            # when it is collected and the plan re-run
            try:
                collect = eng.ctx.table_compact_deferred(bt.table, 1, hint + hint // 8 + 1024, want_hits=want_hits, replayable=bool(env.get("__record__")))
            except abi.SdqhError as exc:
                if exc.code != abi.ERR_UNSUPPORTED:               # (the option "async_result" is off: the waited-for call below)
                    raise
                collect = None

            def resolve():
                try:
                    keys, payload, values, hits, n = collect()
                except abi.SdqhError as exc:
                    if exc.code == abi.ERR_OVERFLOW and getattr(exc, "needed", None):
                        eng.compact_hints[hint_key] = exc.needed     # (the next launch sizes its block from this)
                    raise
                eng.compact_hints[hint_key] = n
                d = DictResult([(f, keys if src == "key" else _decode_column(payload[src], None, bt.payload_dtypes[src])) for f, src in out_key_fields],
                               _value_arrays(vnames, count_idx, [values[j] for j in range(nv)], hits, bt.int_values), key_is_record, val_is_record)
                d.ordered = False
                return d
            if collect is not None:
                p = Pending(resolve)
                if env.get("__record__"):
                    env["__rows_out__"] = collect.rows_out      # (PlanGraph: the recording is launched again only when no view of an earlier result is alive)
                return p
        keys, payload, values, hits, ordered = _fetch_entries(eng, bt.table, 1, hint_key, top, spec, want_hits=want_hits, lazy=lazy)
        values = [values[j] for j in range(nv)]
        if getattr(bt, "key_radix", None) is not None and out_key_fields == [(bt.key_name, "key")]:    # several key fields in one mixed-radix integer
            from . import xplan
            d = DictResult(xplan._decode_radix(keys, bt.key_radix[0], bt.key_radix[1]), _value_arrays(vnames, count_idx, values, hits, bt.int_values), True, val_is_record)
            d.ordered = False
            return d
        if bt.key_parts is not None and out_key_fields == [(bt.key_name, "key")]:        # a composite group key: its two packed parts
            decs = getattr(bt, "key_part_decoders", None) or [None, None]
            d = DictResult([(bt.key_parts[0], _decode_column(keys >> 32, decs[0], np.int64)), (bt.key_parts[1], _decode_column(keys & 0xFFFFFFFF, decs[1], np.int64))],
                           _value_arrays(vnames, count_idx, values, hits, bt.int_values), True, val_is_record)
            d.ordered = ordered
            return d

        def decode(fname, src, sel=None):
            raw = keys if src == "key" else payload[src]
            if sel is not None:
                raw = raw[sel]
            return raw if src == "key" else _decode_column(raw, bt.decoder_of(fields_of.get(fname), src), bt.payload_dtypes[src])
        def decode_late(fname, src):                         # text of a large result is gathered when it is first read
            dec = None if src == "key" else bt.decoder_of(fields_of.get(fname), src)
            return decode_text(payload[src], dec) if dec is not None and dec.dtype.kind == "U" else decode(fname, src)
        if entry_is_group or ordered:
            d = DictResult([(f, decode_late(f, src)) for f, src in out_key_fields], _value_arrays(vnames, count_idx, values, hits, bt.int_values),
                           key_is_record, val_is_record)
            d.ordered = ordered
            if lazy:
                d.ready = eng.ctx.result_wait                 # the rows are still on their way: whoever reads them first waits
            return d
        # The output key does not identify the entry (Q10: one entry per order, the key names customer
        # fields): fold entries that reference the same source rows first — integer work on the row
        # references — then decode, text last and only for the rows that survive ORDER BY / LIMIT.
        slots = sorted({src for _, src in out_key_fields})
        n = len(payload[slots[0]]) if slots else 0
        if n:
            refs = [np.asarray(payload[sl]) for sl in slots]
            lo = [int(r.min()) for r in refs]
            span = [int(r.max()) - l + 1 for r, l in zip(refs, lo)]
            cells = int(np.prod([float(x) for x in span]))
            want_counts = hits is not None and count_idx is not None
            if cells <= _DENSE_MERGE_CELLS:                  # row references span a small rectangle: bucket, no sort
                code = np.zeros(n, np.int64)
                for r, l, w in zip(refs, lo, span):
                    code = code * w + (r - l)
                present = np.zeros(cells, bool); present[code] = True
                cell_of_group = np.nonzero(present)[0]
                group_of_cell = np.cumsum(present) - 1
                gid = group_of_cell[code]
                values = [np.bincount(gid, weights=v, minlength=len(cell_of_group)) for v in values]
                if want_counts:
                    hits = np.bincount(gid, weights=np.asarray(hits), minlength=len(cell_of_group)).astype(np.int64)
                rest, merged = cell_of_group, []
                for l, w in zip(reversed(lo), reversed(span)):
                    merged.append(rest % w + l); rest = rest // w
                payload = dict(zip(slots, [m.astype(r.dtype) for m, r in zip(reversed(merged), refs)]))
            else:
                order = np.lexsort(list(reversed(refs)))
                sorted_refs = [r[order] for r in refs]
                new_group = np.zeros(n, bool); new_group[0] = True
                for c in sorted_refs:
                    new_group[1:] |= c[1:] != c[:-1]
                first = np.nonzero(new_group)[0]
                payload = dict(zip(slots, [c[first] for c in sorted_refs]))
                values = [np.add.reduceat(v[order], first) for v in values]
                if want_counts:
                    hits = np.add.reduceat(np.asarray(hits)[order], first)
        vf = _value_arrays(vnames, count_idx, values, hits, bt.int_values)
        numeric = {f: decode(f, src) for f, src in out_key_fields
                   if bt.decoder_of(fields_of.get(f), src) is None or bt.decoder_of(fields_of.get(f), src).dtype.kind != "U"}
        distinct = any(_all_distinct(a) for a in numeric.values())
        if not distinct:                                     # different rows may still decode to equal fields
            d = _merge_equal_keys(DictResult([(f, numeric[f] if f in numeric else decode(f, src)) for f, src in out_key_fields],
                                             vf, key_is_record, val_is_record))
            d.ordered = False
            return d
        sel = None
        by_name = dict(numeric); by_name.update(dict(vf))
        if top is not None and all(nm in by_name for nm, _ in top[1]):
            sel = ResultSet([nm for nm, _ in top[1]], [by_name[nm] for nm, _ in top[1]]).top_index(top[0], top[1])
            vf = [(nm, a[sel]) for nm, a in vf]
        d = DictResult([(f, (numeric[f] if sel is None else numeric[f][sel]) if f in numeric else decode(f, src, sel)) for f, src in out_key_fields],
                       vf, key_is_record, val_is_record)
        d.ordered = sel is not None
        return d
    if isinstance(value, BuiltTable):
        spec = _device_sort_spec(value, [(value.key_name, "key")] + [(f, src) for f, src in value.val_fields if src != "key"], [], None, top[1]) \
            if top is not None and value.key_parts is None else None
        if top is not None and spec is not None and 1 <= top[0] <= abi.MAX_TOPK and len(spec) <= abi.MAX_SORT_KEYS:
            keys, payload, _, _ = eng.ctx.table_topk(value.table, 0, top[0], spec, want_hits=False)
            ordered = True
        else:
            keys, payload, _, _ = _compact(eng, value.table, 0, hint_key, want_values=False, want_hits=False)
            ordered = False
        vf = [(fname, keys if src == "key" else _decode_column(payload[src], value.decoder_of(fname, src), value.payload_dtypes[src]))
              for fname, src in value.val_fields]
        if value.key_parts is not None:
            decs = getattr(value, "key_part_decoders", None) or [None, None]
            kf = [(value.key_parts[0], _decode_column(keys >> 32, decs[0], np.int64)), (value.key_parts[1], _decode_column(keys & 0xFFFFFFFF, decs[1], np.int64))]
        else:
            kf = [(value.key_name, _decode_column(keys, value.key_decoder, np.int64))]
        d = DictResult(kf, vf, value.key_is_record, value.val_is_record)
        d.ordered = ordered
        return d
    raise UnsupportedQuery("cannot materialise %r" % (value,))


def _finalize(eng, op, env, top=None):
    """K-F.  `top` = (k, [(result column, "asc" | "desc")]) adds ORDER BY ... LIMIT k: on the device
    (sdqh_table_topk) when the source is a device table and the columns are numeric, else on the host."""
    src_val = env[op.source]
    inner_top = top
    if top is not None and op.fields is not None:           # result columns are aliases of single-field sides
        names = {}
        probe = src_val
        if isinstance(probe, tuple) and probe and probe[0] == "aggregated":
            kf_names, vf_names = [f for f, _ in env[probe[1]].agg[0]], list(env[probe[1]].agg[1])
        elif isinstance(probe, BuiltTable):
            kf_names, vf_names = [probe.key_name], [f for f, _ in probe.val_fields]
        else:
            kf_names, vf_names = [n for n, _ in probe.key_fields], [n for n, _ in probe.val_fields]
        for spec in op.fields:
            name, which = spec[0], spec[1]
            side = kf_names if which == 0 else vf_names
            if len(spec) == 3:
                names[name] = spec[2]
            elif len(side) == 1:
                names[name] = side[0]
        inner_top = (top[0], [(names.get(n, n), d) for n, d in top[1]])
    d = _materialize(eng, src_val, env, hint_key=id(op), top=inner_top, lazy_ok=True, defer=op.out in env.get("__defer__", ()))     # (only a ResultSet is made of it: that waits for the rows itself)
    if isinstance(d, Pending):                              # launched, not waited for: the shaping below runs when it is collected
        return Pending(lambda: _shape_result(op, d.resolve(), top))
    return _shape_result(op, d, top)


def _shape_result(op, d, top):
    """K-F's last mile: the (key, value) sides of a DictResult as the record set the plan's reshaping sum names."""
    if op.fields is None:                                   # p[0].concat(p[1])
        fields = d.key_fields + d.val_fields
    else:
        fields = []
        for spec in op.fields:
            name, which = spec[0], spec[1]
            src = d.key_fields if which == 0 else d.val_fields
            if len(spec) == 3:                                  # p[which].field
                hit = [a for n, a in src if n == spec[2]]
                if not hit:
                    raise UnsupportedQuery("line %d: p[%d] has no field '%s'" % (op.lineno, which, spec[2]))
                fields.append((name, hit[0]))
                continue
            if len(src) != 1:
                raise UnsupportedQuery("line %d: p[%d] is a record; name a field or use concat" % (op.lineno, which))
            fields.append((name, src[0][1]))
    rs = ResultSet([n for n, _ in fields], [a for _, a in fields], ready=getattr(d, "ready", None))
    if top is not None and not getattr(d, "ordered", False):
        rs = rs.top(top[0], top[1])
    return rs


def _looked_up(plan):
    """Names of the results that a table loop looks up (`d[key]`, joinProbe index)."""
    found = []
    for op in plan.ops:
        if isinstance(op, ScanOp):
            if op.probe is not None:
                _walk_lookups(op.probe, found)
            for e in list(op.conds) + [op.key, op.val] + [x for _, x, _ in (op.fields or [])] + [c for _, _, fc in (op.fields or []) for c in fc]:
                if e is not None:
                    _walk_lookups(e, found)
    return {lk.dict_name for lk in found}


def _compared_lookups(plan):
    """Names of the results whose looked-up VALUES a table loop compares or tests (`d[k] == "x"`, `d[k].f > 3`,
    startsWith(d[k].f, ...)): their text payloads are worth dictionary codes (a test on a code set), which the
    row-program builds produce; the fixed-shape builds carry text as row references."""
    found = set()

    def mark(e):
        if isinstance(e, PayloadField):
            found.add(e.lookup.dict_name)
        elif isinstance(e, Lookup):
            found.add(e.dict_name)

    def walk(e):
        if isinstance(e, Cmp):
            mark(e.left); mark(e.right); walk(e.left); walk(e.right)
        elif isinstance(e, Call):
            for a in e.args:
                mark(a); walk(a)
        elif isinstance(e, Bin):
            walk(e.left); walk(e.right)
        elif isinstance(e, (And, Or)):
            for t in e.terms:
                walk(t)
        elif isinstance(e, Not):
            walk(e.term)
        elif isinstance(e, IfElse):
            walk(e.cond); walk(e.then); walk(e.other)
        elif isinstance(e, RecordCons):
            for _, x in e.fields:
                walk(x)
        elif isinstance(e, (Lookup, PayloadField)):
            walk(e.key if isinstance(e, Lookup) else e.lookup.key)

    for op in plan.ops:
        if isinstance(op, (ScanOp, HostDictOp)):                  # (a sum over a result dictionary compares looked-up text too: the reference's own q12)
            for e in list(op.conds) + [op.key, op.val]:
                if e is not None:
                    walk(e)
    return found


def _host_dict(eng, op, env, is_result):
    """A sum over a result dictionary with conditions / lookups / arithmetic (frontend.HostDictOp)."""
    from . import xplan
    out = xplan.run_host_dict(eng, op, env, lambda v: _materialize(eng, v, env, hint_key=(id(op), id(v) if isinstance(v, BuiltTable) else 0)))
    top = env.get("__top__") if is_result else None
    if top is not None and isinstance(out, ResultSet):
        out = out.top(top[0], top[1])
    return out


def _prepare_dict_loop(eng, op, is_result, as_table):
    """closure(env) for a sum over a result dictionary: a device loop over the dictionary's entries where the loop has a
    shape for it (xplan.prepare_dict_scan), the host evaluation over the materialised dictionaries otherwise."""
    from . import xplan
    state = {"dev": None}
    if getattr(eng, "dict_programs", True):
        try:
            state["dev"] = xplan.prepare_dict_scan(eng, op, as_table, is_result)
        except UnsupportedQuery:
            pass

    def run(env):
        why = "no device loop for this sum over a result dictionary"
        made = None
        if state["dev"] is not None and isinstance(env.get(op.source), DictResult):
            # the source is a handful of groups the small group-by kernels delivered to the host (Q8's two volume groups): made resident
            # again — a table keyed by the group key with the sums as its accumulators, what a large group-by would have left — so that
            # the walk is the same device loop over a table's K-F columns as for every other source (round-5 review: the last host loop)
            made = _resident_groups(eng, env[op.source], op)
            if made is not None:
                env = dict(env)
                env[op.source + "$resident"] = made
                env[op.source] = ("aggregated", op.source + "$resident")
        try:
            return run_on(env, why)
        finally:
            if made is not None:
                made.table.free()

    def run_on(env, why):
        if state["dev"] is not None:
            try:
                out = state["dev"](env)
                if isinstance(out, BuiltTable) and getattr(out, "record_order", None) is not None:
                    return _record_set(eng, op, out, env)
                if out is not NotImplemented:
                    return out
                why = "the device loop declined this run's source (a handful of host groups)"
            except UnsupportedQuery as exc:                         # known once the source's layout is: the host path from now on
                state["dev"] = None
                why = state["why"] = str(exc)
        else:
            why = state.get("why", why)
        # The loop runs on the HOST over the materialised dictionaries (O(groups), like the reference's serial K-F,
        # generator_par.py:520-568) — never silently: counted per plan step (Engine.stats()), refused under Engine.strict_device.
        root = getattr(eng, "_eng", eng)
        rec = root.host_loops.setdefault((getattr(op, "lineno", 0), op.out), {"runs": 0, "why": why})
        rec["runs"] += 1
        if getattr(root, "strict_device", False):
            raise UnsupportedQuery("line %d: the sum over the result dictionary '%s' would run on the host (%s) and the engine is strict (SDQLPY_AMD_STRICT_DEVICE)"
                                   % (getattr(op, "lineno", 0), getattr(op, "source", op.out), why))
        return _host_dict(eng, op, env, is_result)
    return run


def _resident_groups(eng, d, op):
    """A host DictResult of a few groups — one integer key field, float sums — as an aggregated device table (BuiltTable with .agg), or
    None when it has another shape (text / composite keys, counts: the host evaluation stays)."""
    if len(d.key_fields) != 1 or not d.val_fields or d.size() == 0 or d.size() > abi.MAX_LOOKUP_GROUPS:
        return None
    kname, karr = d.key_fields[0]
    karr = np.asarray(karr)
    if karr.dtype.kind != "i" or any(np.asarray(a).dtype.kind != "f" for _, a in d.val_fields) or len(d.val_fields) > abi.TUPLE_MAX_VALUES:
        return None
    ctx = eng.ctx
    keys = np.ascontiguousarray(karr, np.int64)
    n = len(keys)
    kcol = ctx.upload(keys)
    vcols = [ctx.upload(np.ascontiguousarray(a, np.float64)) for _, a in d.val_fields]
    build = abi.Program()
    build.key = build.op(abi.X_COL, abi.T_I64, col=kcol)
    try:
        table = ctx.xbuild(n, build, int(keys.min()), int(keys.max()), accumulate=True, nsums=len(vcols))
        add = abi.Program()
        k = add.op(abi.X_COL, abi.T_I64, col=kcol)
        look = add.op(abi.X_LOOKUP, abi.T_BOOL, a=k, table=table)
        add.gates = [look]
        add.vals = [add.op(abi.X_COL, abi.T_F64, col=c) for c in vcols]
        ctx.xprobe_aggregate(n, add, look, table)
    except abi.SdqhError as exc:
        if exc.code != abi.ERR_UNSUPPORTED:
            raise
        return None
    bt = BuiltTable(table, kname, d.key_is_record, [], d.val_is_record, [])
    bt.agg = ([(kname, "key")], [nm for nm, _ in d.val_fields], None, d.key_is_record, d.val_is_record, len(vcols))
    bt._keep = [kcol] + vcols
    return bt


def _record_set(eng, op, bt, env):
    """The plan's result as a set of records built on the device (xplan.prepare_dict_scan): K-F of the build — the first k rows
    only when the caller's ORDER BY / LIMIT can be applied on the device — as a ResultSet in the record's field order."""
    try:
        top = env.get("__top__")
        d = _materialize(eng, bt, env, hint_key=(id(op), 1), top=top)
        cols = dict(d.key_fields); cols.update(dict(d.val_fields))
        out = ResultSet(list(bt.record_order), [cols[nm] for nm in bt.record_order])
        if top is not None and not getattr(d, "ordered", False):
            out = out.top(top[0], top[1])
        return out
    finally:
        bt.table.free()


def _membership_only(plan):
    """Names of unique builds that are only used as `tbl[key] != None` / joinProbe index with no
    payload access, and are not the plan's result."""
    builds = {op.out for op in plan.ops if isinstance(op, ScanOp) and op.kind == "dict" and op.unique and op.probe is None
              and not (isinstance(op.val, RecordCons) and any(not (isinstance(e, Col) and isinstance(op.key, Col) and e.name == op.key.name) for _, e in op.val.fields))}
    used_for_payload = set()

    def walk(e, as_cond=False):
        if isinstance(e, PayloadField):
            if e.field is not None or not as_cond:
                used_for_payload.add(e.lookup.dict_name)
            walk(e.lookup.key)
        elif isinstance(e, Lookup):
            if not as_cond:
                used_for_payload.add(e.dict_name)
            walk(e.key)
        elif isinstance(e, Contains):
            walk(e.lookup.key)
        elif isinstance(e, (Bin, Cmp)):
            walk(e.left); walk(e.right)
        elif isinstance(e, (And, Or)):
            for t in e.terms:
                walk(t, as_cond)
        elif isinstance(e, Not):
            walk(e.term, as_cond)
        elif isinstance(e, IfElse):
            walk(e.cond, True); walk(e.then); walk(e.other)
        elif isinstance(e, Call):
            for a in e.args:
                walk(a)
        elif isinstance(e, RecordCons):
            for _, x in e.fields:
                walk(x)

    for op in plan.ops:
        if isinstance(op, ScanOp):
            for c in op.conds:
                walk(c, as_cond=True)
            if op.probe is not None:
                walk(op.probe.key)                               # the joinProbe index itself: membership
            for e in [op.key, op.val] + ([x for _, x, _ in op.fields] if op.fields else []):
                if e is not None:
                    walk(e)
            for _, _, fconds in (op.fields or []):
                for c in fconds:
                    walk(c, as_cond=True)
        elif isinstance(op, FinalizeOp):
            used_for_payload.add(op.source)
        elif isinstance(op, HostDictOp):                        # evaluated on the materialised dictionaries: all of them need entries
            used_for_payload.add(op.source)
            found = []
            for e in list(op.conds) + [op.key, op.val]:
                _walk_lookups(e, found)
            for lk in found:
                used_for_payload.add(lk.dict_name)
    return {b for b in builds if b not in used_for_payload and b != plan.result}


def _select_keys(eng, op, env):
    """HAVING over an aggregated dictionary -> membership-only BuiltTable of the passing keys."""
    src = env[op.source]
    if not (isinstance(src, tuple) and src and src[0] == "aggregated"):
        raise UnsupportedQuery("line %d: '%s' is not an aggregated dictionary" % (op.lineno, op.source))
    bt = env[src[1]]
    _, vnames, count_idx, _, val_is_record, nv = bt.agg
    if val_is_record or count_idx is not None or nv != 1:
        raise UnsupportedQuery("line %d: HAVING needs a dictionary with one summed value" % op.lineno)
    lo, hi = -np.inf, np.inf
    for c in op.conds:
        v = float(c.right.value)
        if c.op == ">": lo = max(lo, abi.gt_float(v))
        elif c.op == ">=": lo = max(lo, v)
        elif c.op == "<": hi = min(hi, abi.lt_float(v))
        elif c.op == "<=": hi = min(hi, v)
        else: lo, hi = max(lo, v), min(hi, v)
    try:
        table = eng.ctx.table_select_keys(bt.table, 1, 0, lo, hi)
    except abi.SdqhError as exc:
        if exc.code != abi.ERR_UNSUPPORTED:
            raise
        # the aggregated table's keys have no dense range (open addressing: keys far apart, or the direct layouts switched off): the
        # library's HAVING writes an exact bitmap over the key range and has none here.  K-F of the groups, the condition on the host
        # (O(groups), like the reference's serial K-F: generator_par.py:520-568), the passing keys built into a key set again — a HOST
        # loop: counted per plan step like the sums over result dictionaries, refused under Engine.strict_device.
        root = getattr(eng, "_eng", eng)
        rec = root.host_loops.setdefault((getattr(op, "lineno", 0), op.out), {"runs": 0, "why": "HAVING over a table without a dense key range: %s" % exc})
        rec["runs"] += 1
        if getattr(root, "strict_device", False):
            raise UnsupportedQuery("line %d: %s (the host-side HAVING is refused: SDQLPY_AMD_STRICT_DEVICE)" % (op.lineno, exc))
        keys, _, values, _ = _compact(eng, bt.table, 1, (id(op), "having"), want_payload=False, want_hits=False)
        v = values[0]
        passing = np.ascontiguousarray(keys[(v >= lo) & (v <= hi)], np.int64)
        col = eng.ctx.upload(passing)
        table = eng.ctx.hash_build_unique(len(passing), abi.make_filter(), [], col, [])      # (keeps `col` alive)
    return BuiltTable(table, bt.key_name, False, [], False, [])


def _in_flight(eng):
    """Results launched through this engine and not collected yet (dropped unread: no longer counted)."""
    return sum(1 for rs in list(getattr(eng, "_eng", eng)._outstanding) if "_thunk" in rs.__dict__)


class PlanGraph:
    """One recording of a prepared plan (abi.Graph) with what collecting its result needs: the step it ends at, that step's Pending
    (its resolve() reads the recording's own result block, after every launch), the host-side values of the recorded run.
    state: "free" (may be launched) / "flying" (launched, result not collected yet)."""

    def __init__(self, graph, at, pending, host_env, rows_out, epoch):
        self.graph, self.at, self.pending, self.host_env, self.rows_out, self.epoch = graph, at, pending, host_env, rows_out, epoch
        self.state = "free"
        self.result = lambda: None

    def free(self):
        if self.graph is not None:
            self.graph.free()
            self.graph = None
        self.pending = None


class RetryPlan(Exception):
    """Raised by a deferred run's precheck: what was launched has to be run again (PreparedPlan.run: on_retry)."""


class PreparedPlan:
    """A plan bound to one engine and one set of tables: every operator lowered to a closure."""

    def __init__(self, eng, plan, args, member_only=None):
        """member_only: the builds to make as key sets, when the caller knows better than the plan alone (the multi-GPU runner: a set
        other ranks' rows look up, whose keys do not suit a bitmap, is built as a table so that its entries can travel)."""
        if len(args) != len(plan.params):
            raise TypeError("%s expects %d tables, got %d" % (plan.name, len(plan.params), len(args)))
        self.eng, self.plan, self.args = eng, plan, tuple(args)       # keeps the tables (and so their ids) alive
        self.generation = eng.generation
        tables = {p: HostTable(p, a) for p, a in zip(plan.params, args)}
        # tables that a later probe-aggregate folds its group-by into must carry accumulators
        from . import xplan
        accumulate_into = {op.probe.dict_name for op in plan.ops
                           if isinstance(op, ScanOp) and op.kind == "dict" and not op.unique and op.probe is not None
                           and (_is_simple(op, tables[op.table], [c.lookup for c in op.conds if isinstance(c, Contains)]) or xplan.groups_by_entry(op))}
        # builds that only ever answer `tbl[k] != None` (never probed for a payload, never materialised):
        # membership-only tables (sdqh_build_key_set)
        # how many sums the probe-aggregate into a table will add per entry, where the plan says (a build then clears that many)
        acc_sums = {}
        for op in plan.ops:
            if isinstance(op, ScanOp) and op.kind == "dict" and not op.unique and op.probe is not None and op.probe.dict_name in accumulate_into:
                fields = [e for _, e in op.val.fields] if isinstance(op.val, RecordCons) else [op.val]
                n = sum(0 if (isinstance(e, Const) and isinstance(e.value, int) and not isinstance(e.value, bool)) else 1 for e in fields)
                acc_sums[op.probe.dict_name] = max(acc_sums.get(op.probe.dict_name, 0), n)
        # (the set of names becomes a dictionary name -> sums per entry, or None where the plan does not say: `in` works as before)
        accumulate_into = {name: (acc_sums.get(name) if 1 <= acc_sums.get(name, 0) <= abi.TUPLE_MAX_VALUES else None) for name in accumulate_into}
        member_only = _membership_only(plan) if member_only is None else set(member_only)
        looked_up = _looked_up(plan)
        compared = _compared_lookups(plan)
        self.steps = []
        self._graphs, self._graph_refused, self._deferred_runs = [], None, 0
        getattr(eng, "_eng", eng)._prepared.add(self)
        getattr(eng, "_eng", eng).prefetch_dicts(plan, tables)
        self.defer_names = self._defer_names(plan)
        for op in plan.ops:
            if isinstance(op, ScanOp):
                self.steps.append((op.out, _prepare_scan(eng, op, tables[op.table], accumulate_into, op.out in member_only, op.out in looked_up,
                                                         coded_text=op.out in compared)))
            elif isinstance(op, SelectKeysOp):
                self.steps.append((op.out, (lambda env, op=op: _select_keys(eng, op, env))))
            elif isinstance(op, ScalarExprOp):
                self.steps.append((op.out, (lambda env, op=op: _eval_scalar_expr(op.expr, env, op.lineno))))
            elif isinstance(op, FinalizeOp):
                is_result = op.out == plan.result
                self.steps.append((op.out, (lambda env, op=op, is_result=is_result: _finalize(eng, op, env, env.get("__top__") if is_result else None))))
            elif isinstance(op, HostDictOp):
                self.steps.append((op.out, _prepare_dict_loop(eng, op, op.out == plan.result, op.out in looked_up)))
            elif isinstance(op, WrapScalarOp):
                self.steps.append((op.out, (lambda env, op=op: ResultSet([n for n, _ in op.fields], [np.array([_eval_scalar_expr(e, env, op.lineno)]) for _, e in op.fields]))))

    @staticmethod
    def _defer_names(plan):
        """The steps whose device call may be launched without being waited for: the plan's LAST table loop when it is an aggregation
        that only the final reshaping sum reads, and that sum itself (its K-F) — nothing else of the plan runs after them."""
        ops = plan.ops
        if len(ops) >= 2 and isinstance(ops[-1], FinalizeOp) and ops[-1].out == plan.result and ops[-1].source == ops[-2].out \
                and isinstance(ops[-2], ScanOp) and ops[-2].kind == "dict" and not ops[-2].unique:
            return frozenset([ops[-2].out, ops[-1].out])
        return frozenset()

    def run(self, top=None, deferred=True, after=None, replace=None, keep_tables=False, on_retry=None, precheck=None, env_extra=None):
        """top = (k, [(result column, "asc" | "desc"), ...]): ORDER BY ... LIMIT k on the final K-F.
        deferred: the plan's last device call may be launched without being waited for (Engine.deferred_results): the result is
        then a DeferredResultSet that finishes the plan when it is first looked at.
        after: {step name: callable(env)} run right behind that step, before the next one (the multi-GPU runner's seams: a built
        table replaced by its replica, foreign rows aggregated into a table before it is finalised); a callable may return a
        replacement for the step's outcome.
        replace: {step name: callable(env)} run INSTEAD of that step (the multi-GPU runner: the probe loop over the rows that reached
        this rank through the exchange, not over the scanned table).
        keep_tables: a deferred run keeps its tables until its result has been collected, so that a K-F whose block turned out too
        small is repeated on them, waited for, by this rank alone — with seams, running the whole plan again would issue collectives
        the other ranks are not in.
        precheck: callable() run when the deferred result is collected, before anything is read (it may raise RetryPlan).
        on_retry: callable() -> result, what a deferred result does when the data decided against what was launched (instead of
        running the plan again here with every call waited for): the multi-GPU runner's collective re-run."""
        env = dict(env_extra) if env_extra else {}                  # (env_extra: hooks of the multi-GPU runner, "__group_fold__")
        if top is not None:
            env["__top__"] = (int(top[0]), [(str(n), str(d)) for n, d in top[1]])
            if any(d not in ("asc", "desc") for _, d in env["__top__"][1]):
                raise ValueError("top: directions are 'asc' or 'desc'")
        if deferred and top is None and self.defer_names and getattr(self.eng, "deferred_results", False) and not self.eng.ctx._profiling:
            env["__defer__"] = self.defer_names
            # A recording starts a query SOONER (one call instead of a dozen): that is worth something when the device is idle — a
            # query that is waited for before the next is launched: 0.33 -> 0.29 ms for Q3 at SF=10.  While other results are still in
            # flight the device is busy anyway, and issuing call by call staggers the queries' big streaming kernels instead of starting
            # them all at once (measured: the q1+q3+q5 step 0.59 ms call by call, 0.62 as three recordings at once —
            # profiles/r05_plan_graphs.txt): then the calls are issued.  SDQLPY_AMD_PLAN_GRAPHS_ALWAYS=1 records regardless.
            if not (after or replace) and getattr(self.eng, "plan_graphs", 0) > 0 and not self._graph_refused and not self.eng.ctx._prof_mode \
                    and (getattr(self.eng, "plan_graphs_always", False) or _in_flight(self.eng) == 0):
                rs = self._run_graph()
                if rs is not None:
                    return rs
        held = False
        try:
            for i, (out, step) in enumerate(self.steps):
                env[out] = replace[out](env) if replace and out in replace else step(env)
                if after and out in after:
                    repl = after[out](env)
                    if repl is not None:
                        env[out] = repl
                if isinstance(env[out], Pending):
                    rs = self._deferred(env, i, top, keep_tables, on_retry, precheck, bool(after or replace))
                    held = keep_tables
                    self._deferred_runs += 1
                    return rs
            res = env[self.plan.result]
            if isinstance(res, (BuiltTable, tuple)):
                res = _materialize(self.eng, res, env, hint_key=id(self.plan))
                if isinstance(res, DictResult) and not res.val_fields:      # {record: True}: a set of records, the reference's result container
                    res = ResultSet([n for n, _ in res.key_fields], [a for _, a in res.key_fields])
                    if top is not None:
                        res = res.top(env["__top__"][0], env["__top__"][1])
            if top is not None and not isinstance(res, ResultSet):
                raise UnsupportedQuery("top(k) applies to queries that end in a result set")
            return res
        finally:
            if not held:
                for v in env.values():                       # release device tables of this run
                    if isinstance(v, BuiltTable):
                        v.table.free()

    # ---- plan graphs ---------------------------------------------------------------------------------------------------------
    def _graph_epoch(self):
        root = getattr(self.eng, "_eng", self.eng).ctx
        return (getattr(root, "option_epoch", 0), self.eng.ctx.handle)

    def drop_graphs(self):
        for g in self._graphs:
            g.free()
        self._graphs = []

    def _run_graph(self):
        """Launch one of this plan's recordings (making one first when the plan has run often enough to be settled: kernels
        specialised, twins and dictionaries built, result blocks sized) and return its deferred result — or None: the caller issues
        the calls itself this time."""
        eng = self.eng
        epoch = self._graph_epoch()
        if self._graphs and self._graphs[0].epoch != epoch:
            self.drop_graphs()                                   # an option changed since they were recorded
        g = None
        for cand in self._graphs:
            if cand.state == "flying" and cand.result() is None:
                # launched, and its result object was dropped unread: nobody will collect — but its completion word may still be on
                # its way: wait for the lane, then it is free
                eng.ctx.synchronize()
                cand.state = "free"
            if cand.state == "free" and not cand.rows_out():
                g = cand
                break
        if g is None:
            if self._deferred_runs < 2 or len(self._graphs) >= eng.plan_graphs:
                return None
            g = self._record(epoch)
            if g is None:
                return None
            self._graphs.append(g)
        g.graph.launch()
        eng.graph_stats["launched"] += 1
        g.state = "flying"
        rs = self._graph_result(g)
        g.result = weakref.ref(rs)
        return rs

    def _record(self, epoch, after=None, env_extra=None, replace=None):
        """Run the plan's steps with the lane's stream in capture mode: every launch is recorded, nothing executes.  A step that has
        to wait for the device refuses (SDQH_ERR_UNSUPPORTED) and the plan is never recorded again.
        after / env_extra (round 6): the multi-GPU runner's seams and hooks (see run) — a settled distributed plan is recorded WITH its
        collectives: torch's current stream is the lane's, so RCCL's work joins the capture and is replayed with the kernels."""
        eng = self.eng
        ctx = eng.ctx
        env = dict(env_extra) if env_extra else {}
        env.update({"__defer__": self.defer_names, "__record__": True})
        at, graph = None, None
        try:
            ctx.graph_begin()
            try:
                for i, (out, step) in enumerate(self.steps):
                    env[out] = replace[out](env) if replace and out in replace else step(env)
                    if after and out in after:
                        repl = after[out](env)
                        if repl is not None:
                            env[out] = repl
                    if isinstance(env[out], Pending):
                        at = i
                        break
                if at is None:
                    raise UnsupportedQuery("the plan's last call was not deferred")
                graph = ctx.graph_end()
            except BaseException:
                ctx.graph_abort()
                raise
        except (abi.SdqhError, UnsupportedQuery) as exc:
            if isinstance(exc, abi.SdqhError) and exc.code == abi.ERR_DEVICE:
                raise                                            # (not a refusal: a launch failed, or the recording could not be taken back)
            self._graph_refused = str(exc) or "refused"
            eng.graph_stats["refused"] += 1
            return None
        finally:
            for v in env.values():                               # the handles go; the memory stays with the recording
                if isinstance(v, BuiltTable):
                    v.table.free()
        eng.graph_stats["recorded"] += 1
        host_env = {k: v for k, v in env.items() if not isinstance(v, BuiltTable)}
        return PlanGraph(graph, at, env[at_name(self, at)], host_env, env.get("__rows_out__") or (lambda: False), epoch)

    def _graph_result(self, g, precheck=None, on_retry=None):
        """precheck / on_retry: as for run (the multi-GPU runner's recordings: the exchange status is read before the rows, and what
        the data decided against is repeated collectively, not by this rank alone)."""
        pending, out, rest = g.pending, at_name(self, g.at), self.steps[g.at + 1:]
        plan = self.plan

        def thunk():
            host_env = dict(g.host_env)
            try:
                try:
                    if precheck is not None:
                        precheck()
                    host_env[out] = pending.resolve()
                finally:
                    g.state = "free"                             # collected (or failed): the block's completion word has been seen
                for name, step in rest:
                    host_env[name] = step(host_env)
                res = host_env[plan.result]
                if isinstance(res, DictResult) and not res.val_fields:
                    res = ResultSet([n for n, _ in res.key_fields], [a for _, a in res.key_fields])
                if not isinstance(res, ResultSet):
                    raise UnsupportedQuery("a deferred plan must end in a result set")
                return res
            except (abi.SdqhError, UnsupportedQuery, RetryPlan) as exc:
                if isinstance(exc, abi.SdqhError) and exc.code not in (abi.ERR_OVERFLOW, abi.ERR_UNSUPPORTED):
                    raise
                if on_retry is not None:
                    return on_retry()                            # (the runner drops its recordings itself and repeats the plan collectively)
                if isinstance(exc, RetryPlan):
                    raise
                # the data decided against what was recorded (a result that outgrew its block, more groups than the kernel's table):
                # the recordings go, the plan runs once more with every call waited for and is recorded again when it has settled
                self.eng.graph_stats["dropped"] += len(self._graphs)
                self.drop_graphs()
                self._deferred_runs = 0
                return self.run(None, deferred=False)
        rs = DeferredResultSet(thunk)
        self.eng._outstanding.add(rs)
        return rs

    def _deferred(self, env, at, top, keep_tables=False, on_retry=None, precheck=None, seams=False):
        """The steps after `at` (host-side: they only reshape) and the hand-over, as the thunk of a DeferredResultSet.  The tables of
        the run are released by the caller now — their memory is reused in stream order, behind the kernels still queued — unless
        keep_tables: then they live until the result is collected (see run)."""
        pending, out = env[at_name(self, at)], at_name(self, at)
        host_env = {k: v for k, v in env.items() if not isinstance(v, BuiltTable)}
        kept = dict(env) if keep_tables else None
        rest = self.steps[at + 1:]
        eng, plan = self.eng, self.plan

        def release():
            if kept is not None:
                for v in kept.values():
                    if isinstance(v, BuiltTable):
                        v.table.free()
                kept.clear()

        def thunk():                                                # (resolve() waits for THIS result — its completion word —, not for what was queued behind it)
            try:
                if precheck is not None:
                    precheck()
                try:
                    host_env[out] = pending.resolve()
                except abi.SdqhError as exc:
                    if exc.code != abi.ERR_OVERFLOW or not kept:
                        raise
                    # K-F's block was sized from a smaller result: K-F again on the run's own tables, waited for — local to this rank
                    again = {k: v for k, v in kept.items() if k != "__defer__"}
                    host_env[out] = self.steps[at][1](again)
                for name, step in rest:
                    host_env[name] = step(host_env)
                res = host_env[plan.result]
                if isinstance(res, DictResult) and not res.val_fields:
                    res = ResultSet([n for n, _ in res.key_fields], [a for _, a in res.key_fields])
                if not isinstance(res, ResultSet):
                    raise UnsupportedQuery("a deferred plan must end in a result set")
                return res
            except (abi.SdqhError, UnsupportedQuery, RetryPlan) as exc:
                if isinstance(exc, abi.SdqhError) and exc.code not in (abi.ERR_OVERFLOW, abi.ERR_UNSUPPORTED):
                    raise
                release()
                if on_retry is not None:
                    return on_retry()
                if seams:
                    # the run had seams (after / replace: the multi-GPU runner's exchange): running it again HERE would build from this
                    # rank's shard alone — or issue collectives the other ranks are not in.  Never silently: the runner gives on_retry.
                    raise RuntimeError("%s: a deferred run with seams cannot be repeated by one rank alone (%s)" % (plan.name, exc))
                # what only the data could decide (more groups than the kernel's table, a result that outgrew its block): the
                # plan once more, every call waited for — its own fall-backs take it from there
                return self.run(top, deferred=False)
            finally:
                release()
        rs = DeferredResultSet(thunk)
        eng._outstanding.add(rs)
        return rs


def at_name(prepared, i):
    return prepared.steps[i][0]


def prepared_plan(eng, plan, args, lane=None, member_only=None):
    """The plan bound to this engine and these tables (prepared once, reused while the engine's columns are the same), on one of the
    engine's lanes: `lane`, or the one the plan was given when it first ran here (round robin over the engine's lanes)."""
    eng = getattr(eng, "_eng", eng)
    cache = plan.__dict__.setdefault("_prepared", {})
    if lane is None:
        lanes = plan.__dict__.setdefault("_lanes", {})
        lane = lanes.get(id(eng))
        if lane is None or lane >= eng.nlanes:
            lane = lanes[id(eng)] = eng.next_lane()
    key = (id(eng), lane) + tuple(id(a) for a in args) + ((frozenset(member_only),) if member_only is not None else ())
    prepared = cache.get(key)
    if prepared is None or prepared.generation != eng.generation or any(x is not y for x, y in zip(prepared.args, args)):
        if len(cache) > 16:
            cache.clear()
        prepared = cache[key] = PreparedPlan(eng.lane(lane), plan, args, member_only=member_only)
    return prepared


def execute_plan(eng, plan, args, top=None, after=None, lane=None):
    return prepared_plan(eng, plan, args, lane=lane).run(top, after=after)
