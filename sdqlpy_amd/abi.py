"""ctypes binding of include/sdqh.h.

`Library(path)` loads one implementation of the C ABI and exposes thin, typed wrappers.  The
product (engine.py) only ever loads sdqlpy_amd/csrc/libsdqlhip.so; the test-suite loads the CPU
oracle through the same class to compare the two implementations call for call.
"""
import ctypes as C
import threading
import weakref
import math

import numpy as np

MAX_SHARE_CELLS = 1 << 28  # SDQH_MAX_SHARE_CELLS
ABI_VERSION = 7            # include/sdqh.h: SDQH_ABI_VERSION (struct layouts below must match the library's)
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_DEVICE, ERR_OVERFLOW, ERR_NOMEM = range(6)
I64, F64, STR = 0, 1, 2
TUPLE_A, TUPLE_AB, TUPLE_A_1MB, TUPLE_PRICING, TUPLE_A_1MB_M_CD, TUPLE_COUNT = 1, 2, 3, 4, 5, 6
TUPLE_MAX_VALUES = 4
SORT_KEY, SORT_PAYLOAD, SORT_VALUE, SORT_HITS = 0, 1, 2, 3
MAX_TOPK, MAX_SORT_KEYS = 128, 3
TUPLE_NVALUES = {TUPLE_A: 1, TUPLE_AB: 1, TUPLE_A_1MB: 1, TUPLE_PRICING: 4, TUPLE_A_1MB_M_CD: 1, TUPLE_COUNT: 0}
TUPLE_NOPERANDS = {TUPLE_A: 1, TUPLE_AB: 2, TUPLE_A_1MB: 2, TUPLE_PRICING: 4, TUPLE_A_1MB_M_CD: 4, TUPLE_COUNT: 0}
MAX_IPRED, MAX_FPRED, MAX_SPRED, MAX_STR_CONST = 4, 4, 1, 64
MAX_PROBE, MAX_PAYLOAD, MAX_GROUPKEYS, MAX_SMALL_GROUPS, MAX_COMPACT_COLS = 2, 4, 2, 64, 6

INT64_MIN, INT64_MAX = -(1 << 63), (1 << 63) - 1


class SdqhError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "sdqh error %d: %s" % (code, message))
        self.code = code


class _IPred(C.Structure):
    _fields_ = [("col", C.c_void_p), ("lo", C.c_int64), ("hi", C.c_int64)]


class _FPred(C.Structure):
    _fields_ = [("col", C.c_void_p), ("lo", C.c_double), ("hi", C.c_double)]


class _SPred(C.Structure):
    _fields_ = [("col", C.c_void_p), ("len", C.c_int32), ("negate", C.c_int32), ("value", C.c_uint32 * MAX_STR_CONST)]


MAX_CPRED = 2
CMP_LT, CMP_LE, CMP_EQ, CMP_NE = 0, 1, 2, 3
STR_EQ, STR_NE, STR_CONTAINS, STR_PREFIX, STR_SUFFIX = 0, 1, 2, 3, 4


class _CPred(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("op", C.c_int32), ("_pad", C.c_int32)]


class Filter(C.Structure):
    _fields_ = [("n_ipred", C.c_int32), ("n_fpred", C.c_int32), ("n_spred", C.c_int32), ("n_cpred", C.c_int32),
                ("ipred", _IPred * MAX_IPRED), ("fpred", _FPred * MAX_FPRED), ("spred", _SPred * MAX_SPRED),
                ("cpred", _CPred * MAX_CPRED)]


class Tuple(C.Structure):
    _fields_ = [("shape", C.c_int32), ("_pad", C.c_int32),
                ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("d", C.c_void_p)]


class Probe(C.Structure):
    _fields_ = [("table", C.c_void_p), ("key", C.c_void_p)]


class SortKey(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32), ("descending", C.c_int32), ("is_f64", C.c_int32)]


SRC_COLUMN, SRC_LOOKUP, SRC_LOOKUP_YEAR = 0, 1, 2
MAX_LOOKUP, MAX_LOOKUP_GROUPS = 3, 256


class Source(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lookup", C.c_int32), ("field", C.c_int32), ("_pad", C.c_int32), ("col", C.c_void_p)]


class Lookup(C.Structure):
    _fields_ = [("table", C.c_void_p), ("nkey", C.c_int32), ("_pad", C.c_int32), ("key", Source * 2)]


def src_col(col):
    """Source: a column of the scanned row."""
    return ("col", col)


def src_lookup(lookup, field, year=False):
    """Source: payload field `field` of the entry matched by lookup step `lookup` (optionally // 10000)."""
    return ("year" if year else "lookup", lookup, field)


def _fill_source(dst, spec):
    if spec[0] == "col":
        dst.kind, dst.col = SRC_COLUMN, spec[1].handle
    else:
        dst.kind, dst.lookup, dst.field = (SRC_LOOKUP_YEAR if spec[0] == "year" else SRC_LOOKUP), spec[1], spec[2]


def _sources(specs):
    arr = (Source * max(1, len(specs)))()
    for i, sp in enumerate(specs):
        _fill_source(arr[i], sp)
    return arr


def _lookups(lookups):
    """lookups: [(Table, [key source specs (1 or 2)])]."""
    if len(lookups) > MAX_LOOKUP:
        raise SdqhError(ERR_UNSUPPORTED, "more than %d lookups in one loop" % MAX_LOOKUP)
    arr = (Lookup * max(1, len(lookups)))()
    for i, (table, keys) in enumerate(lookups):
        arr[i].table, arr[i].nkey = table.handle, len(keys)
        for k, sp in enumerate(keys):
            _fill_source(arr[i].key[k], sp)
    return arr


# ---- row programs (ABI 4) ---------------------------------------------------------------------------
T_I64, T_F64, T_BOOL = 0, 1, 2
X_COL, X_ROWID, X_CONST, X_LOOKUP, X_FIELD, X_ACC = 1, 2, 3, 4, 5, 6
X_ADD, X_SUB, X_MUL, X_DIV, X_NEG, X_I2F, X_YEAR, X_PACK2, X_DIVI, X_MODI = 10, 11, 12, 13, 14, 15, 16, 17, 18, 19
X_LT, X_LE, X_GT, X_GE, X_EQ, X_NE = 20, 21, 22, 23, 24, 25
X_AND, X_OR, X_NOT, X_SELECT = 30, 31, 32, 33
X_STR, X_STRIDX, X_CHAR = 40, 41, 42
MAX_XOPS, MAX_XCOLS, MAX_XTABLES, MAX_XGATES, MAX_XSTR = 96, 16, 6, 16, 256


class XOp(C.Structure):
    _fields_ = [("code", C.c_int32), ("type", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("aux", C.c_int32),
                ("imm_i", C.c_int64), ("imm_f", C.c_double), ("col", C.c_void_p), ("table", C.c_void_p), ("str", C.c_void_p),
                ("slen", C.c_int32), ("_pad", C.c_int32)]


class XProgram(C.Structure):
    _fields_ = [("nops", C.c_int32), ("ngates", C.c_int32), ("ops", C.c_void_p), ("gates", C.c_void_p),
                ("key", C.c_int32), ("nvals", C.c_int32), ("vals", C.c_void_p)]


class Program:
    """A row program under construction: `op(...)` appends an operation and returns its index.
    Columns and tables are given as Column / Table objects (tables may be bound late with `bind_table`:
    a program is built once per plan, its tables are rebuilt every run)."""

    def __init__(self):
        self.ops = []            # dicts
        self.gates, self.vals, self.key = [], [], -1
        self._struct = None

    def op(self, code, typ, a=-1, b=-1, c=-1, aux=0, imm_i=0, imm_f=0.0, col=None, table=None, text=None):
        if len(self.ops) >= MAX_XOPS:
            raise SdqhError(ERR_UNSUPPORTED, "expression program longer than %d operations" % MAX_XOPS)
        self.ops.append(dict(code=code, type=typ, a=a, b=b, c=c, aux=aux, imm_i=int(imm_i), imm_f=float(imm_f), col=col, table=table, text=text))
        self._struct = None
        return len(self.ops) - 1

    def type_of(self, i):
        return self.ops[i]["type"]

    def bind_table(self, i, table):
        self.ops[i]["table"] = table
        if self._struct is not None:
            self._struct[1][i].table = table.handle

    def bind_col(self, i, col):
        """Rebind the column of COL operation i (the entries of a dictionary are fresh columns on every run: sdqh_table_columns)."""
        self.ops[i]["col"] = col
        if self._struct is not None:
            self._struct[1][i].col = col.handle

    def struct(self):
        """The ctypes sdqh_program.  The operation array is cached (table handles are refreshed by
        bind_table, constants by set_const); gates / key / values are re-read on every call."""
        if self._struct is None:
            n = len(self.ops)
            arr = (XOp * max(1, n))()
            keep = []
            for i, o in enumerate(self.ops):
                x = arr[i]
                x.code, x.type, x.a, x.b, x.c, x.aux, x.imm_i, x.imm_f = o["code"], o["type"], o["a"], o["b"], o["c"], o["aux"], o["imm_i"], o["imm_f"]
                x.col = o["col"].handle if o["col"] is not None else None
                x.table = o["table"].handle if o["table"] is not None else None
                if o["text"] is not None:
                    units = (C.c_uint32 * max(1, len(o["text"])))(*[ord(ch) for ch in o["text"]])
                    keep.append(units)
                    x.str, x.slen = C.cast(units, C.c_void_p), len(o["text"])
            self._struct = [XProgram(), arr, None, None, keep]
        p, arr = self._struct[0], self._struct[1]
        gates = (C.c_int32 * max(1, len(self.gates)))(*self.gates)
        vals = (C.c_int32 * max(1, len(self.vals)))(*self.vals)
        self._struct[2], self._struct[3] = gates, vals
        p.nops, p.ngates, p.ops, p.gates = len(self.ops), len(self.gates), C.cast(arr, C.c_void_p), C.cast(gates, C.c_void_p)
        p.key, p.nvals, p.vals = self.key, len(self.vals), C.cast(vals, C.c_void_p)
        return p

    def set_const(self, i, value):
        """Rebind the value of CONST operation i (a scalar computed by an earlier loop of the same run)."""
        o = self.ops[i]
        if o["type"] == T_F64:
            o["imm_f"] = float(value)
        else:
            o["imm_i"] = int(value)
        if self._struct is not None:
            self._struct[1][i].imm_f, self._struct[1][i].imm_i = o["imm_f"], o["imm_i"]


EXPORTS = [
    "sdqh_abi_version", "sdqh_backend_name", "sdqh_create", "sdqh_fork", "sdqh_destroy", "sdqh_last_error", "sdqh_set_threads",
    "sdqh_synchronize", "sdqh_last_device_ms", "sdqh_set_profiling", "sdqh_set_profile_filter", "sdqh_profile_count", "sdqh_profile_entry", "sdqh_profile_entry_bytes",
    "sdqh_stream", "sdqh_set_option", "sdqh_memory_stats",
    "sdqh_column_upload", "sdqh_column_wrap", "sdqh_column_alloc", "sdqh_column_download", "sdqh_column_data",
    "sdqh_column_rows", "sdqh_column_dtype", "sdqh_column_width", "sdqh_column_minmax", "sdqh_column_free",
    "sdqh_scan_filter_sum", "sdqh_scan_probe_sum", "sdqh_groupby_small", "sdqh_hash_build_unique", "sdqh_build_key_set", "sdqh_groupby_key", "sdqh_table_select_keys", "sdqh_table_share_groups", "sdqh_table_size", "sdqh_table_free",
    "sdqh_hash_probe_aggregate", "sdqh_table_compact", "sdqh_table_compact_async", "sdqh_table_compact_deferred", "sdqh_host_wait_word", "sdqh_result_wait", "sdqh_scan_compact", "sdqh_partition_by_key",
    "sdqh_table_export_bitmap", "sdqh_table_from_bitmap", "sdqh_column_copy_out", "sdqh_column_copy_in", "sdqh_column_unpack2", "sdqh_partition_pack", "sdqh_unpack_parts", "sdqh_column_mark_transient", "sdqh_column_set_bounds",
    "sdqh_build", "sdqh_lookup_aggregate", "sdqh_lookup_aggregate_block", "sdqh_table_entries", "sdqh_host_alloc", "sdqh_host_free", "sdqh_table_topk",
    "sdqh_xscan_sum", "sdqh_xgroupby", "sdqh_xgroupby_block_bytes", "sdqh_xgroupby_async", "sdqh_xgroupby_collect", "sdqh_xgroupby_partial", "sdqh_xgroupby_fold", "sdqh_xbuild", "sdqh_xkey_set", "sdqh_xcompact", "sdqh_xprobe_aggregate", "sdqh_table_columns", "sdqh_jit_stats", "sdqh_jit_compile",
    "sdqh_xstage", "sdqh_chunk_words", "sdqh_table_partition_pack", "sdqh_unpack_chunks",
    "sdqh_graph_begin", "sdqh_graph_end", "sdqh_graph_abort", "sdqh_graph_launch", "sdqh_graph_nodes", "sdqh_graph_free",
]
EXCHANGE_STAT_WORDS, STAT_MAX_COUNT, STAT_DETAIL = 32, 0, 8      # include/sdqh.h: SDQH_EXCHANGE_STAT_WORDS, SDQH_STAT_MAX_COUNT, SDQH_STAT_DETAIL


def _np_ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(None)


class Column:
    """Device (or, for the oracle, host) resident column handle."""

    def __init__(self, ctx, handle, nrows, dtype, width, keepalive=None):
        self.ctx, self.handle, self.nrows, self.dtype, self.width = ctx, handle, int(nrows), dtype, width
        self._keepalive = keepalive

    def free(self):
        if self.handle is not None:
            if self.ctx.handle is not None:                      # (a context that was closed has released everything it owned)
                self.ctx.lib.sdqh_column_free(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def download(self, row0=0, nrows=None):
        n = self.nrows - row0 if nrows is None else nrows
        if self.dtype == I64:
            out = np.empty(n, np.int64)
        elif self.dtype == F64:
            out = np.empty(n, np.float64)
        else:
            out = np.empty(n, "<U%d" % self.width)
        self.ctx._check(self.ctx.lib.sdqh_column_download(self.ctx.handle, self.handle, C.c_int64(row0), C.c_int64(n), _np_ptr(out)))
        return out

    def minmax(self):
        lo, hi = C.c_int64(), C.c_int64()
        self.ctx._check(self.ctx.lib.sdqh_column_minmax(self.ctx.handle, self.handle, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def data_ptr(self):
        return self.ctx.lib.sdqh_column_data(self.handle)

    def mark_transient(self):
        """This column lives for one run (rows that arrived through a collective): nothing is derived from it (sdqh_column_mark_transient)."""
        self.ctx._check(self.ctx.lib.sdqh_column_mark_transient(self.ctx.handle, self.handle))
        return self

    def set_bounds(self, lo, hi):
        """Bounds the caller knows (a superset of the values is fine): no minimum / maximum pass will be made (sdqh_column_set_bounds)."""
        self.ctx._check(self.ctx.lib.sdqh_column_set_bounds(self.ctx.handle, self.handle, C.c_int64(int(lo)), C.c_int64(int(hi))))
        return self


class Table:
    def __init__(self, ctx, handle, npayload=0, accumulate=False):
        self.ctx, self.handle, self.npayload, self.accumulate = ctx, handle, npayload, accumulate

    def free(self):
        if self.handle is not None:
            if self.ctx.handle is not None:
                self.ctx.lib.sdqh_table_free(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def size(self):
        n = C.c_int64()
        self.ctx._check(self.ctx.lib.sdqh_table_size(self.ctx.handle, self.handle, C.byref(n)))
        return n.value


def make_filter(ipreds=(), fpreds=(), spreds=(), cpreds=()):
    """ipreds: [(Column, lo, hi)], fpreds: [(Column, lo, hi)], spreds: [(Column, 'text', mode)] with mode
    STR_EQ / STR_NE / STR_CONTAINS / STR_PREFIX / STR_SUFFIX, cpreds: [(Column a, Column b, CMP_*)] meaning a op b."""
    f = Filter()
    if len(ipreds) > MAX_IPRED or len(fpreds) > MAX_FPRED or len(spreds) > MAX_SPRED or len(cpreds) > MAX_CPRED:
        raise SdqhError(ERR_UNSUPPORTED, "too many predicates for one filter")
    f.n_ipred, f.n_fpred, f.n_spred, f.n_cpred = len(ipreds), len(fpreds), len(spreds), len(cpreds)
    for i, (a, b, op) in enumerate(cpreds):
        f.cpred[i].a, f.cpred[i].b, f.cpred[i].op = a.handle, b.handle, int(op)
    for i, (col, lo, hi) in enumerate(ipreds):
        f.ipred[i].col, f.ipred[i].lo, f.ipred[i].hi = col.handle, max(INT64_MIN, lo), min(INT64_MAX, hi)
    for i, (col, lo, hi) in enumerate(fpreds):
        f.fpred[i].col, f.fpred[i].lo, f.fpred[i].hi = col.handle, lo, hi
    for i, (col, text, negate) in enumerate(spreds):
        if len(text) > MAX_STR_CONST:
            raise SdqhError(ERR_UNSUPPORTED, "string constant longer than %d" % MAX_STR_CONST)
        f.spred[i].col, f.spred[i].len, f.spred[i].negate = col.handle, len(text), int(negate)      # 0 ==, 1 !=, 2 substring
        for k, ch in enumerate(text):
            f.spred[i].value[k] = ord(ch)
    f._keep = (ipreds, fpreds, spreds, cpreds)
    return f


def make_tuple(shape, operands=()):
    t = Tuple()
    t.shape = shape
    if len(operands) != TUPLE_NOPERANDS[shape]:
        raise SdqhError(ERR_INVALID, "tuple shape %d takes %d operands" % (shape, TUPLE_NOPERANDS[shape]))
    for name, col in zip("abcd", operands):
        setattr(t, name, col.handle)
    t._keep = operands
    return t


def lt_float(c):
    """hi bound equivalent to `x < c`."""
    return math.nextafter(c, -math.inf)


def gt_float(c):
    """lo bound equivalent to `x > c`."""
    return math.nextafter(c, math.inf)


class Graph:
    """A recorded plan (sdqh_graph): launch() queues every recorded kernel / copy / fill on the context's stream in one call."""

    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle
        self.nodes = int(ctx.lib.sdqh_graph_nodes(handle))

    def launch(self):
        self.ctx._check(self.ctx.lib.sdqh_graph_launch(self.ctx.handle, self.handle))

    def free(self):
        if self.handle is not None:
            if self.ctx.handle is not None:
                self.ctx.lib.sdqh_graph_free(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    def __init__(self, lib, device=0, threads=1, _parent=None):
        self.lib = lib.cdll
        self.library = lib
        h = C.c_void_p()
        if _parent is None:
            rc = self.lib.sdqh_create(C.c_int(device), C.byref(h))
            if rc != OK:
                raise SdqhError(rc, "sdqh_create(device=%d) failed" % device)
        else:
            _parent._check(self.lib.sdqh_fork(_parent.handle, C.byref(h)))
        self.handle = h
        self.device = device
        self.parent = _parent          # sdqh_fork: a context of the parent's family (own stream / pool / result blocks, shared columns)
        self.forks = []
        self._options = {}             # options set so far (a fork starts with its parent's)
        self._profiling = False
        self._prof_mode, self._prof_only = 0, None
        self._graphs = weakref.WeakSet()          # recorded plans (Graph): released before the context goes
        self.kernel_log = []       # [(kernel name, ms)] of every pattern call since the log was cleared (profiling on)
        self.device_log = []       # [(pattern call, device ms)]
        self._check(self.lib.sdqh_set_threads(self.handle, C.c_int(max(1, threads))))
        self._host_pool = {}       # block bytes -> [free block addresses]
        self._host_quarantine = [] # (address, bytes) of released blocks a queued result copy may still write (host_block)
        self._deferred_quarantine = []  # the same for blocks that kernels still queued on the stream may write (synchronize)
        self._sync_epoch = 0            # full synchronisations of the context so far
        self._pool_lock = threading.RLock()   # the pools above: blocks are released by whichever thread drops the last view of a result

    def fork(self):
        """A context of this one's family (sdqh_fork): calls on it run concurrently with this one's and may name this one's columns.
        It starts with the options set here and follows later changes; profiling switches and the call logs are the family's."""
        if self.parent is not None:
            raise SdqhError(ERR_INVALID, "fork the family's first context")
        child = Context(self.library, self.device, 1, _parent=self)
        for name, value in self._options.items():
            child.set_option(name, value)
        if self._prof_mode:
            child.set_profiling(self._prof_mode, self._prof_only)
        child.kernel_log, child.device_log = self.kernel_log, self.device_log
        self.forks.append(child)
        return child

    def close(self):
        for child in list(self.forks):
            child.close()
        self.forks = []
        if self.handle is not None:
            for g in list(self._graphs):
                g.free()
            self.lib.sdqh_synchronize(self.handle)
            for addr, size in self._host_quarantine + self._deferred_quarantine:
                self._host_pool.setdefault(size, []).append(addr)
            self._host_quarantine, self._deferred_quarantine = [], []
            for blocks in self._host_pool.values():
                for addr in blocks:
                    self.lib.sdqh_host_free(self.handle, C.c_void_p(addr))
            self._host_pool = {}
            self.lib.sdqh_destroy(self.handle)
            self.handle = None
        if self.parent is not None and self in self.parent.forks:
            self.parent.forks.remove(self)

    # -- result memory the device can write (sdqh_host_alloc) -------------------------------------
    def host_block(self, nbytes, deferred=False):
        """A ctypes byte array over a device-visible block of at least nbytes.  numpy views made
        from it keep it alive; when the last one dies the block returns to this context's pool (or
        is freed if the context is gone).  deferred: kernels queued on the stream will write the block (xgroupby_async,
        table_compact_deferred): released, it waits for the next full synchronisation of the context before it is handed out again."""
        size = 1 << 16
        while size < nbytes:
            size <<= 1
        if len(self._deferred_quarantine) > 32:                   # blocks of results nobody collected, and nobody synchronised in a while: do it here
            self.synchronize()
        if self._host_quarantine:
            # blocks whose arrays died while a result copy might still have been landing in them: usable once the copies are done
            # (they are, by the time another result is asked for; waiting when the block was RELEASED stalled every query on its own copy)
            with self._pool_lock:
                waiting, self._host_quarantine = self._host_quarantine, []
            self._check(self.lib.sdqh_result_wait(self.handle))
            with self._pool_lock:
                for a, sz in waiting:
                    self._host_pool.setdefault(sz, []).append(a)
        with self._pool_lock:
            free = self._host_pool.get(size)
            addr = free.pop() if free else None
        if addr is None:
            p = C.c_void_p()
            self._check(self.lib.sdqh_host_alloc(self.handle, C.c_size_t(size), C.byref(p)))
            addr = p.value
            C.memset(addr, 0, size)                              # once per block (a recycled block is NOT zeroed again: value slots a K-F result does not use are undefined, include/sdqh.h)
        buf = (C.c_char * size).from_address(addr)
        weakref.finalize(buf, Context._release_block, weakref.ref(self), self.lib, addr, size, self._sync_epoch if deferred else None,
                         deferred if isinstance(deferred, list) else None)
        return buf

    @staticmethod
    def _release_block(ctx_ref, lib, addr, size, launched_at=None, done=None):
        ctx = ctx_ref()
        if ctx is not None and ctx.handle is not None:
            # a result copy may still be landing in the block (host_block waits for the copies) / kernels still queued may write it:
            # not if its result was collected (done), nor if the context has been synchronised since the block was handed out; else
            # the next synchronize returns it to the pool
            with ctx._pool_lock:
                if launched_at is None:
                    ctx._host_quarantine.append((addr, size))
                elif (done is not None and done[0]) or ctx._sync_epoch > launched_at:
                    ctx._host_pool.setdefault(size, []).append(addr)
                else:
                    ctx._deferred_quarantine.append((addr, size))
        else:
            lib.sdqh_host_free(None, C.c_void_p(addr))

    def _check(self, rc):
        if rc != OK:
            raise SdqhError(rc, self.lib.sdqh_last_error(self.handle).decode())

    # -- plumbing ----------------------------------------------------------------------------
    def set_threads(self, n):
        self._check(self.lib.sdqh_set_threads(self.handle, C.c_int(n)))

    def memory_stats(self, family=True):
        """{"used", "cached_free", "graphs", "blocks"} of this context's device-memory pool (sdqh_memory_stats), its forks' added."""
        out = (C.c_int64 * 4)()
        self._check(self.lib.sdqh_memory_stats(self.handle, out, C.c_int(4)))
        tot = {"used": int(out[0]), "cached_free": int(out[1]), "graphs": int(out[2]), "blocks": int(out[3])}
        if family:
            for child in self.forks:
                for k, v in child.memory_stats(family=False).items():
                    tot[k] += v
        return tot

    def synchronize(self):
        """Everything queued on this context — and, on a family's first context, on its forks — has run."""
        for child in self.forks:
            child.synchronize()
        self._check(self.lib.sdqh_synchronize(self.handle))
        with self._pool_lock:
            self._sync_epoch += 1
            if self._deferred_quarantine:                         # nothing queued before this point can write them any more
                for addr, size in self._deferred_quarantine:
                    self._host_pool.setdefault(size, []).append(addr)
                del self._deferred_quarantine[:]

    def last_device_ms(self):
        ms = C.c_double()
        self._check(self.lib.sdqh_last_device_ms(self.handle, C.byref(ms)))
        return ms.value

    def set_profiling(self, mode, only=None):
        """0/False off; 1/True per call (kernel_log / device_log filled after every pattern call,
        synchronising); 2 record only — read everything afterwards with profile().  only = name of
        the one kernel to record (None: all)."""
        mode = int(mode)
        self._check(self.lib.sdqh_set_profile_filter(self.handle, only.encode() if only else None))
        self._check(self.lib.sdqh_set_profiling(self.handle, C.c_int(mode)))
        self._profiling = mode == 1
        self._prof_mode, self._prof_only = mode, only
        self.kernel_log, self.device_log = [], []
        for child in self.forks:                                   # the family is profiled as one: the forks write this context's logs
            child.set_profiling(mode, only)
            child.kernel_log, child.device_log = self.kernel_log, self.device_log

    def _after_call(self, name):
        """With profiling on, log the device time of the call and of each kernel it launched
        (HIP events recorded on the ctx stream around every launch).  Synchronises; off by default."""
        if self._profiling:
            self.device_log.append((name, self.last_device_ms()))
            self.kernel_log.extend(self.profile())

    def profile(self, family=True):
        """[(kernel, ms)] of every recorded launch; on a family's first context: its own, then its forks' (family=False: its own only)."""
        out = []
        for i in range(self.lib.sdqh_profile_count(self.handle)):
            name, ms = C.c_char_p(), C.c_double()
            self._check(self.lib.sdqh_profile_entry(self.handle, C.c_int(i), C.byref(name), C.byref(ms)))
            out.append((name.value.decode(), ms.value))
        if family:
            for child in self.forks:
                out.extend(child.profile())
        return out

    def profile_bytes(self):
        """[(kernel, ms, modelled HBM bytes or 0)] of every recorded launch (sdqh_profile_entry_bytes), the forks' after this context's."""
        out = []
        for i, (name, ms) in enumerate(self.profile(family=False)):
            b = C.c_int64()
            self._check(self.lib.sdqh_profile_entry_bytes(self.handle, C.c_int(i), C.byref(b)))
            out.append((name, ms, int(b.value)))
        for child in self.forks:
            out.extend(child.profile_bytes())
        return out

    def stream(self):
        return self.lib.sdqh_stream(self.handle)

    def set_option(self, name, value):
        self._check(self.lib.sdqh_set_option(self.handle, name.encode(), C.c_int64(int(value))))
        if self._options.get(name) != int(value) and name != "async_copies":
            self.option_epoch = getattr(self, "option_epoch", 0) + 1      # recorded plans (engine.PlanGraph) were made under the old setting
        self._options[name] = int(value)
        for child in self.forks:
            child.set_option(name, value)

    # -- columns -----------------------------------------------------------------------------
    @staticmethod
    def _classify(arr):
        if arr.dtype == np.int64:
            return I64, 0
        if arr.dtype == np.float64:
            return F64, 0
        if arr.dtype.kind == "U":
            return STR, arr.dtype.itemsize // 4
        raise SdqhError(ERR_INVALID, "unsupported column dtype %s (int64 / float64 / '<U n' only)" % arr.dtype)

    def upload(self, arr):
        """numpy column -> resident Column.  Validates what the reference silently assumes
        (reference sdql_compiler.py:644-668 does no dtype / contiguity checks)."""
        if not isinstance(arr, np.ndarray) or arr.ndim != 1:
            raise SdqhError(ERR_INVALID, "a column must be a 1-d numpy array")
        if not arr.flags.c_contiguous:
            raise SdqhError(ERR_INVALID, "a column must be C-contiguous")
        if arr.dtype.kind == "U" and arr.dtype.byteorder == ">":
            raise SdqhError(ERR_INVALID, "big-endian unicode columns are not supported")
        dtype, width = self._classify(arr)
        h = C.c_void_p()
        self._check(self.lib.sdqh_column_upload(self.handle, _np_ptr(arr), C.c_int64(len(arr)), C.c_int(dtype), C.c_int(width), C.byref(h)))
        return Column(self, h, len(arr), dtype, width)

    def wrap(self, device_ptr, nrows, dtype, width=0, keepalive=None):
        h = C.c_void_p()
        self._check(self.lib.sdqh_column_wrap(self.handle, C.c_void_p(device_ptr), C.c_int64(nrows), C.c_int(dtype), C.c_int(width), C.byref(h)))
        return Column(self, h, nrows, dtype, width, keepalive)

    def alloc(self, nrows, dtype, width=0):
        h = C.c_void_p()
        self._check(self.lib.sdqh_column_alloc(self.handle, C.c_int64(nrows), C.c_int(dtype), C.c_int(width), C.byref(h)))
        return Column(self, h, nrows, dtype, width)

    # -- pattern calls -----------------------------------------------------------------------
    def scan_filter_sum(self, nrows, flt, tup):
        vals = (C.c_double * TUPLE_MAX_VALUES)()
        cnt = C.c_int64()
        self._check(self.lib.sdqh_scan_filter_sum(self.handle, C.c_int64(nrows), C.byref(flt), C.byref(tup), vals, C.byref(cnt)))
        self._after_call("scan_filter_sum")
        return list(vals)[: TUPLE_NVALUES[tup.shape]], cnt.value

    def scan_probe_sum(self, nrows, flt, probes, tup):
        """scan_filter_sum over the rows that also pass every (table, key column) semi-join probe."""
        parr = (Probe * max(1, len(probes)))()
        for i, (tbl, kcol) in enumerate(probes):
            parr[i].table, parr[i].key = tbl.handle, kcol.handle
        vals = (C.c_double * TUPLE_MAX_VALUES)()
        cnt = C.c_int64()
        self._check(self.lib.sdqh_scan_probe_sum(self.handle, C.c_int64(nrows), C.byref(flt), C.c_int(len(probes)), parr, C.byref(tup), vals, C.byref(cnt)))
        self._after_call("scan_probe_sum")
        return list(vals)[: TUPLE_NVALUES[tup.shape]], cnt.value

    def groupby_small(self, nrows, flt, keys, tup, max_groups=MAX_SMALL_GROUPS):
        nk = len(keys)
        karr = (C.c_void_p * nk)(*[k.handle for k in keys])
        out_keys = np.zeros((max_groups, nk), np.int64)
        out_vals = np.zeros((max_groups, TUPLE_MAX_VALUES), np.float64)
        out_cnt = np.zeros(max_groups, np.int64)
        ng = C.c_int32()
        self._check(self.lib.sdqh_groupby_small(self.handle, C.c_int64(nrows), C.byref(flt), C.c_int(nk), karr, C.byref(tup),
                                                C.c_int(max_groups), _np_ptr(out_keys), _np_ptr(out_vals), _np_ptr(out_cnt), C.byref(ng)))
        self._after_call("groupby_small")
        n = ng.value
        return out_keys[:n], out_vals[:n, : TUPLE_NVALUES[tup.shape]], out_cnt[:n]

    def hash_build_unique(self, nrows, flt, probes, key, payload=(), accumulate=False):
        parr = (Probe * max(1, len(probes)))()
        for i, (tbl, kcol) in enumerate(probes):
            parr[i].table, parr[i].key = tbl.handle, kcol.handle
        pl = (C.c_void_p * max(1, len(payload)))(*[p.handle for p in payload])
        h = C.c_void_p()
        self._check(self.lib.sdqh_hash_build_unique(self.handle, C.c_int64(nrows), C.byref(flt), C.c_int(len(probes)), parr, key.handle,
                                                    C.c_int(len(payload)), pl, C.c_int(1 if accumulate else 0), C.byref(h)))
        self._after_call("hash_build_unique")
        t = Table(self, h, len(payload), accumulate)
        t._keep = (probes, key, payload)
        return t

    def build_key_set(self, nrows, flt, probes, key):
        """Membership-only build (exact bitmap of the surviving keys); SdqhError(ERR_UNSUPPORTED)
        when the key range does not suit a bitmap."""
        parr = (Probe * max(1, len(probes)))()
        for i, (tbl, kcol) in enumerate(probes):
            parr[i].table, parr[i].key = tbl.handle, kcol.handle
        h = C.c_void_p()
        self._check(self.lib.sdqh_build_key_set(self.handle, C.c_int64(nrows), C.byref(flt), C.c_int(len(probes)), parr, key.handle, C.byref(h)))
        self._after_call("build_key_set")
        t = Table(self, h, 0, False)
        t._keep = (probes, key)
        return t

    def groupby_key(self, nrows, flt, key, tup):
        """Aggregating dictionary keyed by an int column of any cardinality -> accumulating Table."""
        h = C.c_void_p()
        self._check(self.lib.sdqh_groupby_key(self.handle, C.c_int64(nrows), C.byref(flt), key.handle, C.byref(tup), C.byref(h)))
        self._after_call("groupby_key")
        t = Table(self, h, 0, True)
        t._keep = (key, flt, tup)
        return t

    def table_select_keys(self, table, min_hits, value_index, lo, hi):
        """HAVING lo <= accumulator <= hi -> membership-only Table of the keys."""
        h = C.c_void_p()
        self._check(self.lib.sdqh_table_select_keys(self.handle, table.handle, C.c_int64(min_hits), C.c_int(value_index), C.c_double(lo), C.c_double(hi), C.byref(h)))
        self._after_call("table_select_keys")
        return Table(self, h, 0, False)

    def table_share_groups(self, table, fields, lo, span):
        """Entries with equal payload `fields` (values in [lo, lo + span) each) share one accumulator from now on."""
        n = len(fields)
        self._check(self.lib.sdqh_table_share_groups(self.handle, table.handle, C.c_int(n), (C.c_int32 * max(n, 1))(*fields),
                                                     (C.c_int64 * max(n, 1))(*lo), (C.c_int64 * max(n, 1))(*span)))
        self._after_call("table_share_groups")

    def build(self, nrows, flt, lookups, key, payload=(), accumulate=False):
        """Generalised unique build: lookups [(Table, [key sources])], key [1-2 sources], payload [sources]."""
        larr, karr, parr = _lookups(lookups), _sources(key), _sources(payload)
        h = C.c_void_p()
        self._check(self.lib.sdqh_build(self.handle, C.c_int64(nrows), C.byref(flt), C.c_int(len(lookups)), larr, C.c_int(len(key)), karr,
                                        C.c_int(len(payload)), parr, C.c_int(1 if accumulate else 0), C.byref(h)))
        self._after_call("build")
        t = Table(self, h, len(payload), accumulate)
        t._keep = (lookups, key, payload)
        return t

    def build_marshalled(self, nrows, flt, larr, nlookups, karr, nkey, parr, npayload, accumulate, keep):
        """sdqh_build with arrays marshalled once by the caller (engine closures cache them per layout of the looked-up tables)."""
        h = C.c_void_p()
        self._check(self.lib.sdqh_build(self.handle, C.c_int64(nrows), C.byref(flt), C.c_int(nlookups), larr, C.c_int(nkey), karr,
                                        C.c_int(npayload), parr, C.c_int(1 if accumulate else 0), C.byref(h)))
        self._after_call("build")
        t = Table(self, h, npayload, accumulate)
        t._keep = keep
        return t

    def lookup_aggregate(self, nrows, flt, lookups, keys, shape, operands, max_groups=MAX_LOOKUP_GROUPS):
        return self.lookup_aggregate_marshalled(nrows, flt, _lookups(lookups), len(lookups), _sources(keys), len(keys), shape, _sources(operands), max_groups)

    def lookup_aggregate_marshalled(self, nrows, flt, larr, nlookups, karr, nk, shape, oarr, max_groups=MAX_LOOKUP_GROUPS):
        out_keys = np.zeros((max_groups, nk), np.int64)
        out_vals = np.zeros((max_groups, TUPLE_MAX_VALUES), np.float64)
        out_cnt = np.zeros(max_groups, np.int64)
        ng = C.c_int32()
        self._check(self.lib.sdqh_lookup_aggregate(self.handle, C.c_int64(nrows), C.byref(flt), C.c_int(nlookups), larr, C.c_int(nk), karr,
                                                   C.c_int(shape), oarr, C.c_int(max_groups), _np_ptr(out_keys), _np_ptr(out_vals), _np_ptr(out_cnt), C.byref(ng)))
        self._after_call("lookup_aggregate")
        n = ng.value
        return out_keys[:n], out_vals[:n, : TUPLE_NVALUES[shape]], out_cnt[:n]

    def _block_collect(self, buf, done, nk, nvals, max_groups):
        """collect() of a lookup-aggregate block: (keys [n, nk], values, counts) as lookup_aggregate_marshalled (the block's packed
        key taken apart: part k in bits [32k, 32k + 32))."""
        def collect():
            out_keys = np.zeros(max_groups, np.int64)
            out_vals = np.zeros((max_groups, TUPLE_MAX_VALUES), np.float64)
            out_cnt = np.zeros(max_groups, np.int64)
            ng = C.c_int32()
            rc = self.lib.sdqh_xgroupby_collect(self.handle, C.addressof(buf), C.c_int(nvals), C.c_int(max_groups),
                                                _np_ptr(out_keys), _np_ptr(out_vals), _np_ptr(out_cnt), C.byref(ng))
            done[0] = True                                        # (collect waited for this call's kernels: nothing will write the block any more)
            self._check(rc)
            n = ng.value
            packed = out_keys[:n].view(np.uint64)
            keys = np.stack([((packed >> np.uint64(32 * k)) & np.uint64(0xFFFFFFFF)).astype(np.int64) for k in range(nk)], axis=1) if n else np.zeros((0, nk), np.int64)
            return keys, out_vals[:n, :nvals], out_cnt[:n]
        return collect

    def lookup_aggregate_async_marshalled(self, nrows, flt, larr, nlookups, karr, nk, shape, oarr, max_groups=MAX_LOOKUP_GROUPS):
        """sdqh_lookup_aggregate_block into a result block: launched, not waited for; collect() -> what lookup_aggregate_marshalled returns.
        What the data decides (too many groups, a key part out of range) is raised by collect()."""
        done = [False]
        buf = self.host_block(self.lib.sdqh_xgroupby_block_bytes(), deferred=done)
        self._check(self.lib.sdqh_lookup_aggregate_block(self.handle, C.c_int64(nrows), C.byref(flt), C.c_int(nlookups), larr, C.c_int(nk), karr,
                                                         C.c_int(shape), oarr, C.addressof(buf), C.c_int(0)))
        self._after_call("lookup_aggregate")
        return self._block_collect(buf, done, nk, TUPLE_NVALUES[shape], max_groups)

    def lookup_aggregate_folded_marshalled(self, nrows, flt, larr, nlookups, karr, nk, shape, oarr, exchange, max_groups=MAX_LOOKUP_GROUPS):
        """The same over a ROW SHARD, the ranks' partial groups folded on the device (sdqh_lookup_aggregate_block with a device block,
        sdqh_xgroupby_fold): exchange as for xgroupby_folded.  Nothing is waited for."""
        nbytes = self.xgroupby_block_bytes()
        send_ptr, gather = exchange(nbytes)
        self._check(self.lib.sdqh_lookup_aggregate_block(self.handle, C.c_int64(nrows), C.byref(flt), C.c_int(nlookups), larr, C.c_int(nk), karr,
                                                         C.c_int(shape), oarr, C.c_void_p(send_ptr), C.c_int(1)))
        self._after_call("lookup_aggregate")
        recv_ptr, nblocks = gather()
        done = [False]
        buf = self.host_block(nbytes, deferred=done)
        self._check(self.lib.sdqh_xgroupby_fold(self.handle, C.c_void_p(recv_ptr), C.c_int(nblocks), C.addressof(buf)))
        self._after_call("xgroupby_fold")
        return self._block_collect(buf, done, nk, TUPLE_NVALUES[shape], max_groups)

    # -- row programs (ABI 4) ------------------------------------------------------------------------
    def xscan_sum(self, nrows, prog):
        vals = np.zeros(TUPLE_MAX_VALUES, np.float64)
        cnt = C.c_int64()
        self._check(self.lib.sdqh_xscan_sum(self.handle, C.c_int64(nrows), C.byref(prog.struct()), _np_ptr(vals), C.byref(cnt)))
        self._after_call("xscan_sum")
        return vals[:len(prog.vals)], cnt.value

    def xgroupby(self, nrows, prog, max_groups=MAX_LOOKUP_GROUPS):
        out_keys = np.zeros(max_groups, np.int64)
        out_vals = np.zeros((max_groups, TUPLE_MAX_VALUES), np.float64)
        out_cnt = np.zeros(max_groups, np.int64)
        ng = C.c_int32()
        self._check(self.lib.sdqh_xgroupby(self.handle, C.c_int64(nrows), C.byref(prog.struct()), C.c_int(max_groups),
                                           _np_ptr(out_keys), _np_ptr(out_vals), _np_ptr(out_cnt), C.byref(ng)))
        self._after_call("xgroupby")
        n = ng.value
        return out_keys[:n], out_vals[:n, :len(prog.vals)], out_cnt[:n]

    def xgroupby_async(self, nrows, prog, max_groups=MAX_LOOKUP_GROUPS):
        """Launch K-C small and return at once: collect() -> (keys, values, counts) as xgroupby; it waits for THIS call's kernels
        (a completion word in the block), not for what was queued behind them.  What the data decides — too many groups, a
        negative key — is raised by collect(), not here."""
        done = [False]
        buf = self.host_block(self.lib.sdqh_xgroupby_block_bytes(), deferred=done)
        self._check(self.lib.sdqh_xgroupby_async(self.handle, C.c_int64(nrows), C.byref(prog.struct()), C.addressof(buf)))
        self._after_call("xgroupby")
        nvals = len(prog.vals)

        def collect():
            out_keys = np.zeros(max_groups, np.int64)
            out_vals = np.zeros((max_groups, TUPLE_MAX_VALUES), np.float64)
            out_cnt = np.zeros(max_groups, np.int64)
            ng = C.c_int32()
            rc = self.lib.sdqh_xgroupby_collect(self.handle, C.addressof(buf), C.c_int(nvals), C.c_int(max_groups),
                                                _np_ptr(out_keys), _np_ptr(out_vals), _np_ptr(out_cnt), C.byref(ng))
            done[0] = True                                        # (collect waited for this call's kernels: nothing will write the block any more)
            self._check(rc)
            n = ng.value
            return out_keys[:n], out_vals[:n, :nvals], out_cnt[:n]
        return collect

    def xgroupby_block_bytes(self):
        return int(self.lib.sdqh_xgroupby_block_bytes())

    def xgroupby_folded(self, nrows, prog, exchange, max_groups=MAX_LOOKUP_GROUPS):
        """K-C small over a ROW SHARD, the ranks' partial groups folded on the device (sdqh_xgroupby_partial / _fold): nothing is
        waited for.  exchange(block_bytes) -> (send_ptr, gather) hands out this rank's block in the collective's send buffer;
        gather() -> (recv_ptr, nblocks) issues the collective behind the kernel that filled it and names the gathered blocks.
        Returns collect() as xgroupby_async."""
        nbytes = self.xgroupby_block_bytes()
        send_ptr, gather = exchange(nbytes)
        self._check(self.lib.sdqh_xgroupby_partial(self.handle, C.c_int64(nrows), C.byref(prog.struct()), C.c_void_p(send_ptr)))
        self._after_call("xgroupby")
        recv_ptr, nblocks = gather()
        done = [False]
        buf = self.host_block(nbytes, deferred=done)
        self._check(self.lib.sdqh_xgroupby_fold(self.handle, C.c_void_p(recv_ptr), C.c_int(nblocks), C.addressof(buf)))
        self._after_call("xgroupby_fold")
        nvals = len(prog.vals)

        def collect():
            out_keys = np.zeros(max_groups, np.int64)
            out_vals = np.zeros((max_groups, TUPLE_MAX_VALUES), np.float64)
            out_cnt = np.zeros(max_groups, np.int64)
            ng = C.c_int32()
            rc = self.lib.sdqh_xgroupby_collect(self.handle, C.addressof(buf), C.c_int(nvals), C.c_int(max_groups),
                                                _np_ptr(out_keys), _np_ptr(out_vals), _np_ptr(out_cnt), C.byref(ng))
            done[0] = True
            self._check(rc)
            n = ng.value
            return out_keys[:n], out_vals[:n, :nvals], out_cnt[:n]
        return collect

    def xbuild(self, nrows, prog, key_lo=1, key_hi=0, accumulate=False, nsums=None):
        """nsums: how many sums per entry the later probe-aggregate will add, when the caller's plan knows (else room for all four)."""
        h = C.c_void_p()
        acc = 0 if not accumulate else (16 + int(nsums) if nsums is not None and 0 <= int(nsums) <= TUPLE_MAX_VALUES else 1)
        self._check(self.lib.sdqh_xbuild(self.handle, C.c_int64(nrows), C.byref(prog.struct()), C.c_int64(key_lo), C.c_int64(key_hi),
                                         C.c_int(acc), C.byref(h)))
        self._after_call("xbuild")
        return Table(self, h, len(prog.vals), accumulate)

    def xstage(self, nrows, prog):
        """sdqh_xstage: every passing row of the program staged on the device (key, vals; equal keys included; nothing indexed) — the
        source of table_partition_pack.  How many rows it holds stays on the device."""
        h = C.c_void_p()
        self._check(self.lib.sdqh_xstage(self.handle, C.c_int64(nrows), C.byref(prog.struct()), C.byref(h)))
        self._after_call("xstage")
        return Table(self, h, len(prog.vals), False)

    def chunk_words(self, ncols, chunk_rows):
        return int(self.lib.sdqh_chunk_words(C.c_int(ncols), C.c_int64(chunk_rows)))

    def table_partition_pack(self, table, nparts, chunk_rows, packed_ptr, range_upper=None):
        """sdqh_table_partition_pack: the table's staged rows into nparts fixed-capacity chunks at packed_ptr (the layout of an
        equal-split all-to-all), their counts in the chunk headers; nothing is waited for."""
        ru = None
        if range_upper is not None:
            ru = np.ascontiguousarray(range_upper, np.int64)
            assert len(ru) == nparts - 1
        self._check(self.lib.sdqh_table_partition_pack(self.handle, table.handle, C.c_int(nparts), _np_ptr(ru), C.c_int64(chunk_rows), C.c_void_p(packed_ptr)))
        self._after_call("table_partition_pack")

    def unpack_chunks(self, packed_ptr, nparts, dtypes, chunk_rows, pad_key, stat, slot, sent_ptr=None, self_part=0):
        """sdqh_unpack_chunks: the received chunks as Columns of nparts * chunk_rows rows — the sources' rows, then padding rows (key
        pad_key); the step's counts recorded in `stat` (an I64 Column of EXCHANGE_STAT_WORDS rows).  Nothing is waited for."""
        dts = (C.c_int * len(dtypes))(*[int(d) for d in dtypes])
        outs = (C.c_void_p * len(dtypes))()
        self._check(self.lib.sdqh_unpack_chunks(self.handle, C.c_void_p(packed_ptr), C.c_int(nparts), C.c_int(len(dtypes)), dts, C.c_int64(chunk_rows), C.c_int64(pad_key),
                                                C.c_void_p(sent_ptr), C.c_int(self_part), stat.handle if stat is not None else None, C.c_int(slot), outs))
        self._after_call("unpack_chunks")
        return [Column(self, C.c_void_p(outs[i]), nparts * chunk_rows, dtypes[i], 0) for i in range(len(dtypes))]

    # -- plan graphs (sdqh_graph_*): a prepared plan's device calls recorded once, replayed by one call --------------------------
    def graph_begin(self):
        self._check(self.lib.sdqh_graph_begin(self.handle))
        self._capturing = True

    def capturing(self):
        """Between graph_begin and graph_end / graph_abort: the calls are being recorded, nothing executes."""
        return bool(getattr(self, "_capturing", False))

    def graph_end(self):
        """-> Graph, or raises SdqhError(ERR_UNSUPPORTED) when the recording is no graph (the CPU implementation; a call that waited)."""
        h = C.c_void_p()
        self._capturing = False
        self._check(self.lib.sdqh_graph_end(self.handle, C.byref(h)))
        g = Graph(self, h)
        self._graphs.add(g)
        return g

    def graph_abort(self):
        self._capturing = False
        if self.handle is not None:
            # (a recording this runtime cannot take back leaves the stream in capture mode for good: said once, loudly, instead of every
            #  later call failing with "a previous error during capture")
            self._check(self.lib.sdqh_graph_abort(self.handle))

    def xcompact(self, nrows, prog):
        """(Columns [key, vals...] as I64 bit patterns, n): every passing row of the program, duplicate keys included."""
        k = 1 + len(prog.vals)
        outs = (C.c_void_p * k)()
        n = C.c_int64()
        self._check(self.lib.sdqh_xcompact(self.handle, C.c_int64(nrows), C.byref(prog.struct()), outs, C.byref(n)))
        self._after_call("xcompact")
        return [Column(self, C.c_void_p(outs[i]), n.value, I64, 0) for i in range(k)], n.value

    def xkey_set(self, nrows, prog, key_lo, key_hi):
        h = C.c_void_p()
        self._check(self.lib.sdqh_xkey_set(self.handle, C.c_int64(nrows), C.byref(prog.struct()), C.c_int64(key_lo), C.c_int64(key_hi), C.byref(h)))
        self._after_call("xkey_set")
        return Table(self, h, 0, False)

    def xprobe_aggregate(self, nrows, prog, lookup_op, table):
        self._check(self.lib.sdqh_xprobe_aggregate(self.handle, C.c_int64(nrows), C.byref(prog.struct()), C.c_int(lookup_op), table.handle))
        self._after_call("xprobe_aggregate")

    def table_columns(self, table, min_hits=0):
        """(key Column, [payload Columns], [accumulator Columns (F64)], hits Column, n) of the entries with >= min_hits rows."""
        k = 1 + table.npayload + TUPLE_MAX_VALUES + 1
        outs = (C.c_void_p * k)()
        n = C.c_int64()
        self._check(self.lib.sdqh_table_columns(self.handle, table.handle, C.c_int64(min_hits), outs, C.byref(n)))
        cols = [Column(self, C.c_void_p(outs[i]), n.value, F64 if table.npayload < i <= table.npayload + TUPLE_MAX_VALUES else I64, 0) for i in range(k)]
        for c in cols:
            c.source_table = table                                # (HIP build: the columns are views of the table's K-F buffers)
        self._after_call("table_columns")
        return cols[0], cols[1:1 + table.npayload], cols[1 + table.npayload:k - 1], cols[k - 1], n.value

    def jit_compile(self, source):
        """Compile a kept kernel source (sdqlpy_amd/jit_recipes/) into the on-disk cache; nothing is loaded or run."""
        self._check(self.lib.sdqh_jit_compile(self.handle, source.encode()))

    def jit_stats(self):
        """(kernels compiled by hiprtc, kernels loaded from the cache) in this context and its forks."""
        a, b = C.c_int64(), C.c_int64()
        self._check(self.lib.sdqh_jit_stats(self.handle, C.byref(a), C.byref(b)))
        rest = [child.jit_stats() for child in self.forks]
        return a.value + sum(r[0] for r in rest), b.value + sum(r[1] for r in rest)

    def hash_probe_aggregate(self, nrows, flt, table, key, tup):
        self._check(self.lib.sdqh_hash_probe_aggregate(self.handle, C.c_int64(nrows), C.byref(flt), table.handle, key.handle, C.byref(tup)))
        self._after_call("hash_probe_aggregate")

    def table_compact_count(self, table, min_hits):
        """Run the compaction on the device and return the row count; a following table_compact
        with the same min_hits only copies the rows out."""
        n = C.c_int64()
        self._check(self.lib.sdqh_table_compact(self.handle, table.handle, C.c_int64(min_hits), C.c_int64(0), None, None, None, None, C.byref(n)))
        self._after_call("table_compact")
        return n.value

    def table_compact(self, table, min_hits, capacity, want_payload=True, want_values=True, want_hits=True):
        cap = max(1, int(capacity))
        keys = np.empty(cap, np.int64)
        payload = np.empty((max(1, table.npayload), cap), np.int64) if want_payload and table.npayload else None
        values = np.empty((TUPLE_MAX_VALUES, cap), np.float64) if want_values and table.accumulate else None
        hits = np.empty(cap, np.int64) if want_hits else None
        n = C.c_int64()
        self._check(self.lib.sdqh_table_compact(self.handle, table.handle, C.c_int64(min_hits), C.c_int64(cap), _np_ptr(keys),
                                                _np_ptr(payload), _np_ptr(values), _np_ptr(hits), C.byref(n)))
        n = n.value
        return (keys[:n], None if payload is None else payload[:, :n], None if values is None else values[:, :n], None if hits is None else hits[:n])

    def result_wait(self):
        """Wait for the rows of every table_compact_into_block(lazy=True) of this context to be in their arrays."""
        if self.handle is not None:
            self._check(self.lib.sdqh_result_wait(self.handle))

    def table_compact_deferred(self, table, min_hits, capacity_hint, want_payload=True, want_values=True, want_hits=True, replayable=False):
        """K-F with nothing waited for (sdqh_table_compact_deferred): returns collect() -> (keys, payload, values, hits, n), which
        waits for THIS result's copy (its completion word); a result that did not fit the block sized from capacity_hint raises
        SdqhError(ERR_OVERFLOW) there with .needed = its row count (the caller runs the step again, waited for).
        replayable: the call is being recorded into a plan graph — collect() may then be called once after EVERY launch of the graph:
        each time it hands out fresh views of the same block, and collect.rows_out() says whether views of an earlier collection are
        still alive somewhere (the graph must not be launched again while they are: it would overwrite their rows)."""
        if replayable:
            return self._table_compact_replayable(table, min_hits, capacity_hint, want_payload, want_values, want_hits)
        npay = table.npayload if want_payload else 0
        nval = TUPLE_MAX_VALUES if want_values and table.accumulate else 0
        narr = 1 + npay + nval + (1 if want_hits else 0)
        cap = max(1024, int(capacity_hint))
        cap += cap & 1                                        # (whole 16-byte words per array: the library's own copy kernel moves those)
        done = [False]
        buf = self.host_block(narr * cap * 8 + 64, deferred=done)
        flat = np.frombuffer(buf, dtype=np.int64, count=narr * cap + 8)
        cell = flat[narr * cap:narr * cap + 2]                   # [row count, the result's completion word]
        flat = flat[:narr * cap].reshape(narr, cap)
        keys = flat[0]
        payload = flat[1:1 + npay] if npay else None
        values = flat[1 + npay:1 + npay + nval].view(np.float64) if nval else None
        hits = flat[narr - 1] if want_hits else None
        self._check(self.lib.sdqh_table_compact_deferred(self.handle, table.handle, C.c_int64(min_hits), C.c_int64(cap), _np_ptr(keys),
                                                         _np_ptr(payload), _np_ptr(values), _np_ptr(hits), _np_ptr(cell)))
        self._after_call("table_compact")

        def collect():
            if int(cell[1]) == 2:                                 # (no completion word on this device: every pending copy)
                self.result_wait()
            else:
                self._check(self.lib.sdqh_host_wait_word(self.handle, C.c_void_p(cell.ctypes.data + 8), C.c_uint32(1)))
            done[0] = True
            n = int(cell[0])
            if n < 0:
                raise SdqhError(ERR_DEVICE, "table_compact_deferred: the row count never arrived")
            if n > cap:                                           # the block was sized from a bad guess: the rows beyond it were dropped
                exc = SdqhError(ERR_OVERFLOW, "table_compact_deferred: %d rows, room for %d" % (n, cap))
                exc.needed = n
                raise exc
            return (keys[:n], None if payload is None else payload[:, :n], None if values is None else values[:, :n], None if hits is None else hits[:n], n)
        return collect

    def _table_compact_replayable(self, table, min_hits, capacity_hint, want_payload, want_values, want_hits):
        npay = table.npayload if want_payload else 0
        nval = TUPLE_MAX_VALUES if want_values and table.accumulate else 0
        narr = 1 + npay + nval + (1 if want_hits else 0)
        cap = max(1024, int(capacity_hint))
        cap += cap & 1
        nbytes = narr * cap * 8 + 64
        base = self.host_block(nbytes, deferred=[True])          # the recording's own block for as long as `collect` lives (released like any block after)
        addr, size = C.addressof(base), len(base)
        flat = np.frombuffer(base, dtype=np.int64, count=narr * cap + 8)
        cell = flat[narr * cap:narr * cap + 2]
        top = flat[:narr * cap].reshape(narr, cap)
        self._check(self.lib.sdqh_table_compact_deferred(self.handle, table.handle, C.c_int64(min_hits), C.c_int64(cap), _np_ptr(top[0]),
                                                         _np_ptr(top[1:1 + npay] if npay else None), _np_ptr(top[1 + npay:1 + npay + nval] if nval else None),
                                                         _np_ptr(top[narr - 1] if want_hits else None), _np_ptr(cell)))
        self._after_call("table_compact")
        out = [0]                                                 # collections whose views are still alive

        def gone():
            out[0] -= 1

        def collect():
            if int(cell[1]) == 2:
                self.result_wait()
            else:
                self._check(self.lib.sdqh_host_wait_word(self.handle, C.c_void_p(cell.ctypes.data + 8), C.c_uint32(1)))
            n = int(cell[0])
            if n < 0:
                raise SdqhError(ERR_DEVICE, "table_compact_deferred: the row count never arrived")
            if n > cap:
                exc = SdqhError(ERR_OVERFLOW, "table_compact_deferred: %d rows, room for %d" % (n, cap))
                exc.needed = n
                raise exc
            # fresh views over a wrapper of their own: when the last of them dies the block may be written again
            w = (C.c_char * size).from_address(addr)
            w._base = base
            out[0] += 1
            weakref.finalize(w, gone)
            f = np.frombuffer(w, dtype=np.int64, count=narr * cap).reshape(narr, cap)
            keys = f[0]
            payload = f[1:1 + npay] if npay else None
            values = f[1 + npay:1 + npay + nval].view(np.float64) if nval else None
            hits = f[narr - 1] if want_hits else None
            return (keys[:n], None if payload is None else payload[:, :n], None if values is None else values[:, :n], None if hits is None else hits[:n], n)
        collect.rows_out = lambda: out[0] > 0
        return collect

    def table_compact_into_block(self, table, min_hits, capacity_hint, want_payload=True, want_values=True, want_hits=True, lazy=False):
        """One-call K-F: the result arrays are views of an sdqh_host_alloc block sized from
        capacity_hint, which the compaction kernel writes itself; a result that does not fit is
        fetched again with the exact size.  Returns (keys, payload, values, hits, n).
        lazy: the call returns once n is known and the rows arrive behind it (sdqh_table_compact_async): the
        caller must call result_wait() before reading the arrays."""
        npay = table.npayload if want_payload else 0
        nval = TUPLE_MAX_VALUES if want_values and table.accumulate else 0
        narr = 1 + npay + nval + (1 if want_hits else 0)
        cap = max(1024, int(capacity_hint))
        for attempt in (0, 1):
            buf = self.host_block(narr * cap * 8)
            flat = np.frombuffer(buf, dtype=np.int64, count=narr * cap).reshape(narr, cap)
            keys = flat[0]
            payload = flat[1:1 + npay] if npay else None
            values = flat[1 + npay:1 + npay + nval].view(np.float64) if nval else None
            hits = flat[narr - 1] if want_hits else None
            n = C.c_int64()
            fn = self.lib.sdqh_table_compact_async if lazy else self.lib.sdqh_table_compact
            rc = fn(self.handle, table.handle, C.c_int64(min_hits), C.c_int64(cap), _np_ptr(keys),
                    _np_ptr(payload), _np_ptr(values), _np_ptr(hits), C.byref(n))
            if rc == ERR_OVERFLOW and attempt == 0:
                cap = max(1024, n.value)
                continue
            self._check(rc)
            break
        self._after_call("table_compact")
        n = n.value
        return (keys[:n], None if payload is None else payload[:, :n], None if values is None else values[:, :n], None if hits is None else hits[:n], n)

    def table_topk(self, table, min_hits, k, sort, want_hits=True):
        """ORDER BY ... LIMIT k over the entries.  sort: [(kind, index, descending, is_f64)] with kind
        in SORT_KEY / SORT_PAYLOAD / SORT_VALUE / SORT_HITS.  Returns (keys, payload, values, hits) of n <= k rows, in order."""
        k = int(k)
        arr = (SortKey * len(sort))()
        for i, (kind, index, desc, is_f64) in enumerate(sort):
            arr[i].kind, arr[i].index, arr[i].descending, arr[i].is_f64 = int(kind), int(index), int(bool(desc)), int(bool(is_f64))
        keys = np.empty(k, np.int64)
        payload = np.empty((max(1, table.npayload), k), np.int64) if table.npayload else None
        values = np.empty((TUPLE_MAX_VALUES, k), np.float64) if table.accumulate else None
        hits = np.empty(k, np.int64) if want_hits else None
        n = C.c_int64()
        self._check(self.lib.sdqh_table_topk(self.handle, table.handle, C.c_int64(min_hits), C.c_int(k), C.c_int(len(sort)), arr,
                                             _np_ptr(keys), _np_ptr(payload), _np_ptr(values), _np_ptr(hits), C.byref(n)))
        self._after_call("table_topk")
        n = n.value
        return (keys[:n], None if payload is None else payload[:, :n], None if values is None else values[:, :n], None if hits is None else hits[:n])

    def table_entries(self, table):
        """(Columns [key, payload...], n): the table's entries as resident columns."""
        k = 1 + table.npayload
        outs = (C.c_void_p * k)()
        n = C.c_int64()
        self._check(self.lib.sdqh_table_entries(self.handle, table.handle, outs, C.byref(n)))
        return [Column(self, C.c_void_p(outs[i]), n.value, I64, 0) for i in range(k)], n.value

    def scan_compact(self, nrows, flt, probes, cols):
        parr = (Probe * max(1, len(probes)))()
        for i, (tbl, kcol) in enumerate(probes):
            parr[i].table, parr[i].key = tbl.handle, kcol.handle
        carr = (C.c_void_p * len(cols))(*[c.handle for c in cols])
        outs = (C.c_void_p * len(cols))()
        n = C.c_int64()
        self._check(self.lib.sdqh_scan_compact(self.handle, C.c_int64(nrows), C.byref(flt), C.c_int(len(probes)), parr,
                                               C.c_int(len(cols)), carr, outs, C.byref(n)))
        return [Column(self, C.c_void_p(outs[i]), n.value, cols[i].dtype, 0) for i in range(len(cols))], n.value

    def partition_by_key(self, nrows, key, nparts, cols, range_upper=None):
        carr = (C.c_void_p * len(cols))(*[c.handle for c in cols])
        outs = (C.c_void_p * len(cols))()
        counts = np.zeros(nparts, np.int64)
        ru = None
        if range_upper is not None:
            ru = np.ascontiguousarray(range_upper, np.int64)
            assert len(ru) == nparts - 1
        self._check(self.lib.sdqh_partition_by_key(self.handle, C.c_int64(nrows), key.handle, C.c_int(nparts), _np_ptr(ru), C.c_int(len(cols)),
                                                   carr, outs, _np_ptr(counts)))
        return [Column(self, C.c_void_p(outs[i]), nrows, cols[i].dtype, 0) for i in range(len(cols))], counts

    def partition_pack(self, nrows, key, nparts, cols, packed_ptr, range_upper=None):
        """sdqh_partition_pack: the rows of `cols` partitioned by `key` straight into the buffer at packed_ptr (nrows * len(cols) 8-byte
        elements, the all-to-all layout); returns the rows per part."""
        carr = (C.c_void_p * len(cols))(*[c.handle for c in cols])
        counts = np.zeros(nparts, np.int64)
        ru = None
        if range_upper is not None:
            ru = np.ascontiguousarray(range_upper, np.int64)
            assert len(ru) == nparts - 1
        self._check(self.lib.sdqh_partition_pack(self.handle, C.c_int64(nrows), key.handle, C.c_int(nparts), _np_ptr(ru), C.c_int(len(cols)),
                                                 carr, C.c_void_p(packed_ptr), _np_ptr(counts)))
        return counts

    def unpack_parts(self, packed_ptr, part_rows, dtypes):
        """sdqh_unpack_parts: the received buffer (a chunk per source) as Columns of sum(part_rows) rows."""
        rows = np.ascontiguousarray(part_rows, np.int64)
        dts = (C.c_int * len(dtypes))(*[int(d) for d in dtypes])
        outs = (C.c_void_p * len(dtypes))()
        n = C.c_int64()
        self._check(self.lib.sdqh_unpack_parts(self.handle, C.c_void_p(packed_ptr), C.c_int(len(rows)), _np_ptr(rows), C.c_int(len(dtypes)), dts, outs, C.byref(n)))
        return [Column(self, C.c_void_p(outs[i]), n.value, dtypes[i], 0) for i in range(len(dtypes))], n.value

    def copy_out(self, col, row0, nrows, dst_ptr):
        self._check(self.lib.sdqh_column_copy_out(self.handle, col.handle, C.c_int64(row0), C.c_int64(nrows), C.c_void_p(dst_ptr)))

    def copy_in(self, col, row0, nrows, src_ptr):
        self._check(self.lib.sdqh_column_copy_in(self.handle, col.handle, C.c_int64(row0), C.c_int64(nrows), C.c_void_p(src_ptr)))

    def unpack2(self, packed, nrows):
        """(hi, lo) Columns of a column of packed composite keys (sdqh_column_unpack2)."""
        h, l = C.c_void_p(), C.c_void_p()
        self._check(self.lib.sdqh_column_unpack2(self.handle, packed.handle, C.c_int64(nrows), C.byref(h), C.byref(l)))
        return Column(self, h, nrows, I64, 0), Column(self, l, nrows, I64, 0)

    def table_export_bitmap(self, table, lo, hi, into=None):
        """Exact key bitmap of `table` over [lo, hi].  `into`: an I64 Column of at least
        ceil(bits/64) rows to fill (e.g. a wrapped collective buffer); otherwise a new column."""
        h = C.c_void_p(into.handle.value if into is not None else None)
        self._check(self.lib.sdqh_table_export_bitmap(self.handle, table.handle, C.c_int64(lo), C.c_int64(hi), C.byref(h)))
        if into is not None:
            return into
        words32 = ((hi - lo + 1) + 31) // 32
        return Column(self, h, (words32 + 1) // 2, I64, 0)

    def table_from_bitmap(self, words, lo, hi):
        h = C.c_void_p()
        self._check(self.lib.sdqh_table_from_bitmap(self.handle, words.handle, C.c_int64(lo), C.c_int64(hi), C.byref(h)))
        t = Table(self, h, 0, False)
        t._keep = (words,)
        return t


class Library:
    """One loaded implementation of the ABI."""

    def __init__(self, path):
        self.path = path
        self.cdll = C.CDLL(path)
        missing = [s for s in EXPORTS if not hasattr(self.cdll, s)]
        if missing:
            raise OSError("%s does not export: %s" % (path, ", ".join(missing)))
        if self.cdll.sdqh_abi_version() != ABI_VERSION:
            raise OSError("%s implements ABI version %d, this binding is for version %d: rebuild the library"
                          % (path, self.cdll.sdqh_abi_version(), ABI_VERSION))
        L = self.cdll
        L.sdqh_backend_name.restype = C.c_char_p
        L.sdqh_last_error.restype = C.c_char_p
        L.sdqh_last_error.argtypes = [C.c_void_p]
        L.sdqh_destroy.restype = None
        L.sdqh_destroy.argtypes = [C.c_void_p]
        L.sdqh_stream.restype = C.c_void_p
        L.sdqh_stream.argtypes = [C.c_void_p]
        L.sdqh_column_data.restype = C.c_void_p
        L.sdqh_column_data.argtypes = [C.c_void_p]
        L.sdqh_column_rows.restype = C.c_int64
        L.sdqh_column_rows.argtypes = [C.c_void_p]
        L.sdqh_column_dtype.argtypes = [C.c_void_p]
        L.sdqh_column_width.argtypes = [C.c_void_p]
        L.sdqh_column_free.restype = None
        L.sdqh_column_free.argtypes = [C.c_void_p, C.c_void_p]
        L.sdqh_table_free.restype = None
        L.sdqh_table_free.argtypes = [C.c_void_p, C.c_void_p]
        L.sdqh_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.sdqh_synchronize.argtypes = [C.c_void_p]
        L.sdqh_last_device_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.sdqh_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.sdqh_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.sdqh_profile_count.argtypes = [C.c_void_p]
        L.sdqh_profile_entry.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.sdqh_profile_entry_bytes.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.sdqh_column_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.sdqh_column_wrap.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.sdqh_column_alloc.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.sdqh_column_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.sdqh_column_minmax.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_scan_filter_sum.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_groupby_small.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_hash_build_unique.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_int, C.c_void_p]
        L.sdqh_table_size.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_hash_probe_aggregate.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_table_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
        L.sdqh_table_compact_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
        L.sdqh_table_compact_deferred.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p]
        L.sdqh_host_wait_word.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.sdqh_result_wait.argtypes = [C.c_void_p]
        L.sdqh_scan_compact.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
        L.sdqh_partition_by_key.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_column_copy_out.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.sdqh_column_copy_in.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.sdqh_table_export_bitmap.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.sdqh_jit_compile.argtypes = [C.c_void_p, C.c_char_p]
        L.sdqh_column_mark_transient.argtypes = [C.c_void_p, C.c_void_p]
        L.sdqh_column_set_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        L.sdqh_column_unpack2.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.sdqh_partition_pack.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_unpack_parts.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_table_from_bitmap.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.sdqh_table_entries.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_set_profile_filter.argtypes = [C.c_void_p, C.c_char_p]
        L.sdqh_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.sdqh_host_free.restype = None
        L.sdqh_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.sdqh_xscan_sum.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_xgroupby.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_xgroupby_block_bytes.argtypes = []
        L.sdqh_xgroupby_block_bytes.restype = C.c_size_t
        L.sdqh_xgroupby_async.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.sdqh_xgroupby_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_memory_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.sdqh_xgroupby_partial.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.sdqh_xgroupby_fold.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.sdqh_xbuild.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
        L.sdqh_fork.argtypes = [C.c_void_p, C.c_void_p]
        L.sdqh_xkey_set.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        L.sdqh_xcompact.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_xprobe_aggregate.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        L.sdqh_table_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.sdqh_jit_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_xstage.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.sdqh_chunk_words.argtypes = [C.c_int, C.c_int64]
        L.sdqh_chunk_words.restype = C.c_int64
        L.sdqh_table_partition_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        L.sdqh_unpack_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.sdqh_graph_begin.argtypes = [C.c_void_p]
        L.sdqh_graph_end.argtypes = [C.c_void_p, C.c_void_p]
        L.sdqh_graph_abort.argtypes = [C.c_void_p]
        L.sdqh_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
        L.sdqh_graph_nodes.argtypes = [C.c_void_p]
        L.sdqh_graph_free.restype = None
        L.sdqh_graph_free.argtypes = [C.c_void_p, C.c_void_p]
        L.sdqh_build.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.sdqh_lookup_aggregate.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.sdqh_lookup_aggregate_block.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]

    def backend_name(self):
        return self.cdll.sdqh_backend_name().decode()

    def context(self, device=0, threads=1):
        return Context(self, device, threads)
