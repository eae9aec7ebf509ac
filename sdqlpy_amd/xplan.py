"""The open-vocabulary half of the planner: loops whose conditions / values are not one of the
ahead-of-time kernel shapes are compiled into ROW PROGRAMS (include/sdqh.h, ABI 4).

What the reference's generator prints as C++ text for a loop body — boolean structure with `or`
(src/sdqlpy/lib/sdql_compiler.py:277-292), conditional values (lib/sdql_ir.py:294-303), arithmetic on
columns and looked-up fields (lib/sdql_ir_cpp_generator_par.py:712-795), `.at(k)` / `contains(k)`
lookups (85-96), VarChar methods (include/varchar.h:61-124) — is emitted here as a list of typed
operations, and the library specialises a kernel on it (csrc/sdqh_x.hip).  engine.py tries its tuned
fixed-shape calls first and hands a loop over to this module when they refuse it, so the hot TPCH
shapes keep their hand-tuned kernels and everything else still runs on the GPU.

A loop is compiled on its first run (the tables it looks up exist by then, and their layouts — which
payload slot holds which field, which slots are text references — are part of the program) and the
program is reused while those layouts stay the same.
"""
import numpy as np

from . import abi
from .frontend import (And, Bin, Call, Cmp, Col, Const, Contains, IfElse, Lookup, Not, Or, PayloadField, RecordCons,
                       ScalarField, StrIn, UnsupportedQuery, WholeKey)
from .result import Dictionary, DictResult, Pending, ResultSet, TextRefs, decode_text

MAX_CODE_SET = 8          # a text predicate on coded values becomes at most this many equality tests (or one range)


class XV:
    """A value inside a program: operation index, type ('i' | 'f' | 'b'), and what the planner knows about it."""
    __slots__ = ("id", "t", "dec", "rng", "roots", "plain")

    def __init__(self, id, t, dec=None, rng=None, roots=None, plain=None):
        self.id, self.t, self.dec, self.rng = id, t, dec, rng       # dec: text array this int indexes; rng: (lo, hi) of an int
        # roots: the columns of the scanned row the value is a function of (ids of resident columns, "row" for the row number), None:
        # not tracked; plain: the value IS a plain int column / the row number (identifies its root) — engine._share_spec's facts
        self.roots, self.plain = roots, plain


class Text:
    """The text of a column of the scanned row, not yet given a representation."""
    def __init__(self, name, arr):
        self.name, self.arr = name, arr


_T = {"i": abi.T_I64, "f": abi.T_F64, "b": abi.T_BOOL}
_CMP = {"<": abi.X_LT, "<=": abi.X_LE, ">": abi.X_GT, ">=": abi.X_GE, "==": abi.X_EQ, "!=": abi.X_NE}
_ARITH = {"+": abi.X_ADD, "-": abi.X_SUB, "*": abi.X_MUL}


class Compiler:
    """Expressions of one loop -> one abi.Program."""

    def __init__(self, eng, op, htab, env):
        self.eng, self.op, self.htab, self.env = eng, op, htab, env
        self.P = abi.Program()
        self.memo = {}
        self.lookups = {}           # repr(Lookup) -> (operation index, dict name, BuiltTable)
        self.agg_view = {}          # repr(Lookup) -> the name means the aggregation folded into the table, not the build itself (see lookup)
        self.scalars = []           # (CONST operation index, ScalarField): rebound on every run
        self.layout = []            # what the program assumed about the tables it reads: [(dict name, signature)]
        self.entry_cols = []        # (COL operation index, which column of the scanned dictionary's entries): rebound on every run (DictTable)

    def fail(self, why):
        raise UnsupportedQuery("line %d: %s" % (self.op.lineno, why))

    # -- leaves ------------------------------------------------------------------------------------
    def const(self, v):
        key = ("const", type(v).__name__, repr(v))
        if key not in self.memo:
            if isinstance(v, bool):
                self.memo[key] = XV(self.P.op(abi.X_CONST, abi.T_BOOL, imm_i=int(v)), "b")
            elif isinstance(v, (int, np.integer)):
                self.memo[key] = XV(self.P.op(abi.X_CONST, abi.T_I64, imm_i=int(v)), "i", rng=(int(v), int(v)))
            elif isinstance(v, (float, np.floating)):
                self.memo[key] = XV(self.P.op(abi.X_CONST, abi.T_F64, imm_f=float(v)), "f")
            else:
                self.fail("constant %r has no place in an expression" % (v,))
        return self.memo[key]

    def column(self, name):
        arr = self.htab.array(name, self.op)
        if arr.dtype.kind == "U":
            return Text(name, arr)
        col = self.eng.column(arr)
        if arr.dtype == np.int64:
            return XV(self.P.op(abi.X_COL, abi.T_I64, col=col), "i", rng=("col", col), roots=frozenset([id(col)]), plain=True)
        if arr.dtype == np.float64:
            return XV(self.P.op(abi.X_COL, abi.T_F64, col=col), "f", roots=frozenset([id(col)]))
        self.fail("column '%s' has unsupported dtype %s" % (name, arr.dtype))

    def text_as_int(self, tx, coded_ok=True):
        """Dictionary code (low-cardinality column) or row number of a text column: an int the device can
        carry.  A build payload takes the row number (coded_ok = False): every text field of a row then
        shares ONE payload slot and differs only in the array the reference indexes."""
        key = ("textint", tx.name, coded_ok)
        if key not in self.memo:
            coded = self.eng.dict_column(tx.arr) if coded_ok else None
            if coded is not None:
                self.memo[key] = XV(self.P.op(abi.X_COL, abi.T_I64, col=coded[0]), "i", dec=coded[1], rng=(0, max(0, len(coded[1]) - 1)), roots=frozenset([id(coded[0])]))
            else:
                if "rowid" not in self.memo:
                    self.memo["rowid"] = self.P.op(abi.X_ROWID, abi.T_I64)      # one row reference serves every text column of the row
                self.memo[key] = XV(self.memo["rowid"], "i", dec=tx.arr, rng=(0, max(0, self.htab.nrows - 1)), roots=frozenset(["row"]), plain=True)
        return self.memo[key]

    def resolve_rng(self, v):
        """(lo, hi) of an int value, or None."""
        if v.rng is None:
            return None
        if v.rng[0] == "col":
            col = v.rng[1]
            v.rng = col.minmax() if col.nrows else (0, 0)
        return v.rng

    # -- lookups ------------------------------------------------------------------------------------
    def lookup(self, lk):
        key = repr(lk)
        if key in self.lookups:
            return self.lookups[key]
        bt = self.env.get(lk.dict_name)
        # WHICH dictionary the name means.  A probe-aggregate folds its groups into the entries of the table it probes, so two
        # dictionaries of a plan can be one device table: the build itself (key -> payload fields) and the aggregation over it
        # (key -> sums: env holds ("aggregated", table name) for that one).  A lookup by the BUILD's name reads payload fields and
        # finds every entry; a lookup by the AGGREGATION's name reads the sums and finds the entries that received a row.
        aggregated = False
        if not hasattr(bt, "table"):
            if isinstance(bt, tuple) and bt and bt[0] == "aggregated":
                bt = self.env[bt[1]]
                aggregated = True
            else:
                self.fail("'%s' is not a table a loop can look up" % lk.dict_name)
        self.agg_view[key] = aggregated
        parts = [x for _, x in lk.key.fields] if isinstance(lk.key, RecordCons) else [lk.key]
        if getattr(bt, "key_radix", None) is not None:
            self.fail("'%s' is keyed by more than two fields: it cannot be looked up" % lk.dict_name)
        if (bt.key_parts is not None) != (len(parts) == 2) or len(parts) > 2:
            self.fail("lookup into '%s' does not match its key shape" % lk.dict_name)
        vals = [self.as_int(self.value(p), "lookup key") for p in parts]
        kid = vals[0].id if len(vals) == 1 else self.P.op(abi.X_PACK2, abi.T_I64, a=vals[0].id, b=vals[1].id)
        oid = self.P.op(abi.X_LOOKUP, abi.T_BOOL, a=kid, table=bt.table)
        self.lookups[key] = (oid, lk.dict_name, bt, vals)
        self.layout.append((lk.dict_name, _table_signature(bt)))
        return self.lookups[key]

    def found(self, lk):
        """Operation index of `key in dictionary`.  An aggregated dictionary holds the entries that received at least one row (K-F
        keeps hits >= 1): an entry of its table that no row reached — a build row nothing matched, a key of a dense domain no row
        carried — is not in it."""
        oid, _, bt, _ = self.lookup(lk)
        if bt.agg is None or not self.agg_view[repr(lk)]:
            return oid
        key = ("live", oid)
        if key not in self.memo:
            hits = self.P.op(abi.X_ACC, abi.T_I64, a=oid, aux=-1)
            self.memo[key] = self.P.op(abi.X_AND, abi.T_BOOL, a=oid, b=self.P.op(abi.X_GT, abi.T_BOOL, a=hits, b=self.const(0).id))
        return self.memo[key]

    def field(self, lk, fname):
        oid, name, bt, keyvals = self.lookup(lk)
        if bt.agg is not None and self.agg_view[repr(lk)]:          # an aggregated dictionary: its values are the entries' accumulators
            _, vnames, count_idx, _, val_is_record, nv = bt.agg
            if fname is None:
                if val_is_record and len(vnames) != 1:
                    self.fail("the looked-up value of '%s' is a record; name a field" % name)
                i = 0
            elif fname in vnames:
                i = vnames.index(fname)
            else:
                self.fail("'%s' has no value field '%s'" % (name, fname))
            if count_idx is not None and i == count_idx:
                return XV(self.P.op(abi.X_ACC, abi.T_I64, a=oid, aux=-1), "i")
            k = i - (1 if count_idx is not None and count_idx < i else 0)
            return XV(self.P.op(abi.X_ACC, abi.T_F64, a=oid, aux=k), "f")
        if bt.table.npayload == 0 and not bt.val_fields:
            self.fail("'%s' is a key set: it has no fields" % name)
        slot = bt.slot_of(fname)
        if slot is None:
            self.fail("'%s' has no field '%s'" % (name, fname))
        if slot == "key":
            if len(keyvals) != 1:
                self.fail("a field that repeats a composite key cannot be read back")
            return keyvals[0]
        dt = np.dtype(bt.payload_dtypes[slot])
        dec = bt.decoder_of(fname, slot)
        fkey = ("field", oid, slot)                              # one operation per slot: fields that share a slot (text of one row) share it
        if fkey not in self.memo:
            self.memo[fkey] = self.P.op(abi.X_FIELD, abi.T_F64 if dt.kind == "f" else abi.T_I64, a=oid, aux=slot)
        roots = frozenset().union(*[v.roots for v in keyvals]) if all(v.roots is not None for v in keyvals) else None      # a field of the entry the key finds
        if dt.kind == "f":
            return XV(self.memo[fkey], "f", roots=roots)
        rng = (0, len(dec) - 1) if dec is not None else _slot_range(bt, slot)
        return XV(self.memo[fkey], "i", dec=dec, rng=rng, roots=roots)

    # -- expressions ----------------------------------------------------------------------------------
    def as_int(self, v, what):
        if isinstance(v, Text):
            v = self.text_as_int(v)
        if not isinstance(v, XV) or v.t != "i":
            self.fail("%s must be an integer value" % what)
        return v

    def as_float(self, v):
        if isinstance(v, XV) and v.t == "f":
            return v
        if isinstance(v, XV) and v.t == "i" and v.dec is None:
            return XV(self.P.op(abi.X_I2F, abi.T_F64, a=v.id), "f")
        self.fail("a number was expected")

    def value(self, e):
        key = repr(e)
        if key in self.memo and not isinstance(e, Const):
            return self.memo[key]
        v = self._value(e)
        self.memo[key] = v
        return v

    def _value(self, e):
        if isinstance(e, Const):
            if isinstance(e.value, str):
                return e.value
            return self.const(e.value)
        if isinstance(e, Col):
            return self.column(e.name)
        if isinstance(e, WholeKey):
            if not isinstance(self.htab, DictTable):
                self.fail("p[0] / p[1] outside a sum over a dictionary")
            return self.htab.whole(self, e)
        if isinstance(e, ScalarField):
            x = XV(self.P.op(abi.X_CONST, abi.T_F64, imm_f=0.0), "f")       # its own operation: rebound on every run
            self.scalars.append((x.id, e))
            return x
        if isinstance(e, PayloadField):
            return self.field(e.lookup, e.field)
        if isinstance(e, Lookup):
            return self.field(e, None)
        if isinstance(e, (Cmp, And, Or, Not, Contains, StrIn)):
            return self.cond(e)
        if isinstance(e, IfElse):
            c = self.cond(e.cond)
            a, b = self.value(e.then), self.value(e.other)
            if isinstance(a, Text):
                a = self.text_as_int(a)
            if isinstance(b, Text):
                b = self.text_as_int(b)
            if a.t != b.t:
                if {a.t, b.t} == {"i", "f"}:
                    a, b = self.as_float(a), self.as_float(b)
                else:
                    self.fail("the arms of a conditional value have different types")
            if a.dec is not b.dec and (a.dec is not None or b.dec is not None):
                self.fail("the arms of a conditional value are references into different texts")
            rng = None
            if a.t == "i":
                ra, rb = self.resolve_rng(a), self.resolve_rng(b)
                rng = (min(ra[0], rb[0]), max(ra[1], rb[1])) if ra and rb else None
            return XV(self.P.op(abi.X_SELECT, _T[a.t], a=c.id, b=a.id, c=b.id), a.t, dec=a.dec, rng=rng)
        if isinstance(e, Bin):
            a, b = self.value(e.left), self.value(e.right)
            if isinstance(a, (Text, str)) or isinstance(b, (Text, str)):
                self.fail("arithmetic on text")
            if a.t == "b" or b.t == "b":
                self.fail("arithmetic on a condition")
            if e.op == "/":
                a, b = self.as_float(a), self.as_float(b)          # Python's true division
                return XV(self.P.op(abi.X_DIV, abi.T_F64, a=a.id, b=b.id), "f")
            if a.t != b.t:
                a, b = self.as_float(a), self.as_float(b)
            if a.dec is not None or b.dec is not None:
                self.fail("arithmetic on a text reference")
            rng = None
            if a.t == "i":
                ra, rb = self.resolve_rng(a), self.resolve_rng(b)
                if ra and rb and e.op in "+-":
                    rng = (ra[0] + rb[0], ra[1] + rb[1]) if e.op == "+" else (ra[0] - rb[1], ra[1] - rb[0])
            return XV(self.P.op(_ARITH[e.op], _T[a.t], a=a.id, b=b.id), a.t, rng=rng)
        if isinstance(e, Call):
            if e.fn == "extractYear":
                a = self.as_int(self.value(e.args[0]), "extractYear's argument")
                r = self.resolve_rng(a)
                return XV(self.P.op(abi.X_YEAR, abi.T_I64, a=a.id), "i", rng=(r[0] // 10000, r[1] // 10000) if r else None)
            if e.fn == "firstIndex":
                tx = self.value(e.args[0])
                if not isinstance(tx, Text):
                    self.fail("firstIndex needs a text column of the scanned table")
                key = ("stridx", tx.name, e.args[1].value)                 # the same search named twice in a condition is one operation
                if key not in self.memo:
                    self.memo[key] = XV(self.P.op(abi.X_STRIDX, abi.T_I64, col=self.eng.column(tx.arr), text=e.args[1].value), "i", rng=(-1, tx.arr.dtype.itemsize // 4))
                return self.memo[key]
            if e.fn == "substr":
                self.fail("substr() is only supported as a group key")
        self.fail("unsupported expression %r" % (e,))

    # -- conditions -----------------------------------------------------------------------------------
    def fold(self, code, ids):
        out = ids[0]
        for x in ids[1:]:
            out = self.P.op(code, abi.T_BOOL, a=out, b=x)
        return out

    def code_set(self, v, passes, negate=False):
        """`passes(text)` on a coded value: evaluated on the (host) texts the codes index, compiled to tests on the code."""
        hit = np.nonzero(passes(v.dec))[0]
        if negate:
            hit = np.setdiff1d(np.arange(len(v.dec)), hit)
        if len(hit) == 0:
            return self.const(False)
        if len(hit) == len(v.dec):
            return self.const(True)
        if hit[-1] - hit[0] + 1 == len(hit):                          # one run of codes
            lo, hi = self.const(int(hit[0])), self.const(int(hit[-1]))
            if len(hit) == 1:
                return XV(self.P.op(abi.X_EQ, abi.T_BOOL, a=v.id, b=lo.id), "b")
            return XV(self.P.op(abi.X_AND, abi.T_BOOL, a=self.P.op(abi.X_GE, abi.T_BOOL, a=v.id, b=lo.id), b=self.P.op(abi.X_LE, abi.T_BOOL, a=v.id, b=hi.id)), "b")
        if len(hit) <= MAX_CODE_SET:
            return XV(self.fold(abi.X_OR, [self.P.op(abi.X_EQ, abi.T_BOOL, a=v.id, b=self.const(int(h)).id) for h in hit]), "b")
        self.fail("a text condition that selects %d scattered values of %d is not supported" % (len(hit), len(v.dec)))

    def text_pred(self, v, text, mode):
        """mode: abi.STR_*; v: Text (column of the scanned row) or a coded XV."""
        fns = {abi.STR_EQ: lambda d: d == text, abi.STR_NE: lambda d: d != text, abi.STR_CONTAINS: lambda d: np.char.find(d, text) >= 0,
               abi.STR_PREFIX: lambda d: np.char.startswith(d, text), abi.STR_SUFFIX: lambda d: np.char.endswith(d, text)}
        if isinstance(v, Text):
            coded = self.eng.dict_column(v.arr)
            if coded is not None:
                try:
                    return self.code_set(self.text_as_int(v), fns[mode])
                except UnsupportedQuery:
                    pass                                              # too many scattered codes: test the text itself
            return XV(self.P.op(abi.X_STR, abi.T_BOOL, col=self.eng.column(v.arr), aux=mode, text=text), "b")
        if isinstance(v, XV) and v.dec is not None:
            return self.code_set(v, fns[mode])
        self.fail("a text condition needs a text column or a looked-up text field")

    def cond(self, e):
        key = ("cond", repr(e))
        if key not in self.memo:
            self.memo[key] = self._cond(e)
        return self.memo[key]

    def _cond(self, e):
        if isinstance(e, Const) and isinstance(e.value, bool):
            return self.const(e.value)
        if isinstance(e, And):
            return XV(self.fold(abi.X_AND, [self.cond(t).id for t in e.terms]), "b") if e.terms else self.const(True)
        if isinstance(e, Or):
            return XV(self.fold(abi.X_OR, [self.cond(t).id for t in e.terms]), "b")
        if isinstance(e, Not):
            return XV(self.P.op(abi.X_NOT, abi.T_BOOL, a=self.cond(e.term).id), "b")
        if isinstance(e, Contains):
            return XV(self.found(e.lookup), "b")
        if isinstance(e, StrIn):
            return self.text_pred(self.value(e.col), e.needle, {"in": abi.STR_CONTAINS, "prefix": abi.STR_PREFIX, "suffix": abi.STR_SUFFIX}[e.how])
        if isinstance(e, Cmp):
            a, b = self.value(e.left), self.value(e.right)
            if isinstance(a, str):
                a, b, op = b, a, {"<": ">", "<=": ">=", ">": "<", ">=": "<=", "==": "==", "!=": "!="}[e.op]
            else:
                op = e.op
            if isinstance(b, str):
                if op not in ("==", "!="):
                    self.fail("text supports == / != against a literal")
                return self.text_pred(a, b, abi.STR_EQ if op == "==" else abi.STR_NE)
            if isinstance(a, Text) or isinstance(b, Text):
                self.fail("comparing two text columns is not supported")
            if a.t == "b" or b.t == "b":
                if a.t != b.t or op not in ("==", "!="):
                    self.fail("conditions compare with == / != only")
            elif a.t != b.t:
                a, b = self.as_float(a), self.as_float(b)
            elif a.dec is not None or b.dec is not None:
                # integers that stand for text.  Codes of ONE dictionary with pairwise distinct entries compare
                # equal exactly when the texts do; their ORDER is not text order, and row references into a raw text
                # column are not even equal for equal texts in different rows
                if a.dec is not b.dec:
                    self.fail("comparing references into different texts")
                if not isinstance(a.dec, Dictionary):
                    self.fail("comparing two texts given as row references (not dictionary codes) is not supported")
                if op not in ("==", "!="):
                    self.fail("text supports == / != only (dictionary codes are not in text order)")
            return XV(self.P.op(_CMP[op], abi.T_BOOL, a=a.id, b=b.id), "b")
        v = self.value(e)
        if isinstance(v, XV) and v.t == "b":
            return v
        self.fail("%r is not a condition" % (e,))

    # -- group / build keys --------------------------------------------------------------------------------
    def key_parts(self, exprs):
        """[(name, Expr)] -> [XV ints]; substr(col, a, b) expands to one part per code unit."""
        out = []
        for name, e in exprs:
            if isinstance(e, Call) and e.fn == "substr":
                tx = self.value(e.args[0])
                if not isinstance(tx, Text):
                    self.fail("substr needs a text column of the scanned table")
                a, b = e.args[1].value, e.args[2].value
                col = self.eng.column(tx.arr)
                units = [XV(self.P.op(abi.X_CHAR, abi.T_I64, col=col, aux=i), "i", rng=(0, 0x10FFFF)) for i in range(a, b + 1)]
                out.append((name, units, ("chars", b - a + 1)))
                continue
            v = self.value(e)
            v = self.as_int(v, "a key part")
            out.append((name, [v], ("int", v.dec)))
        return out

    def pack_radix(self, parts):
        """One non-negative i64 from several bounded int parts; returns (operation index, [(lo, span)])."""
        flat = [v for _, vs, _ in parts for v in vs]
        if len(flat) == 1:
            r = self.resolve_rng(flat[0])
            if r is not None and r[0] < 0:
                lo = self.const(r[0])
                return self.P.op(abi.X_SUB, abi.T_I64, a=flat[0].id, b=lo.id), [(r[0], r[1] - r[0] + 1)]
            return flat[0].id, [(0, None)]
        radix, total = [], 1
        for v in flat:
            r = self.resolve_rng(v)
            if r is None:
                self.fail("a group key of several parts needs parts with known value ranges")
            radix.append((r[0], r[1] - r[0] + 1))
            total *= r[1] - r[0] + 1
        if total >= 1 << 62:
            self.fail("the group key's value ranges do not fit 62 bits")
        acc = None
        for v, (lo, span) in zip(flat, radix):
            term = v.id if lo == 0 else self.P.op(abi.X_SUB, abi.T_I64, a=v.id, b=self.const(lo).id)
            acc = term if acc is None else self.P.op(abi.X_ADD, abi.T_I64, a=self.P.op(abi.X_MUL, abi.T_I64, a=acc, b=self.const(span).id), b=term)
        return acc, radix

    # -- run-time bindings -----------------------------------------------------------------------------------
    def bind(self, env):
        for oid, name, _, _ in self.lookups.values():
            bt = env[name]
            if isinstance(bt, tuple):
                bt = env[bt[1]]
            self.P.bind_table(oid, bt.table)
        for oid, sf in self.scalars:
            v = env[sf.name]
            self.P.set_const(oid, float(v if sf.field is None else v[sf.field]))
        for oid, which in self.entry_cols:
            self.P.bind_col(oid, self.htab.columns[which])

    def still_valid(self, env):
        for name, sig in self.layout:
            bt = env.get(name)
            if isinstance(bt, tuple):
                bt = env.get(bt[1])
            if not hasattr(bt, "table") or _table_signature(bt) != sig:
                return False
        return True


def _slot_range(bt, slot):
    """(lo, hi) of an int payload slot when the build recorded where its values came from."""
    rng = getattr(bt, "slot_rng", {}).get(slot)
    if rng is not None:
        return rng
    cols = getattr(bt, "slot_cols", None)
    if cols is not None and slot < len(cols) and cols[slot].dtype == abi.I64 and cols[slot].nrows:
        return cols[slot].minmax()
    plain = bt.slot_plain.get(slot) if bt.slot_plain else None
    if plain is not None:
        return (plain[0], plain[0] + plain[1] - 1)
    return None


def _table_signature(bt):
    return (tuple(bt.val_fields), bt.dtype_sig(), bt.key_parts is not None,
            tuple(sorted((k, id(v)) for k, v in bt.decoders.items())), tuple(sorted((k, id(v)) for k, v in bt.field_decoders.items())),
            None if bt.agg is None else (tuple(bt.agg[1]), bt.agg[2], bt.agg[5]), bt.table.npayload, bool(bt.table.accumulate),
            None if getattr(bt, "key_radix", None) is None else tuple(bt.key_radix[1]))


def _decode_radix(keys, parts, radix):
    """Packed group keys -> one array per key field."""
    out = []
    rest = np.asarray(keys, np.int64).copy()
    flat = []
    for (lo, span) in reversed(radix):
        if span is None:
            flat.append(rest + lo); rest = np.zeros_like(rest)
        else:
            flat.append(rest % span + lo); rest = rest // span
    flat.reverse()
    at = 0
    for name, vs, kind in parts:
        cols = flat[at:at + len(vs)]
        at += len(vs)
        if kind[0] == "chars":
            units = np.stack(cols, axis=1).astype(np.uint32)
            out.append((name, np.ascontiguousarray(units).view("<U%d" % kind[1]).reshape(len(keys))))
        else:
            dec = kind[1]
            out.append((name, decode_text(cols[0], dec) if dec is not None else cols[0]))      # text of a large result stays references until read
    return out


def _encoding_fingerprint(parts, radix):
    """What a packed group key MEANS, as a digest: the packing (per part its offset and span) and every decoder's contents.  Ranks whose
    digests are equal may fold their partial groups by packed key on the device (dist.DistributedRunner); a decoder that is a table's own
    text column (row references: large, and local to a rank's shard) gives None — such keys are merged by their decoded values on the
    host."""
    import hashlib
    h = hashlib.sha1(repr([(int(lo), None if span is None else int(span)) for lo, span in radix]).encode())
    for _, vs, kind in parts:
        h.update(("%s/%d;" % (kind[0], len(vs))).encode())
        if kind[0] == "chars":
            continue
        dec = kind[1]
        if dec is None:
            continue
        arr = dec if isinstance(dec, np.ndarray) else None
        if arr is None or len(arr) > (1 << 16):
            return None
        a = np.ascontiguousarray(arr)
        h.update(a.dtype.str.encode()); h.update(a.view(np.uint8).tobytes())
    return h.hexdigest()


def _is_light(c):
    """A comparison between numeric columns / numbers: evaluable on streamed registers."""
    return isinstance(c, Cmp) and all(isinstance(x, (Col, Const)) and not (isinstance(x, Const) and isinstance(x.value, (str, type(None)))) for x in (c.left, c.right))


def groups_by_entry(op):
    """Is the group of this aggregation the entry matched by its joinProbe index (key = the probe key and / or
    fields of that entry)?  Then it is summed into the entry itself (K-C large)."""
    if op.kind != "dict" or op.unique or op.probe is None or not isinstance(op.probe.key, Col):
        return False
    fields = op.key.fields if isinstance(op.key, RecordCons) else [(None, op.key)]
    pk = op.probe.key.name
    return all((isinstance(e, Col) and e.name == pk) or (isinstance(e, PayloadField) and repr(e.lookup) == repr(op.probe)) for _, e in fields)


# =================================================================================================
def prepare_scan(eng, op, htab, accumulate_into, member_only, as_table=False, small_groups_only=False):
    """closure(env) for one table loop, through row programs.
    small_groups_only: an aggregation whose groups do not fit the group-by sinks is refused (UnsupportedQuery) instead of being
    run as build + probe-aggregate — the caller has a better plan for a large group domain (engine.py: the fixed group-by-key call)."""
    from .engine import BuiltTable
    ctx = eng.ctx                                    # (the row count is read when a loop runs: the entries of a dictionary differ from run to run)
    state = {}

    def value_fields(val):
        if isinstance(val, RecordCons):
            return [nm for nm, _ in val.fields], [e for _, e in val.fields], True
        return [None], [val], False

    def summed_values(c, exprs):
        """f64 operations to sum + which field is a plain row count."""
        ids, count_idx = [], None
        for i, e in enumerate(exprs):
            if isinstance(e, Const) and isinstance(e.value, int) and not isinstance(e.value, bool) and e.value == 1 and count_idx is None:
                count_idx = i
                continue
            v = c.value(e)
            if isinstance(v, XV) and v.t == "i" and v.dec is None:    # an integer-valued sum (`1 if c else 0`, an int column): summed as doubles, returned as int64
                if not isinstance(e, Const):
                    state.setdefault("int_values", set()).add(vnames[i])
                v = c.as_float(v)
            if not isinstance(v, XV) or v.t != "f":
                raise UnsupportedQuery("line %d: summed values must be floating-point expressions (or the constant 1 for a count): %r" % (op.lineno, e))
            ids.append(v.id)
        if len(ids) > abi.TUPLE_MAX_VALUES:
            raise UnsupportedQuery("line %d: more than %d summed values in one loop" % (op.lineno, abi.TUPLE_MAX_VALUES))
        return ids, count_idx

    def gates_of(c, conds):
        """Gate operations of the loop: numeric comparisons first (the kernels test those on streamed
        registers), then the joinProbe index, then everything else in source order."""
        light = [x for x in conds if _is_light(x)]
        rest = [x for x in conds if not _is_light(x)]
        ids = [c.cond(x).id for x in light]
        # membership tests keyed by the scanned row alone go in front of the joinProbe index: they are the selective ones (Q18: 57 keys
        # of 15 M) and the kernels test the first lookup on streamed keys; the index is usually a join that every row survives
        from .engine import _walk_lookups
        def own_keyed(x):
            if not isinstance(x, Contains):
                return False
            inner = []
            _walk_lookups(x.lookup.key, inner)
            return not inner
        if op.probe is not None:
            early = [x for x in rest if own_keyed(x)]
            rest = [x for x in rest if not own_keyed(x)]
            ids += [c.cond(x).id for x in early]
        probe_id = None
        if op.probe is not None:
            probe_id = c.lookup(op.probe)[0]
            ids.append(probe_id)
            live = c.found(op.probe)
            if live != probe_id:                                     # (an aggregated dictionary as the index: its entries are those with rows)
                ids.append(live)
        ids += [c.cond(x).id for x in rest]
        if len(ids) > abi.MAX_XGATES:                                # the tail as one gate
            ids = ids[:abi.MAX_XGATES - 1] + [c.fold(abi.X_AND, ids[abi.MAX_XGATES - 1:])]
            if probe_id is not None and probe_id not in ids:
                raise UnsupportedQuery("line %d: too many conditions in one loop" % op.lineno)
        return ids, probe_id

    # ---- scalar sums ----------------------------------------------------------------------------------
    if op.kind in ("scalar", "scalar_record"):
        def compile_scalar(env):
            c = Compiler(eng, op, htab, env)
            gates, _ = gates_of(c, op.conds)
            if op.kind == "scalar":
                names, fields = [None], [(op.val, [])]
            else:
                names, fields = [nm for nm, _, _ in op.fields], [(e, fc) for _, e, fc in op.fields]
            vals, counts = [], []
            for e, fconds in fields:
                if isinstance(e, Const) and isinstance(e.value, int) and not isinstance(e.value, bool) and not fconds:
                    counts.append(float(e.value)); vals.append(None)
                    continue
                v = c.value(e)
                v = c.as_float(v) if isinstance(v, XV) and v.t in "if" else None
                if v is None:
                    raise UnsupportedQuery("line %d: a scalar sum needs a numeric value: %r" % (op.lineno, e))
                if fconds:
                    cnd = c.cond(And(list(fconds)))
                    v = XV(c.P.op(abi.X_SELECT, abi.T_F64, a=cnd.id, b=v.id, c=c.const(0.0).id), "f")
                counts.append(None); vals.append(v.id)
            live = [v for v in vals if v is not None]
            if len(live) > abi.TUPLE_MAX_VALUES:
                raise UnsupportedQuery("line %d: more than %d sums in one scalar record" % (op.lineno, abi.TUPLE_MAX_VALUES))
            c.P.gates, c.P.vals = gates, live
            return c, names, vals, counts

        def run_scalar(env):
            st = state.get("c")
            if st is None or not st[0].still_valid(env):
                st = state["c"] = compile_scalar(env)
            c, names, vals, counts = st
            c.bind(env)
            sums, cnt = ctx.xscan_sum(htab.nrows, c.P)
            it = iter(sums.tolist())
            out = [float(cnt) * k if v is None else next(it) for v, k in zip(vals, counts)]
            return out[0] if op.kind == "scalar" else dict(zip(names, out))
        return run_scalar

    key_is_record = isinstance(op.key, RecordCons)
    key_fields = op.key.fields if key_is_record else [(None, op.key)]

    # ---- unique builds ---------------------------------------------------------------------------------
    if op.unique:
        val_is_record = isinstance(op.val, RecordCons)
        vfields = op.val.fields if val_is_record else ([] if (isinstance(op.val, Const) and op.val.value is True) else [(None, op.val)])
        accumulate = op.out in accumulate_into

        def compile_build(env, coded=True):
            """coded: text payloads of low-cardinality columns travel as dictionary codes (later loops can compare
            and group them cheaply); if that needs more payload slots than an entry has, every text field falls
            back to the shared row reference."""
            c = Compiler(eng, op, htab, env)
            if coded:
                # a row reference is one slot however many text fields share it; a dictionary code is a slot per field.
                # So once one text payload needs the reference (its column has too many distinct values to code), all take it.
                for _, e in vfields:
                    if isinstance(e, Col) and htab.cols.get(e.name) is not None and htab.cols[e.name].dtype.kind == "U" and eng.dict_column(htab.cols[e.name]) is None:
                        coded = False
            gates, _ = gates_of(c, op.conds)
            parts = c.key_parts(key_fields)
            flat = [v for _, vs, _ in parts for v in vs]
            if len(flat) > 2:
                raise UnsupportedQuery("line %d: build keys of more than two parts are not supported" % op.lineno)
            key_names = [nm or (e.name if isinstance(e, Col) else "key%d" % i) for i, (nm, e) in enumerate(key_fields)]
            bounds = (1, 0)
            if len(flat) == 1:
                kid = flat[0].id
                r = c.resolve_rng(flat[0])
                if r is not None:
                    bounds = r
            else:
                kid = c.P.op(abi.X_PACK2, abi.T_I64, a=flat[0].id, b=flat[1].id)
            val_fields, pay_ids, dtypes, decoders, field_decoders, slot_rng = [], [], [], {}, {}, {}
            pay_xvs = []
            for fname, e in vfields:
                if len(key_fields) == 1 and repr(e) == repr(key_fields[0][1]):
                    val_fields.append((fname, "key"))
                    continue
                v = c.value(e)
                if isinstance(v, Text):
                    v = c.text_as_int(v, coded_ok=coded)
                if not isinstance(v, XV) or v.t == "b":
                    raise UnsupportedQuery("line %d: a build payload must be a number or a text: %r" % (op.lineno, e))
                j = next((j for j, pid in enumerate(pay_ids) if pid == v.id), None)
                if j is None:
                    j = len(pay_ids)
                    pay_ids.append(v.id); dtypes.append(np.dtype(np.float64 if v.t == "f" else np.int64)); pay_xvs.append(v)
                    if v.dec is not None:
                        decoders[j] = v.dec
                    elif v.t == "i":
                        r = c.resolve_rng(v)
                        if r is not None:
                            slot_rng[j] = r
                if v.dec is not None:
                    field_decoders[fname] = v.dec                  # text fields of one row share a row-reference slot: the text differs per field
                val_fields.append((fname, j))
            if len(pay_ids) > abi.MAX_PAYLOAD:
                if coded:
                    return compile_build(env, False)
                raise UnsupportedQuery("line %d: more than %d distinct payload values per entry" % (op.lineno, abi.MAX_PAYLOAD))
            c.P.gates, c.P.key, c.P.vals = gates, kid, pay_ids
            # what each payload slot is a function of, for a later aggregation keyed by entry fields only (engine._share_spec, Q10's shape)
            roots, plain = None, {}
            if all(v.roots is not None for v in pay_xvs):
                roots = {j: v.roots for j, v in enumerate(pay_xvs)}
                for j, v in enumerate(pay_xvs):
                    if v.plain and (v.dec is None or "row" in v.roots):
                        r = (0, max(1, htab.nrows)) if "row" in v.roots else c.resolve_rng(v)
                        if r is not None:
                            plain[j] = r if "row" in v.roots else (r[0], r[1] - r[0] + 1)
            state["share_facts"] = (roots, plain)
            return c, key_names, bounds, val_fields, dtypes, (decoders, field_decoders, slot_rng), (flat[0].dec if len(flat) == 1 else [v.dec for v in flat]), len(flat) == 2

        def run_build(env):
            st = state.get("c")
            if st is None or not st[0].still_valid(env):
                st = state["c"] = compile_build(env)
            c, key_names, bounds, val_fields, dtypes, decoders, key_dec, composite = st
            c.bind(env)
            table = None
            dense = bounds[0] <= bounds[1] and bounds[1] - bounds[0] + 1 <= (1 << 31) and bounds[1] - bounds[0] + 1 <= 64 * max(htab.nrows, 1024)
            if member_only and not c.P.vals and not accumulate and dense:
                table = ctx.xkey_set(htab.nrows, c.P, bounds[0], bounds[1])
            if table is None and not c.P.gates and c.P.vals and not accumulate and not composite:
                # every row, keyed by an integer column, payload = integer columns as they are (dictionary codes included): the
                # fixed build keeps such columns in place (no staging; increasing keys: rank = row) — Q12's 15 M orders 0.74 -> 0.15 ms
                ops = c.P.ops
                cols = [ops[i]["col"] if ops[i]["code"] == abi.X_COL and ops[i]["type"] == abi.T_I64 else None for i in [c.P.key] + list(c.P.vals)]
                if all(col is not None for col in cols):
                    table = ctx.hash_build_unique(htab.nrows, abi.make_filter(), [], cols[0], cols[1:])
            if table is None:
                table = ctx.xbuild(htab.nrows, c.P, bounds[0], bounds[1], accumulate=accumulate, nsums=accumulate_into.get(op.out) if accumulate and isinstance(accumulate_into, dict) else None)
            bt = BuiltTable(table, key_names[0], key_is_record, val_fields, val_is_record, dtypes)
            bt.decoders, bt.field_decoders, bt.slot_rng = dict(decoders[0]), dict(decoders[1]), dict(decoders[2])
            if state.get("share_facts") is not None and state["share_facts"][0] is not None:
                bt.slot_roots, bt.slot_plain = dict(state["share_facts"][0]), dict(state["share_facts"][1])
            if composite:
                bt.key_parts = key_names
                bt.key_part_decoders = key_dec
                bt.key_bounds = bounds                      # of the PACKED key, from the parts' ranges (DictTable: a negative packed key cannot be unpacked by division)
            else:
                bt.key_decoder = key_dec
            return bt
        return run_build

    # ---- aggregations --------------------------------------------------------------------------------------
    vnames, vexprs, val_is_record = value_fields(op.val)

    def compile_entry(env):
        """the group is the entry matched by the joinProbe index: summed into the entry itself"""
        c = Compiler(eng, op, htab, env)
        gates, probe_id = gates_of(c, op.conds)
        vals, count_idx = summed_values(c, vexprs)
        c.P.gates, c.P.vals = gates, vals
        return c, probe_id, count_idx

    def compile_groups(env):
        """a small group domain: one packed key, LDS group table"""
        c = Compiler(eng, op, htab, env)
        gates, _ = gates_of(c, op.conds)
        parts = c.key_parts(key_fields)
        kid, radix = c.pack_radix(parts)
        vals, count_idx = summed_values(c, vexprs)
        c.P.gates, c.P.key, c.P.vals = gates, kid, vals
        state["encoding_fp"] = _encoding_fingerprint(parts, radix)
        return c, parts, radix, count_idx

    def compile_large(env):
        """any number of groups: a unique build of the keys with accumulators, then the same rows summed into
        it.  Two programs (the second looks the first one's table up): a program never names a table that
        is not alive when it runs."""
        def half(with_values):
            c = Compiler(eng, op, htab, env)
            gates, _ = gates_of(c, op.conds)
            parts = c.key_parts(key_fields)
            flat = [v for _, vs, _ in parts for v in vs]
            bounds = (1, 0)
            if len(flat) > 2:
                # more than two parts: one mixed-radix integer (the parts' value ranges are known), decoded when the
                # result is read; such a dictionary cannot be looked up by later loops (Q16's distinct combinations)
                kid, radix = c.pack_radix(parts)
                total = 1
                for _, span in radix:
                    total *= span
                if total <= (1 << 31):
                    bounds = (0, total - 1)
                state["radix"] = ([(nm or "key%d" % i, [None] * len(vs), kind) for i, (nm, vs, kind) in enumerate(parts)], radix)
                c.P.gates = gates
                vals, count_idx = summed_values(c, vexprs) if with_values else ([], None)
                return c, kid, vals, count_idx, bounds, [XV(kid, "i")]
            if len(flat) == 1:
                kid = flat[0].id
                r = c.resolve_rng(flat[0])
                if r is not None:
                    bounds = r
            else:
                kid = c.P.op(abi.X_PACK2, abi.T_I64, a=flat[0].id, b=flat[1].id)
                # the packed key's bounds from its parts' ranges (a later loop over the entries unpacks the key by signed division: DictTable)
                r0, r1 = c.resolve_rng(flat[0]), c.resolve_rng(flat[1])
                if r0 is not None and r1 is not None and 0 <= r0[0] <= r0[1] < (1 << 31) and 0 <= r1[0] <= r1[1] < (1 << 32):
                    bounds = ((r0[0] << 32) | r1[0], (r0[1] << 32) | r1[1])
            c.P.gates = gates
            vals, count_idx = summed_values(c, vexprs) if with_values else ([], None)
            return c, kid, vals, count_idx, bounds, flat
        cb, kid, _, _, bounds, flat = half(False)
        cb.P.key = kid
        state["part_decs"] = [v.dec for v in flat] if len(flat) == 2 else None
        cp, pkid, vals, count_idx, _, _ = half(True)
        cp.P.vals = vals
        key_names = [nm or (e.name if isinstance(e, Col) else "key%d" % i) for i, (nm, e) in enumerate(key_fields)]
        return cb, cp, pkid, count_idx, bounds, key_names, (flat[0].dec if len(flat) == 1 else None), len(flat) == 2

    def groups_result(c, parts, radix, count_idx, keys, vals, cnts):
        """The groups of a finished xgroupby as a DictResult (decoded key fields, value columns)."""
        # the same handful of group keys comes back run after run: their decoded fields are kept (a dozen small numpy calls otherwise)
        kcache = state.get("kf_cache")
        kbytes = keys.tobytes()
        if kcache is not None and kcache[0] is c and kcache[1] == kbytes:
            kf = [(nm, a.copy() if isinstance(a, np.ndarray) else a) for nm, a in kcache[2]]
        else:
            kf = _decode_radix(keys, [(nm or "key%d" % i, vs, kind) for i, (nm, vs, kind) in enumerate(parts)], radix)
            state["kf_cache"] = (c, kbytes, [(nm, a.copy() if isinstance(a, np.ndarray) else a) for nm, a in kf])
        vf, at = [], 0
        for i, nm in enumerate(vnames):
            if count_idx is not None and i == count_idx:
                vf.append((nm, np.asarray(cnts, np.int64)))
            else:
                col = np.ascontiguousarray(vals[:, at]); at += 1
                vf.append((nm, np.rint(col).astype(np.int64) if nm in state.get("int_values", ()) else col))
        d = DictResult(kf, vf, key_is_record, val_is_record)
        if any(kind[0] == "int" and kind[1] is not None for _, _, kind in parts):
            from .engine import _merge_equal_keys
            d = _merge_equal_keys(d)                          # two references may decode to the same text
        d.encoding_fp = state.get("encoding_fp")              # (the multi-GPU runner: may the ranks' partial groups be folded by packed key?)
        return d

    def dense_domain(st):
        """(resident column lo..hi, span) when the group key is a single integer over a range much smaller than the row count (Q13: 1.5 M
        customer keys of 15 M orders), else False.  Every key of the range then gets an entry up front (a build over lo..hi: increasing
        keys, rank = row, microseconds) and the rows are summed into it in ONE pass — the conditions are evaluated once, not once to
        find the keys and once to sum.  Keys no row carries keep hits = 0 and are not in the dictionary (K-F's min_hits, Compiler.found)."""
        bounds, composite = st[4], st[7]
        span = bounds[1] - bounds[0] + 1
        if composite or state.get("radix") is not None or not (1 <= span <= max(1 << 16, htab.nrows // 4)):
            return False
        return (eng.iota_column(bounds[0], span), span)

    def run_aggregate(env):
        mode = state.get("mode")
        if mode is None:
            bt = env.get(op.probe.dict_name) if op.probe is not None else None
            entry_ok = groups_by_entry(op) and isinstance(bt, BuiltTable) and bt.table.accumulate and bt.agg is None and not _is_key_set(bt)
            state["entry_ok"] = entry_ok
            if entry_ok and any(isinstance(e, Col) for _, e in key_fields):
                mode = "entry"                                     # the probe key is part of the group: the group IS the matched entry
            elif as_table:
                mode = "large"                                     # a dictionary that later loops look up must be a table
            else:
                mode = "groups"                                    # fields of the entry only: usually a handful of groups (else see below)
            state["mode"] = mode
        if mode == "entry":
            bt = env[op.probe.dict_name]
            st = state.get("c")
            if st is None or not st[0].still_valid(env):
                st = state["c"] = compile_entry(env)
            c, probe_id, count_idx = st
            if bt.agg_spec is None:
                pk = op.probe.key.name
                spec = []
                for fname, e in key_fields:
                    if isinstance(e, Col):
                        spec.append((fname or pk, "key"))
                    else:
                        src = bt.slot_of(e.field)
                        if src is None:
                            raise UnsupportedQuery("line %d: '%s' has no field '%s'" % (op.lineno, op.probe.dict_name, e.field))
                        spec.append((fname or e.field, src))
                        bt.agg_fields[fname or e.field] = e.field
                bt.agg_spec = spec
            c.bind(env)
            ctx.xprobe_aggregate(htab.nrows, c.P, probe_id, bt.table)
            bt.agg = (bt.agg_spec, vnames, count_idx, key_is_record, val_is_record, len(c.P.vals))
            bt.int_values = frozenset(state.get("int_values", ()))
            return ("aggregated", op.probe.dict_name)
        if mode == "groups":
            st = state.get("c")
            try:
                if st is None or not st[0].still_valid(env):
                    st = state["c"] = compile_groups(env)
                c, parts, radix, count_idx = st
                c.bind(env)
                fold = env.get("__group_fold__")
                if fold is not None and not isinstance(htab, DictTable):
                    # a row shard of a multi-GPU run: this rank's partial groups into the collective's buffer, the ranks' blocks folded
                    # on the device behind ONE all-gather (dist.DistributedRunner gives the exchange, for the plan's last loop and where
                    # it has checked that the packed keys mean the same on every rank: the digest is the one it compared).  No exchange:
                    # the loop is waited for and its groups merged on the host by their decoded values.
                    exchange = fold(op.out, state.get("encoding_fp"))
                    if exchange is not None:
                        collect = ctx.xgroupby_folded(htab.nrows, c.P, exchange)
                        return Pending(lambda: groups_result(c, parts, radix, count_idx, *collect()))
                elif op.out in env.get("__defer__", ()) and not isinstance(htab, DictTable):
                    # the plan's last device call: launched, not waited for (engine.PreparedPlan.run finishes the plan when the
                    # result is first looked at; what the data decides — too many groups — is raised there and the plan re-run)
                    collect = ctx.xgroupby_async(htab.nrows, c.P)
                    return Pending(lambda: groups_result(c, parts, radix, count_idx, *collect()))
                return groups_result(c, parts, radix, count_idx, *ctx.xgroupby(htab.nrows, c.P))
            except abi.SdqhError as exc:
                if exc.code != abi.ERR_OVERFLOW:
                    raise
                if small_groups_only and not state.get("entry_ok"):
                    # the caller has a plan of its own for many groups — unless they are a dense integer domain, which is ONE pass here
                    # (below) against its two (Q15: 100 K supplier keys of 2.3 M rows)
                    large = compile_large(env)
                    if not dense_domain(large):
                        raise UnsupportedQuery("line %d: more groups than the group-by sinks of a row program hold" % op.lineno)
                    state["mode"], state["c"], state["look"] = "large", large, None
                    return run_aggregate(env)
                if small_groups_only:
                    raise UnsupportedQuery("line %d: more groups than the group-by sinks of a row program hold" % op.lineno)
                state["mode"], state["c"] = ("entry" if state.get("entry_ok") else "large"), None      # more groups than the LDS table holds
                return run_aggregate(env)
            except UnsupportedQuery:
                if state.get("c") is not None or small_groups_only:
                    raise
                state["mode"] = "entry" if state.get("entry_ok") else "large"
                return run_aggregate(env)
        st = state.get("c")
        if st is None or not (st[0].still_valid(env) and st[1].still_valid(env)):
            st = state["c"] = compile_large(env)
            state["look"] = None
        cb, cp, pkid, count_idx, bounds, key_names, key_dec, composite = st
        cb.bind(env); cp.bind(env)
        dense = state.get("dense")
        if dense is None:
            dense = state["dense"] = dense_domain(st)
        if dense:
            table = ctx.hash_build_unique(dense[1], abi.make_filter(), [], dense[0], [], accumulate=True)
        else:
            table = ctx.xbuild(htab.nrows, cb.P, bounds[0], bounds[1], accumulate=True)
        look = state.get("look")
        if look is None:
            look = state["look"] = cp.P.op(abi.X_LOOKUP, abi.T_BOOL, a=pkid, table=table)
            cp.P.gates = list(cp.P.gates) + [look]
        else:
            cp.P.bind_table(look, table)
        ctx.xprobe_aggregate(htab.nrows, cp.P, look, table)
        vals = cp.P.vals
        bt = BuiltTable(table, key_names[0], key_is_record, [], val_is_record, [])
        if composite:
            bt.key_parts = key_names
            bt.key_part_decoders = state.get("part_decs")
            bt.key_bounds = bounds
        bt.key_decoder = key_dec
        bt.key_radix = state.get("radix")
        bt.agg = ([(key_names[0], "key")], vnames, count_idx, key_is_record, val_is_record, len(vals))
        bt.int_values = frozenset(state.get("int_values", ()))
        hidden = op.out + "$groups"
        env[hidden] = bt
        return ("aggregated", hidden)
    return run_aggregate


def _is_key_set(bt):
    return bt.table.npayload == 0 and not bt.table.accumulate


# =================================================================================================
# Sums over RESULT dictionaries (frontend.HostDictOp) on the device.  The reference compiles such a loop like any
# other (lib/sdql_ir_cpp_generator_par.py:520-568: iteration over a dictionary that is not a database table); here the
# entries of the source dictionary become resident columns (sdqh_table_columns: the table's own K-F buffers, build-row
# order) and the loop body a row program over them — p[0] / p[1] read those columns, a key of several packed fields is
# unpacked with DIVI / MODI.  What this cannot express falls back to run_host_dict below.
# =================================================================================================
PACKED_KEY = "\0packed"     # WholeKey(0, PACKED_KEY): the stored key of a composite-keyed dictionary as one integer (prepare_dict_scan)


class DictTable:
    """The entries of a device-resident dictionary, presented as the table a loop scans."""

    def __init__(self, eng, op):
        self.eng, self.op = eng, op
        self.nrows, self.cols = 0, {}            # (no host columns: prepare_scan looks text payload columns up here)
        self.bt, self.columns = None, None

    def array(self, name, op):
        raise UnsupportedQuery("line %d: a sum over a dictionary has no column '%s'" % (op.lineno, name))

    def source(self, env):
        src = env[self.op.source]
        if isinstance(src, tuple) and src and src[0] == "aggregated":
            bt = env[src[1]]
            if not (any(s == "key" for _, s in bt.agg[0]) or bt.shared_groups):
                raise UnsupportedQuery("line %d: the entries of '%s' are not its groups yet" % (self.op.lineno, self.op.source))
            return bt, 1
        if hasattr(src, "table") and hasattr(src, "val_fields"):
            return src, 0
        raise UnsupportedQuery("line %d: '%s' is not resident on the device" % (self.op.lineno, self.op.source))

    def load(self, env):
        """The entries as columns; every run (the dictionary was rebuilt)."""
        bt, min_hits = self.source(env)
        key, pays, accs, hits, n = self.eng.ctx.table_columns(bt.table, min_hits)
        cols = {"key": key, "hits": hits}
        cols.update({("pay", i): c for i, c in enumerate(pays)})
        cols.update({("acc", i): c for i, c in enumerate(accs)})
        self.bt, self.columns, self.nrows = bt, cols, n

    def release(self):
        self.columns = None                      # (views of the table's buffers: nothing to free but the handles)

    # -- p[0] / p[1] ----------------------------------------------------------------------------------
    def col(self, c, which, typ):
        key = ("entrycol", which)
        if key not in c.memo:
            c.memo[key] = c.P.op(abi.X_COL, typ, col=self.columns[which])
            c.entry_cols.append((c.memo[key], which))
        return c.memo[key]

    def fields(self, c):
        """{side: [(field name, XV)]} of the entries, compiled on first use per program."""
        if "entryfields" in c.memo:
            return c.memo["entryfields"]
        bt, op = self.bt, self.op
        name = op.source
        c.layout.append((name, _table_signature(bt)))

        def text_ok(dec):
            # two entries must never stand for the same key: references into a raw text column may (equal texts in
            # different rows), codes of a dictionary with distinct entries cannot
            if dec is not None and not isinstance(dec, Dictionary):
                raise UnsupportedQuery("line %d: '%s' is keyed by row references into a text column" % (op.lineno, name))
            return dec

        def payload(fname, slot):
            if np.dtype(bt.payload_dtypes[slot]).kind == "f":
                raise UnsupportedQuery("line %d: a floating-point entry field of '%s'" % (op.lineno, name))
            dec = bt.decoder_of(fname, slot)
            rng = (0, len(dec) - 1) if dec is not None else _slot_range(bt, slot)
            return XV(self.col(c, ("pay", slot), abi.T_I64), "i", dec=dec, rng=rng)

        keyv = XV(self.col(c, "key", abi.T_I64), "i")
        c.memo["entrykey"] = XV(keyv.id, "i")                    # (the stored key as it is, whatever it packs)
        kf, vf = [], []
        if bt.agg is not None:
            spec, vnames, count_idx, _, _, nv = bt.agg
        else:
            spec, vnames, count_idx, nv = [(bt.key_name, "key")], [], None, 0
        if getattr(bt, "key_radix", None) is not None and spec == [(bt.key_name, "key")]:
            parts, radix = bt.key_radix                          # one mixed-radix integer: field i = (key / stride_i) % span_i + lo_i
            strides, stride = [], 1
            for lo, span in reversed(radix):
                strides.append(stride)
                stride *= span if span is not None else 1
            strides.reverse()
            at = 0
            for pname, vs, kind in parts:
                if kind[0] != "int" or len(vs) != 1:
                    raise UnsupportedQuery("line %d: '%s' has a substring in its key" % (op.lineno, name))
                (lo, span), st = radix[at], strides[at]
                at += 1
                vid = keyv.id
                if st > 1:
                    vid = c.P.op(abi.X_DIVI, abi.T_I64, a=vid, imm_i=st)
                if span is not None and at > 1:                  # (the leading field needs no remainder)
                    vid = c.P.op(abi.X_MODI, abi.T_I64, a=vid, imm_i=span)
                if lo != 0:
                    vid = c.P.op(abi.X_ADD, abi.T_I64, a=vid, b=c.const(lo).id)
                kf.append((pname, XV(vid, "i", dec=text_ok(kind[1]), rng=None if span is None else (lo, lo + span - 1))))
        elif bt.key_parts is not None and spec == [(bt.key_name, "key")]:
            # (hi << 32) | lo is unpacked by SIGNED division / remainder (SDQH_X_DIVI / _MODI): exact only while the packed key is not
            # negative, i.e. the high part is below 2^31 — known from the build's own bounds, else the host path (shifts and masks) runs
            kb = getattr(bt, "key_bounds", None)
            if kb is None or kb[0] > kb[1] or kb[0] < 0:
                raise UnsupportedQuery("line %d: the parts of '%s's packed key are not known to stay below 2^31" % (op.lineno, op.source))
            decs = getattr(bt, "key_part_decoders", None) or [None, None]
            hi = c.P.op(abi.X_DIVI, abi.T_I64, a=keyv.id, imm_i=1 << 32)
            lo = c.P.op(abi.X_MODI, abi.T_I64, a=keyv.id, imm_i=1 << 32)
            kf = [(bt.key_parts[0], XV(hi, "i", dec=text_ok(decs[0]), rng=(0, 0xFFFFFFFF))), (bt.key_parts[1], XV(lo, "i", dec=text_ok(decs[1]), rng=(0, 0xFFFFFFFF)))]
        else:
            for fname, src in spec:
                if src == "key":
                    keyv.dec = text_ok(bt.key_decoder) if bt.agg is None else None
                    kf.append((fname, keyv))
                else:
                    v = payload(bt.agg_fields.get(fname, fname) if bt.agg is not None else fname, src)
                    text_ok(v.dec)
                    kf.append((fname, v))
        if bt.agg is not None:
            for i, vname in enumerate(vnames):
                if count_idx is not None and i == count_idx:
                    vf.append((vname, XV(self.col(c, "hits", abi.T_I64), "i")))
                else:
                    k = i - (1 if count_idx is not None and count_idx < i else 0)
                    vf.append((vname, XV(self.col(c, ("acc", k), abi.T_F64), "f")))
        else:
            for fname, src in bt.val_fields:
                vf.append((fname, keyv if src == "key" else payload(fname, src)))
        c.memo["entryfields"] = {0: kf, 1: vf}
        return c.memo["entryfields"]

    def whole(self, c, e):
        side = self.fields(c)[e.which]
        if e.which == 0 and e.field == PACKED_KEY:
            return c.memo["entrykey"]
        if e.field is None:
            if len(side) != 1:
                raise UnsupportedQuery("line %d: p[%d] is a record; name a field" % (self.op.lineno, e.which))
            return side[0][1]
        for nm, v in side:
            if nm == e.field:
                return v
        raise UnsupportedQuery("line %d: p[%d] has no field '%s'" % (self.op.lineno, e.which, e.field))


def prepare_dict_scan(eng, op, as_table=False, is_result=False):
    """closure(env) for a HostDictOp as a device loop, or None when its shape is not one a loop has (the closure can
    still raise UnsupportedQuery on its first run, when the source's layout is known: the caller keeps the host path).
    Two shapes: a group-by over the entries ({key: value}, summed); and — as the plan's result only — a set of records
    {unique(record(...)): True} that names the source's own key among its fields: a unique build keyed by the source's key (a
    single field, or the two packed parts of a composite key as they are: Q2's (part, supplier) offers) with the other fields as
    payload, K-F'd (with ORDER BY / LIMIT on the device) by the caller; the BuiltTable carries .record_order."""
    from .frontend import ScanOp
    dtab = DictTable(eng, op)
    record_set = op.unique or isinstance(op.val, Const) and op.val.value is True
    if record_set:
        if not (is_result and isinstance(op.val, Const) and op.val.value is True and isinstance(op.key, RecordCons)):
            return None
        own = [(nm, e) for nm, e in op.key.fields if isinstance(e, WholeKey) and e.which == 0]
        if not own or len(own) == len(op.key.fields):
            return None
        record_order = [nm for nm, _ in op.key.fields]

        def make_run():
            """Known with the source's layout: which of the record's fields ARE the source's key."""
            bt = dtab.bt
            spec = bt.agg[0] if bt.agg is not None else [(bt.key_name, "key")]
            scan = ScanOp(op.out, op.source, op.lineno)
            if getattr(bt, "key_radix", None) is None and bt.key_parts is not None and spec == [(bt.key_name, "key")]:
                # a composite key: the packed integer itself keys the build; its two parts come back out of it when the result is read
                if any(e.field not in bt.key_parts for _, e in own):
                    raise UnsupportedQuery("line %d: p[0] names a field that is not part of the key of '%s'" % (op.lineno, op.source))
                if any(nm != e.field for nm, e in own):
                    raise UnsupportedQuery("line %d: a key part of '%s' under another name" % (op.lineno, op.source))
                key, packed = RecordCons([(bt.key_parts[0], WholeKey(0, PACKED_KEY))]), True
                rest = [(nm, e) for nm, e in op.key.fields if not (isinstance(e, WholeKey) and e.which == 0)]
            else:
                if getattr(bt, "key_radix", None) is not None or bt.key_parts is not None or len(spec) != 1 or len(own) != 1:
                    raise UnsupportedQuery("line %d: '%s' is keyed by several fields" % (op.lineno, op.source))
                key, packed = RecordCons([own[0]]), False
                rest = [(nm, e) for nm, e in op.key.fields if nm != own[0][0]]
            scan.kind, scan.conds, scan.key, scan.val, scan.unique = "dict", list(op.conds), key, RecordCons(rest), True
            return prepare_scan(eng, scan, dtab, {}, False, as_table), packed
    else:
        scan = ScanOp(op.out, op.source, op.lineno)
        scan.kind, scan.conds, scan.key, scan.val, scan.unique = "dict", list(op.conds), op.key, op.val, False
        run_groups = prepare_scan(eng, scan, dtab, {}, False, as_table)
    state = {}

    def run_dict_scan(env):
        dtab.load(env)
        try:
            if dtab.nrows == 0:
                return NotImplemented                            # nothing to launch: the host path shapes the empty result of this run
            if not record_set:
                return run_groups(env)
            if "run" not in state:
                state["run"] = make_run()
            run, packed = state["run"]
            out = run(env)
            if packed:
                if dtab.bt.key_parts is None:
                    raise UnsupportedQuery("line %d: the key of '%s' changed shape" % (op.lineno, op.source))
                out.key_parts, out.key_part_decoders = list(dtab.bt.key_parts), getattr(dtab.bt, "key_part_decoders", None)
            out.record_order = record_order
            return out
        finally:
            dtab.release()
    return run_dict_scan


# =================================================================================================
# Sums over RESULT dictionaries that are more than a reshape (frontend.HostDictOp): O(groups), on the host.
# =================================================================================================
def run_host_dict(eng, op, env, materialize):
    """materialize(value) -> DictResult for a BuiltTable / aggregated dictionary / DictResult."""
    src = materialize(env[op.source])
    n = src.size()
    tables = {}

    def host_table(name):
        if name not in tables:
            d = materialize(env[name])
            if len(d.key_fields) != 1:
                raise UnsupportedQuery("line %d: '%s' has a composite key: it cannot be looked up from a result dictionary" % (op.lineno, name))
            keys = np.asarray(d.key_fields[0][1])
            direct = None
            if len(keys) and keys.dtype.kind == "i":
                lo, hi = int(keys.min()), int(keys.max())
                if hi - lo < 4 * len(keys) + 1024:                       # integer keys over a dense range: row of key k at direct[k - lo]
                    direct = (lo, np.full(hi - lo + 1, -1, np.int64))
                    direct[1][keys[::-1] - lo] = np.arange(len(keys) - 1, -1, -1)      # equal keys: the first row wins, as the stable sort below
            order = None if direct is not None else np.argsort(keys, kind="stable")
            tables[name] = (keys if direct is not None else keys[order], order, d, direct)
        return tables[name]

    looked = {}

    def look(lk):
        memo = (lk.dict_name, repr(lk.key))                              # the fields of one looked-up record share the search
        if memo in looked:
            return looked[memo]
        skeys, order, d, direct = host_table(lk.dict_name)
        k = np.asarray(val(lk.key))
        if direct is not None:
            lo, rows_of = direct
            inside = (k >= lo) & (k < lo + len(rows_of))
            rows = rows_of[np.where(inside, k - lo, 0)]
            hit = inside & (rows >= 0)
            out = (hit, np.where(hit, rows, 0), d)
        else:
            pos = np.searchsorted(skeys, k)
            pos[pos >= len(skeys)] = 0
            hit = (skeys[pos] == k) if len(skeys) else np.zeros(n, bool)
            out = (hit, order[pos] if len(skeys) else np.zeros(n, np.int64), d)
        looked[memo] = out
        return out

    def val(e):
        if isinstance(e, Const):
            return e.value
        if isinstance(e, WholeKey):
            side = src.key_fields if e.which == 0 else src.val_fields
            if e.field is None:
                if len(side) != 1:
                    raise UnsupportedQuery("line %d: p[%d] is a record; name a field" % (op.lineno, e.which))
                return side[0][1] if isinstance(side[0][1], TextRefs) else np.asarray(side[0][1])
            hitf = [a for nm, a in side if nm == e.field]
            if not hitf:
                raise UnsupportedQuery("line %d: p[%d] has no field '%s'" % (op.lineno, e.which, e.field))
            return hitf[0] if isinstance(hitf[0], TextRefs) else np.asarray(hitf[0])      # text of a large result stays references (grouping works on them)
        if isinstance(e, ScalarField):
            v = env[e.name]
            return float(v if e.field is None else v[e.field])
        if isinstance(e, Bin):
            a, b = val(e.left), val(e.right)
            return {"+": lambda: a + b, "-": lambda: a - b, "*": lambda: a * b, "/": lambda: np.true_divide(a, b)}[e.op]()
        if isinstance(e, IfElse):
            return np.where(cond(e.cond), val(e.then), val(e.other))
        if isinstance(e, (PayloadField, Lookup)):
            lk, fname = (e.lookup, e.field) if isinstance(e, PayloadField) else (e, None)
            hit, rows, d = look(lk)
            fields = d.val_fields
            if fname is None:
                if len(fields) != 1:
                    raise UnsupportedQuery("line %d: the looked-up value is a record; name a field" % op.lineno)
                col = np.asarray(fields[0][1])
            else:
                got = [a for nm, a in fields if nm == fname] + [a for nm, a in d.key_fields if nm == fname]
                if not got:
                    raise UnsupportedQuery("line %d: '%s' has no field '%s'" % (op.lineno, lk.dict_name, fname))
                if isinstance(got[0], TextRefs) and hit.all():
                    return got[0][rows]                                   # references of references: still not decoded
                col = np.asarray(got[0])
            if len(col) and col.dtype.kind == "U" and hit.all():
                return decode_text(rows, col)                            # text of a large result: the row references now, the (wide) strings when read
            out = col[rows] if len(col) else np.zeros(n, col.dtype)
            if not hit.all():
                out = out.copy()
                out[~hit] = "" if out.dtype.kind == "U" else 0
            return out
        if isinstance(e, Call) and e.fn == "extractYear":
            return np.asarray(val(e.args[0])) // 10000
        if isinstance(e, (Cmp, And, Or, Not, Contains)):
            return cond(e)
        raise UnsupportedQuery("line %d: unsupported expression over a result dictionary: %r" % (op.lineno, e))

    def cond(e):
        if isinstance(e, Const):
            return np.full(n, bool(e.value))
        if isinstance(e, And):
            out = np.ones(n, bool)
            for t in e.terms:
                out &= cond(t)
            return out
        if isinstance(e, Or):
            out = np.zeros(n, bool)
            for t in e.terms:
                out |= cond(t)
            return out
        if isinstance(e, Not):
            return ~cond(e.term)
        if isinstance(e, Contains):
            return look(e.lookup)[0]
        if isinstance(e, Cmp):
            a, b = val(e.left), val(e.right)
            return {"<": lambda: a < b, "<=": lambda: a <= b, ">": lambda: a > b, ">=": lambda: a >= b, "==": lambda: a == b, "!=": lambda: a != b}[e.op]()
        raise UnsupportedQuery("line %d: unsupported condition over a result dictionary: %r" % (op.lineno, e))

    keep = np.ones(n, bool)
    for c in op.conds:
        keep &= np.asarray(cond(c), bool)

    def arr(v):
        return v if isinstance(v, TextRefs) else np.asarray(v)

    def fields_of(e, default):
        if isinstance(e, RecordCons):
            return [(nm, arr(val(x))) for nm, x in e.fields], True
        return [(default, arr(val(e)))], False

    def broadcast(a):
        return a if isinstance(a, TextRefs) else (np.full(n, a) if np.ndim(a) == 0 else a)
    kf, key_is_record = fields_of(op.key, "key")
    every = bool(keep.all())                                         # nothing filtered: the columns as they are
    kf = [(nm, broadcast(a) if every else broadcast(a)[keep]) for nm, a in kf]
    if isinstance(op.val, Const) and op.val.value is True:
        return ResultSet([nm for nm, _ in kf], [a for _, a in kf])
    vf, val_is_record = fields_of(op.val, "value")
    vf = [(nm, broadcast(a) if every else broadcast(a)[keep]) for nm, a in vf]
    d = DictResult(kf, vf, key_is_record, val_is_record)
    if not op.unique:                                                # a group-by over the dictionary (Q16: combinations -> their number per group)
        from .engine import _merge_equal_keys
        d = _merge_equal_keys(d)
    return d
