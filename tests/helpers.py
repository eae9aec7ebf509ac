"""Shared helpers for the parity tests: golden decoding, input regeneration, result comparison."""
import importlib.util
import math
import os

import numpy as np

from sdqlpy_amd import engine as eng_mod
from sdqlpy_amd import frontend, tpch
from sdqlpy_amd import tpch_queries as Q
from sdqlpy_amd.result import ResultSet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)

_db_cache = {}


class spy_calls:
    """Record the calls of one abi.Context method while a block runs — on the class, so that the calls of every lane of an engine
    (contexts of one family, engine.Engine.lane) are seen.  `calls` gets one note(*args, **kwargs) entry per call."""

    def __init__(self, method, note=lambda *a, **k: 1):
        from sdqlpy_amd import abi
        self.cls, self.method, self.note, self.calls = abi.Context, method, note, []

    def __enter__(self):
        self.real = real = getattr(self.cls, self.method)
        calls, note = self.calls, self.note

        def wrapped(ctx, *a, **k):
            calls.append(note(*a, **k))
            return real(ctx, *a, **k)
        setattr(self.cls, self.method, wrapped)
        return self.calls

    def __exit__(self, *exc):
        setattr(self.cls, self.method, self.real)
        return False


def case_db(case):
    """Regenerate the inputs of a golden case and check they are the ones the reference saw."""
    key = (case["name"], tuple(sorted(case["results"])))
    if key not in _db_cache:
        qs = list(case["results"].keys())
        base = tpch.generate(case["sf"], case["seed"], tables=case["tables"], columns=tpch.columns_for(qs), threads=4)
        db = make_golden.VARIANTS[case["variant"]](base)
        assert tpch.fingerprint(db) == case["fingerprint"], "generator drifted: golden inputs no longer reproducible"
        _db_cache[key] = db
    return _db_cache[key]


def dec(v):
    return float.fromhex(v["f"]) if isinstance(v, dict) else v


def golden_rows(result):
    return [tuple(dec(x) for x in row) for row in result["rows"]]


def run_query(ctx_engine, name, db):
    plan = frontend.lower_function(Q.QUERIES[name])
    return eng_mod.execute_plan(ctx_engine, plan, [db[t] for t in Q.QUERY_TABLES[name]])


def result_rows(res, columns):
    """Rows of a ResultSet in the golden column order, sorted like the golden file."""
    assert isinstance(res, ResultSet)
    assert sorted(res.columns) == sorted(columns), (res.columns, columns)
    cols = [res.column(c).tolist() for c in columns]
    rows = [tuple(r) for r in zip(*cols)]
    return sorted(rows, key=lambda r: [(0, x) if isinstance(x, int) else (1, x) if isinstance(x, str) else (2, x) for x in r])


def assert_rows_match(got, want, rel=0.0, what=""):
    """ints / strings exact; doubles exact when rel == 0 else within rel (relative)."""
    assert len(got) == len(want), "%s: %d rows, expected %d" % (what, len(got), len(want))
    if rel > 0:   # float columns may perturb the sort order of equal-key rows: align on the non-float fields
        key = lambda r: tuple(x for x in r if not isinstance(x, float))
        got, want = sorted(got, key=key), sorted(want, key=key)
    for g, w in zip(got, want):
        assert len(g) == len(w)
        for a, b in zip(g, w):
            if isinstance(b, float):
                assert isinstance(a, float), (what, g, w)
                if rel == 0.0:
                    assert a == b or (math.isnan(a) and math.isnan(b)), "%s: %r != %r in %r vs %r" % (what, a.hex(), b.hex(), g, w)
                else:
                    assert abs(a - b) <= rel * max(abs(a), abs(b), 1e-300), "%s: %r vs %r" % (what, a, b)
            else:
                assert a == b, "%s: %r != %r in %r vs %r" % (what, a, b, g, w)


def check_against_golden(res, gold, rel, what):
    if gold["kind"] == "scalar":
        want = dec(gold["value"])
        got = float(res)
        if rel == 0.0:
            assert got == want, "%s: %s != %s" % (what, got.hex(), want.hex())
        else:
            assert abs(got - want) <= rel * max(abs(want), 1e-300), "%s: %r vs %r" % (what, got, want)
        return
    want = golden_rows(gold)
    if not gold["columns"]:
        # the reference's interpreter raises on an empty aggregate (recorded as an empty set); a scalar
        # expression over empty sums is 0/0 = NaN here, as in the reference's compiled mode
        assert res is None or (isinstance(res, float) and math.isnan(res)) or (not isinstance(res, float) and res.size() == 0), \
            "%s: expected an empty result" % what
        return
    assert_rows_match(result_rows(res, gold["columns"]), want, rel, what)


def check_wide_goldens(ctx_engine, golden_wide, rel, what):
    """q7, q8, q13, q15, q17, q19, q20, q22 (row programs) against the reference's results.  q15: the
    reference's formulation keeps the supplier whose revenue EQUALS the maximum; this package's returns
    every supplier's revenue and TPCH's answer is its top-1 (an equality test on a floating-point sum is
    not reproducible under any other summation order)."""
    n = 0
    for case in golden_wide["cases"]:
        db = case_db(case)
        for q, want in case["results"].items():
            res = run_query(ctx_engine, q, db)
            if q == "q15":
                if not want["rows"]:
                    assert res is None or res.size() == 0
                    continue
                res = res.top(1, [("total_revenue", "desc")])
            check_against_golden(res, want, rel, "%s/%s/%s" % (what, case["name"], q))
            n += 1
    return n


def check_all_goldens(ctx_engine, goldens, rel, rel_q10, what):
    """EVERY golden vector on the planner's default routes."""
    n = 0
    for gold in goldens:
        for case in gold["cases"]:
            db = case_db(case)
            for q, want in case["results"].items():
                res = run_query(ctx_engine, q, db)
                if q == "q15" and want["rows"]:
                    res = res.top(1, [("total_revenue", "desc")])
                check_against_golden(res, want, rel_q10 if q == "q10" else rel, "%s/%s/%s" % (what, case["name"], q))
                n += 1
    return n


def check_all_goldens_as_programs(ctx_engine, goldens, rel, rel_q10, what):
    """EVERY golden vector with every table loop forced through a row program (Engine.force_programs):
    the specialised kernels have to reproduce what the tuned fixed-shape kernels are tested for."""
    ctx_engine.force_programs = True
    try:
        n = 0
        for gold in goldens:
            for case in gold["cases"]:
                db = case_db(case)
                for q, want in case["results"].items():
                    res = run_query(ctx_engine, q, db)
                    if q == "q15" and want["rows"]:
                        res = res.top(1, [("total_revenue", "desc")])
                    check_against_golden(res, want, rel_q10 if q == "q10" else rel, "%s/programs/%s/%s" % (what, case["name"], q))
                    n += 1
        return n
    finally:
        ctx_engine.force_programs = False


def check_tbl_queries(ctx_engine, rel, rel_q10):
    """Load the committed text tables with this package's read_csv, run every recorded query on
    `ctx_engine`, compare with the reference's results on the same files."""
    import json
    from sdqlpy_amd import sdql_lib as L
    tbl = os.path.join(ROOT, "tests", "golden", "tbl")
    with open(os.path.join(ROOT, "tests", "golden", "tbl_query_golden.json")) as fh:
        gold = json.load(fh)
    db = {t: L.read_csv(os.path.join(tbl, t + ".tbl"), tpch.SCHEMAS[t], t) for t in gold["rows"]}
    for t, n in gold["rows"].items():
        assert len(db[t].getContainer()["data"][0]) == n, t
    assert len(gold["results"]) >= 8
    for q, want in gold["results"].items():
        check_against_golden(run_query(ctx_engine, q, db), want, rel_q10 if q == "q10" else rel, "tbl/" + q)


def hash_layout_case(ctx, nb, npr):
    """Unique build + probe-aggregate on keys spread over 2^44 (no bitmap / direct index applies: the
    open-addressing layout), checked against numpy."""
    from sdqlpy_amd import abi
    rng = np.random.default_rng(11)
    keys = (rng.permutation(nb).astype(np.int64) << 20) + (np.int64(1) << 40) + rng.integers(0, 1 << 20, nb)
    pay = rng.integers(0, 1 << 50, nb).astype(np.int64)
    sel = rng.random(nb) < 0.5                                       # build filter: pay parity stands in for a predicate
    pay = np.where(sel, pay | 1, pay & ~np.int64(1))
    pidx = rng.integers(0, nb, npr)
    pk = keys[pidx].copy()
    miss = rng.random(npr) < 0.3
    pk[miss] += 1 << 19                                              # not a build key (low 20 bits of a key are < 2^20; +2^19 may collide rarely: recomputed below)
    pv = rng.random(npr)
    ckeys, cpay, cpk, cpv = ctx.upload(keys), ctx.upload(pay), ctx.upload(pk), ctx.upload(pv)
    odd = ctx.upload((pay & 1).astype(np.int64))
    t = ctx.hash_build_unique(nb, abi.make_filter(ipreds=[(odd, 1, 1)]), [], ckeys, [cpay], accumulate=True)
    assert t.size() == int(sel.sum())
    ctx.hash_probe_aggregate(npr, abi.make_filter(), t, cpk, abi.make_tuple(abi.TUPLE_A, [cpv]))
    cnt = ctx.table_compact_count(t, 1)
    k, p, v, h = ctx.table_compact(t, 1, cnt)
    t.free()
    order = np.argsort(keys)
    skeys = keys[order]
    pos = np.searchsorted(skeys, pk)
    pos[pos >= nb] = 0
    hit = (skeys[pos] == pk) & sel[order][pos]
    owner = order[pos[hit]]
    want_hits = np.bincount(owner, minlength=nb)
    want_sum = np.bincount(owner, weights=pv[hit], minlength=nb)
    live = np.nonzero(want_hits)[0]
    assert cnt == len(live)
    got_order = np.argsort(k)
    want_order = live[np.argsort(keys[live])]
    assert (k[got_order] == keys[want_order]).all()
    assert (p[0][got_order] == pay[want_order]).all()
    assert (h[got_order] == want_hits[want_order]).all()
    np.testing.assert_allclose(v[0][got_order], want_sum[want_order], rtol=1e-10)
    for c in (ckeys, cpay, cpk, cpv, odd):
        c.free()


def xprogram_cases(ctx, n=20000, seed=3):
    """Row programs (ABI 4) through every sdqh_x* entry point against numpy: boolean structure (or /
    not / select), mixed int / float arithmetic, string operations, lookups with payload fields and
    accumulators, composite keys, bounded (direct index) and unbounded (open addressing) builds,
    key sets and anti-joins, probe-aggregate into the matched entry.  Returns the number of checks."""
    from sdqlpy_amd import abi as A
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1000, n).astype(np.int64)
    b = rng.integers(19920101, 19981231, n).astype(np.int64)
    f = np.round(rng.random(n) * 100.0, 2)
    g = np.round(rng.random(n), 2)
    words = np.array(["special requests", "no requests here", "special", "xx special yy requests", "", "requests special", "13-555", "31-777", "29-1"], "<U24")
    txt = words[rng.integers(0, len(words), n)]
    ca, cb, cf, cg, ct = ctx.upload(a), ctx.upload(b), ctx.upload(f), ctx.upload(g), ctx.upload(txt)
    checks = 0

    def near(x, y, what):
        assert abs(x - y) <= 1e-9 * max(abs(y), 1.0), (what, x, y)

    # 1. K-A, register path: (a < 10 or a >= 500) and not (b < 19950101); values f*(1-g) and select(a even, f, 0.0) + a as double
    P = A.Program()
    xa = P.op(A.X_COL, A.T_I64, col=ca); xb = P.op(A.X_COL, A.T_I64, col=cb); xf = P.op(A.X_COL, A.T_F64, col=cf); xg = P.op(A.X_COL, A.T_F64, col=cg)
    c10 = P.op(A.X_CONST, A.T_I64, imm_i=10); c500 = P.op(A.X_CONST, A.T_I64, imm_i=500); cd = P.op(A.X_CONST, A.T_I64, imm_i=19950101)
    lt = P.op(A.X_LT, A.T_BOOL, a=xa, b=c10); ge = P.op(A.X_GE, A.T_BOOL, a=xa, b=c500); o1 = P.op(A.X_OR, A.T_BOOL, a=lt, b=ge)
    early = P.op(A.X_LT, A.T_BOOL, a=xb, b=cd); late = P.op(A.X_NOT, A.T_BOOL, a=early)
    one = P.op(A.X_CONST, A.T_F64, imm_f=1.0); omg = P.op(A.X_SUB, A.T_F64, a=one, b=xg); v1 = P.op(A.X_MUL, A.T_F64, a=xf, b=omg)
    two = P.op(A.X_CONST, A.T_I64, imm_i=2); half = P.op(A.X_YEAR, A.T_I64, a=xa)          # a / 10000 == 0: exercises YEAR
    twice = P.op(A.X_MUL, A.T_I64, a=P.op(A.X_SUB, A.T_I64, a=xa, b=P.op(A.X_MUL, A.T_I64, a=P.op(A.X_ADD, A.T_I64, a=half, b=xa), b=two)), b=two)   # (a - 2a) * 2 = -2a
    neg = P.op(A.X_NEG, A.T_I64, a=twice)                                                  # 2a
    even_t = P.op(A.X_EQ, A.T_BOOL, a=P.op(A.X_SUB, A.T_I64, a=neg, b=P.op(A.X_MUL, A.T_I64, a=two, b=xa)), b=P.op(A.X_CONST, A.T_I64, imm_i=0))   # always true
    zero = P.op(A.X_CONST, A.T_F64, imm_f=0.0)
    big = P.op(A.X_GT, A.T_BOOL, a=xf, b=P.op(A.X_CONST, A.T_F64, imm_f=50.0))
    v2 = P.op(A.X_ADD, A.T_F64, a=P.op(A.X_SELECT, A.T_F64, a=P.op(A.X_AND, A.T_BOOL, a=big, b=even_t), b=xf, c=zero), b=P.op(A.X_I2F, A.T_F64, a=xa))
    P.gates = [o1, late]; P.vals = [v1, v2]
    m = ((a < 10) | (a >= 500)) & ~(b < 19950101)
    vals, cnt = ctx.xscan_sum(n, P)
    assert cnt == int(m.sum())
    near(vals[0], float((f[m] * (1.0 - g[m])).sum()), "x1 v1"); near(vals[1], float((np.where(f[m] > 50.0, f[m], 0.0) + a[m]).sum()), "x1 v2")
    checks += 3

    # 2. K-A with string operations (queue path): firstIndex / contains / prefix / char
    P = A.Program()
    i_sp = P.op(A.X_STRIDX, A.T_I64, col=ct, text="special"); i_rq = P.op(A.X_STRIDX, A.T_I64, col=ct, text="requests")
    has = P.op(A.X_NE, A.T_BOOL, a=i_sp, b=P.op(A.X_CONST, A.T_I64, imm_i=-1))
    after = P.op(A.X_GT, A.T_BOOL, a=i_rq, b=P.op(A.X_ADD, A.T_I64, a=i_sp, b=P.op(A.X_CONST, A.T_I64, imm_i=6)))
    both = P.op(A.X_AND, A.T_BOOL, a=has, b=after)
    p13 = P.op(A.X_STR, A.T_BOOL, col=ct, aux=A.STR_PREFIX, text="13"); p31 = P.op(A.X_STR, A.T_BOOL, col=ct, aux=A.STR_PREFIX, text="31")
    sfx = P.op(A.X_STR, A.T_BOOL, col=ct, aux=A.STR_SUFFIX, text="here"); eq = P.op(A.X_STR, A.T_BOOL, col=ct, aux=A.STR_EQ, text="special")
    anyp = P.op(A.X_OR, A.T_BOOL, a=P.op(A.X_OR, A.T_BOOL, a=both, b=p13), b=P.op(A.X_OR, A.T_BOOL, a=P.op(A.X_OR, A.T_BOOL, a=p31, b=sfx), b=eq))
    ch = P.op(A.X_CHAR, A.T_I64, col=ct, aux=1)
    P.gates = [anyp]; P.vals = [P.op(A.X_I2F, A.T_F64, a=ch), P.op(A.X_COL, A.T_F64, col=cf)]
    sp, rq = np.char.find(txt, "special"), np.char.find(txt, "requests")
    m = ((sp != -1) & (rq > sp + 6)) | np.char.startswith(txt, "13") | np.char.startswith(txt, "31") | np.char.endswith(txt, "here") | (txt == "special")
    second = np.array([ord(t[1]) if len(t) > 1 else 0 for t in txt.tolist()], np.float64)
    vals, cnt = ctx.xscan_sum(n, P)
    assert cnt == int(m.sum()) and cnt > 0
    near(vals[0], float(second[m].sum()), "x2 char"); near(vals[1], float(f[m].sum()), "x2 f")
    checks += 3

    # 3. K-C small, register path: key = (a % ... ) via year of b and a < 500
    P = A.Program()
    xa = P.op(A.X_COL, A.T_I64, col=ca); xb = P.op(A.X_COL, A.T_I64, col=cb); xf = P.op(A.X_COL, A.T_F64, col=cf)
    yr = P.op(A.X_SUB, A.T_I64, a=P.op(A.X_YEAR, A.T_I64, a=xb), b=P.op(A.X_CONST, A.T_I64, imm_i=1992))
    lowa = P.op(A.X_SELECT, A.T_I64, a=P.op(A.X_LT, A.T_BOOL, a=xa, b=P.op(A.X_CONST, A.T_I64, imm_i=500)), b=P.op(A.X_CONST, A.T_I64, imm_i=1), c=P.op(A.X_CONST, A.T_I64, imm_i=0))
    P.key = P.op(A.X_ADD, A.T_I64, a=P.op(A.X_MUL, A.T_I64, a=yr, b=P.op(A.X_CONST, A.T_I64, imm_i=2)), b=lowa)
    P.gates = [P.op(A.X_GE, A.T_BOOL, a=xf, b=P.op(A.X_CONST, A.T_F64, imm_f=5.0))]; P.vals = [xf, P.op(A.X_I2F, A.T_F64, a=xa)]
    keys, vals, cnts = ctx.xgroupby(n, P)
    m = f >= 5.0
    want_key = (b // 10000 - 1992) * 2 + (a < 500)
    for k, v, c in zip(keys.tolist(), vals, cnts.tolist()):
        sel = m & (want_key == k)
        assert c == int(sel.sum()) and c > 0
        near(v[0], float(f[sel].sum()), "x3 f"); near(v[1], float(a[sel].sum()), "x3 a")
    assert sorted(keys.tolist()) == sorted(np.unique(want_key[m]).tolist())
    checks += 1 + 3 * len(keys)

    # 4. K-B: bounded key (direct index) and composite key (open addressing), first row wins; lookups read fields
    bk = rng.permutation(4 * n)[:n].astype(np.int64) + 7          # unique keys in [7, 4n+7)
    bk[100:120] = bk[:20]                                          # duplicates: the first row owns the entry
    pay = rng.integers(0, 1 << 40, n).astype(np.int64)
    cbk, cpay = ctx.upload(bk), ctx.upload(pay)
    P = A.Program()
    k = P.op(A.X_COL, A.T_I64, col=cbk); pv = P.op(A.X_COL, A.T_I64, col=cpay); fv = P.op(A.X_COL, A.T_F64, col=cf)
    P.key = k; P.gates = [P.op(A.X_GE, A.T_BOOL, a=pv, b=P.op(A.X_CONST, A.T_I64, imm_i=1 << 38))]; P.vals = [pv, fv]
    t_direct = ctx.xbuild(n, P, 7, 4 * n + 7, accumulate=True)
    keep = pay >= (1 << 38)
    first = {}
    for i in np.nonzero(keep)[0].tolist():
        first.setdefault(int(bk[i]), i)
    assert t_direct.size() == len(first)
    P2 = A.Program()
    hi = P2.op(A.X_COL, A.T_I64, col=cbk); lo = P2.op(A.X_COL, A.T_I64, col=ca)
    P2.key = P2.op(A.X_PACK2, A.T_I64, a=hi, b=lo); P2.vals = [P2.op(A.X_COL, A.T_F64, col=cg)]
    t_hash = ctx.xbuild(n, P2)
    firstc = {}
    for i in range(n):
        firstc.setdefault((int(bk[i]), int(a[i])), i)
    assert t_hash.size() == len(firstc)
    probe = rng.integers(0, 4 * n + 20, n).astype(np.int64)
    probe[: n // 2] = bk[rng.integers(0, n, n // 2)]
    cpr = ctx.upload(probe)
    P3 = A.Program()
    pk = P3.op(A.X_COL, A.T_I64, col=cpr); lk = P3.op(A.X_LOOKUP, A.T_BOOL, a=pk, table=t_direct)
    ck = P3.op(A.X_PACK2, A.T_I64, a=P3.op(A.X_COL, A.T_I64, col=cbk), b=P3.op(A.X_COL, A.T_I64, col=ca)); lk2 = P3.op(A.X_LOOKUP, A.T_BOOL, a=ck, table=t_hash)
    P3.gates = [lk, lk2]
    P3.vals = [P3.op(A.X_FIELD, A.T_F64, a=lk, aux=1), P3.op(A.X_I2F, A.T_F64, a=P3.op(A.X_FIELD, A.T_I64, a=lk, aux=0)), P3.op(A.X_FIELD, A.T_F64, a=lk2, aux=0)]
    vals, cnt = ctx.xscan_sum(n, P3)
    hit = np.array([int(x) in first for x in probe.tolist()])
    src = np.array([first.get(int(x), 0) for x in probe.tolist()])
    srcc = np.array([firstc[(int(x), int(y))] for x, y in zip(bk.tolist(), a.tolist())])
    assert cnt == int(hit.sum()) and cnt > n // 8
    near(vals[0], float(f[src[hit]].sum()), "x4 field f"); near(vals[1], float(pay[src[hit]].astype(np.float64).sum()), "x4 field i"); near(vals[2], float(g[srcc[hit]].sum()), "x4 composite")
    checks += 6

    # 5. key set + anti-join: rows whose key is NOT in the set
    P = A.Program()
    k = P.op(A.X_COL, A.T_I64, col=cbk); P.key = k
    P.gates = [P.op(A.X_STR, A.T_BOOL, col=ct, aux=A.STR_CONTAINS, text="requests")]
    t_set = ctx.xkey_set(n, P, 0, 4 * n + 7)
    members = set(bk[np.char.find(txt, "requests") >= 0].tolist())
    assert t_set.size() == len(members)
    P = A.Program()
    lk = P.op(A.X_LOOKUP, A.T_BOOL, a=P.op(A.X_COL, A.T_I64, col=cpr), table=t_set)
    P.gates = [P.op(A.X_NOT, A.T_BOOL, a=lk)]; P.vals = [P.op(A.X_COL, A.T_F64, col=cf)]
    vals, cnt = ctx.xscan_sum(n, P)
    notin = np.array([int(x) not in members for x in probe.tolist()])
    assert cnt == int(notin.sum()); near(vals[0], float(f[notin].sum()), "x5 anti")
    checks += 3

    # 6. K-C into the matched entry, then its accumulators / row count read by a later loop
    P = A.Program()
    lk = P.op(A.X_LOOKUP, A.T_BOOL, a=P.op(A.X_COL, A.T_I64, col=cpr), table=t_direct)
    P.gates = [lk, P.op(A.X_LT, A.T_BOOL, a=P.op(A.X_COL, A.T_F64, col=cg), b=P.op(A.X_CONST, A.T_F64, imm_f=0.75))]
    P.vals = [P.op(A.X_COL, A.T_F64, col=cf), P.op(A.X_CONST, A.T_F64, imm_f=1.0)]
    ctx.xprobe_aggregate(n, P, lk, t_direct)
    sel = hit & (g < 0.75)
    acc_sum, acc_cnt = {}, {}
    for x, v in zip(probe[sel].tolist(), f[sel].tolist()):
        acc_sum[x] = acc_sum.get(x, 0.0) + v; acc_cnt[x] = acc_cnt.get(x, 0) + 1
    kcol, pcols, acols, hcol, nent = ctx.table_columns(t_direct, 1)
    assert nent == len(acc_sum)
    got_k, got_s, got_c, got_h = kcol.download(), acols[0].download(), acols[1].download(), hcol.download()
    for kk, ss, cc, hh in zip(got_k.tolist(), got_s.tolist(), got_c.tolist(), got_h.tolist()):
        near(ss, acc_sum[kk], "x6 sum"); assert cc == hh == acc_cnt[kk]
    P = A.Program()
    lk = P.op(A.X_LOOKUP, A.T_BOOL, a=P.op(A.X_COL, A.T_I64, col=cbk), table=t_direct)
    avg = P.op(A.X_DIV, A.T_F64, a=P.op(A.X_ACC, A.T_F64, a=lk, aux=0), b=P.op(A.X_I2F, A.T_F64, a=P.op(A.X_ACC, A.T_I64, a=lk, aux=-1)))
    P.gates = [lk, P.op(A.X_GT, A.T_BOOL, a=P.op(A.X_ACC, A.T_I64, a=lk, aux=-1), b=P.op(A.X_CONST, A.T_I64, imm_i=0)),
               P.op(A.X_GT, A.T_BOOL, a=P.op(A.X_COL, A.T_F64, col=cf), b=avg)]
    P.vals = [P.op(A.X_COL, A.T_F64, col=cf)]
    vals, cnt = ctx.xscan_sum(n, P)
    want = [(fi) for ki, fi in zip(bk.tolist(), f.tolist()) if ki in acc_sum and fi > acc_sum[ki] / acc_cnt[ki]]
    assert cnt == len(want); near(vals[0], float(sum(want)), "x6 acc read")
    checks += 3 + nent

    # 6b. a loop over the ENTRIES (sdqh_table_columns as the scanned table): the key unpacked with DIVI / MODI, the entry's sum and
    # row count as values — a sum over a result dictionary (generator 520-568) as a row program
    P = A.Program()
    kk = P.op(A.X_COL, A.T_I64, col=kcol)
    part_hi = P.op(A.X_DIVI, A.T_I64, a=kk, imm_i=97)
    part_lo = P.op(A.X_MODI, A.T_I64, a=kk, imm_i=7)
    P.key = P.op(A.X_ADD, A.T_I64, a=P.op(A.X_MUL, A.T_I64, a=P.op(A.X_MODI, A.T_I64, a=part_hi, imm_i=5), b=P.op(A.X_CONST, A.T_I64, imm_i=7)), b=part_lo)
    P.gates = [P.op(A.X_GE, A.T_BOOL, a=P.op(A.X_COL, A.T_I64, col=hcol), b=P.op(A.X_CONST, A.T_I64, imm_i=1))]
    P.vals = [P.op(A.X_COL, A.T_F64, col=acols[0]), P.op(A.X_I2F, A.T_F64, a=P.op(A.X_COL, A.T_I64, col=hcol))]
    gk, gv, gc = ctx.xgroupby(nent, P)
    want_g = {}
    for x in acc_sum:
        g7 = ((x // 97) % 5) * 7 + x % 7
        w = want_g.setdefault(g7, [0.0, 0, 0])
        w[0] += acc_sum[x]; w[1] += acc_cnt[x]; w[2] += 1
    assert sorted(gk.tolist()) == sorted(want_g)
    for g7, row, c in zip(gk.tolist(), gv.tolist(), gc.tolist()):
        near(row[0], want_g[g7][0], "x6b sum"); assert row[1] == want_g[g7][1] and c == want_g[g7][2]
    del kcol, pcols, acols, hcol
    checks += 1 + len(want_g)

    # 7. K-C small with lookups (queue path): group by a field of the matched entry's payload (mod 7) and the year
    P = A.Program()
    lk = P.op(A.X_LOOKUP, A.T_BOOL, a=P.op(A.X_COL, A.T_I64, col=cpr), table=t_direct)
    fld = P.op(A.X_FIELD, A.T_I64, a=lk, aux=0)
    bucket = P.op(A.X_SUB, A.T_I64, a=fld, b=P.op(A.X_MUL, A.T_I64, a=P.op(A.X_YEAR, A.T_I64, a=P.op(A.X_MUL, A.T_I64, a=fld, b=P.op(A.X_CONST, A.T_I64, imm_i=10000 // 8 * 0 + 1))), b=P.op(A.X_CONST, A.T_I64, imm_i=10000)))   # fld % 10000
    low = P.op(A.X_LT, A.T_BOOL, a=bucket, b=P.op(A.X_CONST, A.T_I64, imm_i=5000))
    yr = P.op(A.X_SUB, A.T_I64, a=P.op(A.X_YEAR, A.T_I64, a=P.op(A.X_COL, A.T_I64, col=cb)), b=P.op(A.X_CONST, A.T_I64, imm_i=1992))
    P.key = P.op(A.X_ADD, A.T_I64, a=P.op(A.X_MUL, A.T_I64, a=yr, b=P.op(A.X_CONST, A.T_I64, imm_i=2)),
                 b=P.op(A.X_SELECT, A.T_I64, a=low, b=P.op(A.X_CONST, A.T_I64, imm_i=1), c=P.op(A.X_CONST, A.T_I64, imm_i=0)))
    P.gates = [lk]; P.vals = [P.op(A.X_COL, A.T_F64, col=cg)]
    keys, vals, cnts = ctx.xgroupby(n, P)
    wk = (b // 10000 - 1992) * 2 + ((pay[src] % 10000) < 5000)
    for kk, vv, cc in zip(keys.tolist(), vals, cnts.tolist()):
        s_ = hit & (wk == kk)
        assert cc == int(s_.sum()); near(vv[0], float(g[s_].sum()), "x7")
    assert sorted(keys.tolist()) == sorted(np.unique(wk[hit]).tolist())
    checks += 1 + 2 * len(keys)
    for t in (t_direct, t_hash, t_set):
        t.free()
    return checks


def compaction_block_case(ctx):
    """Build + probe-aggregate one table three times and finalise it (a) into a device-writable
    block that fits, (b) into one that overflows and is retried, (c) into pageable arrays; the three
    must agree.  Returns (keys, payload0, value0, hits) of the kept entries."""
    import numpy as np
    from sdqlpy_amd import abi
    rng = np.random.default_rng(5)
    n = 300000
    keys = rng.permutation(n).astype(np.int64) * 3
    pay = rng.integers(0, 1 << 40, n).astype(np.int64)
    pk = rng.integers(0, n, 4 * n).astype(np.int64) * 3
    pv = rng.random(4 * n)
    ckeys, cpay, cpk, cpv = ctx.upload(keys), ctx.upload(pay), ctx.upload(pk), ctx.upload(pv)
    flt = abi.make_filter(ipreds=[(cpay, 0, 1 << 39)])
    got = []
    for cap_hint in (1 << 20, 1000, None):
        t = ctx.hash_build_unique(n, flt, [], ckeys, [cpay], accumulate=True)
        ctx.hash_probe_aggregate(4 * n, abi.make_filter(), t, cpk, abi.make_tuple(abi.TUPLE_A, [cpv]))
        if cap_hint is None:
            cnt = ctx.table_compact_count(t, 2)
            k, p, v, h = ctx.table_compact(t, 2, cnt)
        else:
            k, p, v, h, cnt = ctx.table_compact_into_block(t, 2, cap_hint)
            k2, p2, _, _ = ctx.table_compact(t, 2, cnt, want_values=False, want_hits=False)   # a second call after a direct write
            assert (k2 == k).all() and (p2[0] == p[0]).all()
        assert len(k) == cnt > 1000
        got.append((k.copy(), p[0].copy(), v[0].copy(), h.copy()))
        t.free()
    for g in got[1:]:
        for j, (a, b) in enumerate(zip(got[0], g)):
            if j == 2:                                  # sums: the probe's atomic adds land in any order from run to run
                np.testing.assert_allclose(a, b, rtol=1e-12)
            else:
                assert (a == b).all()
    want = {}
    for kk in pk.tolist():
        want[kk] = want.get(kk, 0) + 1
    sel = pay < (1 << 39) + 1
    expect = [kk for kk, ok in zip(keys.tolist(), sel.tolist()) if ok and want.get(kk, 0) >= 2]
    assert got[0][0].tolist() == expect
    return got[0]


def topk_case(ctx, n=400000, seed=9):
    """sdqh_table_topk on a probed table against a numpy restatement: several sort specs (value desc,
    payload asc/desc incl. a double payload, hits, key), k from 1 to 128, ties broken by build-row
    order, duplicate build keys, and min_hits filtering.  Returns the rows for cross-implementation
    comparison."""
    import numpy as np
    from sdqlpy_amd import abi
    rng = np.random.default_rng(seed)
    keys = rng.permutation(n).astype(np.int64) * 5 + 3
    keys[1000:1200] = keys[:200]                                   # duplicate build keys: the first row owns the entry
    pay_i = rng.integers(0, 50, n).astype(np.int64)                # few distinct values: many ties
    pay_f = (rng.integers(-500, 500, n) / 4.0).astype(np.float64)  # doubles incl. negatives and -0.0/0.0 neighbours
    pk = keys[rng.integers(0, n, 3 * n)]
    pv = rng.integers(1, 1000, 3 * n).astype(np.float64)           # integer-valued: sums are exact in any order
    ck, ci, cf, cpk, cpv = ctx.upload(keys), ctx.upload(pay_i), ctx.upload(pay_f.view(np.int64)), ctx.upload(pk), ctx.upload(pv)
    t = ctx.hash_build_unique(n, abi.make_filter(), [], ck, [ci, cf], accumulate=True)
    ctx.hash_probe_aggregate(3 * n, abi.make_filter(), t, cpk, abi.make_tuple(abi.TUPLE_A, [cpv]))
    # numpy restatement of the table: owner rows in build order
    first = {}
    for i, k in enumerate(keys.tolist()):
        first.setdefault(k, i)
    owners = np.array(sorted(first.values()), np.int64)
    pos = {k: j for j, k in enumerate(keys[owners].tolist())}
    hits = np.zeros(len(owners), np.int64); sums = np.zeros(len(owners))
    idx = np.fromiter((pos[k] for k in pk.tolist()), np.int64, len(pk))
    np.add.at(hits, idx, 1); np.add.at(sums, idx, pv)
    ek, ei, ef = keys[owners], pay_i[owners], pay_f[owners]
    out = []
    specs = [
        (10, 1, [(abi.SORT_VALUE, 0, True, True), (abi.SORT_PAYLOAD, 0, False, False)]),
        (128, 1, [(abi.SORT_PAYLOAD, 0, True, False)]),                                 # heavy ties -> build order decides
        (37, 2, [(abi.SORT_PAYLOAD, 1, False, True), (abi.SORT_HITS, 0, True, False), (abi.SORT_KEY, 0, True, False)]),
        (1, 0, [(abi.SORT_KEY, 0, False, False)]),
        (64, 5, [(abi.SORT_HITS, 0, True, False), (abi.SORT_VALUE, 0, False, True)]),
    ]
    cols = {abi.SORT_KEY: lambda i: ek, abi.SORT_PAYLOAD: lambda i: (ei, ef)[i], abi.SORT_VALUE: lambda i: sums, abi.SORT_HITS: lambda i: hits}
    for k, min_hits, spec in specs:
        gk, gp, gv, gh = ctx.table_topk(t, min_hits, k, spec)
        sel = np.nonzero(hits >= min_hits)[0]
        lex = [sel]                                                 # last key of lexsort is the primary: build order first (least significant)
        for kind, index, desc, _ in reversed(spec):
            a = cols[kind](index)[sel]
            lex.append(-a if desc else a)
        order = sel[np.lexsort(lex)][:k]
        assert gk.tolist() == ek[order].tolist(), (k, spec)
        assert gp[0].tolist() == ei[order].tolist() and gp[1].view(np.float64).tolist() == ef[order].tolist()
        assert gv[0].tolist() == sums[order].tolist() and gh.tolist() == hits[order].tolist()
        out.append((gk.copy(), gv[0].copy()))
    t.free()
    return out


def string_predicate_case(ctx, widths=(1, 2, 3, 7, 10, 25, 31, 32, 33, 55, 64, 65, 100, 128, 129), rows=5000, seed=3, latin=False, key_set=False):
    """The build-side string predicate (==, !=, substring) over fixed-width UCS4 fields of many
    widths, against a plain-Python restatement of VarChar::operator== / contains (reference
    include/varchar.h:61-89: equality = first len units equal and the rest NUL; substring = wcsstr,
    the field ends at its first NUL).  Fields include embedded NULs, needles at both ends,
    overlapping partial matches and non-ASCII units.  latin: every unit of the column is below 256, so the column has a
    byte twin (the needles still include a wider unit); key_set: through build_key_set instead of hash_build_unique."""
    import numpy as np
    from sdqlpy_amd import abi
    rng = np.random.default_rng(seed)
    alphabet = np.array([ord(c) for c in "abgren "] + ([0x00E9, 0x00FF] if latin else [0x00E9, 0x65E5]), np.uint32)      # small: matches are frequent
    checked = 0
    for width in widths:
        raw = alphabet[rng.integers(0, len(alphabet), (rows, width))]
        lens = rng.integers(0, width + 1, rows)
        raw[np.arange(width)[None, :] >= lens[:, None]] = 0                           # zero padding after a random length
        holes = rng.random(rows) < 0.15                                               # embedded NUL with text after it
        pos = rng.integers(0, width, rows)
        raw[holes, pos[holes]] = 0
        needles = ["g", "green", "gre", "ab", "a", "é", "日", "green ab", "nnnnnnnnn", "x"]
        needles = [n for n in needles if len(n) <= max(1, width) and not (latin and n == "日" and width < 2)] + ["".join(chr(c) for c in raw[7, :min(width, 12)] if c)]
        # plant needles at the start, at the very end and after a NUL
        for i, nd in enumerate(needles):
            u = np.array([ord(c) for c in nd], np.uint32)
            if 0 < len(u) <= width and not (latin and u.max() > 255):
                raw[100 + i, :] = 0; raw[100 + i, :len(u)] = u                          # exactly the needle (equality hit)
                raw[200 + i, :] = ord("b"); raw[200 + i, width - len(u):] = u           # at the very end, field full
                if len(u) + 2 <= width:
                    raw[300 + i, :] = 0; raw[300 + i, 0] = ord("b"); raw[300 + i, 2:2 + len(u)] = u   # after an embedded NUL: not visible
        col = np.ascontiguousarray(raw).view("<U%d" % width).reshape(rows)
        fields = [[int(c) for c in r] for r in raw]
        ccol, ckey = ctx.upload(col), ctx.upload(np.arange(rows, dtype=np.int64))
        for nd in needles:
            if not nd:
                continue
            nu = [ord(c) for c in nd]
            for mode, name in ((0, "=="), (1, "!="), (2, "in"), (3, "startsWith"), (4, "endsWith")):
                want = []
                for r, f in enumerate(fields):
                    end = f.index(0) if 0 in f else width
                    if mode == 2:
                        hit = any(f[s:s + len(nu)] == nu for s in range(0, end - len(nu) + 1))
                    elif mode == 3:
                        hit = len(nu) <= end and f[:len(nu)] == nu
                    elif mode == 4:
                        hit = len(nu) <= end and f[end - len(nu):end] == nu
                    else:
                        eq = len(nu) <= width and f[:len(nu)] == nu and not any(f[len(nu):])
                        hit = eq != (mode == 1)
                    if hit:
                        want.append(r)
                if key_set:
                    t = ctx.build_key_set(rows, abi.make_filter(spreds=[(ccol, nd, mode)]), [], ckey)
                    (hit,), nh = ctx.scan_compact(rows, abi.make_filter(), [(t, ckey)], [ckey])
                    got = hit.download(0, nh).tolist() if nh else []
                else:
                    t = ctx.hash_build_unique(rows, abi.make_filter(spreds=[(ccol, nd, mode)]), [], ckey, [])
                    n = ctx.table_compact_count(t, 0)
                    got = ctx.table_compact(t, 0, n, want_values=False, want_hits=False)[0].tolist()
                t.free()
                assert got == want, "width %d, %r %s field: %d rows, expected %d (first difference %s)" % (
                    width, nd, name, len(got), len(want), sorted(set(got) ^ set(want))[:5])
                checked += 1
    return checked


def column_compare_case(ctx, n=70001, seed=12):
    """Column-vs-column predicates (a op b on two int64 or two float64 columns, every operator,
    alone / paired / next to a range predicate) through scan_filter_sum, groupby_small and
    hash_build_unique, against numpy."""
    import numpy as np
    from sdqlpy_amd import abi
    rng = np.random.default_rng(seed)
    ia, ib = rng.integers(-50, 50, n).astype(np.int64), rng.integers(-50, 50, n).astype(np.int64)
    fa = rng.integers(-40, 40, n) / 8.0
    fb = rng.integers(-40, 40, n) / 8.0
    fa[::97] = -0.0; fb[::97] = 0.0                                   # -0.0 == 0.0
    v = rng.integers(1, 100, n).astype(np.float64)
    g = rng.integers(0, 7, n).astype(np.int64)
    cia, cib, cfa, cfb, cv, cg, ck = ctx.upload(ia), ctx.upload(ib), ctx.upload(fa), ctx.upload(fb), ctx.upload(v), ctx.upload(g), ctx.upload(np.arange(n, dtype=np.int64))
    ops = {abi.CMP_LT: np.less, abi.CMP_LE: np.less_equal, abi.CMP_EQ: np.equal, abi.CMP_NE: np.not_equal}
    checked = 0
    for op, fn in ops.items():
        for (ca, cb, a, b) in ((cia, cib, ia, ib), (cfa, cfb, fa, fb)):
            m = fn(a, b)
            vals, cnt = ctx.scan_filter_sum(n, abi.make_filter(cpreds=[(ca, cb, op)]), abi.make_tuple(abi.TUPLE_A, [cv]))
            assert cnt == int(m.sum()) and vals[0] == float(v[m].sum()), (op, cnt, int(m.sum()))
            checked += 1
        # two comparisons and a range predicate together
        m = fn(ia, ib) & (fb < fa) & (g >= 2) & (g <= 5)
        flt = abi.make_filter(ipreds=[(cg, 2, 5)], cpreds=[(cia, cib, op), (cfb, cfa, abi.CMP_LT)])
        keys, vals, cnts = ctx.groupby_small(n, flt, [cg], abi.make_tuple(abi.TUPLE_A, [cv]))
        got = {int(k[0]): (float(x[0]), int(c)) for k, x, c in zip(keys, vals, cnts)}
        want = {int(k): (float(v[m & (g == k)].sum()), int((m & (g == k)).sum())) for k in np.unique(g[m])}
        assert got == want, (op, got, want)
        t = ctx.hash_build_unique(n, flt, [], ck, [])
        cnt = ctx.table_compact_count(t, 0)
        rows = ctx.table_compact(t, 0, cnt, want_values=False, want_hits=False)[0]
        t.free()
        assert rows.tolist() == np.nonzero(m)[0].tolist(), op
        # a comparison as the ONLY predicate of a build on a dense, unique key column: must not be
        # mistaken for an unfiltered build (which indexes the source columns in place)
        m1 = fn(ia, ib)
        t = ctx.hash_build_unique(n, abi.make_filter(cpreds=[(cia, cib, op)]), [], ck, [])
        assert t.size() == int(m1.sum()), op
        cnt = ctx.table_compact_count(t, 0)
        assert ctx.table_compact(t, 0, cnt, want_values=False, want_hits=False)[0].tolist() == np.nonzero(m1)[0].tolist(), op
        t.free()
        checked += 2
    return checked


def key_set_case(ctx, n=200000, seed=8):
    """sdqh_build_key_set: the same membership as sdqh_hash_build_unique gives (probe results,
    size = distinct surviving keys), with duplicates, filters of every kind and a chained probe;
    a key range that does not suit a bitmap is refused with SDQH_ERR_UNSUPPORTED."""
    import numpy as np
    import pytest
    from sdqlpy_amd import abi
    rng = np.random.default_rng(seed)
    keys = np.sort(rng.integers(1000, 1000 + n // 2, n)).astype(np.int64)        # clustered, ~2 rows per key
    a, b = rng.integers(0, 100, n).astype(np.int64), rng.integers(0, 100, n).astype(np.int64)
    txt = np.array(["BUILDING", "MACHINERY", "AUTOMOBILE"], "<U10")[rng.integers(0, 3, n)]
    probe_keys = rng.integers(0, 2000 + n // 2, 3 * n).astype(np.int64)
    v = rng.integers(1, 50, 3 * n).astype(np.float64)
    ck, ca, cb, ct, cpk, cv = ctx.upload(keys), ctx.upload(a), ctx.upload(b), ctx.upload(txt), ctx.upload(probe_keys), ctx.upload(v)
    filters = [
        (abi.make_filter(), np.ones(n, bool)),
        (abi.make_filter(ipreds=[(ca, 10, 60)]), (a >= 10) & (a <= 60)),
        (abi.make_filter(cpreds=[(ca, cb, abi.CMP_LT)]), a < b),
        (abi.make_filter(spreds=[(ct, "BUILDING", abi.STR_EQ)]), txt == "BUILDING"),
        (abi.make_filter(ipreds=[(ca, 0, -1)]), np.zeros(n, bool)),              # nothing passes: empty set
    ]
    for flt, mask in filters:
        members = np.unique(keys[mask])
        s = ctx.build_key_set(n, flt, [], ck)
        assert s.size() == len(members)
        vals, cnt = ctx.scan_probe_sum(3 * n, abi.make_filter(), [(s, cpk)], abi.make_tuple(abi.TUPLE_A, [cv]))
        hit = np.isin(probe_keys, members)
        assert cnt == int(hit.sum()) and vals[0] == float(v[hit].sum())
        # chained: a second set built from the rows whose key is in the first
        s2 = ctx.build_key_set(3 * n, abi.make_filter(), [(s, cpk)], cpk)
        assert s2.size() == len(np.unique(probe_keys[hit]))
        s2.free(); s.free()
    sparse = ctx.upload(np.array([1, 1 << 40], np.int64))
    with pytest.raises(abi.SdqhError) as e:
        ctx.build_key_set(2, abi.make_filter(), [], sparse)
    assert e.value.code == abi.ERR_UNSUPPORTED
    return len(filters)


def groupby_key_case(ctx, n=250000, seed=14):
    """sdqh_groupby_key on clustered and on shuffled keys, dense and sparse key ranges, with a
    filter; then HAVING (sdqh_table_select_keys) and the selected set used as a semi-join.  Checked
    against numpy (integer-valued doubles: sums are exact in any order)."""
    import numpy as np
    import pytest
    from sdqlpy_amd import abi
    rng = np.random.default_rng(seed)
    checked = 0
    for name, keys in (("clustered", np.sort(rng.integers(5, 5 + n // 4, n)).astype(np.int64)),
                       ("shuffled", rng.integers(-1000, 3000, n).astype(np.int64)),
                       ("sparse", (rng.integers(0, 2000, n).astype(np.int64) << 33) + 7)):
        v = rng.integers(1, 60, n).astype(np.float64)
        w = rng.integers(0, 10, n).astype(np.int64)
        ck, cv, cw = ctx.upload(keys), ctx.upload(v), ctx.upload(w)
        mask = (w >= 2)
        t = ctx.groupby_key(n, abi.make_filter(ipreds=[(cw, 2, 99)]), ck, abi.make_tuple(abi.TUPLE_A, [cv]))
        cnt = ctx.table_compact_count(t, 1)
        k, _, vals, hits = ctx.table_compact(t, 1, cnt)
        uk, inv = np.unique(keys[mask], return_inverse=True)
        sums = np.zeros(len(uk)); np.add.at(sums, inv, v[mask])
        counts = np.bincount(inv, minlength=len(uk))
        order = np.argsort(k)
        assert k[order].tolist() == uk.tolist(), name
        assert vals[0][order].tolist() == sums.tolist() and hits[order].tolist() == counts.tolist(), name
        assert t.size() == len(uk)
        thr = float(np.median(sums))
        if name == "sparse":
            with pytest.raises(abi.SdqhError) as e:
                ctx.table_select_keys(t, 1, 0, thr, np.inf)
            assert e.value.code == abi.ERR_UNSUPPORTED
        else:
            sel = ctx.table_select_keys(t, 1, 0, abi.gt_float(thr), np.inf)          # HAVING sum > thr
            want = uk[sums > thr]
            assert sel.size() == len(want)
            pk = rng.integers(int(keys.min()) - 5, int(keys.max()) + 5, n).astype(np.int64)
            vals2, c2 = ctx.scan_probe_sum(n, abi.make_filter(), [(sel, ctx.upload(pk))], abi.make_tuple(abi.TUPLE_A, [cv]))
            hit = np.isin(pk, want)
            assert c2 == int(hit.sum()) and vals2[0] == float(v[hit].sum()), name
            sel.free()
        t.free()
        checked += 1
    return checked


def share_groups_case(ctx, n_build=40000, n_probe=300000, seed=21):
    """sdqh_table_share_groups: entries with equal payload fields become one group (Q10: orders of
    one customer).  Unique and duplicate build keys, dense and sparse key ranges, one and two
    fields, values outside the declared range, then K-F and top-k.  Checked against numpy
    (integer-valued doubles: sums are exact in any order)."""
    import numpy as np
    import pytest
    from sdqlpy_amd import abi
    rng = np.random.default_rng(seed)
    checked = 0
    for name, keys in (("dense", rng.permutation(n_build * 2)[:n_build].astype(np.int64) + 100),
                       ("dups", rng.integers(0, n_build // 2, n_build).astype(np.int64)),
                       ("sparse", (rng.permutation(n_build).astype(np.int64) << 24) + 3)):
        for nfields in (1, 2):
            pa = rng.integers(0, max(4, n_build // 8), n_build).astype(np.int64)          # "customer row"
            pb = (pa % 5 if nfields == 1 else rng.integers(-1, 4, n_build)).astype(np.int64)   # -1: outside the declared range
            ck, ca, cb = ctx.upload(keys), ctx.upload(pa), ctx.upload(pb)
            t = ctx.hash_build_unique(n_build, abi.make_filter(), [], ck, [ca, cb], accumulate=True)
            fields, lo, span = ([0], [0], [int(pa.max()) + 1]) if nfields == 1 else ([0, 1], [0, 0], [int(pa.max()) + 1, 4])
            ctx.table_share_groups(t, fields, lo, span)
            pk = keys[rng.integers(0, n_build, n_probe)].copy()
            pk[rng.random(n_probe) < 0.2] = -7                                              # misses
            v = rng.integers(1, 50, n_probe).astype(np.float64)
            w = rng.integers(0, 10, n_probe).astype(np.int64)
            flt = abi.make_filter(ipreds=[(ctx.upload(w), 1, 99)])
            ctx.hash_probe_aggregate(n_probe, flt, t, ctx.upload(pk), abi.make_tuple(abi.TUPLE_A, [ctx.upload(v)]))
            # expectation: first build row per key owns the entry; groups by the owner's fields
            uk, first_row = np.unique(keys, return_index=True)
            ea, eb = pa[first_row], pb[first_row]
            inside = (eb >= 0) if nfields == 2 else np.ones(len(uk), bool)
            gid_fields = ea * 16 + (eb if nfields == 2 else 0)
            rep = {}
            for e in np.argsort(first_row, kind="stable"):                                 # build-row order
                if inside[e]:
                    rep.setdefault(int(gid_fields[e]), int(e))
            owner_of = np.array([rep[int(g)] if ins else int(e) for e, (g, ins) in enumerate(zip(gid_fields, inside))])
            ok = (w >= 1) & (pk != -7)
            ent = np.searchsorted(uk, pk[ok])
            tgt = owner_of[ent]
            sums = np.bincount(tgt, weights=v[ok], minlength=len(uk)); counts = np.bincount(tgt, minlength=len(uk))
            live = counts > 0
            cnt = ctx.table_compact_count(t, 1)
            k, pay, vals, hits = ctx.table_compact(t, 1, cnt)
            order = np.argsort(k)
            assert k[order].tolist() == uk[live].tolist(), (name, nfields)
            assert vals[0][order].tolist() == sums[live].tolist() and hits[order].tolist() == counts[live].tolist(), (name, nfields)
            assert pay[0][order].tolist() == ea[live].tolist() and pay[1][order].tolist() == eb[live].tolist(), (name, nfields)
            tk, _, tv, _ = ctx.table_topk(t, 1, 5, [(abi.SORT_VALUE, 0, True, True), (abi.SORT_KEY, 0, False, False)])
            want = sorted(zip((-sums[live]).tolist(), uk[live].tolist()))[:5]
            assert list(zip((-tv[0]).tolist(), tk.tolist())) == want, (name, nfields)
            t.free()
            checked += 1
    # argument checking
    ck = ctx.upload(np.arange(10, dtype=np.int64))
    t = ctx.hash_build_unique(10, abi.make_filter(), [], ck, [ck], accumulate=True)
    for fields, lo, span, code in (([1], [0], [10], abi.ERR_INVALID), ([0], [0], [0], abi.ERR_INVALID), ([], [], [], abi.ERR_INVALID),
                                   ([0], [0], [(1 << 28) + 1], abi.ERR_UNSUPPORTED)):
        with pytest.raises(abi.SdqhError) as e:
            ctx.table_share_groups(t, fields, lo, span)
        assert e.value.code == code
    t.free()
    plain = ctx.hash_build_unique(10, abi.make_filter(), [], ck, [ck], accumulate=False)
    with pytest.raises(abi.SdqhError) as e:
        ctx.table_share_groups(plain, [0], [0], [10])
    assert e.value.code == abi.ERR_INVALID
    plain.free()
    return checked


def empty_input_case(ctx):
    """Zero-row inputs through every table-producing entry point and what consumes them."""
    import numpy as np
    from sdqlpy_amd import abi
    e_i, e_f = ctx.upload(np.zeros(0, np.int64)), ctx.upload(np.zeros(0, np.float64))
    keys = ctx.upload(np.arange(100, dtype=np.int64)); vals = ctx.upload(np.ones(100))
    flt = abi.make_filter()
    s = ctx.build_key_set(0, flt, [], e_i)
    assert s.size() == 0
    v, c = ctx.scan_probe_sum(100, flt, [(s, keys)], abi.make_tuple(abi.TUPLE_A, [vals]))
    assert c == 0 and v[0] == 0.0
    v, c = ctx.scan_probe_sum(0, flt, [(s, e_i)], abi.make_tuple(abi.TUPLE_A, [e_f]))
    assert c == 0
    g = ctx.groupby_key(0, flt, e_i, abi.make_tuple(abi.TUPLE_A, [e_f]))
    assert g.size() == 0 and ctx.table_compact_count(g, 1) == 0
    k, p, vv, h = ctx.table_topk(g, 0, 5, [(abi.SORT_VALUE, 0, True, True)])
    assert len(k) == 0
    t = ctx.hash_build_unique(0, flt, [], e_i, [], accumulate=True)
    assert t.size() == 0 and len(ctx.table_topk(t, 0, 3, [(abi.SORT_KEY, 0, False, False)])[0]) == 0
    ctx.hash_probe_aggregate(100, flt, t, keys, abi.make_tuple(abi.TUPLE_A, [vals]))
    assert ctx.table_compact_count(t, 1) == 0
    g2 = ctx.groupby_key(100, abi.make_filter(ipreds=[(keys, 1000, 2000)]), keys, abi.make_tuple(abi.TUPLE_A, [vals]))   # nothing passes
    assert g2.size() == 0
    sel = ctx.table_select_keys(g2, 1, 0, -np.inf, np.inf)
    assert sel.size() == 0
    for x in (s, g, t, g2, sel):
        x.free()
    return True


def fuzz_case(hip, cpu, seed, rounds=12):
    """Differential fuzzing of the pattern calls: random row counts around tile / wave boundaries,
    random filters (int / float ranges, column comparisons, text predicates of every mode), random
    semi-join probes, payload counts, duplicate and sparse / dense / negative keys — the HIP library
    against the CPU implementation of the same ABI.  Values are integer-valued doubles so that sums
    are exact in any order."""
    import numpy as np
    from sdqlpy_amd import abi
    rng = np.random.default_rng(seed)
    sizes = [0, 1, 63, 64, 65, 127, 128, 129, 511, 512, 513, 1023, 1025, 2047, 2049, 4097, 10000, 33333]
    words = np.array(["", "a", "ab", "abc", "b", "ba", "PROMO X", "PROMO", "PRO", "xPROMO", "green", "gre en"], "<U8")
    for rnd in range(rounds):
        n = int(rng.choice(sizes))
        style = rng.integers(0, 4)
        if style == 0:
            keys = rng.permutation(max(n, 1))[:n].astype(np.int64) * 3 - 50                      # unique, dense-ish, some negative
        elif style == 1:
            keys = rng.integers(-20, 40, n).astype(np.int64)                                       # heavy duplicates
        elif style == 2:
            keys = np.sort(rng.integers(0, max(4 * n, 4), n)).astype(np.int64)                     # clustered with duplicates
        else:
            keys = (rng.integers(0, 1 << 20, n).astype(np.int64) << 24) + rng.integers(0, 5, n)   # sparse, wide
        ia, ib = rng.integers(0, 30, n).astype(np.int64), rng.integers(0, 30, n).astype(np.int64)
        fa = rng.integers(0, 40, n) / 4.0
        txt = words[rng.integers(0, len(words), n)]
        pay = rng.integers(-(1 << 40), 1 << 40, n).astype(np.int64)
        v = rng.integers(1, 50, n).astype(np.float64)
        m = int(rng.choice(sizes))
        pk = (keys[rng.integers(0, n, m)] if n else rng.integers(0, 10, m)).astype(np.int64)
        pv = rng.integers(1, 9, m).astype(np.float64)
        pd_ = rng.integers(0, 30, m).astype(np.int64)

        def spec(ctx):
            c = {k: ctx.upload(a) for k, a in dict(keys=keys, ia=ia, ib=ib, fa=fa, txt=txt, pay=pay, v=v, pk=pk, pv=pv, pd=pd_).items()}
            r = np.random.default_rng(seed * 1000 + rnd)                   # the same choices for both implementations
            ip = [(c["ia"], int(r.integers(0, 10)), int(r.integers(10, 30)))] if r.random() < 0.6 else []
            fp = [(c["fa"], float(r.integers(0, 4)), float(r.integers(4, 10)))] if r.random() < 0.4 else []
            cp = [(c["ia"], c["ib"], int(r.integers(0, 4)))] if r.random() < 0.4 else []
            sp = [(c["txt"], str(r.choice(["PROMO", "a", "b", "green", "", "ab"])), int(r.integers(0, 5)))] if r.random() < 0.4 else []
            return c, abi.make_filter(ip, fp, sp, cp), r

        out = []
        for ctx in (hip, cpu):
            c, flt, r = spec(ctx)
            res = []
            npay = int(r.integers(0, 2))
            t = ctx.hash_build_unique(n, flt, [], c["keys"], [c["pay"]][:npay], accumulate=True)
            res.append(("size", t.size()))
            pflt = abi.make_filter(ipreds=[(c["pd"], 0, int(r.integers(5, 30)))]) if r.random() < 0.5 else abi.make_filter()
            ctx.hash_probe_aggregate(m, pflt, t, c["pk"], abi.make_tuple(abi.TUPLE_A, [c["pv"]]))
            cnt = ctx.table_compact_count(t, 1)
            k, p, vals, h = ctx.table_compact(t, 1, cnt)
            order = np.argsort(k, kind="stable")
            res.append(("agg", k[order].tolist(), vals[0][order].tolist(), h[order].tolist(), p[0][order].tolist() if npay else None))
            try:
                s = ctx.build_key_set(n, flt, [], c["keys"])
                res.append(("set", s.size(), ctx.scan_probe_sum(m, abi.make_filter(), [(s, c["pk"])], abi.make_tuple(abi.TUPLE_A, [c["pv"]]))))
                s.free()
            except abi.SdqhError as exc:
                res.append(("set-unsupported", exc.code))
            g = ctx.groupby_key(n, flt, c["keys"], abi.make_tuple(abi.TUPLE_A, [c["v"]]))
            gc = ctx.table_compact_count(g, 1)
            gk, _, gv, gh = ctx.table_compact(g, 1, gc)
            go = np.argsort(gk, kind="stable")
            res.append(("groupby", gk[go].tolist(), gv[0][go].tolist(), gh[go].tolist()))
            kk = int(r.choice([1, 5, 16, 17, 64, 128]))
            tk, _, tv, th = ctx.table_topk(g, 1, kk, [(abi.SORT_VALUE, 0, True, True), (abi.SORT_KEY, 0, False, False)])
            res.append(("topk", tk.tolist(), tv[0].tolist(), th.tolist()))
            res.append(("scan", ctx.scan_filter_sum(n, flt, abi.make_tuple(abi.TUPLE_A, [c["v"]]))))
            t.free(); g.free()
            out.append(res)
        assert out[0] == out[1], "seed %d round %d (n=%d, m=%d, key style %d): %s" % (
            seed, rnd, n, m, style, next((a[0], str(a)[:300], str(b)[:300]) for a, b in zip(out[0], out[1]) if a != b))
    return rounds


def xcode_edge_cases(ctx, n=40000, seed=21):
    """Sorted-dictionary code twins (csrc/sdqh_codes.hip) and the comparisons rewritten in code space (sdqh_x.hip: tight_plan):
    every comparison operator against constants below / equal to / between / above the column's distinct values (and NaN), on
    integer columns with gaps and negatives, two-decimal doubles, columns that cannot be coded (too many values, more decimals,
    beyond int32); values of coded columns through their LDS tables; a dense small group key on the per-lane accumulators.
    Against numpy; the caller sets feature_min_rows to 0 so that small inputs take the coded instances.  Returns the check count."""
    from sdqlpy_amd import abi as A
    rng = np.random.default_rng(seed)
    cols = {
        "gaps": np.array([-5, 3, 4, 1000, 70000], np.int64)[rng.integers(0, 5, n)],                 # non-consecutive, negative: a real dictionary
        "dense": rng.integers(7, 11, n).astype(np.int64),                                           # consecutive: code + offset
        "cents": np.array([0.0, 0.05, 1.25, -3.5, 99.99], np.float64)[rng.integers(0, 5, n)],       # two decimals: coded through the 4-byte twin
        "wide": rng.integers(0, 200000, n).astype(np.int64),                                        # > 65 536 distinct values: no codes, 4-byte twin
        "huge": rng.integers(0, 5, n).astype(np.int64) * (1 << 40),                                 # beyond int32 and a wide range: the column itself
        "fine": np.round(rng.random(n), 5),                                                         # more than two decimals: the column itself
        "dates": (19920101 + rng.integers(0, 2000, n)).astype(np.int64),                            # 2 000 values: 2-byte codes where only compared
    }
    dev = {k: ctx.upload(v) for k, v in cols.items()}
    checks = 0
    ops = [(A.X_LT, np.less), (A.X_LE, np.less_equal), (A.X_GT, np.greater), (A.X_GE, np.greater_equal), (A.X_EQ, np.equal), (A.X_NE, np.not_equal)]
    consts = {"gaps": [-6, -5, 0, 3, 4, 5, 1000, 69999, 70000, 70001], "dense": [6, 7, 9, 10, 11], "cents": [-4.0, -3.5, 0.0, 0.049999, 0.05, 1.25, 99.99, 100.0, float("nan")],
              "wide": [-1, 0, 100000, 199999, 200000], "huge": [0, 1 << 40, (1 << 41) + 1], "fine": [0.0, 0.5, 2.0], "dates": [19920100, 19920101, 19920500, 19921231, 19990101]}
    for name, values in consts.items():
        isf = cols[name].dtype.kind == "f"
        for code, fn in ops:
            for c in values:
                for flipped in (False, True):
                    P = A.Program()
                    x = P.op(A.X_COL, A.T_F64 if isf else A.T_I64, col=dev[name])
                    k = P.op(A.X_CONST, A.T_F64 if isf else A.T_I64, **({"imm_f": c} if isf else {"imm_i": c}))
                    P.gates = [P.op(code, A.T_BOOL, a=k, b=x) if flipped else P.op(code, A.T_BOOL, a=x, b=k)]
                    P.vals = [P.op(A.X_CONST, A.T_F64, imm_f=1.0)]
                    _, cnt = ctx.xscan_sum(n, P)
                    with np.errstate(invalid="ignore"):
                        want = int((fn(c, cols[name]) if flipped else fn(cols[name], c)).sum())
                    assert cnt == want, (name, code, c, flipped, cnt, want)
                    checks += 1
    # values of coded columns (LDS tables, derived tables, code + offset), mixed with an uncoded column
    P = A.Program()
    g = P.op(A.X_COL, A.T_I64, col=dev["gaps"]); d = P.op(A.X_COL, A.T_I64, col=dev["dense"]); c = P.op(A.X_COL, A.T_F64, col=dev["cents"]); f = P.op(A.X_COL, A.T_F64, col=dev["fine"])
    one = P.op(A.X_CONST, A.T_F64, imm_f=1.0)
    v0 = P.op(A.X_MUL, A.T_F64, a=P.op(A.X_SUB, A.T_F64, a=one, b=c), b=f)                                    # (1 - cents) * fine
    v1 = P.op(A.X_I2F, A.T_F64, a=P.op(A.X_ADD, A.T_I64, a=P.op(A.X_MUL, A.T_I64, a=g, b=P.op(A.X_CONST, A.T_I64, imm_i=3)), b=d))   # gaps * 3 + dense
    P.gates = [P.op(A.X_GE, A.T_BOOL, a=P.op(A.X_COL, A.T_I64, col=dev["dates"]), b=P.op(A.X_CONST, A.T_I64, imm_i=19930101))]
    P.vals = [v0, v1, c]
    vals, cnt = ctx.xscan_sum(n, P)
    m = cols["dates"] >= 19930101
    assert cnt == int(m.sum())
    for got, want in zip(vals, [((1.0 - cols["cents"]) * cols["fine"])[m].sum(), (cols["gaps"] * 3 + cols["dense"])[m].sum(), cols["cents"][m].sum()]):
        assert abs(got - want) <= 1e-9 * max(abs(want), 1.0), (got, want)
    checks += 4
    # a dense small key: per-lane accumulators; key = (dense - 7) * 5 + code of gaps' sign ... kept simple: dense alone and dense * 2 + (cents > 0)
    for variant in (0, 1):
        P = A.Program()
        d = P.op(A.X_COL, A.T_I64, col=dev["dense"]); c = P.op(A.X_COL, A.T_F64, col=dev["cents"]); f = P.op(A.X_COL, A.T_F64, col=dev["fine"])
        if variant == 0:
            P.key = d; want_key = cols["dense"]
        else:
            pos = P.op(A.X_SELECT, A.T_I64, a=P.op(A.X_GT, A.T_BOOL, a=c, b=P.op(A.X_CONST, A.T_F64, imm_f=0.0)), b=P.op(A.X_CONST, A.T_I64, imm_i=1), c=P.op(A.X_CONST, A.T_I64, imm_i=0))
            P.key = P.op(A.X_ADD, A.T_I64, a=P.op(A.X_MUL, A.T_I64, a=d, b=P.op(A.X_CONST, A.T_I64, imm_i=2)), b=pos); want_key = cols["dense"] * 2 + (cols["cents"] > 0)
        P.gates = [P.op(A.X_LT, A.T_BOOL, a=f, b=P.op(A.X_CONST, A.T_F64, imm_f=0.9))]
        P.vals = [f, P.op(A.X_MUL, A.T_F64, a=c, b=f)]
        keys, vals, cnts = ctx.xgroupby(n, P)
        m = cols["fine"] < 0.9
        assert sorted(keys.tolist()) == sorted(np.unique(want_key[m]).tolist())
        for k, v, cn in zip(keys.tolist(), vals, cnts.tolist()):
            sel = m & (want_key == k)
            assert cn == int(sel.sum())
            assert abs(v[0] - cols["fine"][sel].sum()) <= 1e-9 * max(abs(cols["fine"][sel].sum()), 1.0)
            assert abs(v[1] - (cols["cents"] * cols["fine"])[sel].sum()) <= 1e-9 * max(abs((cols["cents"] * cols["fine"])[sel].sum()), 1.0)
            checks += 3
    for col in dev.values():
        col.free()
    return checks


def dict_loop_cases(eng, case, rel=0.0):
    """Sums over result dictionaries (frontend.HostDictOp) as device loops (xplan.prepare_dict_scan) against the reference's
    results AND against the host evaluation of the same plans: q16 (group-by over the entries of a dictionary keyed by four
    packed fields), q15 (a record set with looked-up text fields, with and without ORDER BY / LIMIT), q11 (a condition against a
    scalar).  Asserts that the entries really were read as resident columns (sdqh_table_columns)."""
    db = case_db(case)
    checked, on_device = 0, {}
    watch = spy_calls("table_columns")
    calls = watch.__enter__()
    try:
        for q in ("q16", "q15", "q11", "q2"):
            want = case["results"].get(q)
            if want is None:
                continue
            for top in (None, Q.TPCH_ORDER[q]):
                results = {}
                for mode in (True, False):
                    eng.dict_programs = mode
                    before = len(calls)
                    plan = frontend.lower_function(Q.QUERIES[q])
                    res = eng_mod.execute_plan(eng, plan, [db[t] for t in Q.QUERY_TABLES[q]], top)
                    used = len(calls) - before
                    nonempty = bool(want["rows"])
                    assert mode or used == 0, (q, used)
                    if mode and used:                                       # (a source small enough to come back as host groups stays on the host)
                        on_device[q] = on_device.get(q, 0) + 1
                    results[mode] = res
                    if (top is None) != (q == "q15") and nonempty:          # (q15's golden vector is TPCH's top-1 of the set returned here)
                        check_against_golden(res, want, rel, "dict loop %s/%s" % (case["name"], q))
                a, b = results[True], results[False]
                assert a.columns == b.columns, (q, a.columns, b.columns)
                rows_a = list(zip(*[a.column(c).tolist() for c in a.columns]))
                rows_b = list(zip(*[b.column(c).tolist() for c in b.columns]))
                if top is None:
                    rows_a, rows_b = sorted(rows_a), sorted(rows_b)
                assert_rows_match(rows_a, rows_b, rel, "dict loop device vs host %s top=%r" % (q, top))
                checked += 1
    finally:
        eng.dict_programs = True
        watch.__exit__()
    return checked, on_device


DENSE_DOMAIN_SRC = '''
def f(orders, customer):
    per_customer = orders.sum(lambda o: {o[0].o_custkey: record({"n": 1, "total": o[0].o_totalprice})}
                              if o[0].o_totalprice > 10.0 else None)
    with_orders = customer.sum(lambda c: record({"members": 1.0 if per_customer[c[0].c_custkey] != None else 0.0,
                                                  "all": 1.0,
                                                  "spent": per_customer[c[0].c_custkey].total if per_customer[c[0].c_custkey] != None else 0.0}))
    return with_orders
'''


def dense_domain_case(eng, ncust=3000, nord=40000, seed=5):
    """A group-by whose integer key spans a range much smaller than the row count is summed in ONE pass into a table that holds
    every key of the range (xplan: large mode over a dense domain).  Two thirds of the keys carry no row: they must not be members
    of the dictionary (`d[k] != None` is false for them, as in the reference where the key was never inserted), checked against
    numpy — both libraries run the same plan, so their agreement alone would not pin this."""
    from sdqlpy_amd import sdql_lib
    rng = np.random.default_rng(seed)
    custkeys = np.arange(1, ncust + 1, dtype=np.int64)
    present = rng.choice(custkeys, size=ncust // 3, replace=False)
    o_cust = rng.choice(present, size=nord).astype(np.int64)
    o_price = np.round(rng.uniform(0.0, 100.0, nord), 2)
    o_cust[0], o_cust[1] = 1, ncust                                     # the range's ends occur: the domain is the whole range
    orders = sdql_lib.table_from_columns(["o_custkey", "o_totalprice"], [o_cust, o_price])
    customer = sdql_lib.table_from_columns(["c_custkey"], [custkeys])
    plan = frontend.lower_source(DENSE_DOMAIN_SRC, None, 1, None)
    with spy_calls("hash_build_unique", lambda n, *a, **k: (n, k.get("accumulate", False))) as built:
        res = eng_mod.execute_plan(eng, plan, [orders, customer])
    assert (ncust, True) in built, built                                # the domain build: one entry per key of the range, with accumulators
    keep = o_price > 10.0
    assert res["members"] == float(len(np.unique(o_cust[keep]))) and res["all"] == float(ncust), res
    want = float(o_price[keep].sum())
    assert abs(res["spent"] - want) <= 1e-9 * want, (res, want)
    return res


OVERFLOWING_GROUPS_SRC = '''
def f(orders):
    per_key = orders.sum(lambda o: {o[0].o_custkey: record({"n": 1, "total": o[0].o_totalprice})} if o[0].o_totalprice > 10.0 else None)
    out = per_key.sum(lambda g: {unique(g[0].concat(g[1])): True})
    return out
'''


def deferred_result_cases(eng, goldens, rel=0.0):
    """Engine.deferred_results: a plan's last device call is launched and not waited for, the DeferredResultSet finishes the plan when
    it is first looked at.  Every golden query both ways (deferred == waited-for, bit for bit on one implementation); several
    queries launched back to back and read in reverse order; results dropped unread; and what only the data decides — more groups
    than the group-by kernel holds — surfacing at collection time and handled by re-running the plan's other path."""
    from sdqlpy_amd import sdql_lib
    from sdqlpy_amd.result import DeferredResultSet
    checked, deferred_seen = 0, set()

    def rows_of(r):
        return r if isinstance(r, float) else r.rows()
    try:
        for gold in goldens:
            for case in gold["cases"]:
                if case["name"] not in ("small", "medium"):
                    continue
                db = case_db(case)
                plans = {q: frontend.lower_function(Q.QUERIES[q]) for q in case["results"]}
                args = {q: [db[t] for t in Q.QUERY_TABLES[q]] for q in plans}
                eng.deferred_results = False
                want = {q: rows_of(eng_mod.execute_plan(eng, plans[q], args[q])) for q in plans}
                eng.deferred_results = True
                pending = []
                for q in plans:                                         # all launched, none read
                    r = eng_mod.execute_plan(eng, plans[q], args[q])
                    if isinstance(r, DeferredResultSet):
                        deferred_seen.add(q)
                    pending.append((q, r))
                for q in plans:                                         # some results are never looked at
                    eng_mod.execute_plan(eng, plans[q], args[q])
                for q, r in reversed(pending):
                    got = rows_of(r)
                    if rel == 0.0 or isinstance(got, float):
                        assert got == want[q] or (isinstance(got, float) and abs(got - want[q]) <= rel * abs(want[q])), (case["name"], q)
                    else:
                        assert_rows_match(got, want[q], rel, "deferred %s/%s" % (case["name"], q))
                    check_against_golden(r, case["results"][q], max(rel, 1e-12) if q in ("q10", "q15") else rel, "deferred %s/%s" % (case["name"], q)) if q != "q15" else None
                    checked += 1
        # more groups than the kernel's table: decided by the data, found at collection, the plan re-run on its large-domain path
        rng = np.random.default_rng(3)
        n = 50000
        o_cust = rng.integers(1, 3000, n).astype(np.int64)
        o_price = np.round(rng.uniform(0.0, 100.0, n), 2)
        orders = sdql_lib.table_from_columns(["o_custkey", "o_totalprice"], [o_cust, o_price])
        plan = frontend.lower_source(OVERFLOWING_GROUPS_SRC, None, 1, None)
        for run in range(3):
            r = eng_mod.execute_plan(eng, plan, [orders])
            keep = o_price > 10.0
            assert r.size() == len(np.unique(o_cust[keep])), run
            tot = dict(zip(r.column("o_custkey").tolist(), r.column("total").tolist()))
            k0 = int(o_cust[keep][0])
            assert abs(tot[k0] - float(o_price[keep][o_cust[keep] == k0].sum())) <= 1e-9 * tot[k0]
        checked += 1
    finally:
        eng.deferred_results = False
    return checked, deferred_seen


def redistribution_pack_case(ctx, n=70001, seed=21, nparts=5):
    """sdqh_partition_pack / sdqh_unpack_parts / sdqh_column_unpack2 / sdqh_table_export_bitmap on key sets and direct-layout tables,
    against numpy: every chunk of the packed buffer holds exactly its part's rows, column after column; taking the buffer apart
    gives the columns back; a key set's bitmap exported over a wider and a narrower, shifted range names exactly its keys.
    Returns a digest that is the same on both implementations of the ABI."""
    from sdqlpy_amd import abi
    rng = np.random.default_rng(seed)
    key = rng.integers(0, 5000, n).astype(np.int64) * 3 + 7
    val = rng.random(n)
    tag = np.arange(n, dtype=np.int64)
    kc, vc, tc = ctx.upload(key), ctx.upload(val), ctx.upload(tag)
    packed = ctx.alloc(3 * n, abi.I64)
    out = {}
    for mode, upper in (("hash", None), ("range", np.array([2000, 6000, 9000, 12000][:nparts - 1], np.int64))):
        counts = ctx.partition_pack(n, kc, nparts, [kc, vc, tc], packed.data_ptr(), range_upper=upper)
        assert int(counts.sum()) == n
        cols, m = ctx.unpack_parts(packed.data_ptr(), counts, [abi.I64, abi.F64, abi.I64])
        ctx.synchronize()
        assert m == n
        k2, v2, t2 = cols[0].download(0, n), cols[1].download(0, n), cols[2].download(0, n)
        assert np.array_equal(key[t2], k2) and np.array_equal(val[t2], v2) and np.array_equal(np.sort(t2), tag)     # rows intact, none lost
        off = np.concatenate([[0], np.cumsum(counts)])
        for p in range(nparts):
            seg = k2[off[p]:off[p + 1]]
            if upper is not None:
                lo = -1 if p == 0 else upper[p - 1]
                assert ((seg > lo) & ((seg <= upper[p]) if p < nparts - 1 else True)).all()
        out[mode] = (counts.tolist(), [sorted(t2[off[p]:off[p + 1]].tolist())[:5] for p in range(nparts)])
        for c in cols:
            c.free()
    # empty input
    counts = ctx.partition_pack(0, kc, nparts, [kc, vc], packed.data_ptr())
    assert counts.tolist() == [0] * nparts
    cols, m = ctx.unpack_parts(packed.data_ptr(), counts, [abi.I64, abi.F64])
    assert m == 0
    # packed composite keys -> their parts
    hi, lo = rng.integers(0, 1 << 32, 1000).astype(np.uint64), rng.integers(0, 1 << 32, 1000).astype(np.uint64)
    pk = ctx.upload(((hi << np.uint64(32)) | lo).view(np.int64))
    ch, cl = ctx.unpack2(pk, 1000)
    ctx.synchronize()
    assert np.array_equal(ch.download(0, 1000), hi.astype(np.int64)) and np.array_equal(cl.download(0, 1000), lo.astype(np.int64))
    # a key set's bitmap over other ranges (bitmap-only table: its own words, shifted), and a direct-layout table's
    present = sorted(set(key.tolist()))
    ks = ctx.build_key_set(n, abi.make_filter(), [], kc)
    direct = ctx.hash_build_unique(n, abi.make_filter(), [], kc, [vc])
    for t in (ks, direct):
        for lo_r, hi_r in ((7, 7 + 3 * 5000), (-100, 20011), (1000, 1999), (8, 8), (15001, 15100)):
            words = ctx.table_export_bitmap(t, lo_r, hi_r)
            t2 = ctx.table_from_bitmap(words, lo_r, hi_r)
            probe = ctx.upload(np.arange(lo_r - 3, hi_r + 4, dtype=np.int64))
            (hit,), nh = ctx.scan_compact(hi_r - lo_r + 7, abi.make_filter(), [(t2, probe)], [probe])
            ctx.synchronize()
            assert sorted(hit.download(0, nh).tolist()) == [k for k in present if lo_r <= k <= hi_r], (lo_r, hi_r)
    return out


def xcompact_case(ctx, n=70001, seed=33):
    """sdqh_xcompact against numpy: every passing row comes out (equal keys all stay — nothing is indexed), key and values as
    8-byte bit patterns; integer and double conditions, a key-set LOOKUP gate, an arithmetic value; nothing passing; no rows.
    Returns how many configurations were checked."""
    from sdqlpy_amd import abi
    rng = np.random.default_rng(seed)
    key = np.sort(rng.integers(100, 100 + max(n // 4, 1), n)).astype(np.int64)          # ~4 rows per key, clustered like l_orderkey
    d = rng.integers(0, 200, n).astype(np.int64)
    price = rng.integers(100, 100000, n) / 100.0
    disc = rng.integers(0, 11, n) / 100.0
    tag = np.arange(n, dtype=np.int64)
    members = np.unique(rng.integers(100, 100 + max(n // 4, 1), max(n // 16, 1))).astype(np.int64)
    kc, dc, pc, sc, tc, mc = ctx.upload(key), ctx.upload(d), ctx.upload(price), ctx.upload(disc), ctx.upload(tag), ctx.upload(members)
    kset = ctx.build_key_set(len(members), abi.make_filter(), [], mc)
    checked = 0
    for d_lo, p_hi, with_set in ((50, 1e9, True), (50, 500.0, False), (0, 1e9, True), (300, 1e9, False)):
        P = abi.Program()
        cd = P.op(abi.X_COL, abi.T_I64, col=dc)
        cp = P.op(abi.X_COL, abi.T_F64, col=pc)
        ck = P.op(abi.X_COL, abi.T_I64, col=kc)
        gates = [P.op(abi.X_GE, abi.T_BOOL, a=cd, b=P.op(abi.X_CONST, abi.T_I64, imm_i=d_lo)),
                 P.op(abi.X_LE, abi.T_BOOL, a=cp, b=P.op(abi.X_CONST, abi.T_F64, imm_f=p_hi))]
        mask = (d >= d_lo) & (price <= p_hi)
        if with_set:
            gates.append(P.op(abi.X_LOOKUP, abi.T_BOOL, a=ck, table=kset))
            mask &= np.isin(key, members)
        one = P.op(abi.X_CONST, abi.T_F64, imm_f=1.0)
        rev = P.op(abi.X_MUL, abi.T_F64, a=cp, b=P.op(abi.X_SUB, abi.T_F64, a=one, b=P.op(abi.X_COL, abi.T_F64, col=sc)))
        P.gates, P.key = gates, ck
        P.vals = [cp, rev, P.op(abi.X_COL, abi.T_I64, col=tc)]
        for rows in (n, 0):
            cols, m = ctx.xcompact(rows, P)
            ctx.synchronize()
            want = np.nonzero(mask[:rows])[0]
            assert m == len(want), (d_lo, p_hi, with_set, rows, m, len(want))
            if m:
                t = cols[3].download(0, m)
                order = np.argsort(t, kind="stable")
                assert np.array_equal(t[order], want)                                                  # each passing row once
                assert np.array_equal(cols[0].download(0, m)[order], key[want])
                assert np.array_equal(cols[1].download(0, m)[order].view(np.float64), price[want])
                assert np.array_equal(cols[2].download(0, m)[order].view(np.float64), price[want] * (1.0 - disc[want]))
            for c in cols:
                c.free()
            checked += 1
    kset.free()
    return checked


def lanes_case(lib, sf=0.05, rounds=6, queries=("q1", "q3", "q5", "q6", "q9", "q4", "q14"), rel=0.0, threads=1, tight=False):
    """Engine lanes (contexts of one family, abi.Context.fork): the queries launched together, round after round, on an engine with
    three lanes against the same plans on an engine with one — same rows every round; the lanes really were used; the family's
    options follow the first context's; columns forgotten and uploaded again between rounds (twins and dictionaries are rebuilt by
    whichever lane gets there first).  Returns the lanes the plans ran on."""
    db = tpch.generate(sf, tables=sorted(tpch.columns_for(queries)), columns=tpch.columns_for(queries))
    one = eng_mod.Engine(lib.context(threads=threads))
    many = eng_mod.Engine(lib.context(threads=threads))
    one.nlanes, many.nlanes = 1, 3
    try:
        if tight:
            for e in (one, many):
                e.ctx.set_option("feature_min_rows", 0)
        plans = {q: frontend.lower_function(Q.QUERIES[q].__sdql_func__, Q.QUERIES[q].__sdql_in_type__) for q in queries}
        args = {q: [db[t] for t in Q.QUERY_TABLES[q]] for q in queries}

        def rows_of(r):
            if hasattr(r, "wait"):
                r.wait()
            if hasattr(r, "columns"):
                return sorted(zip(*[r.column(c).tolist() for c in r.columns]))
            return r
        want = {q: rows_of(eng_mod.execute_plan(one, plans[q], args[q])) for q in queries}
        used = set()
        for rnd in range(rounds):
            launched = [(q, eng_mod.execute_plan(many, plans[q], args[q])) for q in queries]            # all in flight, then read in reverse
            for q, r in reversed(launched):
                got = rows_of(r)
                if hasattr(r, "columns"):
                    assert_rows_match(got, want[q], rel, "lanes round %d %s" % (rnd, q))
                else:
                    assert got == want[q] or abs(got - want[q]) <= rel * abs(want[q]), (q, got, want[q])
            used |= {eng_mod.prepared_plan(many, plans[q], args[q]).eng.lane_index for q in queries}
            if rnd == rounds // 2:
                many.clear()                                                 # columns, twins, dictionaries gone: the next round uploads again
            if rnd == 1 and tight:
                many.ctx.set_option("feature_min_rows", 1 << 20)             # ... the forks follow
                assert all(c._options.get("feature_min_rows") == 1 << 20 for c in many.ctx.forks)
                many.ctx.set_option("feature_min_rows", 0)
        assert len(many.ctx.forks) == 2 and used == {0, 1, 2}, (len(many.ctx.forks), used)
        return used
    finally:
        many.close()
        one.close()


def plan_graphs_case(lib, sf=0.05, rounds=8, queries=("q1", "q3", "q5", "q6", "q9", "q4", "q14", "q18"), rel=0.0):
    """Plan graphs (engine.PlanGraph over sdqh_graph_*): a settled plan's device calls recorded once and launched by ONE call from then
    on.  The same queries on an engine that records (always: also while other results are in flight) against one that issues every
    call — same rows, round after round: two results of every query in flight at once (two recordings, then the calls), results read
    late and in reverse, results dropped unread, an option changed in between (the recordings made under the old setting go), the
    columns forgotten and uploaded again.  Returns the recording engine's counters."""
    db = tpch.generate(sf, tables=sorted(tpch.columns_for(queries)), columns=tpch.columns_for(queries))
    calls = eng_mod.Engine(lib.context())
    graphs = eng_mod.Engine(lib.context())
    calls.plan_graphs = 0
    graphs.plan_graphs, graphs.plan_graphs_always = 2, True
    try:
        plans = {q: frontend.lower_function(Q.QUERIES[q].__sdql_func__, Q.QUERIES[q].__sdql_in_type__) for q in queries}
        args = {q: [db[t] for t in Q.QUERY_TABLES[q]] for q in queries}

        def rows_of(r):
            if hasattr(r, "wait"):
                r.wait()
            if hasattr(r, "columns"):
                return sorted(zip(*[r.column(c).tolist() for c in r.columns]))
            return r
        want = {q: rows_of(eng_mod.execute_plan(calls, plans[q], args[q])) for q in queries}
        for rnd in range(rounds):
            launched = [(q, eng_mod.execute_plan(graphs, plans[q], args[q])) for q in queries for _ in range(3 if rnd % 2 else 1)]
            if rnd == 3:
                del launched[::2]                                            # dropped unread: their recordings are reusable once the lane has been waited for
            for q, r in reversed(launched):
                got = rows_of(r)
                if hasattr(r, "columns"):
                    assert_rows_match(got, want[q], rel, "plan graphs round %d %s" % (rnd, q))
                else:
                    assert got == want[q] or abs(got - want[q]) <= rel * abs(want[q]), (q, got, want[q])
            del launched
            if rnd == 4:
                graphs.ctx.set_option("narrow", 0)                           # other kernels from here on: what was recorded is void
            if rnd == 5:
                graphs.ctx.set_option("narrow", 1)
            if rnd == 6:
                graphs.clear()
        return dict(graphs.graph_stats)
    finally:
        graphs.close()
        calls.close()


def grouped_index_case(ctx, n=200000, nprobe=500000, seed=17, keep=0.3, per_a=4):
    """A composite-key build (a, b) -> payload over a table stored in the order of a (sdqh_build, nkey 2: the GROUPED layout on the GPU,
    DevTable) with a build filter, duplicate (a, b) pairs (first build row wins), runs of one a that cross wave segments, first parts
    whose every row is filtered out; probed through sdqh_lookup_aggregate (sum of the matched payload per small group, hit count)
    and compacted (entries, size), against numpy.  Returns a digest that is the same on every implementation and layout."""
    from sdqlpy_amd import abi
    rng = np.random.default_rng(seed)
    a = np.sort(rng.integers(5, 5 + max(n // per_a, 1), n)).astype(np.int64)              # non-decreasing, ~per_a rows per value, gaps
    b = rng.integers(1, 50, n).astype(np.int64)                                             # few values: duplicates of (a, b) are common
    cost = rng.integers(1, 1000, n).astype(np.float64)
    flag = (rng.random(n) < keep).astype(np.int64)
    ca, cb, cc, cf = ctx.upload(a), ctx.upload(b), ctx.upload(cost), ctx.upload(flag)
    t = ctx.build(n, abi.make_filter(ipreds=[(cf, 1, 1)]), [], [abi.src_col(ca), abi.src_col(cb)], [abi.src_col(cc)])
    # the reference semantics: first passing row of a key wins
    first = {}
    for i in np.nonzero(flag)[0]:
        first.setdefault((int(a[i]), int(b[i])), i)
    assert t.size() == len(first)
    pa = rng.integers(0, 10 + max(n // per_a, 1), nprobe).astype(np.int64)
    pb = rng.integers(0, 52, nprobe).astype(np.int64)
    grp = rng.integers(0, 7, nprobe).astype(np.int64)
    one = np.ones(nprobe, np.float64)
    cpa, cpb, cg, c1 = ctx.upload(pa), ctx.upload(pb), ctx.upload(grp), ctx.upload(one)
    keys, vals, cnts = ctx.lookup_aggregate(nprobe, abi.make_filter(), [(t, [abi.src_col(cpa), abi.src_col(cpb)])], [abi.src_col(cg)],
                                            abi.TUPLE_AB, [abi.src_lookup(0, 0), abi.src_col(c1)])
    want_sum, want_cnt = np.zeros(7), np.zeros(7, np.int64)
    fa = np.array([k[0] for k in first], np.int64); fb = np.array([k[1] for k in first], np.int64); fi = np.array(list(first.values()), np.int64)
    look = dict(zip(zip(fa.tolist(), fb.tolist()), fi.tolist()))
    for x, y, g in zip(pa.tolist(), pb.tolist(), grp.tolist()):
        i = look.get((x, y))
        if i is not None:
            want_sum[g] += cost[i]; want_cnt[g] += 1
    got = {int(k[0]): (float(v[0]), int(c)) for k, v, c in zip(keys, vals, cnts)}
    for g in range(7):
        if want_cnt[g]:
            assert got[g][1] == want_cnt[g] and abs(got[g][0] - want_sum[g]) <= 1e-9 * want_sum[g], (g, got.get(g), want_sum[g], want_cnt[g])
    assert sum(1 for g in range(7) if want_cnt[g]) == len(got)
    cols, m = ctx.table_entries(t)
    ctx.synchronize()
    ek, ec = cols[0].download(0, m), cols[1].download(0, m).view(np.float64)
    assert m == len(first)
    want_keys = np.sort((fa << 32) | fb)
    order = np.argsort(ek)
    assert np.array_equal(ek[order], want_keys)
    assert np.array_equal(ec[order], cost[fi[np.argsort((fa << 32) | fb)]])
    for c in cols + [ca, cb, cc, cf, cpa, cpb, cg, c1]:
        c.free()
    t.free()
    return int(want_cnt.sum()), float(want_sum.sum()), len(first)


def cluster_pack_case(ctx, n=40000, nprobe=300001, seed=23, nparts=9000, keep=0.06):
    """Q9's final loop through the ABI (sdqh_lookup_aggregate): a first lookup on a composite (part, supplier) table — built over the
    kept parts only, stored in part order, suppliers from a range wide enough that the key rectangle is not linearised — keyed by a
    probe column in NO row order, a second lookup supplier -> nation, six gathered columns (so the loop has a row pack), groups
    (nation, year-like column), value a * (1 - b) - cost * d.  On the GPU the pack is CLUSTERED by the part key (sdqh_aux.hip:
    cluster_pack_build) unless the option says otherwise.  Checked against numpy here; returns (group keys, counts, sums) for
    comparisons between implementations and options."""
    from sdqlpy_amd import abi
    rng = np.random.default_rng(seed)
    nsup = 50021                                                                          # (prime: a part's suppliers below are all different)
    per = max(n // nparts, 1)
    a = np.repeat(np.arange(3, 3 + nparts, dtype=np.int64), per)[:n]
    n = len(a)
    b = (a * 31 + (np.arange(n, dtype=np.int64) % per) * 7919) % nsup + 1
    cost = np.round(rng.random(n) * 1000.0, 2)
    kept_part = rng.random(nparts + 3) < keep
    flag = kept_part[a].astype(np.int64)
    sup_key = np.arange(nsup + 1, dtype=np.int64)
    sup_nat = rng.integers(0, 25, nsup + 1).astype(np.int64)
    ca, cb, cc, cf, csk, csn = (ctx.upload(x) for x in (a, b, cost, flag, sup_key, sup_nat))
    t0 = ctx.build(n, abi.make_filter(ipreds=[(cf, 1, 1)]), [], [abi.src_col(ca), abi.src_col(cb)], [abi.src_col(cc)])
    t1 = ctx.build(nsup + 1, abi.make_filter(), [], [abi.src_col(csk)], [abi.src_col(csn)])
    pa = rng.integers(0, nparts + 6, nprobe).astype(np.int64)                             # no order; some outside the table's range
    pb = (pa * 31 + rng.integers(0, per + 1, nprobe) * 7919) % nsup + 1                   # mostly a supplier of that part
    yr = rng.integers(1992, 1999, nprobe).astype(np.int64)
    v1 = np.round(rng.random(nprobe) * 90000.0 + 900.0, 2)
    v2 = np.round(rng.integers(0, 11, nprobe) * 0.01, 2)
    v3 = rng.integers(1, 51, nprobe).astype(np.float64)
    cols = [ctx.upload(x) for x in (pa, pb, yr, v1, v2, v3)]
    cpa, cpb, cyr, cv1, cv2, cv3 = cols
    keys, vals, cnts = ctx.lookup_aggregate(nprobe, abi.make_filter(), [(t0, [abi.src_col(cpa), abi.src_col(cpb)]), (t1, [abi.src_col(cpb)])],
                                            [abi.src_lookup(1, 0), abi.src_col(cyr)], abi.TUPLE_A_1MB_M_CD,
                                            [abi.src_col(cv1), abi.src_col(cv2), abi.src_lookup(0, 0), abi.src_col(cv3)])
    # numpy: (part, supplier) -> cost of the kept rows (every pair is unique)
    live = np.flatnonzero(flag)
    packed = (a[live] << 32) | b[live]
    assert len(np.unique(packed)) == len(packed)
    order = np.argsort(packed)
    pk = (pa << 32) | pb
    at = np.searchsorted(packed[order], pk)
    at = np.minimum(at, max(len(live) - 1, 0))
    hit = (packed[order][at] == pk) if len(live) else np.zeros(nprobe, bool)
    c = cost[live][order][at] if len(live) else np.zeros(nprobe)
    val = v1 * (1.0 - v2) - c * v3
    g = sup_nat[pb] * 10000 + yr
    want = {}
    for gg, vv in zip(g[hit].tolist(), val[hit].tolist()):
        s0, n0 = want.get(gg, (0.0, 0))
        want[gg] = (s0 + vv, n0 + 1)
    got = {int(k[0]) * 10000 + int(k[1]): (float(v[0]), int(cn)) for k, v, cn in zip(keys, vals, cnts)}
    assert sorted(got) == sorted(want), (len(got), len(want))
    for gg, (s0, n0) in want.items():
        assert got[gg][1] == n0 and abs(got[gg][0] - s0) <= 1e-9 * max(abs(s0), 1.0), (gg, got[gg], s0, n0)
    for col in cols + [ca, cb, cc, cf, csk, csn]:
        col.free()
    t0.free(); t1.free()
    ks = sorted(got)
    return ks, [got[k][1] for k in ks], [got[k][0] for k in ks]


def keyed_probe_case(ctx, keys, members, groups=5, seed=3):
    """A row program over a probe table of `keys` (any int64 key column): `if set[key] != None: out[g] += v` — the loop shape whose
    streamed key the GPU reads through a delta twin where the column's 8-row groups are narrow, and which it runs as a DRIVEN walk where
    the column is stored in its own order and the set holds few of its values.  Members = the key set.  Checked against numpy; returns
    (group keys, counts, sums)."""
    from sdqlpy_amd import abi as A
    rng = np.random.default_rng(seed)
    n = len(keys)
    g = rng.integers(0, groups, n).astype(np.int64)
    v = np.round(rng.random(n) * 100.0, 2)
    ck, cg, cv, cm = ctx.upload(np.ascontiguousarray(keys, np.int64)), ctx.upload(g), ctx.upload(v), ctx.upload(np.ascontiguousarray(members, np.int64))
    lo, hi = (int(min(members.min(), keys.min())), int(max(members.max(), keys.max()))) if len(members) and n else (0, 1)
    P = A.Program()
    P.key = P.op(A.X_COL, A.T_I64, col=cm)
    t_set = ctx.xkey_set(len(members), P, lo, hi)
    P = A.Program()
    lk = P.op(A.X_LOOKUP, A.T_BOOL, a=P.op(A.X_COL, A.T_I64, col=ck), table=t_set)
    P.gates = [lk]
    P.key = P.op(A.X_COL, A.T_I64, col=cg)
    P.vals = [P.op(A.X_COL, A.T_F64, col=cv)]
    out = None
    for _ in range(2):                                                                  # (the second run finds the twins / indexes made)
        gk, gv, gc = ctx.xgroupby(n, P)
        order = np.argsort(gk)
        got = (gk[order].tolist(), gc[order].tolist(), gv[order, 0].tolist())
        assert out is None or (got[0] == out[0] and got[1] == out[1])
        out = got
    hit = np.isin(keys, members)
    want_c = np.bincount(g[hit], minlength=groups)
    want_s = np.bincount(g[hit], weights=v[hit], minlength=groups)
    ks = [k for k in range(groups) if want_c[k]]
    assert out[0] == ks and out[1] == [int(want_c[k]) for k in ks], (out[0], ks)
    assert all(abs(x - want_s[k]) <= 1e-9 * max(abs(want_s[k]), 1.0) for x, k in zip(out[2], ks))
    for c in (ck, cg, cv, cm):
        c.free()
    t_set.free()
    return out
