#!/usr/bin/env python3
"""Generate the golden parity fixtures by running the REFERENCE itself (Python mode).

Runs only in the build container, where the reference checkout is mounted at /root/reference; the
GPU box never sees it.  What is committed is data: for every case the generator parameters and an
input fingerprint, and the reference's result for TPCH q1/q3/q5/q6/q9.

How the expected values are produced (nothing of the reference is copied into this repository):
  * `sdqlpy.sdql_lib` is imported from /root/reference/src, unmodified, with sdqlpy_init(0, 1)
    — the reference's own row-by-row interpreter (reference src/sdqlpy/sdql_lib.py:207-266).
  * the query functions q1, q3, q5, q6, q9 are taken from the reference's own TPCH script
    (reference test/test_all.py:46-62,145-176,215-281,285-295,431-491) by parsing that file with
    `ast` at run time and exec-ing just those FunctionDefs against the reference module.
  * inputs come from this repository's deterministic generator (sdqlpy_amd/tpch.py) and are handed
    to the reference in its own columnar container form (reference sdql_lib.py:115).

Doubles are stored as C99 hex floats (exact).  Result rows are sorted.  A query the reference
cannot finish on a degenerate input (its interpreter raises on an empty aggregate) is recorded as
an empty result, which is what its compiled mode returns for the same input.

    python tests/golden/make_golden.py            # rewrites tests/golden/tpch_golden.json
"""
import ast
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_ROOT = "/root/reference"

from sdqlpy_amd import tpch  # noqa: E402

QUERIES = ["q1", "q3", "q5", "q6", "q9"]
MORE_QUERIES = ["q4", "q10", "q14", "q18"]       # beyond the configured five (test/test_all.py:180-211, 495-558, 695-716, 874-913)
QUERY_TABLES = {   # positional argument order of each reference query (test/test_all.py decorators)
    "q1": ["lineitem"],
    "q3": ["lineitem", "customer", "orders"],
    "q5": ["lineitem", "customer", "orders", "region", "nation", "supplier"],
    "q6": ["lineitem"],
    "q9": ["lineitem", "orders", "nation", "supplier", "part", "partsupp"],
    "q4": ["orders", "lineitem"],
    "q14": ["lineitem", "part"],
    "q18": ["lineitem", "customer", "orders"],
    "q10": ["customer", "orders", "lineitem", "nation"],
}
# SURVEY.md §8f.3, second step: queries that need the open expression vocabulary (test/test_all.py:299-367, 371-427,
# 656-691, 720-756, 831-870, 917-981, 985-1028, 1113-1183).  Argument order = the reference's decorators.
WIDE_QUERIES = ["q2", "q7", "q8", "q11", "q12", "q13", "q15", "q16", "q17", "q19", "q20", "q22"]
QUERY_TABLES.update({
    "q7": ["supplier", "lineitem", "orders", "customer", "nation"],
    "q8": ["part", "supplier", "lineitem", "orders", "customer", "nation", "region"],
    "q2": ["part", "supplier", "partsupp", "nation", "region"],
    "q11": ["partsupp", "supplier", "nation"],
    "q12": ["orders", "lineitem"],
    "q13": ["customer", "orders"],
    "q16": ["partsupp", "part", "supplier"],
    "q15": ["lineitem", "supplier"],
    "q17": ["lineitem", "part"],
    "q19": ["lineitem", "part"],
    "q20": ["supplier", "nation", "partsupp", "part", "lineitem"],
    "q22": ["customer", "orders"],
})
ALL_TABLES = ["lineitem", "customer", "orders", "region", "nation", "supplier", "part", "partsupp"]

# ---- input variants: edge cases the parity tests must cover -------------------------------------

def v_identity(db):
    return db


def _replace(db, table, **cols):
    c = db[table].getContainer()
    data = list(c["data"])
    for name, arr in cols.items():
        data[c["headers"].index(name)] = arr
    out = dict(db)
    out[table] = tpch.table_from_columns(c["headers"], data)
    return out


def v_nothing_passes(db):
    """every lineitem ships in 1999, every order is dated 1999: all date filters fail / all pass."""
    n = len(tpch.column(db["lineitem"], "l_shipdate"))
    db = _replace(db, "lineitem", l_shipdate=np.full(n, 19990101, np.int64))
    return db


def v_no_building(db):
    """build side with zero survivors (q3 build 1), probe with no hits downstream."""
    seg = tpch.column(db["customer"], "c_mktsegment").copy()
    seg[:] = "MACHINERY"
    return _replace(db, "customer", c_mktsegment=seg)


def v_one_group(db):
    """all rows fall into one q1 group."""
    n = len(tpch.column(db["lineitem"], "l_shipdate"))
    return _replace(db, "lineitem", l_returnflag=np.full(n, "N", "<U1"), l_linestatus=np.full(n, "O", "<U1"))


def v_big_keys(db):
    """keys far outside 32 bits (and one at INT64_MAX-ish) to exercise hashing of wide keys."""
    off_o, off_c = np.int64(1) << 40, np.int64(1) << 35
    ok = tpch.column(db["orders"], "o_orderkey") + off_o
    lk = tpch.column(db["lineitem"], "l_orderkey") + off_o
    # move the last order (and its lines) to the top of the int64 range
    last = ok[-1]
    top = np.int64(np.iinfo(np.int64).max - 7)
    lk = np.where(lk == last, top, lk)
    ok = ok.copy(); ok[-1] = top
    db = _replace(db, "orders", o_orderkey=ok, o_custkey=tpch.column(db["orders"], "o_custkey") + off_c)
    db = _replace(db, "lineitem", l_orderkey=lk)
    db = _replace(db, "customer", c_custkey=tpch.column(db["customer"], "c_custkey") + off_c)
    return db


def v_wide_doubles(db):
    """prices spanning ~1e-3 .. 1e12 (still exact 2-decimal text is not needed here) and zero discounts."""
    ep = tpch.column(db["lineitem"], "l_extendedprice").copy()
    n = len(ep)
    scale = np.array([1e-3, 1.0, 1e3, 1e9], np.float64)[np.arange(n) % 4]
    return _replace(db, "lineitem", l_extendedprice=ep * scale)


def v_big_orders(db):
    """quantities x6: many orders pass Q18's HAVING sum(l_quantity) > 300 (on plain data almost none do)."""
    q = tpch.column(db["lineitem"], "l_quantity")
    return _replace(db, "lineitem", l_quantity=q * 6.0)


def v_signed_denormal(db):
    """negative prices (returns / credits), exact zeros, denormals and values next to DBL_MAX/1e300
    scale in the summed columns: sign handling of the order-preserving maps and non-finite-free
    extremes of SUM(double)."""
    ep = tpch.column(db["lineitem"], "l_extendedprice").copy()
    n = len(ep)
    pattern = np.array([1.0, -1.0, 0.0, 5e-324, -2.5e-310, 1e150, 2e150, -3.0], np.float64)[np.arange(n) % 8]    # huge terms share a sign: no catastrophic cancellation
    ep = np.where(pattern == 0.0, 0.0, np.where(np.abs(pattern) < 1e-300, pattern, np.where(np.abs(pattern) > 1e100, pattern, ep * pattern)))
    return _replace(db, "lineitem", l_extendedprice=ep)


VARIANTS = {
    "base": v_identity,
    "nothing_passes": v_nothing_passes,
    "no_building": v_no_building,
    "one_group": v_one_group,
    "big_keys": v_big_keys,
    "wide_doubles": v_wide_doubles,
    "signed_denormal": v_signed_denormal,
    "big_orders": v_big_orders,
}

# (name, sf, variant, queries)
CASES = [
    ("tiny", 0.0003, "base", QUERIES),
    ("tiny_nothing_passes", 0.0003, "nothing_passes", ["q1", "q3", "q6"]),
    ("tiny_no_building", 0.0003, "no_building", ["q3"]),
    ("tiny_one_group", 0.0003, "one_group", ["q1"]),
    ("tiny_big_keys", 0.0003, "big_keys", ["q3"]),
    ("tiny_wide_doubles", 0.0003, "wide_doubles", ["q1", "q3", "q6"]),
    ("small", 0.01, "base", QUERIES),
    ("small_big_keys", 0.01, "big_keys", ["q3"]),
    ("medium", 0.1, "base", QUERIES),
]


# ---- the reference, imported unmodified ---------------------------------------------------------

def load_reference():
    sys.path.insert(0, os.path.join(REF_ROOT, "src"))
    import sdqlpy.sdql_lib as ref
    ref.sdqlpy_init(0, 1)
    src = open(os.path.join(REF_ROOT, "test", "test_all.py")).read()
    tree = ast.parse(src)
    ns = {k: getattr(ref, k) for k in dir(ref) if not k.startswith("__")}
    # the schema dicts the decorators mention (only used as decorator arguments; mode 0 ignores them)
    for node in tree.body:
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id.endswith("_type"):
            exec(compile(ast.Module([node], []), "test_all.py", "exec"), ns)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in QUERIES + MORE_QUERIES + WIDE_QUERIES:
            if node.name == "q15":
                # The reference's Q15 compares each supplier's revenue with the constant 1772627.2087 — the answer
                # for dbgen's SF=1 data, typed in by hand (test/test_all.py:733; the `max` it stands for is commented
                # out on the next line).  On any other data that selects nothing, so the ONE statement that binds
                # the constant is replaced by the maximum of the dictionary the query has just computed.
                for i, st in enumerate(node.body):
                    if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name) and st.targets[0].id == "max_revenue":
                        node.body[i] = ast.copy_location(ast.parse("max_revenue = max(li_aggr.getContainer().values())").body[0], st)
                ast.fix_missing_locations(node)
            exec(compile(ast.Module([node], []), "test_all.py", "exec"), ns)
    return ref, {q: ns[q] for q in QUERIES + MORE_QUERIES + WIDE_QUERIES}


def to_ref_table(ref, table):
    c = table.getContainer()
    return ref.sr_dict({"headers": list(c["headers"]), "data": list(c["data"])}, None, True)


def enc(v):
    if hasattr(v, "getContainer") and not v.getContainer():
        return {"f": (0.0).hex()}            # a scalar sum over no rows is the empty dictionary in the interpreter: the semiring zero
    if isinstance(v, (float, np.floating)):
        return {"f": float(v).hex()}
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (str, np.str_)):
        return str(v)
    if isinstance(v, (bool, np.bool_)):
        return bool(v)
    raise TypeError(type(v))


def sort_key(row):
    return [(0, x) if isinstance(x, int) else (1, x) if isinstance(x, str) else (2, float.fromhex(x["f"])) for x in row]


def encode_result(ref, res):
    if res is None:
        return {"kind": "set", "columns": [], "rows": []}
    if isinstance(res, (float, np.floating, int, np.integer)):
        return {"kind": "scalar", "value": enc(res)}
    cont = res.getContainer()
    cols, rows = None, []
    for rec in cont.keys():
        fields = rec.getContainer()
        if cols is None:
            cols = list(fields.keys())
        rows.append([enc(fields[c]) for c in cols])
    rows.sort(key=sort_key)
    return {"kind": "set", "columns": cols or [], "rows": rows}


MORE_CASES = [
    ("tiny", 0.0003, "base", MORE_QUERIES),
    ("tiny_nothing_passes", 0.0003, "nothing_passes", ["q14"]),
    ("tiny_big_orders", 0.0003, "big_orders", ["q18"]),
    ("small_big_orders", 0.01, "big_orders", ["q18"]),
    ("tiny_signed_denormal", 0.0003, "signed_denormal", ["q1", "q3", "q6", "q14"]),
    ("small_signed_denormal", 0.01, "signed_denormal", ["q1", "q6"]),
    ("small", 0.01, "base", MORE_QUERIES),
    ("medium", 0.05, "base", MORE_QUERIES),
]


WIDE_CASES = [
    ("tiny", 0.0003, "base", WIDE_QUERIES),
    ("small", 0.01, "base", WIDE_QUERIES),
    ("medium", 0.1, "base", WIDE_QUERIES),
]


# SF=1 (6 M lineitem rows) and big keys at SF=0.1: the sizes at which the product's size-dependent paths switch on (device loops over
# result dictionaries, layout choices, walks, delta twins) — an hour of the reference's interpreter, spread over worker processes,
# one (case, query) each.   python tests/golden/make_golden.py --sf1 [--jobs 6]  -> tpch_golden_sf1.json.gz
SF1_CASES = [
    ("sf1", 1.0, "base", QUERIES + MORE_QUERIES + WIDE_QUERIES),
    ("sf01_big_keys", 0.1, "big_keys", ["q3"]),
    ("sf1_big_keys", 1.0, "big_keys", ["q3"]),
]


def _one(job):
    """One (case, query) in a process of its own: the reference imported there, the inputs generated there."""
    name, sf, variant, q = job
    ref, queries = load_reference()
    tables = sorted(set(QUERY_TABLES[q]))
    base = tpch.generate(sf, tpch.DEFAULT_SEED, tables=tables, columns=tpch.columns_for([q]), threads=2)
    db = VARIANTS[variant](base)
    t0 = time.time()
    args = [to_ref_table(ref, db[t]) for t in QUERY_TABLES[q]]
    res = queries[q](*args)
    r = encode_result(ref, res)
    r["reference_seconds"] = round(time.time() - t0, 1)
    print("%-16s %s  %7.1fs  %s" % (name, q, time.time() - t0, r["value"] if r["kind"] == "scalar" else "%d rows" % len(r["rows"])), flush=True)
    return name, q, r


# BASELINE.json's own size: the five configured queries at SF=10 (60 M lineitem rows) through the reference's interpreter — q1 alone takes it
# a quarter of an hour.   python tests/golden/make_golden.py --sf10 [--jobs 5]  -> tpch_golden_sf10.json.gz (q3's 113 K rows: 1.5 MB compressed)
SF10_CASES = [
    ("sf10", 10.0, "base", QUERIES),
]


def main_sf1():
    import multiprocessing as mp
    global SF1_CASES
    sf10 = "--sf10" in sys.argv
    if sf10:
        SF1_CASES = SF10_CASES
    jobs_n = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else 6
    jobs = [(name, sf, variant, q) for name, sf, variant, qs in SF1_CASES for q in qs]
    with mp.get_context("spawn").Pool(jobs_n, maxtasksperchild=1) as pool:
        done = pool.map(_one, jobs, chunksize=1)
    out = {"meta": {"generator_seed": tpch.DEFAULT_SEED,
                    "reference": "edin-dal/sdqlpy Python mode (sdqlpy_init(0,1)), queries from test/test_all.py",
                    "made_by": "tests/golden/make_golden.py " + ("--sf10" if "--sf10" in sys.argv else "--sf1")},
           "cases": []}
    for name, sf, variant, qs in SF1_CASES:
        tables = sorted({t for q in qs for t in QUERY_TABLES[q]})
        base = tpch.generate(sf, tpch.DEFAULT_SEED, tables=tables, columns=tpch.columns_for(qs), threads=4)
        db = VARIANTS[variant](base)
        case = {"name": name, "sf": sf, "seed": tpch.DEFAULT_SEED, "variant": variant,
                "tables": tables, "fingerprint": tpch.fingerprint(db),
                "rows": {t: len(db[t].getContainer()["data"][0]) for t in tables},
                "results": {q: r for n, q, r in done if n == name}}
        out["cases"].append(case)
    import gzip
    path = os.path.join(HERE, "tpch_golden_sf10.json.gz" if sf10 else "tpch_golden_sf1.json.gz")          # (SF=1: q3 11 K rows, q10 39 K, q16 18 K — 9 MB of JSON, 1.8 MB compressed)
    with gzip.GzipFile(path, "wb", mtime=0) as fh:
        fh.write(json.dumps(out, separators=(",", ":")).encode())
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    if "--sf1" in sys.argv or "--sf10" in sys.argv:
        return main_sf1()
    more = "--more" in sys.argv          # python tests/golden/make_golden.py --more  -> tpch_golden_more.json (q4, q14)
    wide = "--wide" in sys.argv          # python tests/golden/make_golden.py --wide  -> tpch_golden_wide.json (q7, q8, q13, q15, q17, q19, q20, q22)
    cases = WIDE_CASES if wide else MORE_CASES if more else CASES
    ref, queries = load_reference()
    out = {"meta": {"generator_seed": tpch.DEFAULT_SEED,
                    "reference": "edin-dal/sdqlpy Python mode (sdqlpy_init(0,1)), queries from test/test_all.py",
                    "made_by": "tests/golden/make_golden.py"},
           "cases": []}
    for name, sf, variant, qs in cases:
        tables = sorted({t for q in qs for t in QUERY_TABLES[q]})
        base = tpch.generate(sf, tpch.DEFAULT_SEED, tables=tables, columns=tpch.columns_for(qs), threads=4)
        db = VARIANTS[variant](base)
        case = {"name": name, "sf": sf, "seed": tpch.DEFAULT_SEED, "variant": variant,
                "tables": tables, "fingerprint": tpch.fingerprint(db),
                "rows": {t: len(db[t].getContainer()["data"][0]) for t in tables}, "results": {}}
        for q in qs:
            t0 = time.time()
            args = [to_ref_table(ref, db[t]) for t in QUERY_TABLES[q]]
            try:
                res = queries[q](*args)
                note = None
            except (AttributeError, TypeError, ValueError) as exc:   # the interpreter cannot sum an empty aggregate
                res, note = None, "reference raised %s: recorded as empty" % type(exc).__name__
            r = encode_result(ref, res)
            if note:
                r["note"] = note
            case["results"][q] = r
            print("%-22s %s  %6.1fs  %s" % (name, q, time.time() - t0,
                                           r["value"] if r["kind"] == "scalar" else "%d rows" % len(r["rows"])), flush=True)
        out["cases"].append(case)
    path = os.path.join(HERE, "tpch_golden_wide.json" if wide else "tpch_golden_more.json" if more else "tpch_golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
