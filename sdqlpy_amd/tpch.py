"""TPCH schemas and the deterministic TPCH-shaped data generator (Python face of csrc/tpchgen.cpp).

The schemas carry the field names and types of the reference's TPCH script (reference
test/test_all.py:26-33), so a query written against them binds columns by the same names.
Generated tables hold only the columns the generator produces (the numeric / date / key columns
and the short strings the hot-path queries touch) — columns are bound by header name, never by
position, so narrower tables are fine.  ``write_tbl`` emits full-width dbgen-style '|' text with
filler text for the remaining schema columns, for loader tests and for feeding the reference.
"""
import ctypes
import os

import numpy as np

from . import build
from .sdql_lib import date, record, string, table_from_columns

DEFAULT_SEED = 20240607

lineitem_type = {record({"l_orderkey": int, "l_partkey": int, "l_suppkey": int, "l_linenumber": int,
                         "l_quantity": float, "l_extendedprice": float, "l_discount": float, "l_tax": float,
                         "l_returnflag": string(1), "l_linestatus": string(1), "l_shipdate": date,
                         "l_commitdate": date, "l_receiptdate": date, "l_shipinstruct": string(25),
                         "l_shipmode": string(10), "l_comment": string(44), "l_NA": string(1)}): bool}
customer_type = {record({"c_custkey": int, "c_name": string(25), "c_address": string(40), "c_nationkey": int,
                         "c_phone": string(15), "c_acctbal": float, "c_mktsegment": string(10),
                         "c_comment": string(117), "c_NA": string(1)}): bool}
order_type = {record({"o_orderkey": int, "o_custkey": int, "o_orderstatus": string(1), "o_totalprice": float,
                      "o_orderdate": date, "o_orderpriority": string(15), "o_clerk": string(15),
                      "o_shippriority": int, "o_comment": string(79), "o_NA": string(1)}): bool}
nation_type = {record({"n_nationkey": int, "n_name": string(25), "n_regionkey": int, "n_comment": string(152),
                       "n_NA": string(1)}): bool}
region_type = {record({"r_regionkey": int, "r_name": string(25), "r_comment": string(152), "r_NA": string(1)}): bool}
part_type = {record({"p_partkey": int, "p_name": string(55), "p_mfgr": string(25), "p_brand": string(10),
                     "p_type": string(25), "p_size": int, "p_container": string(10), "p_retailprice": float,
                     "p_comment": string(23), "p_NA": string(1)}): bool}
partsupp_type = {record({"ps_partkey": int, "ps_suppkey": int, "ps_availqty": float, "ps_supplycost": float,
                         "ps_comment": string(199), "ps_NA": string(1)}): bool}
supplier_type = {record({"s_suppkey": int, "s_name": string(25), "s_address": string(40), "s_nationkey": int,
                         "s_phone": string(15), "s_acctbal": float, "s_comment": string(101), "s_NA": string(1)}): bool}

SCHEMAS = {"lineitem": lineitem_type, "customer": customer_type, "orders": order_type, "nation": nation_type,
           "region": region_type, "part": part_type, "partsupp": partsupp_type, "supplier": supplier_type}

T_REGION, T_NATION, T_SUPPLIER, T_CUSTOMER, T_PART, T_PARTSUPP, T_ORDERS, T_LINEITEM = range(1, 9)

# columns the native generator produces, per table, with their numpy dtype
GENERATED = {
    "region": [("r_regionkey", "i8"), ("r_name", "U25")],
    "nation": [("n_nationkey", "i8"), ("n_name", "U25"), ("n_regionkey", "i8")],
    "supplier": [("s_suppkey", "i8"), ("s_nationkey", "i8"), ("s_acctbal", "f8")],
    "customer": [("c_custkey", "i8"), ("c_nationkey", "i8"), ("c_acctbal", "f8"), ("c_mktsegment", "U10")],
    "part": [("p_partkey", "i8"), ("p_name", "U55"), ("p_retailprice", "f8"), ("p_size", "i8")],
    "partsupp": [("ps_partkey", "i8"), ("ps_suppkey", "i8"), ("ps_availqty", "f8"), ("ps_supplycost", "f8")],
    "orders": [("o_orderkey", "i8"), ("o_custkey", "i8"), ("o_orderdate", "i8"), ("o_shippriority", "i8"),
               ("o_totalprice", "f8")],
    "lineitem": [("l_orderkey", "i8"), ("l_partkey", "i8"), ("l_suppkey", "i8"), ("l_linenumber", "i8"),
                 ("l_quantity", "f8"), ("l_extendedprice", "f8"), ("l_discount", "f8"), ("l_tax", "f8"),
                 ("l_returnflag", "U1"), ("l_linestatus", "U1"), ("l_shipdate", "i8"), ("l_commitdate", "i8"),
                 ("l_receiptdate", "i8")],
}

# Text columns derived in numpy from a generated key column (deterministic hash of the key, TPCH
# value lists): enough for the queries beyond the five configured ones (Q4, Q12, Q14, Q19 ...).
_PRIORITIES = ["1-URGENT", "2-HIGH", "3-MEDIUM", "4-NOT SPECIFIED", "5-LOW"]
_TYPE_1 = ["STANDARD", "SMALL", "MEDIUM", "LARGE", "ECONOMY", "PROMO"]
_TYPE_2 = ["ANODIZED", "BURNISHED", "PLATED", "POLISHED", "BRUSHED"]
_TYPE_3 = ["TIN", "NICKEL", "BRASS", "STEEL", "COPPER"]
_SHIPMODES = ["REG AIR", "AIR", "RAIL", "SHIP", "TRUCK", "MAIL", "FOB"]
_INSTRUCT = ["DELIVER IN PERSON", "COLLECT COD", "NONE", "TAKE BACK RETURN"]


def _mix(a, salt):
    """splitmix64 of an int64 array (+ salt) -> uint64."""
    z = a.astype(np.uint64) + np.uint64((0x9E3779B97F4A7C15 * (salt + 1)) & 0xFFFFFFFFFFFFFFFF)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _pick(values, width, h):
    return np.array(values, "<U%d" % width)[(h % np.uint64(len(values))).astype(np.int64)]


_CONTAINERS = [a + " " + b for a in ("SM", "LG", "MED", "JUMBO", "WRAP") for b in ("CASE", "BOX", "BAG", "JAR", "PKG", "PACK", "CAN", "DRUM")]
_COMMENTS = ["furiously final deposits sleep", "carefully special packages haggle requests", "special requests nag blithely", "requests above the special ideas",
             "ironic accounts boost slyly", "quickly special asymptotes wake", "pending requests use carefully", "express special theodolites; bold requests",
             "even platelets are", "regular pinto beans cajole", "blithely unusual courts", "silent foxes against the requests"]
_SUPP_COMMENTS = ["slyly regular accounts", "Customer insists on Complaints about late ones", "blithely Customer accounts", "Complaints precede the Customer",
                  "furiously bold deposits", "Customer Complaints", "carefully even requests"]


def _numbered(prefix, k, digits, width):
    return np.char.add(prefix, np.char.zfill(k.astype("<U%d" % digits), digits)).astype("<U%d" % width)


DERIVED = {   # table -> {column: (dtype, base columns, function(base arrays...) -> array)}
    "orders": {"o_orderpriority": ("U15", ["o_orderkey"], lambda k: _pick(_PRIORITIES, 15, _mix(k, 1))),
               "o_comment": ("U79", ["o_orderkey"], lambda k: _pick(_COMMENTS, 79, _mix(k, 12)))},
    "supplier": {"s_name": ("U25", ["s_suppkey"], lambda k: _numbered("Supplier#", k, 9, 25)),
                 "s_address": ("U40", ["s_suppkey"], lambda k: np.char.add(np.char.add(_pick(_TYPE_3, 40, _mix(k, 13)), " "), (_mix(k, 14) % np.uint64(99991)).astype("<U5")).astype("<U40")),
                 "s_phone": ("U15", ["s_suppkey", "s_nationkey"], lambda k, n: np.char.add(np.char.add((n + 10).astype("<U2"), "-"), np.char.zfill((_mix(k, 15) % np.uint64(10 ** 10)).astype("<U10"), 10)).astype("<U15")),
                 "s_comment": ("U101", ["s_suppkey"], lambda k: _pick(_SUPP_COMMENTS, 101, _mix(k, 16)))},
    "customer": {"c_name": ("U25", ["c_custkey"], lambda k: np.char.add("Customer#", np.char.zfill(k.astype("<U9"), 9)).astype("<U25")),
                 "c_address": ("U40", ["c_custkey"], lambda k: np.char.add(np.char.add(_pick(_TYPE_2, 40, _mix(k, 7)), " "), (_mix(k, 8) % np.uint64(9973)).astype("<U5")).astype("<U40")),
                 "c_phone": ("U15", ["c_custkey", "c_nationkey"], lambda k, n: np.char.add(np.char.add((n + 10).astype("<U2"), "-"), np.char.zfill((_mix(k, 9) % np.uint64(10 ** 10)).astype("<U10"), 10)).astype("<U15")),
                 "c_comment": ("U117", ["c_custkey"], lambda k: np.char.add(np.char.add(_pick(_INSTRUCT, 117, _mix(k, 10)), " / "), _pick(_SHIPMODES, 117, _mix(k, 11))).astype("<U117"))},
    "part": {"p_mfgr": ("U25", ["p_partkey"], lambda k: np.char.add("Manufacturer#", (_mix(k, 17) % np.uint64(5) + np.uint64(1)).astype("<U1")).astype("<U25")),
             "p_brand": ("U10", ["p_partkey"], lambda k: np.char.add("Brand#", ((_mix(k, 17) % np.uint64(5) + np.uint64(1)) * np.uint64(10) + _mix(k, 18) % np.uint64(5) + np.uint64(1)).astype("<U2")).astype("<U10")),
             "p_container": ("U10", ["p_partkey"], lambda k: _pick(_CONTAINERS, 10, _mix(k, 19))),
             "p_type": ("U25", ["p_partkey"], lambda k: np.char.add(np.char.add(np.char.add(_pick(_TYPE_1, 25, _mix(k, 2)), " "),
                                                                                  np.char.add(_pick(_TYPE_2, 25, _mix(k, 3)), " ")),
                                                                      _pick(_TYPE_3, 25, _mix(k, 4))).astype("<U25"))},
    "lineitem": {"l_shipmode": ("U10", ["l_orderkey", "l_linenumber"], lambda k, n: _pick(_SHIPMODES, 10, _mix(k * 8 + n, 5))),
                 "l_shipinstruct": ("U25", ["l_orderkey", "l_linenumber"], lambda k, n: _pick(_INSTRUCT, 25, _mix(k * 8 + n, 6)))},
}

# the columns each hot-path query references; used to keep SF=10 generation small
QUERY_COLUMNS = {
    "q6": {"lineitem": ["l_shipdate", "l_discount", "l_quantity", "l_extendedprice"]},
    "q1": {"lineitem": ["l_shipdate", "l_returnflag", "l_linestatus", "l_quantity", "l_extendedprice",
                        "l_discount", "l_tax"]},
    "q3": {"lineitem": ["l_shipdate", "l_orderkey", "l_extendedprice", "l_discount"],
           "customer": ["c_mktsegment", "c_custkey"],
           "orders": ["o_orderdate", "o_custkey", "o_orderkey", "o_shippriority"]},
    "q5": {"lineitem": ["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"],
           "customer": ["c_custkey", "c_nationkey"],
           "orders": ["o_orderkey", "o_custkey", "o_orderdate"],
           "region": ["r_regionkey", "r_name"], "nation": ["n_nationkey", "n_name", "n_regionkey"],
           "supplier": ["s_suppkey", "s_nationkey"]},
    "q4": {"lineitem": ["l_orderkey", "l_commitdate", "l_receiptdate"],
           "orders": ["o_orderkey", "o_orderdate", "o_orderpriority"]},
    "q14": {"lineitem": ["l_partkey", "l_shipdate", "l_extendedprice", "l_discount"], "part": ["p_partkey", "p_type"]},
    "q18": {"lineitem": ["l_orderkey", "l_quantity"], "customer": ["c_custkey", "c_name"],
            "orders": ["o_orderkey", "o_custkey", "o_orderdate", "o_totalprice"]},
    "q10": {"lineitem": ["l_orderkey", "l_returnflag", "l_extendedprice", "l_discount"],
            "customer": ["c_custkey", "c_name", "c_acctbal", "c_address", "c_nationkey", "c_phone", "c_comment"],
            "orders": ["o_orderkey", "o_custkey", "o_orderdate"], "nation": ["n_nationkey", "n_name"]},
    "q7": {"supplier": ["s_suppkey", "s_nationkey"], "lineitem": ["l_suppkey", "l_orderkey", "l_shipdate", "l_extendedprice", "l_discount"],
           "orders": ["o_orderkey", "o_custkey"], "customer": ["c_custkey", "c_nationkey"], "nation": ["n_nationkey", "n_name"]},
    "q8": {"part": ["p_partkey", "p_type"], "supplier": ["s_suppkey", "s_nationkey"],
           "lineitem": ["l_partkey", "l_suppkey", "l_orderkey", "l_extendedprice", "l_discount"],
           "orders": ["o_orderkey", "o_custkey", "o_orderdate"], "customer": ["c_custkey", "c_nationkey"],
           "nation": ["n_nationkey", "n_name", "n_regionkey"], "region": ["r_regionkey", "r_name"]},
    "q2": {"part": ["p_partkey", "p_size", "p_type", "p_mfgr"],
           "supplier": ["s_suppkey", "s_nationkey", "s_acctbal", "s_name", "s_address", "s_phone", "s_comment"],
           "partsupp": ["ps_partkey", "ps_suppkey", "ps_supplycost"], "nation": ["n_nationkey", "n_regionkey", "n_name"],
           "region": ["r_regionkey", "r_name"]},
    "q11": {"partsupp": ["ps_partkey", "ps_suppkey", "ps_supplycost", "ps_availqty"], "supplier": ["s_suppkey", "s_nationkey"],
            "nation": ["n_nationkey", "n_name"]},
    "q12": {"orders": ["o_orderkey", "o_orderpriority"],
            "lineitem": ["l_orderkey", "l_shipmode", "l_shipdate", "l_commitdate", "l_receiptdate"]},
    "q13": {"customer": ["c_custkey"], "orders": ["o_custkey", "o_comment"]},
    "q16": {"part": ["p_partkey", "p_brand", "p_type", "p_size"], "supplier": ["s_suppkey", "s_comment"],
            "partsupp": ["ps_partkey", "ps_suppkey"]},
    "q15": {"lineitem": ["l_suppkey", "l_shipdate", "l_extendedprice", "l_discount"], "supplier": ["s_suppkey", "s_name", "s_address", "s_phone"]},
    "q17": {"lineitem": ["l_partkey", "l_quantity", "l_extendedprice"], "part": ["p_partkey", "p_brand", "p_container"]},
    "q19": {"lineitem": ["l_partkey", "l_quantity", "l_extendedprice", "l_discount", "l_shipinstruct", "l_shipmode"],
            "part": ["p_partkey", "p_brand", "p_size", "p_container"]},
    "q20": {"supplier": ["s_suppkey", "s_name", "s_address", "s_nationkey"], "nation": ["n_nationkey", "n_name"],
            "partsupp": ["ps_partkey", "ps_suppkey", "ps_availqty"], "part": ["p_partkey", "p_name"],
            "lineitem": ["l_partkey", "l_suppkey", "l_quantity", "l_shipdate"]},
    "q22": {"customer": ["c_custkey", "c_phone", "c_acctbal"], "orders": ["o_custkey"]},
    "q9": {"lineitem": ["l_orderkey", "l_partkey", "l_suppkey", "l_quantity", "l_extendedprice", "l_discount"],
           "orders": ["o_orderkey", "o_orderdate"], "nation": ["n_nationkey", "n_name"],
           "supplier": ["s_suppkey", "s_nationkey"], "part": ["p_partkey", "p_name"],
           "partsupp": ["ps_partkey", "ps_suppkey", "ps_supplycost"]},
}


def columns_for(queries):
    """Union of the per-table column lists the given queries need, in generator order."""
    need = {}
    for q in queries:
        for t, cols in QUERY_COLUMNS[q].items():
            need.setdefault(t, set()).update(cols)
    return {t: [c for c, _ in GENERATED[t] if c in cs] + [c for c in DERIVED.get(t, {}) if c in cs] for t, cs in need.items()}


_lib = None


def _gen():
    global _lib
    if _lib is None:
        path = build.build_tpchgen()
        lib = ctypes.CDLL(path)
        lib.tpchgen_rows.restype = ctypes.c_int64
        lib.tpchgen_rows.argtypes = [ctypes.c_int, ctypes.c_double]
        lib.tpchgen_lineitem_rows.restype = ctypes.c_int64
        lib.tpchgen_lineitem_rows.argtypes = [ctypes.c_double, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
        lib.tpchgen_lineitem.restype = ctypes.c_int64
        _lib = lib
    return _lib


def table_rows(table, sf):
    ids = {"region": T_REGION, "nation": T_NATION, "supplier": T_SUPPLIER, "customer": T_CUSTOMER,
           "part": T_PART, "partsupp": T_PARTSUPP, "orders": T_ORDERS}
    return int(_gen().tpchgen_rows(ids[table], float(sf)))


def lineitem_rows(sf, seed=DEFAULT_SEED, order_range=None, threads=None):
    b, e = order_range if order_range is not None else (0, table_rows("orders", sf))
    return int(_gen().tpchgen_lineitem_rows(float(sf), seed, b, e, threads or os.cpu_count() or 1))


def _alloc(table, want, nrows):
    out = {}
    for name, dt in GENERATED[table]:
        if want is None or name in want:
            out[name] = np.empty(nrows, dtype="<" + dt if dt[0] != "U" else "<" + dt)
    return out


def _ptr(arrs, name):
    a = arrs.get(name)
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(None)


def generate_table(table, sf, seed=DEFAULT_SEED, columns=None, row_range=None, threads=None):
    """One table (or the row range [b, e) of it; for lineitem the range is in *orders*) as a
    columnar sr_dict.  ``columns`` restricts the generated columns."""
    lib = _gen()
    threads = int(threads or os.cpu_count() or 1)
    sf = float(sf)
    want = set(columns) if columns is not None else None
    cs, cu, cd, ci = ctypes.c_uint64, ctypes.c_int64, ctypes.c_double, ctypes.c_int
    if table == "lineitem":
        ob, oe = row_range if row_range is not None else (0, table_rows("orders", sf))
        n = int(lib.tpchgen_lineitem_rows(sf, seed, ob, oe, threads))
        a = _alloc(table, want, n)
        order = ["l_orderkey", "l_partkey", "l_suppkey", "l_linenumber", "l_quantity", "l_extendedprice",
                 "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate", "l_commitdate", "l_receiptdate"]
        got = lib.tpchgen_lineitem(cd(sf), cs(seed), cu(ob), cu(oe), cu(n), *[_ptr(a, c) for c in order], ci(threads))
        if got != n:
            raise RuntimeError("lineitem generator row-count mismatch")
    else:
        total = table_rows(table, sf)
        b, e = row_range if row_range is not None else (0, total)
        if table in ("region", "nation") and (b, e) != (0, total):
            raise ValueError("region/nation are generated whole")
        n = e - b
        a = _alloc(table, want, n)
        if table == "region":
            lib.tpchgen_region(_ptr(a, "r_regionkey"), _ptr(a, "r_name"))
        elif table == "nation":
            lib.tpchgen_nation(_ptr(a, "n_nationkey"), _ptr(a, "n_name"), _ptr(a, "n_regionkey"))
        elif table == "supplier":
            lib.tpchgen_supplier(cd(sf), cs(seed), cu(b), cu(e), _ptr(a, "s_suppkey"), _ptr(a, "s_nationkey"),
                                 _ptr(a, "s_acctbal"), ci(threads))
        elif table == "customer":
            lib.tpchgen_customer(cd(sf), cs(seed), cu(b), cu(e), _ptr(a, "c_custkey"), _ptr(a, "c_nationkey"),
                                 _ptr(a, "c_acctbal"), _ptr(a, "c_mktsegment"), ci(threads))
        elif table == "part":
            lib.tpchgen_part(cd(sf), cs(seed), cu(b), cu(e), _ptr(a, "p_partkey"), _ptr(a, "p_name"),
                             _ptr(a, "p_retailprice"), _ptr(a, "p_size"), ci(threads))
        elif table == "partsupp":
            lib.tpchgen_partsupp(cd(sf), cs(seed), cu(b), cu(e), _ptr(a, "ps_partkey"), _ptr(a, "ps_suppkey"),
                                 _ptr(a, "ps_availqty"), _ptr(a, "ps_supplycost"), ci(threads))
        elif table == "orders":
            lib.tpchgen_orders(cd(sf), cs(seed), cu(b), cu(e), _ptr(a, "o_orderkey"), _ptr(a, "o_custkey"),
                               _ptr(a, "o_orderdate"), _ptr(a, "o_shippriority"), _ptr(a, "o_totalprice"), ci(threads))
        else:
            raise KeyError(table)
    headers = [c for c, _ in GENERATED[table] if c in a]
    return table_from_columns(headers, [a[h] for h in headers])


def _with_derived(table, sf, seed, columns, row_range, threads):
    """generate_table + the DERIVED text columns that were asked for (or all, with columns=None)."""
    derived = DERIVED.get(table, {})
    want_d = [c for c in derived if columns is None or c in columns]
    if not want_d:
        return generate_table(table, sf, seed, columns, row_range, threads)
    base = None if columns is None else sorted(set(c for c in columns if c not in derived) | {b for c in want_d for b in derived[c][1]})
    t = generate_table(table, sf, seed, base, row_range, threads).getContainer()
    cols = dict(zip(t["headers"], t["data"]))
    for c in want_d:
        cols[c] = derived[c][2](*[cols[b] for b in derived[c][1]])
    order = [c for c in next(iter(SCHEMAS[table].keys())).getContainer() if c in cols and (columns is None or c in columns)]
    return table_from_columns(order, [cols[c] for c in order])


def generate(sf, seed=DEFAULT_SEED, tables=("lineitem", "customer", "orders"), columns=None, threads=None,
             shard=None):
    """Dict name -> columnar table.  ``columns`` = {table: [names]} (see columns_for).  ``shard`` =
    (rank, world): the big tables (lineitem/orders/customer/supplier/part/partsupp) are cut into
    ``world`` contiguous row ranges — lineitem along order boundaries so an order's lines stay
    together — and only range ``rank`` is produced; region and nation are always whole."""
    db = {}
    for t in tables:
        cols = None if columns is None else columns.get(t)
        rr = None
        if shard is not None and t not in ("region", "nation"):
            rank, world = shard
            total = table_rows("orders" if t == "lineitem" else t, sf)
            rr = (total * rank // world, total * (rank + 1) // world)
        db[t] = _with_derived(t, sf, seed, cols, rr, threads)
        if rr is not None:
            db[t].shard = tuple(shard)          # this rank holds a row range of the table, not all of it (dist.DistributedRunner)
    return db


def column(table, name):
    """The numpy array of column ``name`` of a columnar table."""
    c = table.getContainer()
    return c["data"][c["headers"].index(name)]


def fingerprint(db):
    """Order-sensitive 64-bit checksum over every column of every table (tables by name) — pins
    the generator's output in the golden fixtures without storing the data."""
    import hashlib
    h = hashlib.sha256()
    for t in sorted(db):
        c = db[t].getContainer()
        for name, arr in zip(c["headers"], c["data"]):
            h.update(name.encode())
            h.update(np.ascontiguousarray(arr).view(np.uint8).tobytes())
    return h.hexdigest()[:16]


def _fmt_date(v):
    v = int(v)
    return "%04d-%02d-%02d" % (v // 10000, v // 100 % 100, v % 100)


def write_tbl(directory, db, file_names=None):
    """Write dbgen-style '|'-terminated text files with the full reference schema.  Generated
    columns are written exactly (2-decimal money, yyyy-mm-dd dates); the others get short filler."""
    os.makedirs(directory, exist_ok=True)
    paths = {}
    for t, table in db.items():
        fields = next(iter(SCHEMAS[t].keys())).getContainer()
        c = table.getContainer()
        have = dict(zip(c["headers"], c["data"]))
        n = len(c["data"][0])
        path = os.path.join(directory, (file_names or {}).get(t, t + ".tbl"))
        with open(path, "w", newline="\n") as fh:
            for i in range(n):
                cells = []
                for name, typ in fields.items():
                    if name.endswith("_NA"):
                        continue
                    if name in have:
                        v = have[name][i]
                        if typ == date:
                            cells.append(_fmt_date(v))
                        elif typ == float:
                            cells.append("%.2f" % float(v))
                        else:
                            cells.append(str(v))
                    elif typ == int:
                        cells.append(str(i % 7))
                    elif typ == float:
                        cells.append("0.00")
                    elif typ == date:
                        cells.append("1992-01-01")
                    else:
                        cells.append(("x%d" % (i % 97))[: typ.max_size])
                fh.write("|".join(cells) + "|\n")
        paths[t] = path
    return paths
