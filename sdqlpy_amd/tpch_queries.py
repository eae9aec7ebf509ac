"""The TPCH queries of the hot path, written in the sdqlpy DSL against sdqlpy_amd.

These are the workload, the way SQL text is for a SQL engine: q6, q1, q3 (then q5, q9) expressed
with the same combinators, filters and arithmetic as the reference's TPCH script, so that the
golden results captured from the reference (tests/golden) apply to them verbatim
(reference test/test_all.py: q1 46-62, q3 145-176, q5 215-281, q6 285-295, q9 431-491; beyond the
configured five, SURVEY.md §8f.3: q4 180-211, q10 495-558, q14 695-716, q18 874-913).
Call ``sdqlpy_init(3)`` (or 1) before running them.
"""
from .sdql_lib import *      # noqa: F401,F403
from .tpch import (customer_type, lineitem_type, nation_type, order_type, part_type, partsupp_type,
                   region_type, supplier_type)


@sdql_compile({"li": lineitem_type})
def q6(li):
    results = li.sum(lambda p: p[0].l_extendedprice * p[0].l_discount
                     if (p[0].l_shipdate >= 19940101) and (p[0].l_shipdate < 19950101)
                     and (p[0].l_discount >= 0.05) and (p[0].l_discount <= 0.07) and (p[0].l_quantity < 24.0)
                     else 0.0)
    return results


@sdql_compile({"li": lineitem_type})
def q1(li):
    lineitem_probed = li.sum(lambda p: {
        record({"l_returnflag": p[0].l_returnflag, "l_linestatus": p[0].l_linestatus}):
        record({"sum_qty": p[0].l_quantity,
                "sum_base_price": p[0].l_extendedprice,
                "sum_disc_price": (p[0].l_extendedprice * (1.0 - p[0].l_discount)),
                "sum_charge": ((p[0].l_extendedprice * (1.0 - p[0].l_discount)) * (1.0 + p[0].l_tax)),
                "count_order": 1})
    } if p[0].l_shipdate <= 19980902 else None)
    results = lineitem_probed.sum(lambda p: {unique(p[0].concat(p[1])): True})
    return results


@sdql_compile({"li": lineitem_type, "cu": customer_type, "ord": order_type})
def q3(li, cu, ord):
    building = "BUILDING"
    customer_indexed = cu.joinBuild("c_custkey", lambda p: p[0].c_mktsegment == building, [])
    order_probed = ord.joinProbe(
        customer_indexed, "o_custkey",
        lambda p: p[0].o_orderdate < 19950315,
        lambda indexedDictValue, probeDictKey: {
            probeDictKey.o_orderkey:
            record({"o_orderdate": probeDictKey.o_orderdate, "o_shippriority": probeDictKey.o_shippriority})},
        False)
    lineitem_probed = li.joinProbe(
        order_probed, "l_orderkey",
        lambda p: p[0].l_shipdate > 19950315,
        lambda indexedDictValue, probeDictKey: {
            record({"l_orderkey": probeDictKey.l_orderkey, "o_orderdate": indexedDictValue.o_orderdate,
                    "o_shippriority": indexedDictValue.o_shippriority}):
            record({"revenue": probeDictKey.l_extendedprice * (1.0 - probeDictKey.l_discount)})})
    results = lineitem_probed.sum(lambda p: {unique(p[0].concat(p[1])): True})
    return results


@sdql_compile({"li": lineitem_type, "cu": customer_type, "ord": order_type, "re": region_type,
               "na": nation_type, "su": supplier_type})
def q5(li, cu, ord, re, na, su):
    asia = "ASIA"
    region_indexed = re.joinBuild("r_regionkey", lambda p: p[0].r_name == asia, [])
    nation_probed = na.joinProbe(
        region_indexed, "n_regionkey", lambda p: True,
        lambda indexedDictValue, probeDictKey: {probeDictKey.n_nationkey: probeDictKey.n_name},
        False)
    customer_probed = cu.joinProbe(
        nation_probed, "c_nationkey", lambda p: True,
        lambda indexedDictValue, probeDictKey: {
            probeDictKey.c_custkey: record({"n_name": indexedDictValue, "c_nationkey": probeDictKey.c_nationkey})},
        False)
    order_probed = ord.joinProbe(
        customer_probed, "o_custkey",
        lambda p: (p[0].o_orderdate < 19950101) * (p[0].o_orderdate >= 19940101),
        lambda indexedDictValue, probeDictKey: {
            probeDictKey.o_orderkey:
            record({"n_name": indexedDictValue.n_name, "c_nationkey": indexedDictValue.c_nationkey})},
        False)
    supplier_project = su.sum(lambda p: {
        unique(record({"s_suppkey": p[0].s_suppkey, "s_nationkey": p[0].s_nationkey})): True})
    lineitem_probed = li.joinProbe(
        order_probed, "l_orderkey", lambda p: True,
        lambda indexedDictValue, probeDictKey: {
            indexedDictValue.n_name: probeDictKey.l_extendedprice * (1.0 - probeDictKey.l_discount)}
        if supplier_project[record({"l_suppkey": probeDictKey.l_suppkey,
                                    "c_nationkey": indexedDictValue.c_nationkey})] != None      # noqa: E711
        else None)
    results = lineitem_probed.sum(lambda p: {unique(record({"n_name": p[0], "revenue": p[1]})): True})
    return results


@sdql_compile({"li": lineitem_type, "ord": order_type, "na": nation_type, "su": supplier_type,
               "pa": part_type, "ps": partsupp_type})
def q9(li, ord, na, su, pa, ps):
    nation_indexed = na.joinBuild("n_nationkey", lambda p: True, ["n_name"])
    supplier_probed = su.sum(lambda p: {unique(p[0].s_suppkey): nation_indexed[p[0].s_nationkey].n_name})
    green = "green"
    part_indexed = pa.joinBuild("p_partkey", lambda p: green in p[0].p_name, [])
    partsupp_probe = ps.joinProbe(
        part_indexed, "ps_partkey", lambda p: True,
        lambda indexedDictValue, probeDictKey: {
            record({"ps_partkey": probeDictKey.ps_partkey, "ps_suppkey": probeDictKey.ps_suppkey}):
            record({"n_name": supplier_probed[probeDictKey.ps_suppkey], "ps_supplycost": probeDictKey.ps_supplycost})},
        False)
    ord_indexed = ord.sum(lambda p: {dense(6000000, unique(p[0].o_orderkey)): p[0].o_orderdate})
    li_probed = li.sum(lambda p: {
        record({"nation": partsupp_probe[record({"ps_partkey": p[0].l_partkey, "ps_suppkey": p[0].l_suppkey})].n_name,
                "o_year": extractYear(ord_indexed[p[0].l_orderkey])}):
        record({"sum_profit": p[0].l_extendedprice * (1.0 - p[0].l_discount)
                - partsupp_probe[record({"ps_partkey": p[0].l_partkey, "ps_suppkey": p[0].l_suppkey})].ps_supplycost
                * p[0].l_quantity})
    } if partsupp_probe[record({"ps_partkey": p[0].l_partkey, "ps_suppkey": p[0].l_suppkey})] != None      # noqa: E711
        else None)
    results = li_probed.sum(lambda p: {unique(p[0].concat(p[1])): True})
    return results


@sdql_compile({"ord": order_type, "li": lineitem_type})
def q4(ord, li):
    li_indexed = li.sum(lambda p: {dense(6000000, unique(p[0].l_orderkey)): True}
                        if p[0].l_commitdate < p[0].l_receiptdate else None)
    ord_probed = ord.joinProbe(
        li_indexed, "o_orderkey",
        lambda p: p[0].o_orderdate >= 19930701 and p[0].o_orderdate < 19931001,
        lambda indexedDictValue, probeDictKey: {probeDictKey.o_orderpriority: 1})
    results = ord_probed.sum(lambda p: {unique(record({"o_orderpriority": p[0], "order_count": p[1]})): True})
    return results


@sdql_compile({"li": lineitem_type, "pa": part_type})
def q14(li, pa):
    promo = "PROMO"
    pa_indexed = pa.joinBuild("p_partkey", lambda p: startsWith(p[0].p_type, promo), [])
    li_probed = li.sum(lambda p: record({
        "A": p[0].l_extendedprice * (1.0 - p[0].l_discount) if pa_indexed[p[0].l_partkey] != None else 0.0,      # noqa: E711
        "B": p[0].l_extendedprice * (1.0 - p[0].l_discount)})
        if p[0].l_shipdate >= 19950901 and p[0].l_shipdate < 19951001 else None)
    results = (100.0 * li_probed.A) / li_probed.B
    return results


@sdql_compile({"li": lineitem_type, "cu": customer_type, "ord": order_type})
def q18(li, cu, ord):
    li_aggregated = li.sum(lambda b: {b[0].l_orderkey: b[0].l_quantity})
    li_filtered = li_aggregated.sum(lambda z: {unique(z[0]): True} if z[1] > 300 else None)
    cu_indexed = cu.joinBuild("c_custkey", lambda p: True, ["c_name"])
    order_probed = ord.joinProbe(
        cu_indexed, "o_custkey",
        lambda p: li_filtered[p[0].o_orderkey] != None,      # noqa: E711
        lambda indexedDictValue, probeDictKey: {
            probeDictKey.o_orderkey:
            record({"c_name": indexedDictValue.c_name, "o_custkey": probeDictKey.o_custkey,
                    "o_orderkey": probeDictKey.o_orderkey, "o_orderdate": probeDictKey.o_orderdate,
                    "o_totalprice": probeDictKey.o_totalprice})},
        False)
    li_probed = li.joinProbe(
        order_probed, "l_orderkey", lambda p: True,
        lambda indexedDictValue, probeDictKey: {
            record({"c_name": indexedDictValue.c_name, "o_custkey": indexedDictValue.o_custkey,
                    "o_orderkey": indexedDictValue.o_orderkey, "o_orderdate": indexedDictValue.o_orderdate,
                    "o_totalprice": indexedDictValue.o_totalprice}):
            record({"quantitysum": probeDictKey.l_quantity})})
    results = li_probed.sum(lambda p: {unique(p[0].concat(p[1])): True})
    return results


@sdql_compile({"cu": customer_type, "ord": order_type, "li": lineitem_type, "na": nation_type})
def q10(cu, ord, li, na):
    r = "R"
    na_indexed = na.joinBuild("n_nationkey", lambda p: True, ["n_name"])
    cu_indexed = cu.joinBuild("c_custkey", lambda p: True,
                              ["c_custkey", "c_name", "c_acctbal", "c_address", "c_nationkey", "c_phone", "c_comment"])
    ord_probed = ord.joinProbe(
        cu_indexed, "o_custkey",
        lambda p: p[0].o_orderdate >= 19931001 and p[0].o_orderdate < 19940101,
        lambda indexedDictValue, probeDictKey: {
            probeDictKey.o_orderkey:
            record({"c_custkey": indexedDictValue.c_custkey, "c_name": indexedDictValue.c_name,
                    "c_acctbal": indexedDictValue.c_acctbal, "c_address": indexedDictValue.c_address,
                    "c_phone": indexedDictValue.c_phone, "c_comment": indexedDictValue.c_comment,
                    "n_name": na_indexed[indexedDictValue.c_nationkey].n_name})},
        False)
    li_probed = li.joinProbe(
        ord_probed, "l_orderkey",
        lambda p: p[0].l_returnflag == r,
        lambda indexedDictValue, probeDictKey: {
            record({"c_custkey": indexedDictValue.c_custkey, "c_name": indexedDictValue.c_name,
                    "c_acctbal": indexedDictValue.c_acctbal, "n_name": indexedDictValue.n_name,
                    "c_address": indexedDictValue.c_address, "c_phone": indexedDictValue.c_phone,
                    "c_comment": indexedDictValue.c_comment}):
            probeDictKey.l_extendedprice * (1.0 - probeDictKey.l_discount)},
        True)
    results = li_probed.sum(lambda p: {unique(record({
        "c_custkey": p[0].c_custkey, "c_name": p[0].c_name, "revenue": p[1], "c_acctbal": p[0].c_acctbal,
        "n_name": p[0].n_name, "c_address": p[0].c_address, "c_phone": p[0].c_phone, "c_comment": p[0].c_comment})): True})
    return results


# positional table order of each query (the decorator dict order == call order)
QUERY_TABLES = {
    "q6": ["lineitem"],
    "q1": ["lineitem"],
    "q3": ["lineitem", "customer", "orders"],
    "q5": ["lineitem", "customer", "orders", "region", "nation", "supplier"],
    "q9": ["lineitem", "orders", "nation", "supplier", "part", "partsupp"],
    "q4": ["orders", "lineitem"],
    "q14": ["lineitem", "part"],
    "q18": ["lineitem", "customer", "orders"],
    "q10": ["customer", "orders", "lineitem", "nation"],
}
QUERIES = {"q6": q6, "q1": q1, "q3": q3, "q5": q5, "q9": q9, "q4": q4, "q14": q14, "q18": q18, "q10": q10}


def run(name, db, top=None):
    """Run a query on a database dict; top = (k, [(column, "asc" | "desc")]) adds ORDER BY ... LIMIT k."""
    args = [db[t] for t in QUERY_TABLES[name]]
    return QUERIES[name].top(*top)(*args) if top is not None else QUERIES[name](*args)


# TPCH's own ORDER BY / LIMIT for the queries above (the reference's versions return unordered sets)
TPCH_ORDER = {
    "q1": (100, [("l_returnflag", "asc"), ("l_linestatus", "asc")]),
    "q3": (10, [("revenue", "desc"), ("o_orderdate", "asc")]),
    "q5": (100, [("revenue", "desc")]),
    "q9": (128, [("nation", "asc"), ("o_year", "desc")]),
    "q4": (100, [("o_orderpriority", "asc")]),
    "q18": (100, [("o_totalprice", "desc"), ("o_orderdate", "asc")]),
    "q10": (20, [("revenue", "desc")]),
}
