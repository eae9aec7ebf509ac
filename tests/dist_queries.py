"""Query variants for the multi-rank tests: q18 with a HAVING threshold that leaves rows at the tiny
scale factor the CPU tests run at (the TPCH constant 300 selects nothing below SF ~0.05)."""
from sdqlpy_amd.sdql_lib import *          # noqa: F401,F403
from sdqlpy_amd.tpch import customer_type, lineitem_type, order_type
from sdqlpy_amd import tpch_queries as Q


@sdql_compile({"li": lineitem_type, "cu": customer_type, "ord": order_type})
def q18_low(li, cu, ord):
    li_aggregated = li.sum(lambda b: {b[0].l_orderkey: b[0].l_quantity})
    li_filtered = li_aggregated.sum(lambda z: {unique(z[0]): True} if z[1] > 230 else None)
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


def register():
    Q.QUERIES["q18_low"] = q18_low
    Q.QUERY_TABLES["q18_low"] = Q.QUERY_TABLES["q18"]
    Q.TPCH_ORDER["q18_low"] = Q.TPCH_ORDER["q18"]
