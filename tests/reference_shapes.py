"""This package's OWN formulations of the TPCH queries whose shipped formulation (sdqlpy_amd/tpch_queries.py) lowers to OTHER loops
than the reference's formulation does (tests/golden/reference_lowering.json: same_plan_as_shipped_formulation false): the same
builds in the reference's order, payloads carried as the reference carries them, the same table chosen as the probed index.  A user
who arrives with the reference's script launches THESE loops; the reference's text itself never travels, so the GPU suite runs
these instead: tests/test_frontend_cpu.py asserts that each one's name-free plan (frontend.Plan.fingerprint) has the digest
recorded for the reference's own text, tests/test_hip_parity.py runs them against the goldens.

Written from the plans (what is built, probed, summed), in this package's spelling; nothing here is the reference's text."""
from sdqlpy_amd.sdql_lib import *      # noqa: F401,F403
from sdqlpy_amd.tpch import (customer_type, lineitem_type, nation_type, order_type, part_type, partsupp_type, region_type, supplier_type)


@sdql_compile({"part": part_type, "lineitem": lineitem_type})
def q19(part, lineitem):
    boxed = part.sum(
        lambda p: {unique(p[0].p_partkey): record({"p_brand": p[0].p_brand, "p_size": p[0].p_size, "p_container": p[0].p_container})}
        if (p[0].p_brand == "Brand#12"
            and (p[0].p_container == "SM CASE" or p[0].p_container == "SM BOX" or p[0].p_container == "SM PACK" or p[0].p_container == "SM PKG")
            and 1 <= p[0].p_size <= 5)
        or (p[0].p_brand == "Brand#23"
            and (p[0].p_container == "MED BAG" or p[0].p_container == "MED BOX" or p[0].p_container == "MED PACK" or p[0].p_container == "MED PKG")
            and 1 <= p[0].p_size <= 10)
        or (p[0].p_brand == "Brand#34"
            and (p[0].p_container == "LG CASE" or p[0].p_container == "LG BOX" or p[0].p_container == "LG PACK" or p[0].p_container == "LG PKG")
            and 1 <= p[0].p_size <= 15)
        else None)
    in_person = lineitem.joinProbe(
        boxed, "l_partkey",
        lambda l: l[0].l_shipinstruct == "DELIVER IN PERSON" and (l[0].l_shipmode == "AIR" or l[0].l_shipmode == "AIR REG"),
        lambda box, item: item.l_extendedprice * (1.0 - item.l_discount)
        if (box.p_brand == "Brand#12" and 1 <= item.l_quantity <= 11)
        or (box.p_brand == "Brand#23" and 10 <= item.l_quantity <= 20)
        or (box.p_brand == "Brand#34" and 20 <= item.l_quantity <= 30)
        else 0.0)
    out = sr_dict({record({"revenue": in_person}): True})
    return out


@sdql_compile({"part": part_type, "nation": nation_type, "supplier": supplier_type, "lineitem": lineitem_type, "partsupp": partsupp_type})
def q20(part, nation, supplier, lineitem, partsupp):
    wooded = part.joinBuild("p_partkey", lambda p: startsWith(p[0].p_name, "forest"), [])
    one_nation = nation.joinBuild("n_nationkey", lambda n: n[0].n_name == "CANADA", [])
    its_suppliers = supplier.joinBuild("s_suppkey", lambda s: one_nation[s[0].s_nationkey] != None, [])      # noqa: E711
    half_of_1994 = lineitem.joinProbe(
        wooded, "l_partkey",
        lambda l: 19940101 <= l[0].l_shipdate < 19950101 and its_suppliers[l[0].l_suppkey] != None,      # noqa: E711
        lambda hit, l: {record({"l_partkey": l.l_partkey, "l_suppkey": l.l_suppkey}): 0.5 * l.l_quantity})
    in_excess = partsupp.sum(
        lambda ps: {unique(ps[0].ps_suppkey): True}
        if half_of_1994[record({"l_partkey": ps[0].ps_partkey, "l_suppkey": ps[0].ps_suppkey})] != None      # noqa: E711
        and ps[0].ps_availqty > half_of_1994[record({"l_partkey": ps[0].ps_partkey, "l_suppkey": ps[0].ps_suppkey})]
        else None)
    named = supplier.joinProbe(in_excess, "s_suppkey", lambda s: True,
                               lambda hit, s: {unique(record({"s_name": s.s_name, "s_address": s.s_address})): True}, False)
    return named


QUERIES = {"q19": q19, "q20": q20}
TABLES = {"q19": ["part", "lineitem"], "q20": ["part", "nation", "supplier", "lineitem", "partsupp"]}
