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


@sdql_compile({"nation": nation_type, "customer": customer_type, "orders": order_type, "supplier": supplier_type, "lineitem": lineitem_type})
def q7(nation, customer, orders, supplier, lineitem):
    pair = nation.joinBuild("n_nationkey", lambda n: n[0].n_name == "FRANCE" or n[0].n_name == "GERMANY", ["n_name"])
    buyer_nation = customer.joinProbe(pair, "c_nationkey", lambda c: True, lambda nat, c: {unique(c.c_custkey): nat.n_name}, False)
    order_nation = orders.joinProbe(buyer_nation, "o_custkey", lambda o: True, lambda nat, o: {unique(o.o_orderkey): nat}, False)
    seller_nation = supplier.joinProbe(pair, "s_nationkey", lambda s: True, lambda nat, s: {unique(s.s_suppkey): nat.n_name}, False)
    traded = lineitem.joinProbe(
        order_nation, "l_orderkey",
        lambda l: 19950101 <= l[0].l_shipdate <= 19961231 and seller_nation[l[0].l_suppkey] != None      # noqa: E711
        and ((order_nation[l[0].l_orderkey] == "FRANCE" and seller_nation[l[0].l_suppkey] == "GERMANY")
             or (order_nation[l[0].l_orderkey] == "GERMANY" and seller_nation[l[0].l_suppkey] == "FRANCE")),
        lambda nat, l: {record({"supp_nation": seller_nation[l.l_suppkey], "cust_nation": order_nation[l.l_orderkey], "l_year": extractYear(l.l_shipdate)}):
                        record({"revenue": l.l_extendedprice * (1.0 - l.l_discount)})})
    flat = traded.sum(lambda g: {unique(g[0].concat(g[1])): True})
    return flat


@sdql_compile({"region": region_type, "nation": nation_type, "supplier": supplier_type, "customer": customer_type, "part": part_type,
               "orders": order_type, "lineitem": lineitem_type})
def q8(region, nation, supplier, customer, part, orders, lineitem):
    one_region = region.joinBuild("r_regionkey", lambda r: r[0].r_name == "AMERICA", [])
    its_nations = nation.joinProbe(one_region, "n_regionkey", lambda n: True, lambda hit, n: {unique(n.n_nationkey): True}, False)
    names = nation.joinBuild("n_nationkey", lambda n: True, ["n_name"])
    seller = supplier.joinBuild("s_suppkey", lambda s: True, ["s_nationkey"])
    buyer_nation = customer.sum(lambda c: {unique(c[0].c_custkey): c[0].c_nationkey})
    steel = part.joinBuild("p_partkey", lambda p: p[0].p_type == "ECONOMY ANODIZED STEEL", [])
    two_years = orders.joinBuild("o_orderkey", lambda o: 19950101 <= o[0].o_orderdate <= 19961231, ["o_custkey", "o_orderdate"])
    per_year = lineitem.joinProbe(
        steel, "l_partkey",
        lambda l: two_years[l[0].l_orderkey] != None and its_nations[buyer_nation[two_years[l[0].l_orderkey].o_custkey]] != None,      # noqa: E711
        lambda hit, l: {extractYear(two_years[l.l_orderkey].o_orderdate):
                        record({"A": l.l_extendedprice * (1.0 - l.l_discount) if names[seller[l.l_suppkey].s_nationkey].n_name == "BRAZIL" else 0.0,
                                "B": l.l_extendedprice * (1.0 - l.l_discount)})})
    share = per_year.sum(lambda g: {unique(record({"o_year": g[0], "mkt_share": g[1].A / g[1].B})): True})
    return share


@sdql_compile({"nation": nation_type, "supplier": supplier_type, "partsupp": partsupp_type})
def q11(nation, supplier, partsupp):
    one_nation = nation.joinBuild("n_nationkey", lambda n: n[0].n_name == "GERMANY", [])
    its_suppliers = supplier.joinProbe(one_nation, "s_nationkey", lambda s: True, lambda hit, s: {unique(s.s_suppkey): True}, False)
    bar = partsupp.joinProbe(its_suppliers, "ps_suppkey", lambda ps: True, lambda hit, ps: (ps.ps_supplycost * ps.ps_availqty) * 0.0001)
    stock = partsupp.joinProbe(its_suppliers, "ps_suppkey", lambda ps: True, lambda hit, ps: {ps.ps_partkey: ps.ps_supplycost * ps.ps_availqty})
    above = stock.sum(lambda g: {unique(record({"ps_partkey": g[0], "value": g[1]})): True} if g[1] > bar else None)
    return above


@sdql_compile({"lineitem": lineitem_type, "orders": order_type})
def q12(lineitem, orders):
    # (the reference nests a dictionary per order in a dictionary per ship mode; the front end flattens that to one dictionary keyed by
    #  both — outer key, inner key — and turns the join round: this is that flat form written out)
    late = lineitem.sum(
        lambda l: {record({"nest_outer": l[0].l_orderkey, "nest_inner": l[0].l_shipmode}): 1}
        if (l[0].l_shipmode == "MAIL" or l[0].l_shipmode == "SHIP") and l[0].l_commitdate < l[0].l_receiptdate
        and l[0].l_shipdate < l[0].l_commitdate and 19940101 <= l[0].l_receiptdate < 19950101 else None)
    urgency = orders.joinBuild("o_orderkey", lambda o: True, ["o_orderpriority"])
    per_mode = late.sum(
        lambda g: {record({"l_shipmode": g[0].nest_inner}):
                   record({"high_line_count": g[1] if urgency[g[0].nest_outer].o_orderpriority == "1-URGENT" or urgency[g[0].nest_outer].o_orderpriority == "2-HIGH" else 0,
                           "low_line_count": g[1] if urgency[g[0].nest_outer].o_orderpriority != "1-URGENT" and urgency[g[0].nest_outer].o_orderpriority != "2-HIGH" else 0})}
        if urgency[g[0].nest_outer] != None else None)      # noqa: E711
    flat = per_mode.sum(lambda g: {unique(g[0].concat(g[1])): True})
    return flat


@sdql_compile({"part": part_type, "supplier": supplier_type, "partsupp": partsupp_type})
def q16(part, supplier, partsupp):
    wanted = part.sum(
        lambda p: {unique(p[0].p_partkey): record({"p_brand": p[0].p_brand, "p_type": p[0].p_type, "p_size": p[0].p_size})}
        if p[0].p_brand != "Brand#45" and not startsWith(p[0].p_type, "MEDIUM POLISHED")
        and (p[0].p_size == 49 or p[0].p_size == 14 or p[0].p_size == 23 or p[0].p_size == 45
             or p[0].p_size == 19 or p[0].p_size == 3 or p[0].p_size == 36 or p[0].p_size == 9)
        else None)
    grumbled_about = supplier.sum(
        lambda s: {unique(s[0].s_suppkey): True}
        if firstIndex(s[0].s_comment, "Customer") != -1
        and firstIndex(s[0].s_comment, "Complaints") > firstIndex(s[0].s_comment, "Customer") + 7
        else None)
    distinct = partsupp.sum(
        lambda ps: {record({"p_brand": wanted[ps[0].ps_partkey].p_brand, "p_type": wanted[ps[0].ps_partkey].p_type,
                            "p_size": wanted[ps[0].ps_partkey].p_size, "nest_inner": ps[0].ps_suppkey}): 1}
        if wanted[ps[0].ps_partkey] != None and grumbled_about[ps[0].ps_suppkey] == None      # noqa: E711
        else None)
    per_group = distinct.sum(
        lambda g: {record({"p_brand": g[0].p_brand, "p_type": g[0].p_type, "p_size": g[0].p_size}): record({"supplier_cnt": 1})})
    flat = per_group.sum(lambda g: {unique(g[0].concat(g[1])): True})
    return flat


@sdql_compile({"region": region_type, "nation": nation_type, "supplier": supplier_type, "part": part_type, "partsupp": partsupp_type})
def q2(region, nation, supplier, part, partsupp):
    one_region = region.joinBuild("r_regionkey", lambda r: r[0].r_name == "EUROPE", [])
    its_nations = nation.joinProbe(one_region, "n_regionkey", lambda n: True, lambda hit, n: {unique(n.n_nationkey): n.n_name}, False)
    its_suppliers = supplier.joinProbe(
        its_nations, "s_nationkey", lambda s: True,
        lambda nat, s: {unique(s.s_suppkey): record({"s_acctbal": s.s_acctbal, "s_name": s.s_name, "n_name": nat, "s_address": s.s_address,
                                                       "s_phone": s.s_phone, "s_comment": s.s_comment})}, False)
    brass = part.joinBuild("p_partkey", lambda p: p[0].p_size == 15 and endsWith(p[0].p_type, "BRASS"), ["p_mfgr"])
    regional_cost = partsupp.joinProbe(
        its_suppliers, "ps_suppkey", lambda ps: brass[ps[0].ps_partkey] != None,      # noqa: E711
        lambda sup, ps: {ps.ps_partkey: ps.ps_supplycost})
    offers = partsupp.sum(
        lambda ps: {record({"ps_partkey": ps[0].ps_partkey, "ps_suppkey": ps[0].ps_suppkey}): 1}
        if regional_cost[ps[0].ps_partkey] != None and regional_cost[ps[0].ps_partkey] == ps[0].ps_supplycost      # noqa: E711
        and its_suppliers[ps[0].ps_suppkey] != None else None)      # noqa: E711
    best = offers.sum(lambda g: {unique(record({
        "s_acctbal": its_suppliers[g[0].ps_suppkey].s_acctbal, "s_name": its_suppliers[g[0].ps_suppkey].s_name,
        "n_name": its_suppliers[g[0].ps_suppkey].n_name, "p_partkey": g[0].ps_partkey, "p_mfgr": brass[g[0].ps_partkey].p_mfgr,
        "s_address": its_suppliers[g[0].ps_suppkey].s_address, "s_phone": its_suppliers[g[0].ps_suppkey].s_phone,
        "s_comment": its_suppliers[g[0].ps_suppkey].s_comment})): True})
    return best


QUERIES = {"q2": q2, "q7": q7, "q8": q8, "q11": q11, "q12": q12, "q16": q16, "q19": q19, "q20": q20}
TABLES = {"q2": ["region", "nation", "supplier", "part", "partsupp"], "q7": ["nation", "customer", "orders", "supplier", "lineitem"],
          "q8": ["region", "nation", "supplier", "customer", "part", "orders", "lineitem"], "q11": ["nation", "supplier", "partsupp"],
          "q12": ["lineitem", "orders"], "q16": ["part", "supplier", "partsupp"], "q19": ["part", "lineitem"],
          "q20": ["part", "nation", "supplier", "lineitem", "partsupp"]}
