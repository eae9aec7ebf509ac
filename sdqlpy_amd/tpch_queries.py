"""TPCH workload for the HIP backend, written against sdqlpy_amd's own DSL surface.

These are this package's own formulations of the TPCH queries on the hot path (SURVEY.md §8:
q6, q1, q3, then q5 / q9, and the first widening step q4 / q10 / q14 / q18).  Each one states the
TPCH query with the package's combinators — dictionary-valued `sum`, `joinBuild`, `joinProbe` —
and produces the same result *columns* (names and order) as the reference's formulation of the same
TPCH query, which is what the golden vectors under tests/golden pin: results are compared as sets
of rows, so only the relational meaning and the association order of the floating-point
expressions are shared with the reference (they are the parity contract, SURVEY.md §7 "FP parity").

Conventions used below (all of them lowered by frontend.py):
  * builds that only answer membership are written as `{unique(key): True}` sums,
  * text literals appear in place, range predicates as chained comparisons,
  * key arrays are sized from the data — no `dense(N, ...)` size hints,
  * thresholds that vary between runs are parameters of a query factory (see `large_orders`).

Call ``sdqlpy_init(3)`` before running them.
"""
from .sdql_lib import *      # noqa: F401,F403
from .tpch import (customer_type, lineitem_type, nation_type, order_type, part_type, partsupp_type,
                   region_type, supplier_type)


# ---- q6: forecast revenue change — one filtered scalar sum ---------------------------------------
@sdql_compile({"lineitem": lineitem_type})
def q6(lineitem):
    gain = lineitem.sum(
        lambda row: row[0].l_extendedprice * row[0].l_discount
        if 19940101 <= row[0].l_shipdate < 19950101 and 0.05 <= row[0].l_discount <= 0.07 and row[0].l_quantity < 24.0
        else 0.0)
    return gain


# ---- q1: pricing summary — group-by over (returnflag, linestatus) ---------------------------------
@sdql_compile({"lineitem": lineitem_type})
def q1(lineitem):
    last_ship_day = 19980902          # 1998-12-01 minus the query's 90-day interval
    per_status = lineitem.sum(
        lambda row: {
            record({"l_returnflag": row[0].l_returnflag, "l_linestatus": row[0].l_linestatus}):
            record({
                "sum_qty": row[0].l_quantity,
                "sum_base_price": row[0].l_extendedprice,
                "sum_disc_price": row[0].l_extendedprice * (1.0 - row[0].l_discount),
                "sum_charge": (row[0].l_extendedprice * (1.0 - row[0].l_discount)) * (1.0 + row[0].l_tax),
                "count_order": 1,
            })
        } if row[0].l_shipdate <= last_ship_day else None)
    summary = per_status.sum(lambda g: {unique(record({
        "l_returnflag": g[0].l_returnflag, "l_linestatus": g[0].l_linestatus,
        "sum_qty": g[1].sum_qty, "sum_base_price": g[1].sum_base_price, "sum_disc_price": g[1].sum_disc_price,
        "sum_charge": g[1].sum_charge, "count_order": g[1].count_order})): True})
    return summary


# ---- q3: shipping priority — customer semi-join, orders build, lineitem probe + group-by ----------
@sdql_compile({"customer": customer_type, "orders": order_type, "lineitem": lineitem_type})
def q3(customer, orders, lineitem):
    cutoff = 19950315
    segment_customers = customer.sum(
        lambda c: {unique(c[0].c_custkey): True} if c[0].c_mktsegment == "BUILDING" else None)
    open_orders = orders.sum(
        lambda o: {unique(o[0].o_orderkey): record({"o_orderdate": o[0].o_orderdate, "o_shippriority": o[0].o_shippriority})}
        if o[0].o_orderdate < cutoff and segment_customers[o[0].o_custkey] != None      # noqa: E711
        else None)
    revenue_per_order = lineitem.joinProbe(
        open_orders, "l_orderkey",
        lambda l: l[0].l_shipdate > cutoff,
        lambda order, item: {
            record({"l_orderkey": item.l_orderkey, "o_orderdate": order.o_orderdate, "o_shippriority": order.o_shippriority}):
            record({"revenue": item.l_extendedprice * (1.0 - item.l_discount)})})
    shipping_priority = revenue_per_order.sum(lambda g: {unique(g[0].concat(g[1])): True})
    return shipping_priority


# ---- q5: local supplier volume — region > nation > customer > orders chain, supplier pair set ------
@sdql_compile({"region": region_type, "nation": nation_type, "customer": customer_type, "orders": order_type,
               "supplier": supplier_type, "lineitem": lineitem_type})
def q5(region, nation, customer, orders, supplier, lineitem):
    asia = region.sum(lambda r: {unique(r[0].r_regionkey): True} if r[0].r_name == "ASIA" else None)
    asian_nations = nation.sum(
        lambda n: {unique(n[0].n_nationkey): n[0].n_name} if asia[n[0].n_regionkey] != None else None)      # noqa: E711
    asian_customers = customer.joinProbe(
        asian_nations, "c_nationkey", lambda c: True,
        lambda nation_name, cust: {cust.c_custkey: record({"n_name": nation_name, "c_nationkey": cust.c_nationkey})},
        False)
    orders_1994 = orders.joinProbe(
        asian_customers, "o_custkey", lambda o: 19940101 <= o[0].o_orderdate < 19950101,
        lambda cust, order: {order.o_orderkey: record({"n_name": cust.n_name, "c_nationkey": cust.c_nationkey})},
        False)
    supplier_nations = supplier.sum(
        lambda s: {unique(record({"s_suppkey": s[0].s_suppkey, "s_nationkey": s[0].s_nationkey})): True})
    volume = lineitem.joinProbe(
        orders_1994, "l_orderkey", lambda l: True,
        lambda order, item: {order.n_name: item.l_extendedprice * (1.0 - item.l_discount)}
        if supplier_nations[record({"s_suppkey": item.l_suppkey, "s_nationkey": order.c_nationkey})] != None      # noqa: E711
        else None)
    local_volume = volume.sum(lambda g: {unique(record({"n_name": g[0], "revenue": g[1]})): True})
    return local_volume


# ---- q9: product type profit — green parts, (part, supplier) costs, order year, lineitem group-by ---
@sdql_compile({"nation": nation_type, "supplier": supplier_type, "part": part_type, "partsupp": partsupp_type,
               "orders": order_type, "lineitem": lineitem_type})
def q9(nation, supplier, part, partsupp, orders, lineitem):
    nation_names = nation.joinBuild("n_nationkey", lambda n: True, ["n_name"])
    supplier_nation = supplier.sum(lambda s: {unique(s[0].s_suppkey): nation_names[s[0].s_nationkey].n_name})
    green_parts = part.sum(lambda p: {unique(p[0].p_partkey): True} if "green" in p[0].p_name else None)
    green_costs = partsupp.sum(
        lambda ps: {
            unique(record({"ps_partkey": ps[0].ps_partkey, "ps_suppkey": ps[0].ps_suppkey})):
            record({"n_name": supplier_nation[ps[0].ps_suppkey], "ps_supplycost": ps[0].ps_supplycost})}
        if green_parts[ps[0].ps_partkey] != None else None)      # noqa: E711
    order_dates = orders.sum(lambda o: {unique(o[0].o_orderkey): o[0].o_orderdate})
    profit = lineitem.sum(
        lambda l: {
            record({"nation": green_costs[record({"ps_partkey": l[0].l_partkey, "ps_suppkey": l[0].l_suppkey})].n_name,
                    "o_year": extractYear(order_dates[l[0].l_orderkey])}):
            record({"sum_profit": l[0].l_extendedprice * (1.0 - l[0].l_discount)
                    - green_costs[record({"ps_partkey": l[0].l_partkey, "ps_suppkey": l[0].l_suppkey})].ps_supplycost * l[0].l_quantity})}
        if green_costs[record({"ps_partkey": l[0].l_partkey, "ps_suppkey": l[0].l_suppkey})] != None      # noqa: E711
        else None)
    profit_by_nation_year = profit.sum(lambda g: {unique(g[0].concat(g[1])): True})
    return profit_by_nation_year


# ---- q4: order priority checking — EXISTS(late lineitem) as a key set, count per priority ----------
@sdql_compile({"lineitem": lineitem_type, "orders": order_type})
def q4(lineitem, orders):
    late_orders = lineitem.sum(
        lambda l: {unique(l[0].l_orderkey): True} if l[0].l_commitdate < l[0].l_receiptdate else None)
    per_priority = orders.joinProbe(
        late_orders, "o_orderkey", lambda o: 19930701 <= o[0].o_orderdate < 19931001,
        lambda late, order: {order.o_orderpriority: 1})
    priority_counts = per_priority.sum(lambda g: {unique(record({"o_orderpriority": g[0], "order_count": g[1]})): True})
    return priority_counts


# ---- q14: promotion effect — two conditional sums and their ratio ---------------------------------
@sdql_compile({"part": part_type, "lineitem": lineitem_type})
def q14(part, lineitem):
    promo_parts = part.sum(lambda p: {unique(p[0].p_partkey): True} if startsWith(p[0].p_type, "PROMO") else None)
    september = lineitem.sum(
        lambda l: record({
            "A": l[0].l_extendedprice * (1.0 - l[0].l_discount) if promo_parts[l[0].l_partkey] != None else 0.0,      # noqa: E711
            "B": l[0].l_extendedprice * (1.0 - l[0].l_discount)})
        if 19950901 <= l[0].l_shipdate < 19951001 else None)
    promo_share = (100.0 * september.A) / september.B
    return promo_share


# ---- q18: large volume customers — HAVING on a row-keyed group-by, then two joins ------------------
def large_orders(min_quantity):
    """q18 with its HAVING threshold as a parameter (TPCH: 300; tests at tiny scale factors use less)."""

    @sdql_compile({"lineitem": lineitem_type, "orders": order_type, "customer": customer_type})
    def q18(lineitem, orders, customer):
        quantity_per_order = lineitem.sum(lambda l: {l[0].l_orderkey: l[0].l_quantity})
        big_orders = quantity_per_order.sum(lambda g: {unique(g[0]): True} if g[1] > min_quantity else None)
        customer_names = customer.joinBuild("c_custkey", lambda c: True, ["c_name"])
        big_order_rows = orders.joinProbe(
            customer_names, "o_custkey",
            lambda o: big_orders[o[0].o_orderkey] != None,      # noqa: E711
            lambda cust, order: {
                order.o_orderkey:
                record({"c_name": cust.c_name, "o_custkey": order.o_custkey, "o_orderkey": order.o_orderkey,
                        "o_orderdate": order.o_orderdate, "o_totalprice": order.o_totalprice})},
            False)
        quantity = lineitem.joinProbe(
            big_order_rows, "l_orderkey", lambda l: True,
            lambda order, item: {
                record({"c_name": order.c_name, "o_custkey": order.o_custkey, "o_orderkey": order.o_orderkey,
                        "o_orderdate": order.o_orderdate, "o_totalprice": order.o_totalprice}):
                record({"quantitysum": item.l_quantity})})
        large_volume = quantity.sum(lambda g: {unique(g[0].concat(g[1])): True})
        return large_volume

    return q18


q18 = large_orders(300)


# ---- q10: returned item reporting — late materialisation of seven customer fields -------------------
@sdql_compile({"nation": nation_type, "customer": customer_type, "orders": order_type, "lineitem": lineitem_type})
def q10(nation, customer, orders, lineitem):
    nation_names = nation.joinBuild("n_nationkey", lambda n: True, ["n_name"])
    customers = customer.joinBuild(
        "c_custkey", lambda c: True,
        ["c_custkey", "c_name", "c_acctbal", "c_address", "c_nationkey", "c_phone", "c_comment"])
    quarter_orders = orders.joinProbe(
        customers, "o_custkey", lambda o: 19931001 <= o[0].o_orderdate < 19940101,
        lambda cust, order: {
            order.o_orderkey:
            record({"c_custkey": cust.c_custkey, "c_name": cust.c_name, "c_acctbal": cust.c_acctbal,
                    "c_address": cust.c_address, "c_phone": cust.c_phone, "c_comment": cust.c_comment,
                    "n_name": nation_names[cust.c_nationkey].n_name})},
        False)
    returned = lineitem.joinProbe(
        quarter_orders, "l_orderkey", lambda l: l[0].l_returnflag == "R",
        lambda order, item: {
            record({"c_custkey": order.c_custkey, "c_name": order.c_name, "c_acctbal": order.c_acctbal,
                    "n_name": order.n_name, "c_address": order.c_address, "c_phone": order.c_phone,
                    "c_comment": order.c_comment}):
            item.l_extendedprice * (1.0 - item.l_discount)})
    lost_revenue = returned.sum(lambda g: {unique(record({
        "c_custkey": g[0].c_custkey, "c_name": g[0].c_name, "revenue": g[1], "c_acctbal": g[0].c_acctbal,
        "n_name": g[0].n_name, "c_address": g[0].c_address, "c_phone": g[0].c_phone, "c_comment": g[0].c_comment})): True})
    return lost_revenue


# =================================================================================================
# Queries whose loops need the open expression vocabulary (row programs, xplan.py): `or`, conditional
# values, text functions, conditions on looked-up fields and on aggregated values.
# =================================================================================================

# ---- q2: the European supplier whose offer for a size-15 brass part equals the part's total over European suppliers ------
# (the reference SUMS the supply costs of a part — its "## Min" stays a comment — so the offer that "equals the minimum" is, in
#  effect, the only European offer of the part; the formulation keeps that meaning.  The reference builds the eight-field result
#  record in the partsupp loop; here that loop yields the (part, supplier) pairs and the fields are looked up per pair.)
@sdql_compile({"region": region_type, "nation": nation_type, "supplier": supplier_type, "part": part_type, "partsupp": partsupp_type})
def q2(region, nation, supplier, part, partsupp):
    europe = region.sum(lambda r: {unique(r[0].r_regionkey): True} if r[0].r_name == "EUROPE" else None)
    european_nations = nation.sum(lambda n: {unique(n[0].n_nationkey): n[0].n_name} if europe[n[0].n_regionkey] != None else None)      # noqa: E711
    european_suppliers = supplier.sum(
        lambda s: {unique(s[0].s_suppkey): record({"s_acctbal": s[0].s_acctbal, "s_name": s[0].s_name, "n_name": european_nations[s[0].s_nationkey],
                                                   "s_address": s[0].s_address, "s_phone": s[0].s_phone, "s_comment": s[0].s_comment})}
        if european_nations[s[0].s_nationkey] != None else None)      # noqa: E711
    brass_parts = part.sum(lambda p: {unique(p[0].p_partkey): p[0].p_mfgr} if p[0].p_size == 15 and endsWith(p[0].p_type, "BRASS") else None)
    european_cost = partsupp.sum(
        lambda ps: {ps[0].ps_partkey: ps[0].ps_supplycost}
        if brass_parts[ps[0].ps_partkey] != None and european_suppliers[ps[0].ps_suppkey] != None else None)      # noqa: E711
    offers = partsupp.sum(
        lambda ps: {record({"p_partkey": ps[0].ps_partkey, "s_suppkey": ps[0].ps_suppkey}): 1}
        if european_cost[ps[0].ps_partkey] != None and european_cost[ps[0].ps_partkey] == ps[0].ps_supplycost      # noqa: E711
        and european_suppliers[ps[0].ps_suppkey] != None else None)      # noqa: E711
    best = offers.sum(lambda g: {unique(record({
        "s_acctbal": european_suppliers[g[0].s_suppkey].s_acctbal, "s_name": european_suppliers[g[0].s_suppkey].s_name,
        "n_name": european_suppliers[g[0].s_suppkey].n_name, "p_partkey": g[0].p_partkey, "p_mfgr": brass_parts[g[0].p_partkey],
        "s_address": european_suppliers[g[0].s_suppkey].s_address, "s_phone": european_suppliers[g[0].s_suppkey].s_phone,
        "s_comment": european_suppliers[g[0].s_suppkey].s_comment})): True})
    return best


# ---- q11: important stock of one nation: parts whose stock value exceeds a ten-thousandth of the nation's total -----------
# (the reference sums a record of a scalar and a dictionary in one loop; here the total and the per-part values are two sums)
@sdql_compile({"nation": nation_type, "supplier": supplier_type, "partsupp": partsupp_type})
def q11(nation, supplier, partsupp):
    germany = nation.sum(lambda n: {unique(n[0].n_nationkey): True} if n[0].n_name == "GERMANY" else None)
    german_suppliers = supplier.sum(lambda s: {unique(s[0].s_suppkey): True} if germany[s[0].s_nationkey] != None else None)      # noqa: E711
    threshold = partsupp.sum(
        lambda ps: (ps[0].ps_supplycost * ps[0].ps_availqty) * 0.0001 if german_suppliers[ps[0].ps_suppkey] != None else None)      # noqa: E711
    value_per_part = partsupp.sum(
        lambda ps: {ps[0].ps_partkey: ps[0].ps_supplycost * ps[0].ps_availqty} if german_suppliers[ps[0].ps_suppkey] != None else None)      # noqa: E711
    important = value_per_part.sum(lambda g: {unique(record({"ps_partkey": g[0], "value": g[1]})): True} if g[1] > threshold else None)
    return important


# ---- q7: volume shipping between two nations, per year --------------------------------------------------
@sdql_compile({"nation": nation_type, "supplier": supplier_type, "customer": customer_type, "orders": order_type, "lineitem": lineitem_type})
def q7(nation, supplier, customer, orders, lineitem):
    two_nations = nation.sum(
        lambda n: {unique(n[0].n_nationkey): n[0].n_name} if n[0].n_name == "FRANCE" or n[0].n_name == "GERMANY" else None)
    supplier_nation = supplier.sum(
        lambda s: {unique(s[0].s_suppkey): two_nations[s[0].s_nationkey]} if two_nations[s[0].s_nationkey] != None else None)      # noqa: E711
    customer_nation = customer.sum(
        lambda c: {unique(c[0].c_custkey): two_nations[c[0].c_nationkey]} if two_nations[c[0].c_nationkey] != None else None)      # noqa: E711
    order_nation = orders.sum(
        lambda o: {unique(o[0].o_orderkey): customer_nation[o[0].o_custkey]} if customer_nation[o[0].o_custkey] != None else None)  # noqa: E711
    shipped = lineitem.sum(
        lambda l: {
            record({"supp_nation": supplier_nation[l[0].l_suppkey], "cust_nation": order_nation[l[0].l_orderkey],
                    "l_year": extractYear(l[0].l_shipdate)}):
            record({"revenue": l[0].l_extendedprice * (1.0 - l[0].l_discount)})}
        if 19950101 <= l[0].l_shipdate <= 19961231
        and supplier_nation[l[0].l_suppkey] != None and order_nation[l[0].l_orderkey] != None      # noqa: E711
        and ((supplier_nation[l[0].l_suppkey] == "FRANCE" and order_nation[l[0].l_orderkey] == "GERMANY")
             or (supplier_nation[l[0].l_suppkey] == "GERMANY" and order_nation[l[0].l_orderkey] == "FRANCE"))
        else None)
    volume = shipped.sum(lambda g: {unique(g[0].concat(g[1])): True})
    return volume


# ---- q8: national market share within a region, per year --------------------------------------------------
@sdql_compile({"region": region_type, "nation": nation_type, "customer": customer_type, "supplier": supplier_type,
               "part": part_type, "orders": order_type, "lineitem": lineitem_type})
def q8(region, nation, customer, supplier, part, orders, lineitem):
    america = region.sum(lambda r: {unique(r[0].r_regionkey): True} if r[0].r_name == "AMERICA" else None)
    american_nations = nation.sum(lambda n: {unique(n[0].n_nationkey): True} if america[n[0].n_regionkey] != None else None)      # noqa: E711
    nation_names = nation.sum(lambda n: {unique(n[0].n_nationkey): n[0].n_name})
    supplier_nation = supplier.sum(lambda s: {unique(s[0].s_suppkey): s[0].s_nationkey})
    customer_nation = customer.sum(lambda c: {unique(c[0].c_custkey): c[0].c_nationkey})
    steel_parts = part.sum(lambda p: {unique(p[0].p_partkey): True} if p[0].p_type == "ECONOMY ANODIZED STEEL" else None)
    orders_95_96 = orders.sum(
        lambda o: {unique(o[0].o_orderkey): record({"o_custkey": o[0].o_custkey, "o_orderdate": o[0].o_orderdate})}
        if 19950101 <= o[0].o_orderdate <= 19961231 else None)
    volume = lineitem.sum(
        lambda l: {
            extractYear(orders_95_96[l[0].l_orderkey].o_orderdate):
            record({"A": l[0].l_extendedprice * (1.0 - l[0].l_discount)
                    if nation_names[supplier_nation[l[0].l_suppkey]] == "BRAZIL" else 0.0,
                    "B": l[0].l_extendedprice * (1.0 - l[0].l_discount)})}
        if steel_parts[l[0].l_partkey] != None and orders_95_96[l[0].l_orderkey] != None      # noqa: E711
        and american_nations[customer_nation[orders_95_96[l[0].l_orderkey].o_custkey]] != None      # noqa: E711
        else None)
    share = volume.sum(lambda g: {unique(record({"o_year": g[0], "mkt_share": g[1].A / g[1].B})): True})
    return share


# ---- q12: late line items of two ship modes by the priority class of their order ----------------------------------
# (the reference nests a dictionary per order, {l_orderkey: {l_shipmode: n}}, and walks it from the orders side; here the
#  order's priority is looked up from the lineitem side, so the loop is one group-by with two conditional counts)
@sdql_compile({"orders": order_type, "lineitem": lineitem_type})
def q12(orders, lineitem):
    priority = orders.sum(lambda o: {unique(o[0].o_orderkey): o[0].o_orderpriority})
    late_lines = lineitem.sum(
        lambda l: {
            record({"l_shipmode": l[0].l_shipmode}):
            record({"high_line_count": 1 if priority[l[0].l_orderkey] == "1-URGENT" or priority[l[0].l_orderkey] == "2-HIGH" else 0,
                    "low_line_count": 1 if priority[l[0].l_orderkey] != "1-URGENT" and priority[l[0].l_orderkey] != "2-HIGH" else 0})}
        if (l[0].l_shipmode == "MAIL" or l[0].l_shipmode == "SHIP")
        and 19940101 <= l[0].l_receiptdate < 19950101
        and l[0].l_shipdate < l[0].l_commitdate and l[0].l_commitdate < l[0].l_receiptdate
        and priority[l[0].l_orderkey] != None      # noqa: E711
        else None)
    by_mode = late_lines.sum(lambda g: {unique(g[0].concat(g[1])): True})
    return by_mode


# ---- q13: customer distribution by number of (non-special) orders --------------------------------------------
@sdql_compile({"orders": order_type, "customer": customer_type})
def q13(orders, customer):
    orders_per_customer = orders.sum(
        lambda o: {o[0].o_custkey: 1}
        if not (firstIndex(o[0].o_comment, "special") != -1
                and firstIndex(o[0].o_comment, "requests") > firstIndex(o[0].o_comment, "special") + 6)
        else None)
    distribution = customer.sum(
        lambda c: {record({"c_count": orders_per_customer[c[0].c_custkey] if orders_per_customer[c[0].c_custkey] != None else 0}):      # noqa: E711
                   record({"custdist": 1})})
    histogram = distribution.sum(lambda g: {unique(g[0].concat(g[1])): True})
    return histogram


# ---- q15: revenue per supplier for one quarter; TPCH's "top supplier" is `.top(1, [("total_revenue", "desc")])` ----
@sdql_compile({"lineitem": lineitem_type, "supplier": supplier_type})
def q15(lineitem, supplier):
    revenue = lineitem.sum(
        lambda l: {l[0].l_suppkey: l[0].l_extendedprice * (1.0 - l[0].l_discount)}
        if 19960101 <= l[0].l_shipdate < 19960401 else None)
    suppliers = supplier.joinBuild("s_suppkey", lambda s: True, ["s_name", "s_address", "s_phone"])
    ranked = revenue.sum(lambda g: {unique(record({
        "s_suppkey": g[0], "s_name": suppliers[g[0]].s_name, "s_address": suppliers[g[0]].s_address,
        "s_phone": suppliers[g[0]].s_phone, "total_revenue": g[1]})): True})
    return ranked


# ---- q16: suppliers per (brand, type, size) of the wanted parts, complaint-ridden suppliers left out -----------------
# (COUNT(DISTINCT ps_suppkey): the distinct (brand, type, size, supplier) combinations first, then their number per group)
@sdql_compile({"part": part_type, "supplier": supplier_type, "partsupp": partsupp_type})
def q16(part, supplier, partsupp):
    wanted_parts = part.sum(
        lambda p: {unique(p[0].p_partkey): record({"p_brand": p[0].p_brand, "p_type": p[0].p_type, "p_size": p[0].p_size})}
        if p[0].p_brand != "Brand#45" and not startsWith(p[0].p_type, "MEDIUM POLISHED")
        and (p[0].p_size == 49 or p[0].p_size == 14 or p[0].p_size == 23 or p[0].p_size == 45
             or p[0].p_size == 19 or p[0].p_size == 3 or p[0].p_size == 36 or p[0].p_size == 9)
        else None)
    complained_about = supplier.sum(
        lambda s: {unique(s[0].s_suppkey): True}
        if firstIndex(s[0].s_comment, "Customer") != -1
        and firstIndex(s[0].s_comment, "Complaints") > firstIndex(s[0].s_comment, "Customer") + 7
        else None)
    offers = partsupp.sum(
        lambda ps: {record({"p_brand": wanted_parts[ps[0].ps_partkey].p_brand, "p_type": wanted_parts[ps[0].ps_partkey].p_type,
                            "p_size": wanted_parts[ps[0].ps_partkey].p_size, "ps_suppkey": ps[0].ps_suppkey}): 1}
        if wanted_parts[ps[0].ps_partkey] != None and complained_about[ps[0].ps_suppkey] == None      # noqa: E711
        else None)
    suppliers_per_group = offers.sum(
        lambda g: {record({"p_brand": g[0].p_brand, "p_type": g[0].p_type, "p_size": g[0].p_size}): record({"supplier_cnt": 1})})
    counted = suppliers_per_group.sum(lambda g: {unique(g[0].concat(g[1])): True})
    return counted


# ---- q17: small-quantity-order revenue: rows below a fifth of their part's average quantity ------------------------
@sdql_compile({"part": part_type, "lineitem": lineitem_type})
def q17(part, lineitem):
    boxed = part.sum(lambda p: {unique(p[0].p_partkey): True}
                     if p[0].p_brand == "Brand#23" and p[0].p_container == "MED BOX" else None)
    quantity_per_part = lineitem.joinProbe(
        boxed, "l_partkey", lambda l: True,
        lambda hit, item: {item.l_partkey: record({"l_quantity": item.l_quantity, "count": 1.0})})
    small_orders = lineitem.joinProbe(
        quantity_per_part, "l_partkey", lambda l: True,
        lambda part_total, item: item.l_extendedprice
        if 0.2 * (part_total.l_quantity / part_total.count) > item.l_quantity else 0.0)
    yearly = small_orders / 7.0
    return yearly


# ---- q19: discounted revenue: three brand / container / quantity / size combinations -----------------------------------
@sdql_compile({"part": part_type, "lineitem": lineitem_type})
def q19(part, lineitem):
    candidates = part.sum(
        lambda p: {unique(p[0].p_partkey): record({"p_brand": p[0].p_brand})}
        if (p[0].p_brand == "Brand#12" and 1 <= p[0].p_size <= 5
            and (p[0].p_container == "SM CASE" or p[0].p_container == "SM BOX" or p[0].p_container == "SM PACK" or p[0].p_container == "SM PKG"))
        or (p[0].p_brand == "Brand#23" and 1 <= p[0].p_size <= 10
            and (p[0].p_container == "MED BAG" or p[0].p_container == "MED BOX" or p[0].p_container == "MED PKG" or p[0].p_container == "MED PACK"))
        or (p[0].p_brand == "Brand#34" and 1 <= p[0].p_size <= 15
            and (p[0].p_container == "LG CASE" or p[0].p_container == "LG BOX" or p[0].p_container == "LG PACK" or p[0].p_container == "LG PKG"))
        else None)
    discounted = lineitem.joinProbe(
        candidates, "l_partkey",
        lambda l: l[0].l_shipinstruct == "DELIVER IN PERSON" and (l[0].l_shipmode == "AIR" or l[0].l_shipmode == "AIR REG"),
        lambda cand, item: item.l_extendedprice * (1.0 - item.l_discount)
        if (cand.p_brand == "Brand#12" and 1 <= item.l_quantity <= 11)
        or (cand.p_brand == "Brand#23" and 10 <= item.l_quantity <= 20)
        or (cand.p_brand == "Brand#34" and 20 <= item.l_quantity <= 30)
        else 0.0)
    result = sr_dict({record({"revenue": discounted}): True})
    return result


# ---- q20: potential part promotion: suppliers of one nation with excess stock of forest parts ------------------------
@sdql_compile({"nation": nation_type, "supplier": supplier_type, "part": part_type, "partsupp": partsupp_type, "lineitem": lineitem_type})
def q20(nation, supplier, part, partsupp, lineitem):
    canada = nation.sum(lambda n: {unique(n[0].n_nationkey): True} if n[0].n_name == "CANADA" else None)
    canadian_suppliers = supplier.sum(lambda s: {unique(s[0].s_suppkey): True} if canada[s[0].s_nationkey] != None else None)      # noqa: E711
    forest_parts = part.sum(lambda p: {unique(p[0].p_partkey): True} if startsWith(p[0].p_name, "forest") else None)
    half_shipped_1994 = lineitem.sum(
        lambda l: {record({"l_partkey": l[0].l_partkey, "l_suppkey": l[0].l_suppkey}): 0.5 * l[0].l_quantity}
        if 19940101 <= l[0].l_shipdate < 19950101
        and forest_parts[l[0].l_partkey] != None and canadian_suppliers[l[0].l_suppkey] != None      # noqa: E711
        else None)
    overstocked = partsupp.sum(
        lambda ps: {unique(ps[0].ps_suppkey): True}
        if half_shipped_1994[record({"l_partkey": ps[0].ps_partkey, "l_suppkey": ps[0].ps_suppkey})] != None      # noqa: E711
        and ps[0].ps_availqty > half_shipped_1994[record({"l_partkey": ps[0].ps_partkey, "l_suppkey": ps[0].ps_suppkey})]
        else None)
    promotion = supplier.sum(
        lambda s: {unique(record({"s_name": s[0].s_name, "s_address": s[0].s_address})): True}
        if overstocked[s[0].s_suppkey] != None else None)      # noqa: E711
    return promotion


# ---- q22: global sales opportunity: well-funded customers of seven country codes without orders -------------------------
@sdql_compile({"orders": order_type, "customer": customer_type})
def q22(orders, customer):
    customers_with_orders = orders.sum(lambda o: {unique(o[0].o_custkey): True})
    positive = customer.sum(
        lambda c: record({"c_acctbal": c[0].c_acctbal, "count": 1.0})
        if c[0].c_acctbal > 0.0
        and (startsWith(c[0].c_phone, "13") or startsWith(c[0].c_phone, "31") or startsWith(c[0].c_phone, "23")
             or startsWith(c[0].c_phone, "29") or startsWith(c[0].c_phone, "30") or startsWith(c[0].c_phone, "18")
             or startsWith(c[0].c_phone, "17"))
        else None)
    average_balance = positive.c_acctbal / positive.count
    idle_rich = customer.sum(
        lambda c: {record({"cntrycode": substr(c[0].c_phone, 0, 1)}): record({"numcust": 1, "totalacctbal": c[0].c_acctbal})}
        if c[0].c_acctbal > average_balance and customers_with_orders[c[0].c_custkey] == None      # noqa: E711
        and (startsWith(c[0].c_phone, "13") or startsWith(c[0].c_phone, "31") or startsWith(c[0].c_phone, "23")
             or startsWith(c[0].c_phone, "29") or startsWith(c[0].c_phone, "30") or startsWith(c[0].c_phone, "18")
             or startsWith(c[0].c_phone, "17"))
        else None)
    opportunity = idle_rich.sum(lambda g: {unique(g[0].concat(g[1])): True})
    return opportunity


QUERIES = {"q6": q6, "q1": q1, "q3": q3, "q5": q5, "q9": q9, "q4": q4, "q14": q14, "q18": q18, "q10": q10,
           "q2": q2, "q11": q11, "q7": q7, "q8": q8, "q12": q12, "q13": q13, "q15": q15, "q16": q16, "q17": q17, "q19": q19, "q20": q20, "q22": q22}

_TABLE_OF_PARAM = {"lineitem": "lineitem", "orders": "orders", "customer": "customer", "supplier": "supplier", "part": "part",
                   "partsupp": "partsupp", "nation": "nation", "region": "region"}


def tables_of(query):
    """Database table names in the positional order of a decorated query (decorator dict order == call order)."""
    return [_TABLE_OF_PARAM[p] for p in query.__sdql_in_type__]


QUERY_TABLES = {name: tables_of(fn) for name, fn in QUERIES.items()}


def register(name, query, order=None):
    """Add a query (e.g. `large_orders(230)`) under `name`; `order` = its TPCH ORDER BY / LIMIT."""
    QUERIES[name] = query
    QUERY_TABLES[name] = tables_of(query)
    if order is not None:
        TPCH_ORDER[name] = order


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
    "q7": (100, [("supp_nation", "asc"), ("cust_nation", "asc"), ("l_year", "asc")]),
    "q8": (100, [("o_year", "asc")]),
    "q2": (100, [("s_acctbal", "desc"), ("n_name", "asc"), ("s_name", "asc"), ("p_partkey", "asc")]),
    "q11": (100, [("value", "desc")]),
    "q12": (100, [("l_shipmode", "asc")]),
    "q13": (100, [("custdist", "desc"), ("c_count", "desc")]),
    "q16": (100, [("supplier_cnt", "desc"), ("p_brand", "asc"), ("p_type", "asc"), ("p_size", "asc")]),
    "q15": (1, [("total_revenue", "desc")]),
    "q20": (100, [("s_name", "asc")]),
    "q22": (100, [("cntrycode", "asc")]),
}
