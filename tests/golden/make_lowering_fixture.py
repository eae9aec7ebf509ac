#!/usr/bin/env python3
"""Does the REFERENCE's own query text lower through this package's front end?  (build container only)

The shipped workload (sdqlpy_amd/tpch_queries.py) is this package's own formulation of the TPCH
queries.  A user of the reference arrives with the reference's formulation — its TPCH script
test/test_all.py — so this script feeds THAT text, parsed from /root/reference at run time, to
`frontend.lower_source` and records, per query, whether it lowers, the planner's refusal if not, and
a digest of the name-free plan (`Plan.fingerprint`).  Only this data is committed
(tests/golden/reference_lowering.json); nothing of the reference's text is.

    python tests/golden/make_lowering_fixture.py
"""
import ast
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_SCRIPT = "/root/reference/test/test_all.py"

from sdqlpy_amd import frontend  # noqa: E402
from sdqlpy_amd import tpch_queries as Q  # noqa: E402


def digest(text):
    return hashlib.sha1(text.encode()).hexdigest()[:16]


TABLE_OF_TYPE = {"lineitem_type": "lineitem", "customer_type": "customer", "order_type": "orders", "nation_type": "nation", "region_type": "region",
                 "part_type": "part", "partsupp_type": "partsupp", "supplier_type": "supplier"}


def run_against_goldens(plans, records):
    """The plans lowered from the reference's OWN text, executed by the CPU implementation (one thread) on the inputs of every golden
    case and compared with the reference's results bit for bit (q10: 1e-12, two-level summation): `matches_reference_results` per
    query, or the first difference / refusal.  q15 is expected to differ: the reference's text keeps the supplier whose revenue
    EQUALS a constant typed in for dbgen's SF=1 data (test/test_all.py:733), which the golden script replaces."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from sdqlpy_amd import abi, engine
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    eng = engine.Engine(abi.Library(os.path.join(ROOT, "oracle", "libsdqloracle.so")).context(threads=1))
    status = {}
    for fname in ("tpch_golden.json", "tpch_golden_more.json", "tpch_golden_wide.json"):
        with open(os.path.join(HERE, fname)) as fh:
            gold = json.load(fh)
        for case in gold["cases"]:
            db = helpers.case_db(case)
            for q, want in case["results"].items():
                if q not in plans or status.get(q, True) is not True:
                    continue
                plan, tabs = plans[q]
                try:
                    res = engine.execute_plan(eng, plan, [db[t] for t in tabs])
                    helpers.check_against_golden(res, want, 1e-12 if q == "q10" else 0.0, "reference text/%s/%s" % (case["name"], q))
                    status[q] = True
                except (AssertionError, frontend.UnsupportedQuery, abi.SdqhError) as exc:
                    status[q] = "%s: %s" % (type(exc).__name__, str(exc).split("\n")[0][:160])
            eng.clear()
    eng.close()
    for q, st in status.items():
        records[q]["matches_reference_results"] = st
        print(q, "reference text vs goldens:", st)


def main():
    src = open(REF_SCRIPT).read()
    tree = ast.parse(src)
    lines = src.splitlines()
    shipped = {}
    for name, fn in Q.QUERIES.items():
        shipped[name] = digest(frontend.lower_function(fn).fingerprint())
    out = {"meta": {"made_by": "tests/golden/make_lowering_fixture.py",
                    "source": "reference test/test_all.py, parsed at run time (not stored)"},
           "queries": {}}
    plans = {}
    for node in tree.body:
        if not (isinstance(node, ast.FunctionDef) and node.name.startswith("q") and node.name[1:].isdigit()):
            continue
        first = min([d.lineno for d in node.decorator_list] + [node.lineno])
        text = "\n".join(lines[first - 1:node.end_lineno])
        rec = {"ref_lines": [first, node.end_lineno]}
        try:
            plan = frontend.lower_source(text, node.name, first)
            plans[node.name] = (plan, [TABLE_OF_TYPE[v.id] for v in node.decorator_list[0].args[0].values])
            rec["lowers"] = True
            rec["scan_loops"] = sum(isinstance(o, frontend.ScanOp) for o in plan.ops)
            rec["plan_digest"] = digest(plan.fingerprint())
            if node.name in shipped:
                rec["same_plan_as_shipped_formulation"] = rec["plan_digest"] == shipped[node.name]
        except frontend.UnsupportedQuery as exc:
            rec["lowers"] = False
            rec["refused"] = str(exc).split("\n")[0].split(": ", 1)[-1]
        out["queries"][node.name] = rec
        print(node.name, rec)
    run_against_goldens(plans, out["queries"])
    out["shipped_plan_digests"] = shipped
    path = os.path.join(HERE, "reference_lowering.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
