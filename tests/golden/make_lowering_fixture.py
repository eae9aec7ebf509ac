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
    for node in tree.body:
        if not (isinstance(node, ast.FunctionDef) and node.name.startswith("q") and node.name[1:].isdigit()):
            continue
        first = min([d.lineno for d in node.decorator_list] + [node.lineno])
        text = "\n".join(lines[first - 1:node.end_lineno])
        rec = {"ref_lines": [first, node.end_lineno]}
        try:
            plan = frontend.lower_source(text, node.name, first)
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
    out["shipped_plan_digests"] = shipped
    path = os.path.join(HERE, "reference_lowering.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()
