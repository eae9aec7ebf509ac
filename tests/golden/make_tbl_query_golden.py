#!/usr/bin/env python3
"""Reference results for the whole ingest -> query path on the committed text tables.  (build container only)

The REFERENCE does everything here: its own `read_csv` (src/sdqlpy/sdql_lib.py:69-129) loads
tests/golden/tbl/*.tbl with its own schema markers, and its own TPCH queries (parsed from
test/test_all.py at run time, Python mode) run on what it loaded.  Committed: the results only
(tests/golden/tbl_query_golden.json).  tests/test_hip_parity.py then checks
`.tbl -> sdqlpy_amd.read_csv -> HIP kernels` against them.

    python tests/golden/make_tbl_query_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G            # noqa: E402  (load_reference / encode_result; imports the reference unmodified)
import make_loader_golden as LG    # noqa: E402

TBL = os.path.join(HERE, "tbl")
QUERIES = ["q1", "q3", "q5", "q6", "q9", "q4", "q10", "q14"]


def main():
    ref, queries = G.load_reference()
    tables = {}
    for t in G.ALL_TABLES:
        tables[t] = ref.read_csv(os.path.join(TBL, t + ".tbl"), LG.ref_schema(ref, LG.fields_of(t)), t)
    out = {"meta": {"made_by": "tests/golden/make_tbl_query_golden.py",
                    "reference": "edin-dal/sdqlpy: its read_csv on tests/golden/tbl/*.tbl, its queries (test/test_all.py) in Python mode"},
           "rows": {t: len(tables[t].getContainer()["data"][0]) for t in tables}, "results": {}}
    for q in QUERIES:
        args = [tables[t] for t in G.QUERY_TABLES[q]]
        try:
            res = queries[q](*args)
        except (AttributeError, TypeError):
            res = None
        out["results"][q] = G.encode_result(ref, res)
        r = out["results"][q]
        print(q, r["value"] if r["kind"] == "scalar" else "%d rows" % len(r["rows"]))
    with open(os.path.join(HERE, "tbl_query_golden.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))


if __name__ == "__main__":
    main()
