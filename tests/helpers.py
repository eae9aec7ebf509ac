"""Shared helpers for the parity tests: golden decoding, input regeneration, result comparison."""
import importlib.util
import math
import os

import numpy as np

from sdqlpy_amd import engine as eng_mod
from sdqlpy_amd import frontend, tpch
from sdqlpy_amd import tpch_queries as Q
from sdqlpy_amd.result import ResultSet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)

_db_cache = {}


def case_db(case):
    """Regenerate the inputs of a golden case and check they are the ones the reference saw."""
    key = case["name"]
    if key not in _db_cache:
        qs = list(case["results"].keys())
        base = tpch.generate(case["sf"], case["seed"], tables=case["tables"], columns=tpch.columns_for(qs), threads=4)
        db = make_golden.VARIANTS[case["variant"]](base)
        assert tpch.fingerprint(db) == case["fingerprint"], "generator drifted: golden inputs no longer reproducible"
        _db_cache[key] = db
    return _db_cache[key]


def dec(v):
    return float.fromhex(v["f"]) if isinstance(v, dict) else v


def golden_rows(result):
    return [tuple(dec(x) for x in row) for row in result["rows"]]


def run_query(ctx_engine, name, db):
    plan = frontend.lower_function(Q.QUERIES[name])
    return eng_mod.execute_plan(ctx_engine, plan, [db[t] for t in Q.QUERY_TABLES[name]])


def result_rows(res, columns):
    """Rows of a ResultSet in the golden column order, sorted like the golden file."""
    assert isinstance(res, ResultSet)
    assert sorted(res.columns) == sorted(columns), (res.columns, columns)
    cols = [res.column(c).tolist() for c in columns]
    rows = [tuple(r) for r in zip(*cols)]
    return sorted(rows, key=lambda r: [(0, x) if isinstance(x, int) else (1, x) if isinstance(x, str) else (2, x) for x in r])


def assert_rows_match(got, want, rel=0.0, what=""):
    """ints / strings exact; doubles exact when rel == 0 else within rel (relative)."""
    assert len(got) == len(want), "%s: %d rows, expected %d" % (what, len(got), len(want))
    if rel > 0:   # float columns may perturb the sort order of equal-key rows: align on the non-float fields
        key = lambda r: tuple(x for x in r if not isinstance(x, float))
        got, want = sorted(got, key=key), sorted(want, key=key)
    for g, w in zip(got, want):
        assert len(g) == len(w)
        for a, b in zip(g, w):
            if isinstance(b, float):
                assert isinstance(a, float), (what, g, w)
                if rel == 0.0:
                    assert a == b or (math.isnan(a) and math.isnan(b)), "%s: %r != %r in %r vs %r" % (what, a.hex(), b.hex(), g, w)
                else:
                    assert abs(a - b) <= rel * max(abs(a), abs(b), 1e-300), "%s: %r vs %r" % (what, a, b)
            else:
                assert a == b, "%s: %r != %r in %r vs %r" % (what, a, b, g, w)


def check_against_golden(res, gold, rel, what):
    if gold["kind"] == "scalar":
        want = dec(gold["value"])
        got = float(res)
        if rel == 0.0:
            assert got == want, "%s: %s != %s" % (what, got.hex(), want.hex())
        else:
            assert abs(got - want) <= rel * max(abs(want), 1e-300), "%s: %r vs %r" % (what, got, want)
        return
    want = golden_rows(gold)
    if not gold["columns"]:
        assert res is None or res.size() == 0, "%s: expected an empty result" % what
        return
    assert_rows_match(result_rows(res, gold["columns"]), want, rel, what)
