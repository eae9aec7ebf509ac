"""Table ingest against the reference's own loader (fixtures: tests/golden/make_loader_golden.py).

`tests/golden/tbl_expected.npz` holds what the reference's `read_csv` (reference
src/sdqlpy/sdql_lib.py:69-129) returned for every file under tests/golden/tbl/.  Both ingest paths of
this repository — the native parser and the csv-module path — must return the same arrays: same
order, dtype, shape and values (bit-exact, doubles included)."""
import os

import numpy as np
import pytest

from sdqlpy_amd import loader, tpch
from sdqlpy_amd import sdql_lib as L

HERE = os.path.dirname(os.path.abspath(__file__))
TBL = os.path.join(HERE, "golden", "tbl")
EDGE = {L.record({"e_key": int, "e_val": float, "e_day": L.date, "e_txt": L.string(6), "e_NA": L.string(1)}): bool}


def schema_of(fname):
    return EDGE if fname.startswith("edge_") else tpch.SCHEMAS[fname[:-4]]


def expected():
    z = np.load(os.path.join(HERE, "golden", "tbl_expected.npz"))
    out = {}
    for key in z.files:
        f, col = key.split("/")
        out.setdefault(f, {})[col] = z[key]
    return out


def same(a, b):
    if a.shape != b.shape:
        return False
    if a.size == 0:
        return True              # the reference's empty int columns come out as float64 (np.array([])): shape is what matters
    if a.dtype != b.dtype:
        return False
    if a.dtype.kind == "f":
        return bool((a.view(np.int64) == b.view(np.int64)).all() or (np.isnan(a) == np.isnan(b)).all() and (a[~np.isnan(a)] == b[~np.isnan(b)]).all())
    return bool((a == b).all())


@pytest.mark.parametrize("fname", sorted(expected()))
def test_read_csv_matches_the_reference_loader(fname, capsys):
    want = expected()[fname]
    table = L.read_csv(os.path.join(TBL, fname), schema_of(fname), fname)
    assert "Finished." in capsys.readouterr().out          # the reference prints the same progress line
    c = table.getContainer()
    assert c["headers"] == list(next(iter(schema_of(fname).keys())).getContainer().keys())
    assert set(c["headers"]) == set(want)
    for h, a in zip(c["headers"], c["data"]):
        assert same(a, want[h]), (fname, h, a[:5], want[h][:5])


@pytest.mark.parametrize("fname", ["lineitem.tbl", "part.tbl", "edge_plain.tbl", "edge_unterminated.tbl", "edge_empty.tbl"])
def test_native_parser_takes_plain_files(fname):
    types = list(next(iter(schema_of(fname).keys())).getContainer().values())
    cols = loader.read_text_native(os.path.join(TBL, fname), types)            # must not decline
    want = expected()[fname]
    for h, a in zip(next(iter(schema_of(fname).keys())).getContainer().keys(), cols):
        assert same(a, want[h]), (fname, h)


def test_native_parser_declines_what_only_python_reads():
    types = list(next(iter(EDGE.keys())).getContainer().values())
    with pytest.raises(loader.Declined):
        loader.read_text_native(os.path.join(TBL, "edge_general.tbl"), types)
    general = loader._read_text_general(os.path.join(TBL, "edge_general.tbl"), types)
    want = expected()["edge_general.tbl"]
    for h, a in zip(next(iter(EDGE.keys())).getContainer().keys(), general):
        assert same(a, want[h]), h


def test_general_path_agrees_with_native_on_generated_tables(tmp_path):
    db = tpch.generate(0.002, tables=["lineitem", "orders", "customer"])
    paths = tpch.write_tbl(str(tmp_path), db)
    for t, p in paths.items():
        types = list(next(iter(tpch.SCHEMAS[t].keys())).getContainer().values())
        a, b = loader.read_text_native(p, types, threads=3), loader._read_text_general(p, types)
        assert all(same(x, y) for x, y in zip(a, b)), t
        have = dict(zip(db[t].getContainer()["headers"], db[t].getContainer()["data"]))
        for h, col in zip(next(iter(tpch.SCHEMAS[t].keys())).getContainer().keys(), a):
            if h in have:
                assert (col == have[h]).all(), (t, h)          # text round trip is exact (2-decimal money, yyyymmdd dates)


def test_ragged_lines_are_reported(tmp_path):
    p = tmp_path / "bad.tbl"
    p.write_text("1|0.5|1992-01-01|a|\n2|0.5|1992-01-01|\n")
    with pytest.raises(ValueError):
        L.read_csv(str(p), EDGE, "bad")
    with pytest.raises(OSError):
        L.read_csv(str(tmp_path / "missing.tbl"), EDGE, "missing")


def test_binary_column_format_round_trip(tmp_path):
    db = tpch.generate(0.002, tables=["lineitem"])
    d = L.write_columns(str(tmp_path / "lineitem.cols"), db["lineitem"])
    back = L.read_columns(d)
    a, b = db["lineitem"].getContainer(), back.getContainer()
    assert a["headers"] == b["headers"]
    for x, y in zip(a["data"], b["data"]):
        assert x.dtype == y.dtype and (x == y).all() and not y.flags.writeable      # memory-mapped read-only
    schema = {L.record({"l_shipdate": L.date, "l_quantity": float}): bool}
    sub = L.read_columns(d, schema, mmap=False).getContainer()
    assert sub["headers"] == ["l_shipdate", "l_quantity"] and len(sub["data"][0]) == len(a["data"][0])
    with pytest.raises(KeyError):
        L.read_columns(d, {L.record({"nope": int}): bool})


def test_dictionary_encoding_equals_numpy_unique():
    """loader.dict_encode (what text group keys and low-cardinality text predicates run on): the same
    codes and sorted dictionary as np.unique, None beyond the distinct-value limit."""
    rng = np.random.default_rng(3)
    vals = np.array(["MAIL", "SHIP", "AIR", "REG AIR", "日本", "", "é", "TRUCK", "FOB", "RAIL", "A", "AB", "a"], "<U7")
    a = vals[rng.integers(0, len(vals), 200000)]
    codes, dic = loader.dict_encode(a, threads=3)
    u, inv = np.unique(a, return_inverse=True)
    assert dic.dtype == a.dtype and (dic == u).all() and (codes == inv).all()
    assert loader.dict_encode(np.array(["x%d" % i for i in range(300)], "<U5"), max_distinct=256) is None
    one = loader.dict_encode(np.array(["same"] * 10, "<U4"))
    assert one[0].tolist() == [0] * 10 and one[1].tolist() == ["same"]
    empty = loader.dict_encode(np.array([], "<U3"))
    assert len(empty[0]) == 0 and len(empty[1]) == 0


def test_text_tables_to_query_results_match_the_reference(oracle_lib):
    """.tbl -> read_csv -> front end / planner -> C ABI (CPU implementation) against what the reference
    gets from ITS read_csv + ITS queries on the same files (tests/golden/make_tbl_query_golden.py)."""
    import helpers
    from sdqlpy_amd import engine
    eng = engine.Engine(oracle_lib.context(threads=1))
    try:
        helpers.check_tbl_queries(eng, 0.0, 1e-12)
    finally:
        eng.close()
