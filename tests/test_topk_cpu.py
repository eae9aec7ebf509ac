"""ORDER BY ... LIMIT k (BASELINE config 3 "hash joins + top-k") on the CPU
implementation of the ABI and through the engine: the device operator, the host ordering used for
small / non-table results, and their agreement."""
import os

import pytest

from sdqlpy_amd import abi, engine, frontend, tpch
from sdqlpy_amd import tpch_queries as Q

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def oracle():
    eng = engine.Engine(abi.Library(os.path.join(ROOT, "oracle", "libsdqloracle.so")).context(threads=4))
    yield eng
    eng.close()


def test_table_topk_against_numpy(oracle):
    from helpers import topk_case
    topk_case(oracle.ctx, n=60000)


def test_topk_arguments_are_checked(oracle):
    import numpy as np
    ctx = oracle.ctx
    t = ctx.hash_build_unique(10, abi.make_filter(), [], ctx.upload(np.arange(10, dtype=np.int64)), [], accumulate=False)
    for k, spec in ((0, [(abi.SORT_KEY, 0, False, False)]), (abi.MAX_TOPK + 1, [(abi.SORT_KEY, 0, False, False)])):
        with pytest.raises(abi.SdqhError) as e:
            ctx.table_topk(t, 0, k, spec)
        assert e.value.code == abi.ERR_UNSUPPORTED
    for spec in ([(abi.SORT_PAYLOAD, 0, False, False)], [(abi.SORT_VALUE, 0, False, True)], [(abi.SORT_HITS, 0, False, False)]):
        with pytest.raises(abi.SdqhError) as e:
            ctx.table_topk(t, 0, 3, spec)
        assert e.value.code == abi.ERR_INVALID
    k, _, _, _ = ctx.table_topk(t, 0, 3, [(abi.SORT_KEY, 0, True, False)], want_hits=False)
    assert k.tolist() == [9, 8, 7]
    t.free()


@pytest.mark.parametrize("q", ["q1", "q3", "q5", "q9", "q10", "q18"])
def test_query_top_equals_ordering_the_full_result(oracle, q):
    qs = (q,)
    db = tpch.generate(0.02, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
    plan = frontend.lower_function(Q.QUERIES[q].__sdql_func__, Q.QUERIES[q].__sdql_in_type__)
    args = [db[t] for t in Q.QUERY_TABLES[q]]
    full = engine.execute_plan(oracle, plan, args)
    k, order = Q.TPCH_ORDER[q]
    for kk in (k, 3, 500):                                  # 500 > SDQH_MAX_TOPK: ordered on the host
        got = engine.execute_plan(oracle, plan, args, top=(kk, order))
        want = full.top(kk, order)
        assert got.columns == want.columns and got.ordered_rows() == want.ordered_rows()
        assert got.size() == min(kk, full.size())
    assert sorted(full.top(10 ** 6, order).ordered_rows()) == full.rows()      # ordering loses nothing
    with pytest.raises(KeyError):
        engine.execute_plan(oracle, plan, args, top=(3, [("no_such_column", "asc")]))
    with pytest.raises(ValueError):
        engine.execute_plan(oracle, plan, args, top=(3, [(order[0][0], "up")]))


def test_q3_top10_runs_on_the_device_operator(oracle):
    db = tpch.generate(0.02, tables=["lineitem", "customer", "orders"], columns=tpch.columns_for(("q3",)))
    plan = frontend.lower_function(Q.QUERIES["q3"].__sdql_func__, Q.QUERIES["q3"].__sdql_in_type__)
    import helpers
    with helpers.spy_calls("table_topk", lambda *a, **kw: a[2:4]) as calls:
        engine.execute_plan(oracle, plan, [db[t] for t in Q.QUERY_TABLES["q3"]], top=Q.TPCH_ORDER["q3"])
    assert calls == [(10, [(abi.SORT_VALUE, 0, True, True), (abi.SORT_PAYLOAD, 0, False, False)])]


def test_result_set_has_the_reference_containers_surface(oracle, capsys):
    """size / to_dict / get / set / from_dict / print, as the reference's fastd wrapper offers them
    (reference src/sdqlpy/fastd.py:31-51)."""
    from sdqlpy_amd.sdql_lib import record
    db = tpch.generate(0.005, tables=["lineitem"], columns=tpch.columns_for(("q1",)))
    plan = frontend.lower_function(Q.QUERIES["q1"].__sdql_func__, Q.QUERIES["q1"].__sdql_in_type__)
    res = engine.execute_plan(oracle, plan, [db["lineitem"]])
    assert res.size() == len(res) == 4
    d = res.to_dict().getContainer()
    assert len(d) == 4 and all(v is True for v in d.values())
    some = next(iter(d))
    assert res.get(some) is True
    other = record({**some.getContainer(), "count_order": -1})
    assert res.get(other) is None
    res.set(other, True)
    assert res.size() == 5 and res.get(other) is True
    res.print()
    assert "count_order" in capsys.readouterr().out
    back = res.from_dict(res.to_dict())
    assert back.size() == 5
    with pytest.raises(KeyError):
        res.get(record({"nope": 1}))


def test_q10_group_sharing_and_host_fold_agree(oracle, monkeypatch):
    """Q10's output key names customer fields of the matched order.  Default: the orders of one
    customer share an accumulator on the device (sdqh_table_share_groups), so ORDER BY / LIMIT runs
    there too.  Without it the entries are folded on the host on the row references — bucketed, or
    sort-based for large reference rectangles.  All three agree."""
    qs = ("q10",)
    db = tpch.generate(0.05, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
    plan = frontend.lower_function(Q.q10.__sdql_func__, Q.q10.__sdql_in_type__)
    args = [db[t] for t in Q.QUERY_TABLES["q10"]]
    calls = []
    real_share, real_topk = abi.Context.table_share_groups, abi.Context.table_topk       # (on the class: whichever lane of the engine runs the plan)
    monkeypatch.setattr(abi.Context, "table_share_groups", lambda ctx, *a: (calls.append("share"), real_share(ctx, *a))[1])
    monkeypatch.setattr(abi.Context, "table_topk", lambda ctx, *a, **k: (calls.append("topk"), real_topk(ctx, *a, **k))[1])
    want = engine.execute_plan(oracle, plan, args)
    top_want = engine.execute_plan(oracle, plan, args, top=Q.TPCH_ORDER["q10"])
    assert calls == ["share", "share", "topk"] and want.size() > 100
    monkeypatch.setattr(engine, "_share_spec", lambda eng, bt: None)
    for cells in (1 << 26, 0):
        monkeypatch.setattr(engine, "_DENSE_MERGE_CELLS", cells)
        got = engine.execute_plan(oracle, plan, args)
        assert got.columns == want.columns and got.size() == want.size()
        for a, b in zip(got.rows(), want.rows()):
            assert a[:2] == b[:2] and a[3:] == b[3:] and abs(a[2] - b[2]) <= 1e-12 * abs(b[2])
        top_got = engine.execute_plan(oracle, plan, args, top=Q.TPCH_ORDER["q10"])
        assert [r[0] for r in top_got.ordered_rows()] == [r[0] for r in top_want.ordered_rows()]
    assert calls == ["share", "share", "topk"]


def test_text_columns_of_large_results_are_decoded_on_first_read(oracle, monkeypatch):
    """Large results keep their text columns as row references until they are read (result.TextRefs):
    the same rows either way, through every accessor, and ordering / slicing keeps them undecoded."""
    from sdqlpy_amd import result
    qs = ("q10", "q18")
    db = tpch.generate(0.1, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
    for q in qs:
        plan = frontend.lower_function(Q.QUERIES[q].__sdql_func__, Q.QUERIES[q].__sdql_in_type__)
        args = [db[t] for t in Q.QUERY_TABLES[q]]
        monkeypatch.setattr(result, "LAZY_TEXT_ROWS", 1 << 30)
        eager = engine.execute_plan(oracle, plan, args)
        assert not any(isinstance(a, result.TextRefs) for a in eager._cols)
        monkeypatch.setattr(result, "LAZY_TEXT_ROWS", 1)
        lazy = engine.execute_plan(oracle, plan, args)
        assert any(isinstance(a, result.TextRefs) for a in lazy._cols) and lazy.size() == eager.size() > 0
        k, order = Q.TPCH_ORDER[q]
        top = lazy.top(3, order)                                        # ordering reads the sort columns only
        assert any(isinstance(a, result.TextRefs) for a in top._cols) and top.ordered_rows() == eager.top(3, order).ordered_rows()
        name = next(c for c in lazy.columns if lazy._cols[lazy.columns.index(c)].dtype.kind == "U")
        assert lazy.column(name).tolist() == eager.column(name).tolist()
        assert lazy.rows() == eager.rows() and lazy.to_dict() == eager.to_dict()
        assert not any(isinstance(a, result.TextRefs) for a in lazy._cols)


def test_deferred_result_set_finishes_on_first_use():
    """result.DeferredResultSet: nothing runs until something looks at the result; every way of looking runs the thunk exactly
    once; an exception of the thunk surfaces at that first use (and again if asked again: the thunk is gone, the object stays unfinished)."""
    import numpy as np
    import pytest
    from sdqlpy_amd.result import DeferredResultSet, ResultSet
    calls = []

    def make():
        calls.append(1)
        return ResultSet(["k", "v"], [np.array([3, 1, 2]), np.array([0.5, 1.5, 2.5])])
    for use in (lambda r: r.size(), lambda r: len(r), lambda r: r.columns, lambda r: r.column("v"), lambda r: r.rows(), lambda r: r.arrays,
                lambda r: r.wait(), lambda r: r.top(2, [("k", "desc")]).rows()):
        del calls[:]
        r = DeferredResultSet(make)
        assert isinstance(r, ResultSet) and calls == []
        use(r)
        use(r)
        assert calls == [1]
        assert r.rows() == [(1, 1.5), (2, 2.5), (3, 0.5)] and r.size() == 3
    nested = DeferredResultSet(lambda: DeferredResultSet(make))
    assert nested.size() == 3
    bad = DeferredResultSet(lambda: (_ for _ in ()).throw(ValueError("decided by the data")))
    with pytest.raises(ValueError):
        bad.size()
