"""CPU-side checks of the drop-in boundary: both libraries export every symbol include/sdqh.h
declares; the HIP library refuses to create a context without a GPU (no silent fall-back)."""
import os
import re

import pytest

from sdqlpy_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sdqh.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdqh_[a-z_0-9]+)\s*\(", text)))


def test_binding_lists_every_declared_symbol():
    assert sorted(abi.EXPORTS) == declared_symbols()


def test_integration_notes_count_the_declared_symbols():
    """INTEGRATION.md states how many entry points the boundary has, and names each one the header declares."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"\((\d+) `extern \"C\"` entry points", text)
    assert m and int(m.group(1)) == len(declared_symbols())


def test_oracle_exports_every_symbol(oracle_lib):
    for s in declared_symbols():
        assert hasattr(oracle_lib.cdll, s), s
    assert oracle_lib.backend_name() == "cpu-oracle"


def test_hip_library_builds_loads_and_exports_every_symbol(hip_lib):
    for s in declared_symbols():
        assert hasattr(hip_lib.cdll, s), s
    assert hip_lib.backend_name() == "hip-gfx950"


def test_hip_library_fails_loudly_without_a_gpu(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(abi.SdqhError):
        hip_lib.context(device=0)


def test_python_mode_is_not_a_fallback():
    from sdqlpy_amd.sdql_lib import sdqlpy_init, sr_dict
    from sdqlpy_amd import tpch_queries as Q
    sdqlpy_init(0, 1)
    with pytest.raises(NotImplementedError):
        Q.q6(sr_dict({"headers": [], "data": []}, None, True))
    with pytest.raises(NotImplementedError):
        sr_dict({}).sum(lambda p: p)


def test_oracle_compaction_into_blocks(oracle_lib):
    """sdqh_host_alloc / one-call K-F with overflow retry on the CPU implementation of the ABI."""
    from helpers import compaction_block_case
    ctx = oracle_lib.context(threads=2)
    try:
        compaction_block_case(ctx)
    finally:
        ctx.close()


def test_oracle_string_predicates_follow_varchar_semantics(oracle_lib):
    """==, != and substring on fixed-width UCS4 fields against a plain-Python restatement of the
    reference's VarChar (include/varchar.h:61-89), embedded NULs included."""
    from helpers import string_predicate_case
    ctx = oracle_lib.context(threads=2)
    try:
        assert string_predicate_case(ctx, widths=(1, 3, 10, 33, 55), rows=1200) > 100
    finally:
        ctx.close()


def test_oracle_column_comparisons(oracle_lib):
    from helpers import column_compare_case
    ctx = oracle_lib.context(threads=3)
    try:
        assert column_compare_case(ctx) == 16
    finally:
        ctx.close()


def test_oracle_key_sets(oracle_lib):
    from helpers import key_set_case
    ctx = oracle_lib.context(threads=3)
    try:
        assert key_set_case(ctx, n=20000) == 5
    finally:
        ctx.close()


def test_oracle_share_groups(oracle_lib):
    from helpers import share_groups_case
    ctx = oracle_lib.context(threads=3)
    try:
        assert share_groups_case(ctx, n_build=6000, n_probe=40000) == 6
    finally:
        ctx.close()


def test_oracle_groupby_key_and_having(oracle_lib):
    from helpers import groupby_key_case
    ctx = oracle_lib.context(threads=3)
    try:
        assert groupby_key_case(ctx, n=30000) == 3
    finally:
        ctx.close()


def test_oracle_empty_inputs(oracle_lib):
    from helpers import empty_input_case
    ctx = oracle_lib.context(threads=2)
    try:
        assert empty_input_case(ctx)
    finally:
        ctx.close()


def test_host_arrays_are_frozen_until_invalidated(oracle_lib):
    """A column handed to a query lives on as a device copy: the host array becomes read-only, so an
    in-place edit raises instead of silently leaving stale answers; after invalidate() the edit is
    allowed and the next run sees it (the reference re-reads the caller's buffers on every call)."""
    import numpy as np
    import pytest
    from sdqlpy_amd import engine, frontend, tpch
    from sdqlpy_amd import tpch_queries as Q
    eng = engine.Engine(oracle_lib.context(threads=1))
    try:
        db = tpch.generate(0.001, tables=["lineitem"], columns=tpch.columns_for(["q6", "q1"]), threads=1)
        plan6, plan1 = frontend.lower_function(Q.q6), frontend.lower_function(Q.q1)
        before6 = engine.execute_plan(eng, plan6, [db["lineitem"]])
        before1 = engine.execute_plan(eng, plan1, [db["lineitem"]])
        price = tpch.column(db["lineitem"], "l_extendedprice")
        flag = tpch.column(db["lineitem"], "l_returnflag")
        with pytest.raises(ValueError):
            price[:] = 0.0
        eng.invalidate(db["lineitem"])
        price *= 2.0                                   # exact in binary: every product and sum doubles
        flag[:] = "Z"                                  # the dictionary / group keys must be rebuilt too
        after6 = engine.execute_plan(eng, plan6, [db["lineitem"]])
        after1 = engine.execute_plan(eng, plan1, [db["lineitem"]])
        assert after6 == 2.0 * before6 and before6 != 0.0
        assert set(after1.column("l_returnflag").tolist()) == {"Z"} and set(before1.column("l_returnflag").tolist()) != {"Z"}
        assert abs(after1.column("sum_base_price").sum() - 2.0 * before1.column("sum_base_price").sum()) <= 1e-9 * abs(after1.column("sum_base_price").sum())
    finally:
        eng.close()


def test_hash_layout_case_on_the_cpu_implementation(oracle_lib):
    """The scale test of the open-addressing layout (tests/test_hip_parity.py runs it at 12 M keys on
    the GPU), small, against the CPU implementation: pins the test's own numpy model."""
    import helpers
    ctx = oracle_lib.context(threads=2)
    try:
        helpers.hash_layout_case(ctx, 50_000, 200_000)
    finally:
        ctx.close()


def test_row_programs_on_the_cpu_implementation(oracle_lib):
    """ABI 4 (row programs) through every sdqh_x* entry point of the CPU implementation against numpy:
    pins the interpreter the GPU specialisations are compared with (tests/test_hip_parity.py runs the same cases)."""
    import helpers
    for threads in (1, 3):
        ctx = oracle_lib.context(threads=threads)
        try:
            assert helpers.xprogram_cases(ctx) > 40
        finally:
            ctx.close()


def test_specialised_kernels_compile_without_a_gpu(hip_lib):
    """Build check of the run-time specialisation path on a host without a GPU: a compile-only
    context (sdqh_create(-1)) generates and compiles the kernel of each sink for gfx950 and stops."""
    from sdqlpy_amd import abi as A
    ctx = hip_lib.context(device=-1)
    try:
        n = 1000
        ci, cf, cs = ctx.wrap(0x10000, n, A.I64), ctx.wrap(0x20000, n, A.F64), ctx.wrap(0x30000, n, A.STR, 12)
        P = A.Program()
        x = P.op(A.X_COL, A.T_I64, col=ci); v = P.op(A.X_COL, A.T_F64, col=cf)
        P.gates = [P.op(A.X_OR, A.T_BOOL, a=P.op(A.X_LT, A.T_BOOL, a=x, b=P.op(A.X_CONST, A.T_I64, imm_i=5)), b=P.op(A.X_STR, A.T_BOOL, col=cs, aux=A.STR_PREFIX, text="ab"))]
        P.vals = [v]

        def compiled(call):
            with pytest.raises(A.SdqhError) as exc:
                call()
            assert exc.value.code == A.ERR_DEVICE and "kernel specialised" in str(exc.value), str(exc.value)[:2000]
        compiled(lambda: ctx.xscan_sum(n, P))
        P.key = x
        compiled(lambda: ctx.xgroupby(n, P))
        table = ctx.xbuild(n, P, 0, 100, accumulate=True)                     # placeholder table: later programs can name it
        Q = A.Program()
        lk = Q.op(A.X_LOOKUP, A.T_BOOL, a=Q.op(A.X_COL, A.T_I64, col=ci), table=table)
        Q.gates = [lk]; Q.vals = [Q.op(A.X_ADD, A.T_F64, a=Q.op(A.X_FIELD, A.T_F64, a=lk, aux=0), b=Q.op(A.X_ACC, A.T_F64, a=lk, aux=0))]
        compiled(lambda: ctx.xprobe_aggregate(n, Q, lk, table))
        Q.key, Q.vals = Q.op(A.X_COL, A.T_I64, col=ci), []
        assert ctx.xkey_set(n, Q, 0, 100).handle is not None
        assert sum(ctx.jit_stats()) >= 5
        # a malformed program is refused before anything is generated
        bad = A.Program()
        bad.vals = [bad.op(A.X_ADD, A.T_F64, a=bad.op(A.X_COL, A.T_I64, col=ci), b=bad.op(A.X_COL, A.T_F64, col=cf))]
        with pytest.raises(A.SdqhError) as exc:
            ctx.xscan_sum(n, bad)
        assert exc.value.code == A.ERR_INVALID
    finally:
        ctx.close()


def test_code_space_comparison_cases_on_the_cpu_implementation(oracle_lib):
    """The comparison / dictionary-value / dense-key cases the GPU's coded streaming kernels are checked with
    (helpers.xcode_edge_cases, tests/test_hip_parity.py) against numpy on the CPU implementation: pins the cases themselves."""
    import helpers
    ctx = oracle_lib.context(threads=2)
    try:
        assert helpers.xcode_edge_cases(ctx, n=3000) > 500
    finally:
        ctx.close()


def test_tight_kernels_compile_without_a_gpu(hip_lib, monkeypatch):
    """Build check of the round-3 skeletons on a host without a GPU: with SDQLPY_AMD_FAKE_CODES the compile-only context pretends
    every column has dictionary codes / twins, so the generator emits — and hiprtc compiles for gfx950 — x_tight with the per-lane
    group sink, code-space comparisons and LDS dictionary tables, and x_queue8 with the batched 32-bit bitmap prefilter."""
    from sdqlpy_amd import abi as A
    monkeypatch.setenv("SDQLPY_AMD_FAKE_CODES", "1")
    ctx = hip_lib.context(device=-1)
    try:
        n = 1 << 22
        ship, qty, ep, disc, flag, key = (ctx.wrap(0x10000 * (i + 1), n, dt) for i, dt in enumerate((A.I64, A.F64, A.F64, A.F64, A.I64, A.I64)))

        def compiled(call):
            with pytest.raises(A.SdqhError) as exc:
                call()
            assert exc.value.code == A.ERR_DEVICE and "kernel specialised" in str(exc.value), str(exc.value)[:3000]
        before = sum(ctx.jit_stats())
        P = A.Program()
        s = P.op(A.X_COL, A.T_I64, col=ship); q = P.op(A.X_COL, A.T_F64, col=qty); e = P.op(A.X_COL, A.T_F64, col=ep); d = P.op(A.X_COL, A.T_F64, col=disc); f = P.op(A.X_COL, A.T_I64, col=flag)
        one = P.op(A.X_CONST, A.T_F64, imm_f=1.0)
        P.gates = [P.op(A.X_LE, A.T_BOOL, a=s, b=P.op(A.X_CONST, A.T_I64, imm_i=19980902)), P.op(A.X_LT, A.T_BOOL, a=P.op(A.X_CONST, A.T_F64, imm_f=0.02), b=d)]
        P.vals = [q, P.op(A.X_MUL, A.T_F64, a=e, b=P.op(A.X_SUB, A.T_F64, a=one, b=d))]
        compiled(lambda: ctx.xscan_sum(n, P))                                       # x_tight + XSum
        P.key = P.op(A.X_ADD, A.T_I64, a=P.op(A.X_MUL, A.T_I64, a=f, b=P.op(A.X_CONST, A.T_I64, imm_i=2)), b=f)
        compiled(lambda: ctx.xgroupby(n, P))                                        # x_tight + XGroupLane
        K = A.Program()
        K.gates = [K.op(A.X_LT, A.T_BOOL, a=K.op(A.X_COL, A.T_I64, col=ship), b=K.op(A.X_CONST, A.T_I64, imm_i=19950315))]
        K.key = K.op(A.X_COL, A.T_I64, col=key)
        members = ctx.xkey_set(n, K, 0, 1000)                                       # x_queue8 + XKeySet (placeholder table)
        B = A.Program()
        B.gates = [B.op(A.X_LT, A.T_BOOL, a=B.op(A.X_COL, A.T_I64, col=ship), b=B.op(A.X_CONST, A.T_I64, imm_i=19950315)),
                   B.op(A.X_LOOKUP, A.T_BOOL, a=B.op(A.X_COL, A.T_I64, col=flag), table=members)]
        B.key = B.op(A.X_COL, A.T_I64, col=key); B.vals = [B.op(A.X_COL, A.T_I64, col=ship)]
        built = ctx.xbuild(n, B, 0, 100000, accumulate=True)                        # x_queue8 + XStage, prefilter on the key set's bitmap
        R = A.Program()
        lk = R.op(A.X_LOOKUP, A.T_BOOL, a=R.op(A.X_COL, A.T_I64, col=key), table=built)
        R.gates = [R.op(A.X_GT, A.T_BOOL, a=R.op(A.X_COL, A.T_I64, col=ship), b=R.op(A.X_CONST, A.T_I64, imm_i=19950315)), lk]
        R.vals = [R.op(A.X_MUL, A.T_F64, a=R.op(A.X_COL, A.T_F64, col=ep), b=R.op(A.X_COL, A.T_F64, col=disc))]
        compiled(lambda: ctx.xprobe_aggregate(n, R, lk, built))                     # x_queue8 + XEntry
        monkeypatch.setenv("SDQLPY_AMD_FAKE_WINDOW", "1")                           # ... and with the lane's 8 rows tested against one 16-byte window of the bitmap
        compiled(lambda: ctx.xprobe_aggregate(n, R, lk, built))
        assert sum(ctx.jit_stats()) - before >= 6
    finally:
        ctx.close()


def test_packed_exchange_helpers_on_the_cpu_implementation(oracle_lib):
    """sdqh_partition_pack / sdqh_unpack_parts / sdqh_column_unpack2 / bitmap export from key sets: the CPU implementation against numpy
    (the HIP library runs the same case in tests/test_hip_parity.py and must agree with this one)."""
    import helpers
    ctx = oracle_lib.context(threads=2)
    try:
        out = helpers.redistribution_pack_case(ctx, n=20011)
        assert sum(out["hash"][0]) == 20011 and sum(out["range"][0]) == 20011
        assert helpers.xcompact_case(ctx, n=20011) == 8           # sdqh_xcompact: the probe side's rows as a row program, equal keys kept
        assert helpers.xcompact_case(ctx, n=1, seed=2) == 8
    finally:
        ctx.close()


def test_context_families_share_columns_on_the_cpu_implementation(oracle_lib):
    """sdqh_fork: a forked context names its parent's columns; engine lanes on top of it give the rows one lane gives."""
    import numpy as np
    import helpers
    ctx = oracle_lib.context(threads=2)
    try:
        child = ctx.fork()
        a = np.arange(1000, dtype=np.int64)
        v = np.arange(1000, dtype=np.float64) / 4
        ca, cv = ctx.upload(a), ctx.upload(v)
        flt = abi.make_filter(ipreds=[(ca, 10, 499)])
        want = ctx.scan_filter_sum(1000, flt, abi.make_tuple(abi.TUPLE_A, [cv]))
        got = child.scan_filter_sum(1000, flt, abi.make_tuple(abi.TUPLE_A, [cv]))
        assert got == want and got[1] == 490
        with pytest.raises(abi.SdqhError):
            child.fork()                                            # the family's first context forks
        child.close()
        assert ctx.forks == []
    finally:
        ctx.close()
    assert helpers.lanes_case(oracle_lib, sf=0.02, rounds=4) == {0, 1, 2}


def test_composite_key_builds_over_sorted_first_parts_on_the_cpu_implementation(oracle_lib):
    """The case the GPU build's grouped layout is for (helpers.grouped_index_case), on the CPU implementation: checked against numpy inside."""
    import helpers
    ctx = oracle_lib.context(threads=2)
    try:
        assert helpers.grouped_index_case(ctx, n=20000, nprobe=40000)[2] > 1000
        assert helpers.grouped_index_case(ctx, n=3000, nprobe=5000, keep=0.02, per_a=1)[2] > 10
    finally:
        ctx.close()


def test_plan_recording_is_refused_by_the_cpu_implementation_and_the_calls_go_on(oracle_lib):
    """Plan graphs (include/sdqh.h, ABI 5) are the HIP library's: the CPU implementation records nothing and says so at
    sdqh_graph_end (SDQH_ERR_UNSUPPORTED).  An engine told to record all the same (Engine.plan_graphs) tries once per prepared plan,
    is refused, and keeps issuing the calls: same rows, the refusal counted."""
    import helpers
    from sdqlpy_amd import engine, tpch
    ctx = oracle_lib.context()
    ctx.graph_begin()
    with pytest.raises(abi.SdqhError) as exc:
        ctx.graph_end()
    assert exc.value.code == abi.ERR_UNSUPPORTED
    ctx.graph_abort()
    ctx.close()
    qs = ("q1", "q3")
    db = tpch.generate(0.01, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
    plain, recording = engine.Engine(oracle_lib.context()), engine.Engine(oracle_lib.context())
    recording.deferred_results, recording.plan_graphs, recording.plan_graphs_always = True, 2, True
    try:
        from sdqlpy_amd import frontend
        from sdqlpy_amd import tpch_queries as Q
        for q in qs:
            want = sorted(helpers.run_query(plain, q, db).rows())
            plan, args = frontend.lower_function(Q.QUERIES[q]), [db[t] for t in Q.QUERY_TABLES[q]]      # ONE plan object: prepared once, run again and again
            for _ in range(5):
                assert sorted(engine.execute_plan(recording, plan, args).rows()) == want, q
        # (q3 ends in a deferred K-F and is tried; q1's small group-by is a waited-for call on this implementation and never is)
        assert recording.graph_stats["recorded"] == 0 and recording.graph_stats["launched"] == 0 and recording.graph_stats["refused"] >= 1, recording.graph_stats
    finally:
        plain.close()
        recording.close()
