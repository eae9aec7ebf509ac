#!/usr/bin/env python3
"""Plan graphs against calls issued one by one (GPU): the same queries, the same rows, wall time per query and per step.
   python tools/graph_try.py [sf] [queries]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdqlpy_amd import engine, tpch
from sdqlpy_amd import tpch_queries as Q
from sdqlpy_amd.sdql_lib import sdqlpy_init

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
qs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["q1", "q3", "q5"]
sdqlpy_init(3, 1, device=0)
eng = engine.default_engine(device=0)
db = tpch.generate(sf, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))


def fin(r):
    return r.wait() if hasattr(r, "wait") else r


def rows_of(r):
    return sorted(r.rows()) if hasattr(r, "rows") else r


def step(n, waited):
    eng.ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        rs = []
        for q in qs:
            r = Q.run(q, db)
            if waited:
                fin(r)
            else:
                rs.append(r)
        for r in rs:
            fin(r)
    eng.ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for mode in (0, 2, 0, 2):
    eng.plan_graphs = mode
    for _ in range(4):
        step(1, False)
    a = step(200, False)
    b = step(200, True)
    per = {}
    for q in qs:
        eng.ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fin(Q.run(q, db))
        per[q] = round((time.perf_counter() - t0) * 10, 4)
    print("plan_graphs=%d  step %.4f ms  each waited %.4f ms  per query %s  stats %s" % (mode, a, b, per, eng.graph_stats), flush=True)
# same rows both ways
eng.plan_graphs = 0
want = {q: rows_of(fin(Q.run(q, db))) for q in qs}
eng.plan_graphs = 2
for rnd in range(5):
    held = [(q, Q.run(q, db)) for q in qs for _ in range(3)]          # three results of each query in flight / alive at once: two recordings + the calls
    for q, r in held:
        got = rows_of(fin(r))
        assert len(got) == len(want[q]), (q, len(got), len(want[q]))
        for a, b in zip(got, want[q]):
            for x, y in zip(a, b):
                assert (abs(x - y) <= 1e-12 * max(abs(x), abs(y))) if isinstance(y, float) else x == y, (q, a, b)
print("same rows with graphs, results held:", eng.graph_stats, flush=True)
