#!/usr/bin/env python3
"""Kernel / wall time of the plans the REFERENCE's own text lowers to (tests/reference_shapes.py) beside the shipped formulations, at a
given SF on the GPU.   python tools/reference_shapes_sf10.py [sf]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sdqlpy_amd import engine, frontend, tpch
from sdqlpy_amd import tpch_queries as Q
from sdqlpy_amd.sdql_lib import sdqlpy_init
import reference_shapes as shapes

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
sdqlpy_init(3, 1, device=0)
eng = engine.default_engine(device=0)
qs = sorted(shapes.QUERIES, key=lambda q: int(q[1:]))
db = tpch.generate(sf, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))


def fin(r):
    return r.wait() if hasattr(r, "wait") else r


def timed(run):
    for _ in range(3):
        fin(run())
    eng.plan_graphs = 0
    eng.ctx.set_profiling(2, only=None)
    fin(run())
    eng.ctx.synchronize()
    launches = eng.ctx.profile()
    eng.ctx.set_profiling(0)
    eng.plan_graphs = 2
    for _ in range(3):
        fin(run())
    t0 = time.perf_counter()
    for _ in range(20):
        fin(run())
    return (time.perf_counter() - t0) * 50, sum(ms for _, ms in launches), len(launches)


print("SF=%g; per query: wall ms (each run waited for), sum of kernel ms, launches" % sf)
for q in qs:
    plan = frontend.lower_function(shapes.QUERIES[q])
    args = [db[t] for t in shapes.TABLES[q]]
    a = timed(lambda: engine.execute_plan(eng, plan, args))
    b = timed(lambda: Q.run(q, db))
    ra, rb = fin(engine.execute_plan(eng, plan, args)), fin(Q.run(q, db))
    same = sorted(ra.rows()) == sorted(rb.rows()) if ra.size() < 200000 else ra.size() == rb.size()
    print("%-4s reference-shaped plan %.3f ms wall, %.3f ms of kernels in %d launches | shipped formulation %.3f ms wall, %.3f ms of kernels in %d launches | rows %d / %d %s"
          % (q, a[0], a[1], a[2], b[0], b[1], b[2], ra.size(), rb.size(), "equal" if same else "(sums differ in the last bits: compared in the test suite)"), flush=True)
print("engine stats:", eng.stats())
