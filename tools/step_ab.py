#!/usr/bin/env python3
"""The bench's step (q1, q3, q5 launched, then their results finished) timed under tuning options of the library:
python tools/step_ab.py copy_kernel=0 side_priority=1      (every name=value goes to sdqh_set_option before the first query)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdqlpy_amd import engine, tpch, tpch_queries as Q
from sdqlpy_amd.sdql_lib import sdqlpy_init

qs = ["q1", "q3", "q5"]
sdqlpy_init(3, 1, device=0)
eng = engine.default_engine(device=0)
for kv in sys.argv[1:]:
    name, value = kv.split("=")
    eng.ctx.set_option(name, int(value))
db = tpch.generate(10, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
for _ in range(6):
    for r in [Q.run(q, db) for q in qs]:
        r.wait()
eng.ctx.synchronize()
best = None
for rep in range(3):
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        for r in [Q.run(q, db) for q in qs]:
            r.wait()
    eng.ctx.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    best = ms if best is None else min(best, ms)
print("options %s: step %.4f ms (best of 3 x 100)" % (" ".join(sys.argv[1:]) or "(defaults)", best))
