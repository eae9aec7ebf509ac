#!/usr/bin/env python3
"""q3 at a given SF: full result vs ORDER BY revenue desc, o_orderdate asc LIMIT k (wall ms + top-k kernel times)."""
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdqlpy_amd import engine, tpch
from sdqlpy_amd import tpch_queries as Q
from sdqlpy_amd.sdql_lib import sdqlpy_init

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
sdqlpy_init(3, 1, device=0)
db = tpch.generate(sf, tables=["lineitem", "customer", "orders"], columns=tpch.columns_for(("q3",)))
order = Q.TPCH_ORDER["q3"][1]
for top in (None, (10, order), (100, order)):
    for _ in range(5):
        r = Q.run("q3", db, top)
    t0 = time.perf_counter()
    for _ in range(50):
        r = Q.run("q3", db, top)
    print("q3 top=%s: %.3f ms, %d rows" % (top and top[0], (time.perf_counter() - t0) * 20, r.size()), r.ordered_rows()[:2] if top else "", flush=True)
eng = engine.default_engine()
eng.ctx.set_profiling(True)
for top in ((10, order), (100, order)):
    eng.ctx.kernel_log = []
    Q.run("q3", db, top)
    print([(k, round(ms, 4)) for k, ms in eng.ctx.kernel_log if "topk" in k], flush=True)
