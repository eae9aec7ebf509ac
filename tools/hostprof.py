#!/usr/bin/env python3
"""Where does the host time of a query call go?  cProfile over many calls at SF=10."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdqlpy_amd import tpch
from sdqlpy_amd import tpch_queries as Q
from sdqlpy_amd.sdql_lib import sdqlpy_init

qs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["q3"]
TOP = len(sys.argv) > 2 and sys.argv[2] == "top"          # python tools/hostprof.py q10 top: with the query's ORDER BY / LIMIT
sdqlpy_init(3, 1, device=0)
db = tpch.generate(10, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
for q in qs:
    for _ in range(5):
        (lambda r: r.wait() if hasattr(r, 'wait') else r)(Q.run(q, db, top=Q.TPCH_ORDER[q] if TOP else None))
    t0 = time.perf_counter()
    for _ in range(100):
        (lambda r: r.wait() if hasattr(r, 'wait') else r)(Q.run(q, db, top=Q.TPCH_ORDER[q] if TOP else None))
    print(q, "mean wall ms", (time.perf_counter() - t0) * 10)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        (lambda r: r.wait() if hasattr(r, 'wait') else r)(Q.run(q, db, top=Q.TPCH_ORDER[q] if TOP else None))
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
