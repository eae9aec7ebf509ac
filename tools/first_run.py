#!/usr/bin/env python3
"""Where a query's FIRST run goes (upload, twins, dictionaries, plan lowering): wall time per ABI call of the first run of q1 / q3 / q5
at a given SF, from a cold engine.   python tools/first_run.py [sf] [queries]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdqlpy_amd import engine, tpch
from sdqlpy_amd import tpch_queries as Q
from sdqlpy_amd.sdql_lib import sdqlpy_init

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
qs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["q1", "q3", "q5"]
sdqlpy_init(3, 1, device=0)
eng = engine.default_engine(device=0)
db = tpch.generate(sf, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
eng.ctx.synchronize()
for q in qs:
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    r = Q.run(q, db)
    if hasattr(r, "wait"):
        r.wait()
    eng.ctx.synchronize()
    pr.disable()
    print("==", q, "first run %.1f ms, resident %.2f GB" % ((time.perf_counter() - t0) * 1e3, eng.resident_bytes / 1e9), flush=True)
    pstats.Stats(pr).sort_stats("tottime").print_stats(10)
