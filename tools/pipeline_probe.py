#!/usr/bin/env python3
"""Where does the host block when the queries of a step are launched back to back and their results finished afterwards?
Host time of every call of the step (launch q1, q3, q5; finish r1, r3, r5), averaged: python tools/pipeline_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdqlpy_amd import engine, tpch, tpch_queries as Q
from sdqlpy_amd.sdql_lib import sdqlpy_init

qs = (sys.argv[1] if len(sys.argv) > 1 else "q1,q3,q5").split(",")
if os.environ.get("PROBE_TORCH"):                        # bench.py has torch loaded and its device initialised
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
sdqlpy_init(3, 1, device=0)
eng = engine.default_engine(device=0)
db = tpch.generate(10, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
for _ in range(5):
    for q in qs:
        r = Q.run(q, db)
        r.wait()
n = 50
acc = [0.0] * (2 * len(qs))
if len(sys.argv) > 2:                                    # the bench's timed region records events around one kernel (profiling mode 2)
    eng.ctx.set_profiling(2, only=sys.argv[2])
eng.ctx.synchronize()
t_all = time.perf_counter()
for _ in range(n):
    rs = []
    for i, q in enumerate(qs):
        t0 = time.perf_counter()
        rs.append(Q.run(q, db))
        acc[i] += time.perf_counter() - t0
    for i, r in enumerate(rs):
        t0 = time.perf_counter()
        r.wait()
        acc[len(qs) + i] += time.perf_counter() - t0
eng.ctx.synchronize()
total = (time.perf_counter() - t_all) / n * 1e3
print("step %.4f ms" % total)
for i, q in enumerate(qs):
    print("  launch %-4s %8.1f us" % (q, acc[i] / n * 1e6))
for i, q in enumerate(qs):
    print("  finish %-4s %8.1f us" % (q, acc[len(qs) + i] / n * 1e6))
print("types", [type(r).__name__ for r in rs])
