#!/usr/bin/env python3
"""Which shipped queries run a sum over a result dictionary on the HOST (Engine.stats()["host_loops"]), at a given SF, and what a
strict engine (SDQLPY_AMD_STRICT_DEVICE=1) says instead.   python tools/host_loops.py [sf]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdqlpy_amd import engine, frontend, tpch
from sdqlpy_amd import tpch_queries as Q
from sdqlpy_amd.sdql_lib import sdqlpy_init

sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sdqlpy_init(3, 1, device=0)
eng = engine.default_engine(device=0)
qs = sorted(Q.QUERIES, key=lambda q: int(q[1:]) if q[1:].isdigit() else 99)
db = tpch.generate(sf, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
for q in qs:
    eng.host_loops.clear()
    for _ in range(2):
        r = Q.run(q, db)
        if hasattr(r, "wait"):
            r.wait()
    loops = eng.stats()["host_loops"]
    eng.strict_device = True
    try:
        r = Q.run(q, db)
        if hasattr(r, "wait"):
            r.wait()
        strict = "runs"
    except frontend.UnsupportedQuery as exc:
        strict = "refused: " + str(exc).splitlines()[0][:150]
    finally:
        eng.strict_device = False
    print(q, "host loops:", [(l["line"], l["result"], l["runs"], l["why"][:90]) for l in loops] or "none", "| strict:", strict, flush=True)
