#!/usr/bin/env python3
"""Every query at a sweep of scale factors, the GPU engine against the CPU implementation of the ABI on the same generated tables —
the planner's routes switch on sizes (row counts, group counts, key ranges, result sizes), and the golden vectors sit at three sizes
only.  python tools/sweep_queries.py 0.05,0.3,2 [q1,q3,...] [option=value,...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from sdqlpy_amd import abi, build, engine, frontend, tpch
from sdqlpy_amd import tpch_queries as Q

sfs = [float(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0.05,0.3,2").split(",")]
qs = (sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] else sorted(Q.QUERIES, key=lambda s: int(s[1:])))
hip = engine.Engine(abi.Library(build.HIP_LIB).context(device=0))
cpu = engine.Engine(abi.Library(os.path.join(ROOT, "oracle", "libsdqloracle.so")).context(threads=32))
for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] else []):      # options of the HIP context: "direct_index=0,row_index=0,grouped_index=0,feature_min_rows=0"
    k, v = kv.split("=")
    hip.ctx.set_option(k, int(v))
    print("option", k, "=", v, flush=True)
bad = 0
for sf in sfs:
    db = tpch.generate(sf, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
    t0 = time.time()
    for q in qs:
        plan = frontend.lower_function(Q.QUERIES[q])
        args = [db[t] for t in Q.QUERY_TABLES[q]]
        try:
            want = engine.execute_plan(cpu, plan, args)
            for run in range(2):                             # the second run takes the cached routes (hints, deferred results)
                got = engine.execute_plan(hip, plan, args)
                if isinstance(want, float):
                    assert abs(got - want) <= 1e-9 * max(abs(want), 1e-300), (got, want)
                else:
                    helpers.assert_rows_match(helpers.result_rows(got, want.columns), helpers.result_rows(want, want.columns), 1e-9, "sf %g %s run %d" % (sf, q, run))
            if q in Q.TPCH_ORDER and not isinstance(want, float) and want.size():        # ORDER BY ... LIMIT k: on the device where it can be
                k, order = Q.TPCH_ORDER[q]
                got = engine.execute_plan(hip, plan, args, (k, order)).ordered_rows()
                ref = want.top(k, order).ordered_rows()
                assert len(got) == len(ref), (len(got), len(ref))
                key = lambda r: tuple(x for x in r if not isinstance(x, float))
                helpers.assert_rows_match(sorted(got, key=key), sorted(ref, key=key), 1e-9, "sf %g %s top" % (sf, q))
        except Exception as exc:                             # noqa: BLE001
            bad += 1
            print("MISMATCH sf %g %s: %s: %s" % (sf, q, type(exc).__name__, str(exc)[:300]), flush=True)
    print("sf %g: %d queries in %.1f s" % (sf, len(qs), time.time() - t0), flush=True)
    hip.clear(); cpu.clear()
print("sweep done: %d mismatches" % bad)
sys.exit(1 if bad else 0)
