#!/usr/bin/env python3
"""A/B: the tuned fixed-shape kernels against the run-time specialised ones (Engine.force_programs) on the same queries.
    python tools/probe_programs.py --sf 10 --queries q1,q3,q5,q6,q9 --iters 5"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--queries", default="q1,q3,q5,q6,q9")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    qs = args.queries.split(",")
    from sdqlpy_amd import engine, frontend, tpch
    from sdqlpy_amd import tpch_queries as Q
    from sdqlpy_amd.sdql_lib import sdqlpy_init
    sdqlpy_init(3, 1, device=0)
    eng = engine.default_engine(device=0)
    db = tpch.generate(args.sf, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
    for q in qs:
        for forced in (False, True):
            eng.force_programs = forced
            plan = frontend.lower_function(Q.QUERIES[q])          # a fresh plan: prepared steps are bound to the route
            a = [db[t] for t in Q.QUERY_TABLES[q]]
            try:
                for _ in range(2):
                    engine.execute_plan(eng, plan, a)
                t0 = time.perf_counter()
                for _ in range(args.iters):
                    engine.execute_plan(eng, plan, a)
                wall = (time.perf_counter() - t0) / args.iters * 1e3
                eng.ctx.set_profiling(True)
                logs = []
                for _ in range(args.iters):
                    eng.ctx.kernel_log = []
                    engine.execute_plan(eng, plan, a)
                    logs.append(list(eng.ctx.kernel_log))
                eng.ctx.set_profiling(False)
                print("== %s %s  wall %.3f ms" % (q, "programs" if forced else "tuned   ", wall))
                for i in range(len(logs[0])):
                    ms = statistics.median(l[i][1] for l in logs if i < len(l))
                    if ms >= 0.02:
                        print("     %-18s %8.4f ms" % (logs[0][i][0], ms))
                print("     %-18s %8.4f ms" % ("sum of kernels", sum(statistics.median(l[i][1] for l in logs if i < len(l)) for i in range(len(logs[0])))))
            except Exception as exc:
                print("== %s %s  FAILED %s: %s" % (q, "programs" if forced else "tuned", type(exc).__name__, str(exc)[:300]))
    eng.force_programs = False


if __name__ == "__main__":
    main()
