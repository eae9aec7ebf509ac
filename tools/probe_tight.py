#!/usr/bin/env python3
"""A/B of the streaming loops (q1, q6): row programs on tight encodings (x_tight) against the fixed-shape kernels on 4-byte twins.
Same engine, same resident columns; results compared (1e-12) with each other and at SF <= 1 with the CPU implementation.
    python tools/probe_tight.py --sf 10"""
import argparse
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rows_of(res):
    return res if isinstance(res, float) else sorted(res.rows())


def close(a, b, rel):
    if isinstance(a, float):
        return abs(a - b) <= rel * abs(b)
    if len(a) != len(b):
        return False
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            if isinstance(v, float):
                if abs(u - v) > rel * max(abs(u), abs(v)):
                    return False
            elif u != v:
                return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--queries", default="q1,q6")
    ap.add_argument("--iters", type=int, default=7)
    args = ap.parse_args()
    qs = args.queries.split(",")
    from sdqlpy_amd import abi, engine, frontend, tpch
    from sdqlpy_amd import tpch_queries as Q
    from sdqlpy_amd.sdql_lib import sdqlpy_init
    sdqlpy_init(3, 1, device=0)
    eng = engine.default_engine(device=0)
    db = tpch.generate(args.sf, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
    print("rows", {t: len(db[t].getContainer()["data"][0]) for t in db})
    results = {}
    for mode in ("fixed", "tight"):
        eng.stream_programs = mode == "tight"
        for q in qs:
            plan = frontend.lower_function(Q.QUERIES[q])          # a fresh plan per mode: prepared loops are cached on the plan
            run = lambda: engine.execute_plan(eng, plan, [db[t] for t in Q.QUERY_TABLES[q]])   # noqa: E731
            t0 = time.perf_counter(); res = run(); first = (time.perf_counter() - t0) * 1e3
            run()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                run()
            wall = (time.perf_counter() - t0) / args.iters * 1e3
            eng.ctx.set_profiling(True)
            logs = []
            for _ in range(args.iters):
                eng.ctx.kernel_log, eng.ctx.device_log = [], []
                run()
                logs.append(list(eng.ctx.kernel_log))
            eng.ctx.set_profiling(False)
            results[(mode, q)] = rows_of(res)
            print("== %s %-5s first run %.1f ms, wall %.4f ms" % (q, mode, first, wall))
            for i in range(len(logs[0])):
                print("   %-22s %8.4f ms" % (logs[0][i][0], statistics.median(l[i][1] for l in logs)))
    for q in qs:
        print(q, "tight == fixed (1e-12):", close(results[("tight", q)], results[("fixed", q)], 1e-12))
        print("  ", results[("tight", q)] if isinstance(results[("tight", q)], float) else results[("tight", q)][:2])
    if args.sf <= 1.0:
        oracle = engine.Engine(abi.Library(os.path.join(ROOT, "oracle", "libsdqloracle.so")).context(threads=8))
        for q in qs:
            want = rows_of(engine.execute_plan(oracle, frontend.lower_function(Q.QUERIES[q]), [db[t] for t in Q.QUERY_TABLES[q]]))
            print(q, "tight == cpu (1e-10):", close(results[("tight", q)], want, 1e-10))
        oracle.close()
    print("jit:", eng.ctx.jit_stats() if hasattr(eng.ctx, "jit_stats") else "")


if __name__ == "__main__":
    main()
