#!/usr/bin/env python3
"""Per-launch kernel timings (HIP events on the ctx stream) for the hot-path queries at a given SF.
Tuning aid: python tools/probe.py --sf 10 --queries q1,q3 --iters 7"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--queries", default="q1,q3,q6")
    ap.add_argument("--iters", type=int, default=7)
    ap.add_argument("--configs", default="", help="';'-separated tuning configs, each 'name=value,name=value' (sdqh_set_option)")
    args = ap.parse_args()
    qs = args.queries.split(",")
    from sdqlpy_amd import engine, tpch
    from sdqlpy_amd import tpch_queries as Q
    from sdqlpy_amd.sdql_lib import sdqlpy_init
    sdqlpy_init(3, 1, device=0)
    eng = engine.default_engine(device=0)
    eng.plan_graphs = 0                                  # per-launch timings need the calls issued one by one (a recorded plan has no place for events)
    db = tpch.generate(args.sf, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
    rows = {t: len(db[t].getContainer()["data"][0]) for t in db}
    print("rows", rows)
    for config in (args.configs.split(";") if args.configs else [""]):
        for kv in filter(None, config.split(",")):
            name, value = kv.split("=")
            eng.ctx.set_option(name, int(value))
        print("#### config: %s" % (config or "defaults"))
        run_all(args, qs, Q, eng, db)


def run_all(args, qs, Q, eng, db):
    import statistics

    def run(q):                                          # every result finished before the next run (a query may return with its last call queued)
        r = Q.run(q, db)
        return r.wait() if hasattr(r, "wait") else r
    for q in qs:
        run(q)
        run(q)
        t0 = time.perf_counter()
        for _ in range(args.iters):
            run(q)
        wall = (time.perf_counter() - t0) / args.iters * 1e3
        eng.ctx.set_profiling(True)
        logs = []
        for _ in range(args.iters):
            eng.ctx.kernel_log, eng.ctx.device_log = [], []
            Q.run(q, db)
            logs.append((list(eng.ctx.kernel_log), list(eng.ctx.device_log)))
        eng.ctx.set_profiling(False)
        nk = len(logs[0][0])
        print("== %s  wall %.3f ms (unprofiled)" % (q, wall))
        tot = 0.0
        for i in range(nk):
            ms = statistics.median(l[0][i][1] for l in logs)
            tot += ms
            print("   %-18s %8.4f ms" % (logs[0][0][i][0], ms))
        print("   %-18s %8.4f ms   (device, per call: %s)" % ("sum of kernels", tot, ", ".join(
            "%s %.3f" % (logs[0][1][i][0], statistics.median(l[1][i][1] for l in logs)) for i in range(len(logs[0][1])))))


if __name__ == "__main__":
    main()
