#!/usr/bin/env python3
"""Where the host spends a query's wall time: cProfile over N warm runs of one query (tools/host_profile.py q5 --sf 10)."""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("query")
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    from sdqlpy_amd import sdql_lib, tpch, tpch_queries as Q
    sdql_lib.sdqlpy_init(3)
    db = tpch.generate(args.sf, tables=Q.QUERY_TABLES[args.query], columns=tpch.columns_for([args.query]))
    tables = [db[t] for t in Q.QUERY_TABLES[args.query]]
    fn = Q.QUERIES[args.query]
    for _ in range(5):
        fn(*tables)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        fn(*tables)
    wall = (time.perf_counter() - t0) / args.iters
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.iters):
        fn(*tables)
    pr.disable()
    print("%s: %.3f ms per run (unprofiled)" % (args.query, wall * 1e3))
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
