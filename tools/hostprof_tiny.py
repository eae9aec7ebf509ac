#!/usr/bin/env python3
"""cProfile of a query's host side at a tiny scale factor (kernels take no time there): python tools/hostprof_tiny.py q5"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdqlpy_amd import sdql_lib, tpch, tpch_queries as Q

q = sys.argv[1] if len(sys.argv) > 1 else "q5"
sdql_lib.sdqlpy_init(3)
db = tpch.generate(0.002, tables=Q.QUERY_TABLES[q], columns=tpch.columns_for([q]))
tables = [db[t] for t in Q.QUERY_TABLES[q]]
fn = Q.QUERIES[q]
for _ in range(30):
    fn(*tables)
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    fn(*tables)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
