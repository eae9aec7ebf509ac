#!/usr/bin/env python3
"""Host time per ABI call of a query at a tiny scale factor (the kernels do nothing there: what is left is the host side
of every launch and call).  tools/call_times.py q5"""
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    q = sys.argv[1] if len(sys.argv) > 1 else "q5"
    sf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.002
    from sdqlpy_amd import abi, sdql_lib, tpch, tpch_queries as Q
    sdql_lib.sdqlpy_init(3)
    db = tpch.generate(sf, tables=Q.QUERY_TABLES[q], columns=tpch.columns_for([q]))
    tables = [db[t] for t in Q.QUERY_TABLES[q]]
    fn = Q.QUERIES[q]
    for _ in range(20):
        fn(*tables)
    acc, cnt = defaultdict(float), defaultdict(int)

    def wrap(name):
        orig = getattr(abi.Context, name)

        def timed(self, *a, **k):
            t0 = time.perf_counter()
            try:
                return orig(self, *a, **k)
            finally:
                acc[name] += time.perf_counter() - t0
                cnt[name] += 1
        setattr(abi.Context, name, timed)
    for name in ("build", "build_marshalled", "build_key_set", "hash_build_unique", "hash_probe_aggregate", "lookup_aggregate_marshalled", "groupby_small",
                 "scan_filter_sum", "scan_probe_sum", "groupby_key", "table_select_keys", "table_share_groups", "table_compact", "table_compact_count",
                 "table_compact_into_block", "table_topk", "table_entries", "host_block", "xbuild", "xgroupby", "xscan_sum", "xprobe_aggregate", "xkey_set"):
        if hasattr(abi.Context, name):
            wrap(name)
    orig_free = abi.Table.free

    def timed_free(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return orig_free(self, *a, **k)
        finally:
            acc["Table.free"] += time.perf_counter() - t0
            cnt["Table.free"] += 1
    abi.Table.free = timed_free
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        fn(*tables)
    wall = (time.perf_counter() - t0) / n
    print("%s at SF %g: %.1f us per run" % (q, sf, wall * 1e6))
    inside = 0.0
    for name in sorted(acc, key=lambda k: -acc[k]):
        print("  %-22s %5.1f calls/run  %7.1f us per call  %7.1f us per run" % (name, cnt[name] / n, acc[name] / cnt[name] * 1e6, acc[name] / n * 1e6))
        inside += acc[name] / n
    print("  outside the ABI calls: %.1f us per run" % ((wall - inside) * 1e6))


if __name__ == "__main__":
    main()
