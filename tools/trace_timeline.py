#!/usr/bin/env python3
"""Timeline of ONE steady-state run of a query out of a rocprofv3 kernel trace: every dispatch in order with its start (relative to the
run's first kernel), its duration and the idle gap in front of it — where a query's wall time goes between its big kernels.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl/q5 -- python3 tools/run_queries.py --sf 10 --queries q5 --iters 8
    python3 tools/trace_timeline.py gpurun_out/tl/q5
"""
import csv
import glob
import os
import sys


def short(name):
    name = name.split("(")[0].split("<")[0]
    for p in ("void sdqh::", "sdqh::", "void "):
        if name.startswith(p):
            name = name[len(p):]
    return name.strip()


def main(root):
    recs = []
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            recs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short(row["Kernel_Name"])))
    recs.sort()
    if not recs:
        print("no kernel trace under", root)
        return
    last = recs[-1][2]
    ends = [i for i, r in enumerate(recs) if r[2] == last]
    if len(ends) < 3:
        print("fewer than three runs in the trace")
        return
    run = recs[ends[-2] + 1: ends[-1] + 1]                      # the last whole run
    t0 = run[0][0]
    prev_end = t0
    total = 0
    print("# %d dispatches, first start -> last end %.4f ms" % (len(run), (run[-1][1] - t0) / 1e6))
    print("# %-28s %10s %10s %10s" % ("kernel", "start ms", "dur ms", "gap ms"))
    for s, e, n in run:
        print("  %-28s %10.4f %10.4f %10.4f" % (n[:28], (s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6))
        total += e - s
        prev_end = max(prev_end, e)
    print("# sum of durations %.4f ms, idle %.4f ms" % (total / 1e6, (run[-1][1] - t0 - total) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
