#!/usr/bin/env python3
"""The last steps of a traced `bench.py --force-dist` run as a timeline per queue: every kernel (and copy) with its start relative to the
step's first kernel, its duration and the gap since the previous activity on the same queue — where a recorded distributed plan's wall
time goes between its kernels.
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r06/dtrace -- python3 bench.py --force-dist --trivial-collectives ...
    python tools/dist_timeline.py gpurun_out/r06/dtrace [window_ms]"""
import csv
import glob
import os
import sys


def main(root, window_ms=2.5):
    rows = []
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q%s" % r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][:44]))
    for path in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", "%s %s" % (r.get("Direction", "copy"), r.get("Bytes", ""))))
    rows.sort()
    if not rows:
        print("no trace under", root)
        return
    # the timed steady-state region is the densest part of the trace: take a window in its last third
    t_end = rows[-1][1]
    t0 = t_end - int(0.25e9)                                  # 0.25 s before the end: well inside the steady-state loop of 1500 steps
    sel = [r for r in rows if t0 <= r[0] < t0 + int(window_ms * 1e6)]
    base = sel[0][0]
    last = {}
    print("%9s %8s %8s  %-6s %s" % ("start_us", "dur_us", "gap_us", "queue", "what"))
    for s, e, q, name in sel:
        gap = (s - last[q]) / 1e3 if q in last else 0.0
        last[q] = e
        print("%9.1f %8.1f %8.1f  %-6s %s" % ((s - base) / 1e3, (e - s) / 1e3, gap, q, name))
    busy = {}
    for s, e, q, name in sel:
        busy[q] = busy.get(q, 0) + (e - s)
    print("busy per queue over %.2f ms: %s" % (window_ms, {q: round(v / 1e6, 3) for q, v in busy.items()}))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 2.5)
