#!/usr/bin/env python3
"""The overlapped step (q1, q3, q5 launched on three lanes, then finished) dispatch by dispatch, out of a rocprofv3 kernel trace of
bench.py: for ONE step of the timed region every kernel with its lane (queue), start and end relative to the step's first kernel, and
the step's union of busy time — where the lanes overlap, where the chip idles, which kernels stretch when they share the chip.

    rocprofv3 --kernel-trace --output-format csv -d out/step -- python3 bench.py --no-cpu-baseline --no-reference-width --extra-queries "" --steady-steps 0 --steps 40
    python tools/step_timeline.py out/step [which step: N from the end (default 3), -N from the start — the timed region follows the first pass and the warm-up]
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


def main(root, back=3):
    recs = []
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            recs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short(row["Kernel_Name"]), row.get("Queue_Id", "?")))
    recs.sort()
    # a step starts with q1's kernel: cut at every xk_group_lane_tight
    starts = [i for i, r in enumerate(recs) if r[2] == "xk_group_lane_tight"]
    if len(starts) < back + 2:
        print("too few steps in the trace")
        return
    a, b = (starts[-back - 1], starts[-back]) if back > 0 else (starts[-back], starts[-back + 1])      # back < 0: the |back|-th step from the START of the trace
    step = recs[a:b]
    t0 = step[0][0]
    queues = sorted({r[3] for r in step})
    print("# one step: %d dispatches on %d queues, first start -> last end %.4f ms" % (len(step), len(queues), (max(r[1] for r in step) - t0) / 1e6))
    print("# %-6s %-26s %9s %9s %9s" % ("queue", "kernel", "start ms", "end ms", "dur ms"))
    for s, e, n, q in step:
        print("  %-6s %-26s %9.4f %9.4f %9.4f" % (queues.index(q), n[:26], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
    ev = sorted([(s, 1) for s, e, n, q in step] + [(e, -1) for s, e, n, q in step])
    busy, depth, last = 0, 0, None
    for t, d in ev:
        if depth > 0:
            busy += t - last
        depth += d
        last = t
    tot = sum(e - s for s, e, n, q in step)
    print("# sum of durations %.4f ms, union of busy time %.4f ms, idle inside the step %.4f ms" % (tot / 1e6, busy / 1e6, (max(r[1] for r in step) - t0 - busy) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
