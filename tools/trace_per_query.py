#!/usr/bin/env python3
"""Device time of one query from a rocprofv3 kernel trace: the sum of its kernels' durations (End - Start of each
dispatch as the command processor stamps it), averaged over the steady-state runs.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/qtrace/q9 -- python3 tools/run_queries.py --sf 10 --queries q9 --iters 8
    python3 tools/trace_per_query.py gpurun_out/qtrace q1,q3,q5,q6,q9 '<rows json>' > profiles/r02_rocprofv3_per_query_kernel_sums.txt

The run is cut into query runs at the kernel every run ends with (launched once per run, seen last); the first two runs
(uploads, twins, plan lowering, hiprtc) are dropped.  `copies` are the runtime's blit kernels (result read-backs) inside
those runs: reported apart, they are not launched by the library.  bench.py's own per-query figure puts a HIP event pair
around every launch, which reads 2-3 us long on each of the 15-20 tiny kernels of a join query; this is the same sum without them."""
import csv
import glob
import json
import os
import sys

ALGO = {"q1": lambda r: 48 * r["lineitem"], "q6": lambda r: 32 * r["lineitem"],
        "q3": lambda r: 48 * r["customer"] + 32 * r["orders"] + 32 * r["lineitem"],
        "q5": lambda r: 16 * r["customer"] + 24 * r["orders"] + 16 * r["supplier"] + 32 * r["lineitem"],
        "q9": lambda r: 228 * r["part"] + 24 * r["partsupp"] + 16 * r["orders"] + 16 * r["supplier"] + 48 * r["lineitem"]}


def short(name):
    name = name.split("(")[0].split("<")[0]
    for p in ("void sdqh::", "sdqh::"):
        if name.startswith(p):
            name = name[len(p):]
    return name.strip()


def main(root, queries, rows_json):
    rows = json.loads(rows_json)
    print("# sum of kernel durations per query run, rocprofv3 --kernel-trace (ms); frac = algorithmic bytes / sum / 8 TB/s")
    for q in queries.split(","):
        recs = []
        for path in glob.glob(os.path.join(root, q, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                recs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short(row["Kernel_Name"])))
        if not recs:
            continue
        recs.sort()
        # the kernel every run ends with: among the kernels launched once per run, the one seen last — the
        # very last kernel of the trace may be one a run launches several times (a deferred result's completion store)
        own_recs = [r for r in recs if not r[2].startswith("__amd_rocclr")]
        counts = {}
        for r in own_recs:
            counts[r[2]] = counts.get(r[2], 0) + 1
        per_count = {}
        for c in counts.values():
            if c >= 4:                                           # (what only the first runs launch — twins, bounds, fills — says nothing about the runs)
                per_count[c] = per_count.get(c, 0) + 1
        if not per_count:
            continue
        once_per_run = max(per_count, key=lambda c: (per_count[c], c))          # most kernels of a plan are launched once per run: that count IS the number of runs
        last = [r for r in own_recs if counts[r[2]] == once_per_run][-1][2]
        runs, cur = [], []
        for r in recs:
            cur.append(r)
            if r[2] == last:
                runs.append(cur); cur = []
        steady = runs[2:]
        if not steady:
            continue
        own = [sum(e - s for s, e, k in run if not k.startswith("__amd_rocclr")) / 1e6 for run in steady]
        cop = [sum(e - s for s, e, k in run if k.startswith("__amd_rocclr")) / 1e6 for run in steady]
        span = [(run[-1][1] - run[0][0]) / 1e6 for run in steady]
        nk = [sum(1 for _, _, k in run if not k.startswith("__amd_rocclr")) for run in steady]
        ms = sum(own) / len(own)
        algo = ALGO[q](rows) if q in ALGO else 0
        print("%-4s runs %d  kernels/run %d  sum %.4f ms (min %.4f)  copies %.4f ms  first start -> last end %.4f ms  algorithmic %.3f GB -> %.2f TB/s  frac %.3f"
              % (q, len(steady), round(sum(nk) / len(nk)), ms, min(own), sum(cop) / len(cop), sum(span) / len(span), algo / 1e9, algo / ms / 1e9, algo / ms / 1e9 / 8.0))


if __name__ == "__main__":
    main(*sys.argv[1:4])
