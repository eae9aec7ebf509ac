#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: per-kernel average of each counter (or duration for kernel traces)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "", 1) if name.startswith(("void (anonymous", "(anonymous")) else name
    name = name.split("(")[0]
    for p in ("void sdqh::", "sdqh::"):
        if name.startswith(p):
            name = name[len(p):]
    return name[:70]


def main(d):
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(path)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("##", path)
        for k in sorted(acc):
            print("  %-72s" % k, "  ".join("%s=%.4g (n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(acc[k].items())))
    for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        acc = defaultdict(list)
        for row in csv.DictReader(open(path)):
            acc[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        print("##", path)
        for k in sorted(acc):
            v = acc[k]
            print("  %-72s n=%-4d avg=%9.2f us  min=%9.2f us" % (k, len(v), sum(v) / len(v), min(v)))


if __name__ == "__main__":
    main(sys.argv[1])
