#!/usr/bin/env python3
"""profiles/r01_pmc_traffic.json from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py.

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB): on gfx950 FETCH_SIZE
counts 128-byte fabric requests at 64 bytes, i.e. exactly half of a 16-byte-per-lane streaming
read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE reads the bytes exactly."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0].split("<")[0]
    for p in ("void sdqh::", "sdqh::"):
        if name.startswith(p):
            name = name[len(p):]
    return name.strip()


def avg_counter(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main(fetch_dir, write_dir, rows_json, out):
    fetch, write = avg_counter(fetch_dir, "FETCH_SIZE"), avg_counter(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                      "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    rec = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `python3 bench.py`",
           "correction": "hbm = 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of a wide coalesced read)",
           "rows": json.loads(rows_json), "kernels": kernels}
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
