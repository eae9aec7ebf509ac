#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json: HBM bytes per query and per kernel, from rocprofv3 PMC passes.

Collect on the GPU box (separate passes per counter, as MI355X_MICROARCH.md prescribes; the program
after `--` is python itself):

    for q in q1 q3 q5 q6 q9; do for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc/${q}_$c -- python3 tools/run_queries.py --sf 10 --queries $q --iters 5
    done; done
    python3 tools/pmc_per_query.py gpurun_out/pmc 5 '<rows json>' profiles/r02_pmc_traffic.json

HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB): on gfx950 FETCH_SIZE counts a
128-byte fabric request at 64 bytes, i.e. half of a 16-byte-per-lane streaming read; WRITE_SIZE is
exact.  One-off launches of the first iteration (k_minmax, the upload's copy kernels) are left out;
everything else is summed over the run and divided by the iteration count.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

TABLES = {"q1": ["lineitem"], "q6": ["lineitem"], "q3": ["lineitem", "orders", "customer"],
          "q5": ["lineitem", "orders", "customer", "supplier"], "q9": ["lineitem", "orders", "part", "partsupp", "supplier"]}
ONE_OFF = ("k_minmax", "k_check_increasing", "k_check_nondecreasing", "k_check_pair_increasing", "k_narrow_", "k_delta8", "k_code_", "__amd_rocclr")      # twins, dictionaries and column facts: built at the first run only
BUILT_ONCE = ("k_interleave", "k_rs_", "k_lower_bounds", "k_run_index")      # resident structures built at the first run only (row packs — in row order or clustered: the radix passes —, run indexes): reported apart


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "", 1) if name.startswith(("void (anonymous", "(anonymous")) else name
    name = name.split("(")[0].split("<")[0]
    for p in ("void sdqh::", "sdqh::"):
        if name.startswith(p):
            name = name[len(p):]
    return name.strip()


def counter_sums(d, counter):
    total, n = defaultdict(float), defaultdict(int)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            if k.startswith(ONE_OFF):
                continue
            total[k] += float(row["Counter_Value"])
            n[k] += 1
    return total, n


def main(root, iters, rows_json, out):
    iters = int(iters)
    rec = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes per query over `python3 tools/run_queries.py --sf 10 --queries <q> --iters %d`" % iters,
           "correction": "hbm = 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of a wide coalesced read); KiB -> bytes",
           "rows": json.loads(rows_json), "queries": {},
           # which build the counters were collected on (the box has no .git: the caller passes the commit in the environment)
           "commit": os.environ.get("SDQLPY_COMMIT", "")}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    rec["source_digest"] = bench.source_digest()      # the sources the counters were collected on: bench.py says `STALE` when it runs on others
    for q in sorted(TABLES):
        fd, wd = os.path.join(root, q + "_FETCH_SIZE"), os.path.join(root, q + "_WRITE_SIZE")
        if not (os.path.isdir(fd) and os.path.isdir(wd)):
            continue
        f, fn = counter_sums(fd, "FETCH_SIZE")
        w, _ = counter_sums(wd, "WRITE_SIZE")
        kernels, once, run_bytes = {}, {}, 0.0
        for k in sorted(set(f) | set(w)):
            b = (2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024
            launches = fn.get(k, 0)
            if k.startswith(BUILT_ONCE):
                once[k] = {"hbm_bytes_per_launch": int(b / max(1, launches)), "note": "built at the first run, resident afterwards (left out of hbm_bytes_per_run)"}
                continue
            kernels[k] = {"launches_per_run": round(launches / iters, 2), "hbm_bytes_per_launch": int(b / max(1, launches)),
                          "hbm_bytes_per_run": int(b / iters)}
            run_bytes += b / iters
        rec["queries"][q] = {"tables": TABLES[q], "hbm_bytes_per_run": int(run_bytes), "kernels": kernels}
        if once:
            rec["queries"][q]["built_once"] = once
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps({q: v["hbm_bytes_per_run"] for q, v in rec["queries"].items()}))


if __name__ == "__main__":
    main(*sys.argv[1:5])
