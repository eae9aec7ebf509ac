#!/bin/bash
# Where the bytes of a build kernel go (round-5 review: Q5's xk_build_tight moves 3.4 x its modelled stream, Q3's xk_build_values 1.58 x):
# FETCH_SIZE / WRITE_SIZE per launch (separate passes), launch by launch — Q5's two builds (customer, orders) apart — on the code as it is
# and with the build loops cut short (X8_EXP_BUILD=1: the timing experiments of sdqh_xkernels.hpp applied to the BUILD loops):
#   X8_EXP=3  survivors queued, never drained: the stream and the first lookup's bitmap tests only — no gathers of the looked-up entry, no sink
#   X8_EXP=5  drained and evaluated (the entry's fields gathered), nothing reaches the sink
#   bash tools/build_bytes.sh q5 gpurun_out/r06/build_bytes_q5.txt
Q=${1:-q5}; OUT=${2:-gpurun_out/build_bytes_$Q.txt}; ITERS=${3:-4}
D=$(dirname $OUT)/bb_$Q
mkdir -p $D
export TMPDIR=/tmp
: > $OUT
VARIANTS=${VARIANTS:-"base|X8_EXP_BUILD=1 X8_EXP=3|X8_EXP_BUILD=1 X8_EXP=5|XS_EXP=1|XS_EXP=2|XS_EXP=4|XS_EXP=7"}
#   XS_EXP (the stage sink's parts, bits): 1 no bitmap atomics, 2 no row-index notes, 4 nothing stored to the stage arrays (results ARE wrong)
IFS='|' read -ra VS <<< "$VARIANTS"
for variant in "${VS[@]}"; do
  defs=""; [ "$variant" != "base" ] && defs="$variant"
  tag=$(echo "$variant" | tr ' =' '__')
  for c in FETCH_SIZE WRITE_SIZE; do
    SDQLPY_AMD_X_DEFINES="$defs" SDQLPY_AMD_JIT_CACHE=/tmp/jit_bb_$tag rocprofv3 --pmc $c --output-format csv -d $D/${tag}_$c -- python3 tools/run_queries.py --sf 10 --queries $Q --iters $ITERS > $D/${tag}_$c.log 2>&1
  done
  echo "######## variant: $variant" >> $OUT
  python3 - "$D" "$tag" >> $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d, tag = sys.argv[1], sys.argv[2]
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for path in glob.glob(os.path.join(d, tag + "_" + c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == c and ("xk_build" in row["Kernel_Name"] or "xk_group" in row["Kernel_Name"] or "xk_probe" in row["Kernel_Name"]):
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"].split("(")[0], int(row.get("Grid_Size", 0) or 0), float(row["Counter_Value"])))
    rows.sort()
    per[c] = rows
# launches of one kernel name alternate (customer build, orders build, ...): group by (name, grid size)
acc = defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for c, rows in per.items():
    for _, name, grid, v in rows:
        acc[(name, grid)][c].append(v)
for (name, grid), v in sorted(acc.items()):
    f = sum(v["FETCH_SIZE"][1:]) / max(1, len(v["FETCH_SIZE"][1:])) if len(v["FETCH_SIZE"]) > 1 else (v["FETCH_SIZE"] or [0])[0]
    w = sum(v["WRITE_SIZE"][1:]) / max(1, len(v["WRITE_SIZE"][1:])) if len(v["WRITE_SIZE"]) > 1 else (v["WRITE_SIZE"] or [0])[0]
    print("  %-22s grid %-8d launches %d  FETCH_SIZE %9.1f KiB  WRITE_SIZE %9.1f KiB  hbm = 2*F + W = %7.1f MB (read %7.1f MB, written %6.1f MB)"
          % (name, grid, len(v["FETCH_SIZE"]), f, w, (2 * f + w) * 1024 / 1e6, 2 * f * 1024 / 1e6, w * 1024 / 1e6))
PY
done
rm -rf $D
cat $OUT
