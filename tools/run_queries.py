#!/usr/bin/env python3
"""Run the hot-path queries a few times with no instrumentation of our own — the target program for
rocprofv3 (--kernel-trace --stats, or --pmc in a separate run).
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 tools/run_queries.py --sf 10 --queries q1,q3 --iters 10"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--queries", default="q1,q3")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--reference-width", action="store_true", help="the columns in the reference's own 8-byte form (no 4-byte twins, no dictionary codes of numeric "
                    "columns), the loops on the fixed-shape kernels: bench.py's `reference_width` leg, for the counter passes that price it")
    args = ap.parse_args()
    qs = args.queries.split(",")
    from sdqlpy_amd import engine, tpch
    from sdqlpy_amd import tpch_queries as Q
    from sdqlpy_amd.sdql_lib import sdqlpy_init
    sdqlpy_init(3, 1, device=0)
    if args.reference_width:
        import bench
        bench.set_reference_width(engine.default_engine(device=0), True)      # (exactly what bench.py's leg runs)
    db = tpch.generate(args.sf, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
    for _ in range(args.iters):
        for q in qs:
            r = Q.run(q, db)
            if hasattr(r, "wait"):
                r.wait()                                 # (a query may return with its last call queued: finished here, run by run)


if __name__ == "__main__":
    main()
