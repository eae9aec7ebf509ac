"""Experiment: the bench step (q1 + q3 + q5 launched, then finished) with every query on a context — a stream, a pool, result
blocks — of its own, against all three on one context.  Each context uploads its own copy of the columns its query reads (an
experiment's shortcut: a product version would share the resident columns).
    python tools/step_lanes.py --sf 10 --steps 300"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=10.0)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--queries", default="q1,q3,q5")
    args = ap.parse_args()
    from sdqlpy_amd import engine, tpch
    from sdqlpy_amd import tpch_queries as Q
    qs = args.queries.split(",")
    need = tpch.columns_for(qs)
    db = tpch.generate(args.sf, tables=sorted(need), columns=need)
    rows = sum(len(db[t].getContainer()["data"][0]) for q in qs for t in Q.QUERY_TABLES[q])
    lib = engine.load_hip_library()
    one = engine.Engine(lib.context(device=0))
    lanes = {q: engine.Engine(lib.context(device=0)) for q in qs}

    def step(engs):
        rs = []
        for q in qs:
            engine.use_engine(engs[q])
            rs.append(Q.run(q, db))
        for r in rs:
            r.wait() if hasattr(r, "wait") else None
        return rs

    def sync(engs):
        for e in set(engs.values()):
            e.ctx.synchronize()

    out = {}
    for name, engs in (("one context", {q: one for q in qs}), ("a context per query", lanes), ("one context", {q: one for q in qs}), ("a context per query", lanes)):
        for _ in range(10):
            step(engs)
        sync(engs)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(engs)
        sync(engs)
        ms = (time.perf_counter() - t0) * 1e3 / args.steps
        print("%-22s %.4f ms per step   %.1f G rows/s" % (name, ms, rows / ms / 1e6), flush=True)
        out.setdefault(name, []).append(ms)
    a = [str(r) for r in step({q: one for q in qs})]
    b = [str(r) for r in step(lanes)]
    print("results identical:", a == b)


if __name__ == "__main__":
    main()
