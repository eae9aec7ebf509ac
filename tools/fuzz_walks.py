#!/usr/bin/env python3
"""Extra rounds of the round-5 cases with seeds and sizes the suite does not use: the clustered row pack and its run walk (tests/helpers.
cluster_pack_case: numpy inside, the CPU implementation beside) with the pack forced on and off, and Q3 / Q5 / Q9 / Q10 / Q12 on generated
tables of odd sizes with the walks and the delta twins at their most eager against the CPU implementation.
python tools/fuzz_walks.py [first_seed] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from sdqlpy_amd import abi, build, engine, tpch

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000), (int(sys.argv[2]) if len(sys.argv) > 2 else 12)
hip = engine.Engine(abi.Library(build.HIP_LIB).context(device=0))
cpu = engine.Engine(abi.Library(os.path.join(ROOT, "oracle", "libsdqloracle.so")).context(threads=16))
hip.ctx.set_option("feature_min_rows", 0)
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    kw = dict(seed=seed, nparts=int(rng.integers(50, 200000)), nprobe=int(rng.integers(1, 3000000)), keep=float(rng.choice([0.01, 0.05, 0.3, 1.0])))
    kw["n"] = int(kw["nparts"] * rng.integers(1, 6))
    want = helpers.cluster_pack_case(cpu.ctx, **kw)
    for opt in (2, 0):
        hip.ctx.set_option("cluster_pack", opt)
        got = helpers.cluster_pack_case(hip.ctx, **kw)
        ok = got[0] == want[0] and got[1] == want[1] and all(abs(x - y) <= 1e-10 * max(abs(y), 1.0) for x, y in zip(got[2], want[2]))
        bad += 0 if ok else 1
        print("cluster_pack=%d %r %s" % (opt, kw, "ok" if ok else "MISMATCH"), flush=True)
hip.ctx.set_option("cluster_pack", 1)
qs = ("q3", "q5", "q9", "q10", "q12", "q4")
for seed, sf in ((first, 0.07), (first + 1, 0.61), (first + 2, 3.3)):
    db = tpch.generate(sf, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs), seed=seed) if "seed" in tpch.generate.__code__.co_varnames else tpch.generate(sf, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
    for ratio, d8 in ((1, 1), (64, 1), (0, 0)):
        hip.ctx.set_option("x_driven", ratio); hip.ctx.set_option("delta8", d8)
        hip.clear()
        for q in qs:
            want = helpers.run_query(cpu, q, db)
            for _ in range(2):
                got = helpers.run_query(hip, q, db)
                try:
                    helpers.assert_rows_match(helpers.result_rows(got, want.columns), helpers.result_rows(want, want.columns), 1e-10, "sf=%s %s" % (sf, q))
                except AssertionError as exc:
                    bad += 1
                    print("MISMATCH sf=%s x_driven=%d delta8=%d %s: %s" % (sf, ratio, d8, q, str(exc)[:200]), flush=True)
    print("sf %s: %d queries x 3 settings ok" % (sf, len(qs)), flush=True)
    cpu.clear()
print("fuzz_walks done: %d mismatches" % bad)
sys.exit(1 if bad else 0)
