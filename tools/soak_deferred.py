import os, sys, time, resource
sys.path.insert(0, os.getcwd())
from sdqlpy_amd import engine, tpch, tpch_queries as Q
from sdqlpy_amd.sdql_lib import sdqlpy_init
qs = ["q3", "q1", "q5"]
sdqlpy_init(3, 1, device=0)
eng = engine.default_engine(device=0)
db = tpch.generate(10, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs))
ref = {q: Q.run(q, db).wait().rows() for q in qs}
import random
random.seed(1)
t0 = time.perf_counter()
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
for step in range(STEPS):
    rs = [Q.run(q, db) for q in qs]
    mode = random.random()
    if mode < 0.6:
        for r in rs: r.wait()
    elif mode < 0.8:
        rs[1].wait()                      # the others dropped unread
    # else: all dropped unread
    if step % EVERY == EVERY - 1:
        eng.ctx.synchronize()
        per_step = (time.perf_counter() - t0) / EVERY * 1e3
        got = {q: Q.run(q, db).wait().rows() for q in qs}
        ok = all(got[q] == ref[q] for q in qs)
        if not ok:                                   # what differs: keys / counts (never allowed) or sums in their last bits (the order of atomic adds)?
            for q in qs:
                a, b = sorted(got[q]), sorted(ref[q])
                worst, exact_ok = 0.0, len(a) == len(b)
                for ra, rb in zip(a, b):
                    for x, y in zip(ra, rb):
                        if isinstance(x, float) or isinstance(y, float):
                            worst = max(worst, abs(x - y) / max(abs(y), 1e-300))
                        elif x != y:
                            exact_ok = False
                if a != b:
                    print("   %s: %d rows, keys / counts equal %s, largest relative difference of a sum %.3g" % (q, len(a), exact_ok, worst), flush=True)
                    assert exact_ok and worst < 1e-12, q
        print("step %d: %.3f ms/step, maxrss %.0f MB, pool %d blocks, quarantine %d/%d, identical %s" % (step + 1, per_step,
              resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, sum(len(v) for v in eng.ctx._host_pool.values()), len(eng.ctx._host_quarantine), len(eng.ctx._deferred_quarantine), ok), flush=True)
        t0 = time.perf_counter()
