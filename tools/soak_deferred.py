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
for step in range(6000):
    rs = [Q.run(q, db) for q in qs]
    mode = random.random()
    if mode < 0.6:
        for r in rs: r.wait()
    elif mode < 0.8:
        rs[1].wait()                      # the others dropped unread
    # else: all dropped unread
    if step % 1000 == 999:
        eng.ctx.synchronize()
        ok = all(Q.run(q, db).wait().rows() == ref[q] for q in qs)
        print("step %d: %.3f ms/step, maxrss %.0f MB, pool %d blocks, quarantine %d/%d, identical %s" % (step + 1, (time.perf_counter() - t0) / 1000 * 1e3,
              resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, sum(len(v) for v in eng.ctx._host_pool.values()), len(eng.ctx._host_quarantine), len(eng.ctx._deferred_quarantine), ok), flush=True)
        t0 = time.perf_counter()
