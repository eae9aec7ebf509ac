#!/usr/bin/env python3
"""Host time of ONE launch of each recorded distributed plan (RCCL group of one, collectives issued) and the nodes its recording holds:
is the recorded step the device's or the host's?
    python tools/dist_graph_launch_times.py [q1,q3,q5]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from sdqlpy_amd import engine, tpch
from sdqlpy_amd import dist as sdist

qs = (sys.argv[1] if len(sys.argv) > 1 else "q1,q3,q5").split(",")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29617", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = engine.Engine(engine.load_hip_library().context(device=0))
runner = sdist.DistributedRunner(eng, 0, 1, skip_trivial=False, partition=os.environ.get("PARTITION", "hash"))
cols = tpch.columns_for(qs)
db = tpch.generate(10.0, tables=sorted(cols), columns=cols, shard=(0, 1))
for _ in range(6):
    for q in qs:
        r = runner.run(q, db)
        r.size()
torch.cuda.synchronize()
N = 200
for q in qs:
    t_launch = t_total = 0.0
    for _ in range(N):
        t0 = time.perf_counter()
        r = runner.run(q, db)
        t1 = time.perf_counter()
        r.size()
        t2 = time.perf_counter()
        t_launch += t1 - t0; t_total += t2 - t0
    print("%s: run() returns after %.1f us, result collected after %.1f us" % (q, t_launch / N * 1e6, t_total / N * 1e6), flush=True)
# the step: three launches, then three collections
t_l = t_s = 0.0
for _ in range(N):
    t0 = time.perf_counter()
    rs = [runner.run(q, db) for q in qs]
    t1 = time.perf_counter()
    for r in rs:
        r.size()
    t2 = time.perf_counter()
    t_l += t1 - t0; t_s += t2 - t0
print("step: launches %.1f us, whole %.1f us" % (t_l / N * 1e6, t_s / N * 1e6))
pgs = []
for name, (fn, plan) in list(runner._plans.items()):
    for cache in ("_dist_prepared", "_dist_chain"):
        for key, st in (plan.__dict__.get(cache) or {}).items():
            for rec in st.__dict__.get("recordings", []) or []:
                print("%s: recording of %d nodes" % (getattr(plan, "name", name), rec["pg"].graph.nodes))
                pgs.append((getattr(plan, "name", name), rec["pg"]))
                break
# the bare launches (hipGraphLaunch through sdqh_graph_launch), one after the other and from a thread each
eng.synchronize(); torch.cuda.synchronize()
import threading
for label, threaded in (("one after the other", False), ("a thread each", True)):
    per = {n: 0.0 for n, _ in pgs}
    tot = 0.0
    for _ in range(N):
        t0 = time.perf_counter()
        if threaded:
            ths = [threading.Thread(target=pg.graph.launch) for _, pg in pgs]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        else:
            for n, pg in pgs:
                t1 = time.perf_counter()
                pg.graph.launch()
                per[n] += time.perf_counter() - t1
        tot += time.perf_counter() - t0
        eng.synchronize(); torch.cuda.synchronize()
    print("bare launches %s: %.1f us per step %s" % (label, tot / N * 1e6, {n: round(v / N * 1e6, 1) for n, v in per.items()} if not threaded else ""))
print("recordings %d launches %d" % (runner.graph_recordings, runner.graph_launches))
runner.close(); torch.cuda.synchronize(); dist.destroy_process_group(); eng.close()
