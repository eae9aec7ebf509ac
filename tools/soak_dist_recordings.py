#!/usr/bin/env python3
"""Soak of the recorded distributed plans (RCCL group of one, collectives issued): the chains of q1 / q5 / q9 and the partitioned join of
q3 are settled, recorded and replayed, and every few rounds a plan's recordings are made stale (its chunk bounds nudged), so that it
settles and is recorded AGAIN — hundreds of captures with eager collectives right before them, the pattern that dumped core a few per
cent of the time while torch's watchdog could still poll an event of the captured stream (dist.DistributedRunner._coll).  Results are
compared with the single-GPU plan's every round; the pools' size is printed at the start and at the end (a recording owns pool memory).
    python tools/soak_dist_recordings.py [rounds=150] [sf=1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
import helpers
from sdqlpy_amd import engine, tpch
from sdqlpy_amd import dist as sdist

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
sf = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29631", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = engine.Engine(engine.load_hip_library().context(device=0))
runner = sdist.DistributedRunner(eng, 0, 1, skip_trivial=False, partition="hash")
qs = ("q1", "q3", "q5", "q9")
cols = tpch.columns_for(qs)
db = tpch.generate(sf, tables=sorted(cols), columns=cols, shard=(0, 1))
want = {}
for q in qs:
    r = helpers.run_query(eng, q, db)
    want[q] = sorted(map(tuple, (r.wait() if hasattr(r, "wait") else r).rows()))


def check(q, got, tag):
    a = sorted(map(tuple, got.rows()))
    assert len(a) == len(want[q]), (q, tag, len(a), len(want[q]))
    for x, y in zip(a, want[q]):
        for u, v in zip(x, y):
            assert u == v or (isinstance(u, float) and abs(u - v) <= 1e-9 * max(1.0, abs(v))), (q, tag, x, y)


def states(q):
    fn, plan, _ = runner._resolve(q, db)
    out = []
    for cache in ("_dist_chain", "_dist_prepared"):
        for key, st in (plan.__dict__.get(cache) or {}).items():
            if key[0] == id(runner):
                out.append(st)
    return out


mem0 = None
t0 = time.time()
for i in range(rounds):
    rs = [(q, runner.run(q, db)) for q in qs]
    for q, r in rs:
        check(q, r, i)
    if i == 8:
        mem0 = eng.ctx.memory_stats(family=True)
    if i >= 8 and i % 3 == 0:
        # make one plan's recordings stale: its bounds grow by a row (every rank would do the same: the bounds are all-reduced facts)
        q = ("q5", "q9", "q3")[(i // 3) % 3]
        for st in states(q):
            caps = st.caps
            if isinstance(caps, dict):
                for name in caps:
                    caps[name] = int(caps[name]) + 8
            elif caps is not None:
                st.caps = tuple(int(c) + 8 for c in caps)
mem1 = eng.ctx.memory_stats(family=True)
print("rounds %d in %.1f s: recordings %d, launches %d, settled runs %d, repeated with exact sizes %d" % (rounds, time.time() - t0, runner.graph_recordings, runner.graph_launches, runner.fast_runs, runner.fast_retries))
print("pools after round 8:", mem0)
print("pools at the end:   ", mem1)
runner.close(); torch.cuda.synchronize(); dist.destroy_process_group(); eng.close()
print("ok")
