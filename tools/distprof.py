#!/usr/bin/env python3
"""Host profile of the distributed plan on one rank (RCCL group of size 1): where the fixed cost of
the multi-GPU path goes.  python tools/distprof.py q3,q5"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from sdqlpy_amd import dist as sdist, engine, tpch
from sdqlpy_amd.sdql_lib import sdqlpy_init

qs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["q3", "q5"]
sdqlpy_init(3, 1, device=0)
db = tpch.generate(10, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs), shard=(0, 1))      # marked as row shards: the partitioned plans run
runner = sdist.DistributedRunner(engine.default_engine(device=0), 0, 1, partition=os.environ.get("PARTITION", "hash"),
                                 skip_trivial=os.environ.get("TRIVIAL", "0") != "1")


def finished(r):
    return r.wait() if hasattr(r, "wait") else r

for q in qs:
    for _ in range(5):
        finished(runner.run(q, db))
    t0 = time.perf_counter()
    for _ in range(50):
        finished(runner.run(q, db))
    print(q, "mean wall ms", (time.perf_counter() - t0) * 20, flush=True)
    # one run with events around every launch (waited for: profiling switches the deferred K-F off), in launch order
    ctx = engine.default_engine(device=0).ctx
    ctx.set_profiling(2, only=None)
    finished(runner.run(q, db))
    ctx.synchronize()
    launches = ctx.profile()
    ctx.set_profiling(0)
    print(q, "launches:", " | ".join("%s %.4f" % (n, ms) for n, ms in launches), "| sum %.4f ms in %d launches" % (sum(ms for _, ms in launches), len(launches)), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        finished(runner.run(q, db))
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
runner.close()
torch.cuda.synchronize()
dist.destroy_process_group()
