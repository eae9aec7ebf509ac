#!/usr/bin/env python3
"""Host time per ABI call of the hash-partitioned Q3 on an RCCL group of one (device-sized exchanges), SF=10: where the runner's
host time goes.   python tools/dist_call_times.py"""
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from sdqlpy_amd import abi, dist as sdist, engine, tpch
from sdqlpy_amd.sdql_lib import sdqlpy_init

sdqlpy_init(3, 1, device=0)
db = tpch.generate(10, tables=sorted(tpch.columns_for(["q3"])), columns=tpch.columns_for(["q3"]), shard=(0, 1))
eng = engine.default_engine(device=0)
runner = sdist.DistributedRunner(eng, 0, 1, partition="hash")
for _ in range(10):
    runner.run("q3", db).wait()
acc, cnt = defaultdict(float), defaultdict(int)


def wrap(cls, name):
    orig = getattr(cls, name)

    def timed(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return orig(self, *a, **k)
        finally:
            acc[cls.__name__ + "." + name] += time.perf_counter() - t0
            cnt[cls.__name__ + "." + name] += 1
    setattr(cls, name, timed)


for name in ("build_key_set", "xbuild", "xstage", "table_partition_pack", "unpack_chunks", "hash_probe_aggregate", "xprobe_aggregate", "table_compact_deferred",
             "chunk_words", "wrap", "set_option", "host_block", "synchronize", "hash_build_unique"):
    wrap(abi.Context, name)
for name in ("free",):
    wrap(abi.Table, name)
    wrap(abi.Column, name)
wrap(abi.Column, "set_bounds")
n = 200
launch = 0.0
t0 = time.perf_counter()
for _ in range(n):
    t1 = time.perf_counter()
    r = runner.run("q3", db)
    launch += time.perf_counter() - t1
    r.wait()
wall = (time.perf_counter() - t0) / n
print("q3 hash-partitioned on a group of one: %.1f us per run, %.1f us to launch" % (wall * 1e6, launch / n * 1e6))
inside = 0.0
for name in sorted(acc, key=lambda k: -acc[k]):
    print("  %-34s %5.1f calls/run  %7.1f us per call  %7.1f us per run" % (name, cnt[name] / n, acc[name] / cnt[name] * 1e6, acc[name] / n * 1e6))
    inside += acc[name] / n
print("  outside these calls: %.1f us per run" % ((wall - inside) * 1e6))
runner.close()
torch.cuda.synchronize()
dist.destroy_process_group()
