#!/usr/bin/env python3
"""Host time per call of the settled distributed plans (q1, q5, q3) on an RCCL group of one with its collectives issued, SF=10: where the
runner's launch time goes — ABI calls, torch collectives, torch tensor ops, and what is left (Python between them).
python tools/dist_call_times2.py [q5,q1,q3]"""
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29592")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from sdqlpy_amd import abi, dist as sdist, engine, tpch
from sdqlpy_amd.sdql_lib import sdqlpy_init

qs = (sys.argv[1] if len(sys.argv) > 1 else "q5,q1,q3").split(",")
sdqlpy_init(3, 1, device=0)
db = tpch.generate(10, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs), shard=(0, 1))
eng = engine.default_engine(device=0)
runner = sdist.DistributedRunner(eng, 0, 1, partition="hash", skip_trivial=False)
for q in qs:
    for _ in range(10):
        r = runner.run(q, db); r.wait() if hasattr(r, "wait") else None
acc, cnt = defaultdict(float), defaultdict(int)


def wrap(owner, name, label=None):
    orig = getattr(owner, name)
    label = label or (getattr(owner, "__name__", str(owner)) + "." + name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t0
            cnt[label] += 1
    setattr(owner, name, timed)

for name in [n for n in dir(abi.Context) if not n.startswith("_") and callable(getattr(abi.Context, n)) and n not in ("stream", "profile", "close")]:
    wrap(abi.Context, name)
for name in ("free",):
    wrap(abi.Table, name); wrap(abi.Column, name)
wrap(abi.Column, "set_bounds")
for name in ("all_gather_into_tensor", "all_reduce", "all_to_all_single"):
    wrap(dist, name, "torch.distributed." + name)
for name in ("copy_", "amax"):
    wrap(torch.Tensor, name, "torch.Tensor." + name)
wrap(torch, "empty", "torch.empty")
n = 300
for q in qs:
    acc.clear(); cnt.clear()
    launch = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        t1 = time.perf_counter()
        r = runner.run(q, db)
        launch += time.perf_counter() - t1
        r.wait() if hasattr(r, "wait") else None
    wall = (time.perf_counter() - t0) / n
    print("%s settled on a group of one, collectives issued: %.1f us per run, %.1f us to launch" % (q, wall * 1e6, launch / n * 1e6))
    inside = 0.0
    for name in sorted(acc, key=lambda k: -acc[k]):
        if acc[name] / n * 1e6 < 1.0:
            continue
        print("  %-44s %5.1f calls/run  %7.1f us per call  %7.1f us per run" % (name, cnt[name] / n, acc[name] / cnt[name] * 1e6, acc[name] / n * 1e6))
        inside += acc[name] / n
    print("  outside these calls (Python between them): %.1f us per run of %.1f us to launch" % ((launch / n - inside) * 1e6, launch / n * 1e6))
runner.close()
torch.cuda.synchronize()
dist.destroy_process_group()
