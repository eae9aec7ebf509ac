#!/usr/bin/env python3
"""Can a collective of an RCCL group be RECORDED into a HIP graph and replayed (what a recording of a settled distributed plan would need:
DESIGN.md section 8)?  An experiment, not product code: torch's own graph capture around all_gather_into_tensor / all_reduce on a group of
one, replayed 100 times, results checked.  Run it under `timeout`: a capture that RCCL does not support may hang rather than fail.
    timeout 120 python tools/exp_rccl_in_graph.py"""
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29594")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
inp = torch.arange(4096, dtype=torch.int64, device=dev)
out = torch.zeros(4096, dtype=torch.int64, device=dev)
red = torch.ones(4, dtype=torch.int64, device=dev)
# warm up the communicator outside any capture
dist.all_gather_into_tensor(out, inp); dist.all_reduce(red, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print("eager collectives ok", flush=True)
try:
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            tmp = inp * 2
            dist.all_gather_into_tensor(out, tmp)
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
            res = out + red[0]
    torch.cuda.synchronize()
    print("captured: a graph with the two collectives inside", flush=True)
    t0 = time.perf_counter()
    for i in range(100):
        inp.add_(1)
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    want = (inp * 2) + 1
    print("replayed 100 times: %.1f us per replay, result %s" % (dt * 1e6, "right" if torch.equal(res, want) else "WRONG"), flush=True)
except Exception as exc:                                  # noqa: BLE001
    print("capture / replay refused: %s: %s" % (type(exc).__name__, str(exc)[:400]), flush=True)
dist.destroy_process_group()
