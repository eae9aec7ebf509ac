#!/usr/bin/env python3
"""Does torch's collective watchdog break a stream capture that follows eager collectives on the same stream?  (An experiment, not
product code.)  On this runtime hipEventQuery of an event recorded on a stream BEFORE its capture began invalidates the capture for good
(tools/exp/capture_abort.hip, case 4) — and the watchdog polls the end events of eager collectives every 100 ms until they are reaped.
The race is forced here: eager collectives on stream S, then a capture on S held open for 0.4 s.
    timeout 120 python tools/exp_watchdog_capture.py same|side|drain"""
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29593")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else "same"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
inp = torch.arange(4096, dtype=torch.int64, device=dev)
out = torch.zeros(4096, dtype=torch.int64, device=dev)
s = torch.cuda.Stream()
side = torch.cuda.Stream()
dist.all_gather_into_tensor(out, inp)
torch.cuda.synchronize()
time.sleep(0.5)                                            # (the warm-up's work is reaped)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        if mode == "side":                                 # eager collectives on a stream of their own, forked from / joined to S
            side.wait_stream(s)
            with torch.cuda.stream(side):
                dist.all_gather_into_tensor(out, inp)
            s.wait_stream(side)
        else:
            dist.all_gather_into_tensor(out, inp)
    if mode == "drain":
        torch.cuda.synchronize()
        time.sleep(0.3)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
            tmp = inp * 2
            dist.all_gather_into_tensor(out, tmp)
            time.sleep(0.4)                                # (the watchdog wakes at least three times in here)
            res = out + 1
        torch.cuda.synchronize()
        inp.add_(1)
        g.replay()
        torch.cuda.synchronize()
        print("%s: captured and replayed, result %s" % (mode, "right" if torch.equal(res, inp * 2 + 1) else "WRONG"), flush=True)
    except Exception as exc:                               # noqa: BLE001
        print("%s: capture failed: %s: %s" % (mode, type(exc).__name__, str(exc)[:300]), flush=True)
dist.destroy_process_group()
