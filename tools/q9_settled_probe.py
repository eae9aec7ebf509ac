import os, sys, time, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, torch.distributed as dist
from sdqlpy_amd import abi, engine, tpch, dist as sdist
import helpers
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1, device_id=torch.device("cuda", 0))
lib = engine.load_hip_library()
eng = engine.Engine(lib.context(device=0))
runner = sdist.DistributedRunner(eng, 0, 1, skip_trivial=False)
qs = ("q9",)
cols = tpch.columns_for(qs)
db = tpch.generate(10.0, tables=sorted(cols), columns=cols, shard=(0, 1))
want = helpers.run_query(eng, "q9", db)
want = want.wait() if hasattr(want, "wait") else want
if os.environ.get("PROBE_STATUS") == "1":
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    def status(what):
        out = []
        for k, lc in enumerate([eng.ctx] + list(eng.ctx.forks)):
            cs = C.c_int(-1); e = hip.hipStreamIsCapturing(C.c_void_p(int(lc.stream())), C.byref(cs)); out.append((e, cs.value))
        print("   [status] %-28s %s" % (what, out), flush=True)
    def wrap(obj, name):
        fn = getattr(obj, name)
        def w(*a, **k):
            r = fn(*a, **k); status(name); return r
        setattr(obj, name, w)
    import sdqlpy_amd.abi as A
    for nm in ("graph_begin", "table_partition_pack", "unpack_chunks", "unpack2", "build", "build_marshalled"):
        wrap(A.Context, nm)
    wrap(dist, "all_gather_into_tensor")
for i in range(6):
    runner.reset_collectives(); runner.last_chain = None
    torch.cuda.synchronize(); t = time.perf_counter()
    got = runner.run("q9", db)
    got = got.wait() if hasattr(got, "wait") else got
    dt = time.perf_counter() - t
    print(i, "%.3f ms" % (dt * 1e3), dict(runner.last_chain or {}), {k: v[:2] for k, v in runner.collectives.items()}, flush=True)
    a = sorted(map(tuple, got.rows())); b = sorted(map(tuple, want.rows()))
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x[:-1] == y[:-1] and abs(x[-1] - y[-1]) <= 1e-9 * abs(y[-1]), (x, y)
N = 200
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(N):
    r = runner.run("q9", db); r = r.wait() if hasattr(r, "wait") else r
print("steady %.3f ms" % ((time.perf_counter() - t) / N * 1e3), runner.fast_runs, runner.fast_retries, runner.graph_recordings, runner.graph_launches)
runner.close(); dist.destroy_process_group(); eng.close()
