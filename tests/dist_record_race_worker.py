"""Worker for test_recordings_survive_the_collective_watchdog (tests/test_hip_parity.py): an RCCL group of one, the settled chains of q1 and
q5 and the partitioned join of q3 recorded with their collectives inside — every recording HELD OPEN for 0.35 s, so that torch's
collective watchdog (a poll every 100 ms) certainly runs while the stream is in capture mode.  On this runtime a query of an event that
was recorded on the stream before its capture began breaks the capture and the stream for good (tools/exp/capture_abort.hip case 4); the
runner keeps every eager collective off the recorded streams (dist.DistributedRunner._coll).  Exit code 0 and "ok" = recorded, replayed,
rows equal to the single-GPU plan's."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(port, sf):
    import torch
    import torch.distributed as dist
    import helpers
    from sdqlpy_amd import abi, engine, tpch
    from sdqlpy_amd import dist as sdist
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    lib = engine.load_hip_library()
    eng = engine.Engine(lib.context(device=0))
    runner = sdist.DistributedRunner(eng, 0, 1, skip_trivial=False)
    end = abi.Context.graph_end

    def slow_end(self):
        time.sleep(0.35)
        return end(self)
    abi.Context.graph_end = slow_end
    qs = ("q1", "q5", "q3")
    cols = tpch.columns_for(qs)
    db = tpch.generate(sf, tables=sorted(cols), columns=cols, shard=(0, 1))
    want = {}
    for q in qs:
        r = helpers.run_query(eng, q, db)
        want[q] = r.wait() if hasattr(r, "wait") else r
    for again in range(7):
        for q in qs:
            got = runner.run(q, db)
            got = got.wait() if hasattr(got, "wait") else got
            a, b = sorted(map(tuple, got.rows())), sorted(map(tuple, want[q].rows()))
            assert len(a) == len(b), (q, again, len(a), len(b))
            for x, y in zip(a, b):
                for u, v in zip(x, y):
                    assert u == v or (isinstance(u, float) and abs(u - v) <= 1e-9 * max(1.0, abs(v))), (q, again, x, y)
    assert runner.graph_recordings >= 2 and runner.graph_launches >= 6, (runner.graph_recordings, runner.graph_launches)
    print("ok recordings=%d launches=%d" % (runner.graph_recordings, runner.graph_launches), flush=True)
    runner.close()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main(int(sys.argv[1]), float(sys.argv[2]))
