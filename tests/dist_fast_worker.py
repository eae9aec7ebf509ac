"""Worker for tests/test_dist_cpu.py::test_device_sized_exchanges_*: one rank of a gloo group running the hash-partitioned join of
sdqlpy_amd/dist.py again and again — its first run exchanges exact sizes through the host, later runs device-sized chunks with
nothing waited for (DistributedRunner._hash_join_device_sized) — and forcing each thing that can go wrong with a bound:
  * chunks too small for what arrives (every rank learns it from the all-reduced status and the join is repeated collectively),
  * a K-F result block too small on every rank / on ONE rank only (K-F is repeated on that rank's kept tables: no collective),
and the range-partitioned form of the same (the engine's deferred run with the runner's seams)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(rank, world, port, sf, out_path):
    import torch.distributed as dist
    from sdqlpy_amd import abi, engine, tpch
    from sdqlpy_amd import dist as sdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    lib = abi.Library(os.path.join(ROOT, "oracle", "libsdqloracle.so"))
    eng = engine.Engine(lib.context(threads=2))
    eng.deferred_results = True                       # (on by default on the GPU: the plan's last call launched, collected later)
    cols = tpch.columns_for(["q3"])
    db = tpch.generate(sf, tables=sorted(cols), columns=cols, threads=2, shard=(rank, world))
    out = {"rank": rank}

    def states(runner):
        from sdqlpy_amd import tpch_queries as Q
        fn, plan = runner._plans[id(Q.QUERIES["q3"])]
        return [st for k, st in plan.__dict__.get("_dist_prepared", {}).items() if k[0] == id(runner)]

    def shrink_result_blocks():
        for k in list(eng.compact_hints):
            eng.compact_hints[k] = 0

    for mode in ("hash", "range"):
        runner = sdist.DistributedRunner(eng, rank, world, partition={"hash": "hash", "range": "auto"}[mode])
        log = []

        def once(label):
            r = runner.run("q3", db)
            deferred = type(r).__name__ == "DeferredResultSet"
            rows = runner.gather_rows(r)
            log.append({"label": label, "rows": rows, "local_rows": r.size(), "deferred": deferred, "fast_runs": runner.fast_runs, "retries": runner.fast_retries,
                        "exchanged": dict(runner.exchanged_rows), "partitioning": runner.last_partitioning})
        once("first")                                  # exact sizes through the host; learns the bounds (and K-F's block size)
        once("second")                                 # hash: device-sized chunks
        once("third")
        if mode == "hash":
            for st in states(runner):
                st.caps = (4, 2)                       # bounds far too small on every rank
            once("chunks too small")
            once("after the collective re-run")
        shrink_result_blocks()
        once("result block too small on every rank")
        if rank == 0:
            shrink_result_blocks()
        once("result block too small on rank 0 only")
        once("last")
        out[mode] = log
        runner.close()
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump(out, fh)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), sys.argv[5])
