"""Worker for tests/test_hip_parity.py::test_two_ranks_share_one_gpu: one rank of a GLOO process group driving the HIP library on
device 0 — the multi-GPU runner's hybrid mode (sdqlpy_amd/dist.py): collective buffers in pinned host memory, which the library's
kernels read and write through their device-visible addresses and gloo moves between the processes.  RCCL refuses two ranks on one
device; this is how the N > 1 kernels (multi-part packs, chunks of several sources, folds of several blocks) run on the hardware."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def result_digest(res):
    """A large result as a few numbers that add up over key partitions: rows, sums of the integer columns, sums of the float columns."""
    import numpy as np
    out = [float(res.size())]
    for c in res.columns:
        a = np.asarray(res.column(c))
        out.append(float(a.astype(np.float64).sum()) if a.dtype.kind in "if" else 0.0)
    return out


def main(rank, world, port, sf, out_path, qs=None, parts=None, digest=False, shuffled=False):
    import numpy as np
    import torch.distributed as dist
    from sdqlpy_amd import engine, tpch
    from sdqlpy_amd import dist as sdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    eng = engine.Engine(engine.load_hip_library().context(device=0))
    qs = qs or ["q6", "q1", "q5", "q9", "q4", "q14", "q3"]
    cols = tpch.columns_for(qs)
    if shuffled:
        # the rows of every table dealt to the ranks at random: key ranges overlap, nothing is co-partitioned — every build that another
        # table's rows look up is replicated, the join is hash-partitioned
        full = tpch.generate(sf, tables=sorted(cols), columns=cols, threads=4)
        rng = np.random.default_rng(5)
        db = {}
        for t in sorted(full):
            if t in ("region", "nation"):
                db[t] = full[t]
                continue
            c = full[t].getContainer()
            n = len(c["data"][0])
            mine = np.sort(rng.permutation(n)[n * rank // world: n * (rank + 1) // world])
            db[t] = tpch.table_from_columns(c["headers"], [np.ascontiguousarray(a[mine]) for a in c["data"]], shard=(rank, world))
        del full
    else:
        db = tpch.generate(sf, tables=sorted(cols), columns=cols, threads=4, shard=(rank, world))
    out = {"rank": rank, "runs": {}}
    for part in (parts or ("hash", "auto")):
        runner = sdist.DistributedRunner(eng, rank, world, partition=part)
        assert runner.hybrid and runner.backend == "gloo"
        for q in qs:
            for again in range(3):                                  # (first run: exact sizes; later runs: device-sized exchanges, folded groups)
                runner.last_chain = None
                tag = "%s/%s/%d" % (part, q, again)
                try:
                    r = runner.run(q, db)
                except Exception as exc:                            # (recorded, every rank alike: the test says which refusals it expects)
                    from sdqlpy_amd import frontend
                    if not isinstance(exc, frontend.UnsupportedQuery):
                        raise
                    out["runs"][tag] = {"unsupported": str(exc)}
                    continue
                if isinstance(r, float):
                    out["runs"][tag] = {"scalar": r}
                    continue
                if q == "q3" and digest:
                    # (a million rows per run: this rank's key partition as a digest, the ranks' digests added)
                    r = r.wait() if hasattr(r, "wait") else r
                    parts_d = runner._all_gather_array(np.array(result_digest(r), np.float64))
                    out["runs"][tag] = {"columns": r.columns, "digest": [float(x) for x in np.sum(parts_d, axis=0)], "local_rows": r.size(),
                                        "partitioning": runner.last_partitioning, "exchanged": dict(runner.exchanged_rows)}
                    continue
                rows = runner.gather_rows(r) if q == "q3" else sorted(r.rows())
                out["runs"][tag] = {"columns": r.columns, "rows": rows, "seams": dict(runner.last_chain or {}),
                                    "partitioning": runner.last_partitioning if q == "q3" else None,
                                    "exchanged": dict(runner.exchanged_rows) if q == "q3" else None}
        out[part] = {"fast_runs": runner.fast_runs, "fast_retries": runner.fast_retries,
                     "collectives": {k: v[:2] for k, v in runner.collectives.items()}}
        runner.close()
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump(out, fh)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), sys.argv[5],
         qs=sys.argv[6].split(",") if len(sys.argv) > 6 and sys.argv[6] else None,
         parts=sys.argv[7].split(",") if len(sys.argv) > 7 and sys.argv[7] else None,
         digest=len(sys.argv) > 8 and sys.argv[8] == "digest", shuffled=len(sys.argv) > 9 and sys.argv[9] == "shuffled")
