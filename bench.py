#!/usr/bin/env python3
"""bench.py — TPCH Q1 + Q3 + Q5 at SF=10 per GPU on MI355X (BASELINE.json metric: rows/sec + ms/query).

A step = one pass of the hot path over the resident tables: q1(lineitem), q3(customer, orders,
lineitem), q5(six tables), through the public decorator API (front end -> planner -> C ABI -> HIP
kernels), result materialised on the host.  Inputs are synthetic (sdqlpy_amd/tpch.py, seed fixed)
and already resident in HBM when the timed region starts.  rows/sec = rows of every table scanned
by the step / step time.  `--extra-queries` (q6, q9: BASELINE configs[0] and [4]) are measured in the
same run with the same protocol after the timed region; they are reported per query and are not part
of `value`.

    python bench.py                       # 1 GPU, defaults finish in a few minutes
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N              # no launcher: bench.py starts its N ranks itself (spawn_ranks) and relays rank 0's line
    ... bench.py --gpus 8 --global-sf 100     # BASELINE configs[3] / [4]: SF=100 over the 8 GPUs

N > 1 is weak scaling by default (every rank holds an SF=10 shard of a global SF=10*N database); q1
shards by rows, q5 runs the chain plan, q3 is the partitioned join of sdqlpy_amd/dist.py.  The timed
step HASH-partitions q3 on o_orderkey — build survivors and filtered probe rows really travel through
the RCCL all-to-all — because that is the redistribution step the metric names; the range shortcut
(dbgen-clustered shards: nothing moves) is timed beside it and reported, not counted in `value`.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROW_RESULTS = ("q3", "q10", "q18")    # queries whose result is a set of rows (K-F + a device-to-host copy) rather than a handful of groups
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sf", type=float, default=10.0, help="scale factor PER GPU")
    ap.add_argument("--global-sf", type=float, default=0.0, help="scale factor of the WHOLE database, cut over the ranks (overrides --sf; "
                                                                 "--gpus 8 --global-sf 100 = BASELINE configs[3]/[4])")
    ap.add_argument("--queries", default="q1,q3,q5", help="the timed step (BASELINE metric: Q1/Q3/Q5)")
    ap.add_argument("--launch-order", default="given", help="order in which a step launches its queries: 'given' (as --queries), 'rows-first' (the queries whose "
                    "result is a set of rows — K-F, a device-to-host copy — before those that end in a handful of groups: the copy then runs beside the "
                    "next query's big kernel instead of its one-workgroup builds; measured 2 %% faster at SF=10, 20 %% at SF=100, not the default: under "
                    "rocprofv3 the overlapped kernel is the dominant one and its traced duration no longer is what the events of an unprofiled run "
                    "measure), or a comma-separated permutation of --queries")
    ap.add_argument("--extra-queries", default=None, help="measured after the timed region, reported per query only (default q6,q9 at N=1, none at N>1)")
    ap.add_argument("--profile-iters", type=int, default=5, help="extra untimed passes with per-kernel HIP events")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=0, help="engine lanes — contexts of one family, a stream each; a plan is bound to one (0: the engine's default, 3)")
    ap.add_argument("--force-dist", action="store_true", help="use the distributed plan even with one rank (exercises the RCCL path)")
    ap.add_argument("--trivial-collectives", action="store_true", help="with --force-dist: a group of ONE still issues every collective (the RCCL calls themselves on one GPU); "
                    "by default collectives over a group of one — which move nothing — are skipped, as every rank of such a group knows to")
    ap.add_argument("--share-gpu", action="store_true", help="with --gpus N: the N ranks SHARE device 0 (gloo between the processes, collective buffers in pinned host memory: "
                    "dist.DistributedRunner's hybrid mode) — the N > 1 plans and kernels on a box with one GPU; a functional run, not a scaling figure")
    ap.add_argument("--partition", default="hash", choices=["auto", "range", "hash"], help="q3's partitioning in the timed step at N > 1")
    ap.add_argument("--cpu-sample-sf", type=float, default=0.0, help="0 = pick so the CPU leg takes ~10-30 s")
    ap.add_argument("--steady-steps", type=int, default=1500, help="steps of a second, longer leg after the timed region (reported as `steady_state`, not part of `value`); 0 = none")
    ap.add_argument("--no-hash-path", action="store_true", help="skip the leg that prices the open-addressing layouts (direct layouts off; Q3 on keys beyond 2^40 in shuffled rows)")
    ap.add_argument("--no-scan-form", action="store_true", help="skip the leg that re-runs the step with every loop streaming every row (no driven walk, no clustered pack, no delta twins)")
    ap.add_argument("--no-reference-width", action="store_true", help="skip the leg that re-runs q1/q3/q6 on the reference's 8-byte columns (twins and codes off)")
    return ap.parse_args(argv)


def algorithmic_bytes(q, rows):
    """every referenced column read once at the reference's own widths."""
    if q == "q1":
        return 48 * rows["lineitem"]
    if q == "q3":
        return 48 * rows["customer"] + 32 * rows["orders"] + 32 * rows["lineitem"]
    if q == "q6":
        return 32 * rows["lineitem"]
    if q == "q5":
        return 16 * rows["customer"] + 24 * rows["orders"] + 16 * rows["supplier"] + 32 * rows["lineitem"]
    if q == "q9":
        return 228 * rows["part"] + 24 * rows["partsupp"] + 16 * rows["orders"] + 16 * rows["supplier"] + 48 * rows["lineitem"]
    raise KeyError(q)


def scanned_rows(q, rows):
    return {"q1": lambda: rows["lineitem"], "q6": lambda: rows["lineitem"],
            "q3": lambda: rows["lineitem"] + rows["customer"] + rows["orders"],
            "q5": lambda: rows["lineitem"] + rows["customer"] + rows["orders"] + rows["supplier"] + rows["nation"] + rows["region"],
            "q9": lambda: rows["lineitem"] + rows["orders"] + rows["part"] + rows["partsupp"] + rows["supplier"] + rows["nation"]}[q]()


# the kernel that streams the big table of each query, and the algorithmic bytes one launch of it covers
# (fallback name of the kernel that streams the big table of each query, and the algorithmic bytes one launch of it covers)
DOMINANT = {"q1": ("xk_group_lane_tight", lambda r: 48 * r["lineitem"]), "q3": ("k_probe_agg", lambda r: 32 * r["lineitem"]),
            "q6": ("xk_sum_tight", lambda r: 32 * r["lineitem"]), "q5": ("k_lookup_agg", lambda r: 32 * r["lineitem"]),
            "q9": ("k_lookup_agg", lambda r: 48 * r["lineitem"])}


def _context_launches(c):
    """[(kernel, ms, modelled bytes)] recorded on ONE context (not its forks)."""
    import ctypes as C
    out = []
    for i, (name, ms) in enumerate(c.profile(family=False)):
        b = C.c_int64()
        nbytes = 0
        if c.lib.sdqh_profile_entry_bytes(c.handle, C.c_int(i), C.byref(b)) == 0:
            nbytes = int(b.value)
        out.append((name, ms, nbytes))
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very program as child processes (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, exactly what torch.distributed.run would set), wait for them, relay
    rank 0's single JSON line, and return non-zero if any rank failed (the others are then stopped: a rank waiting in a collective
    for a dead peer would hang).  The parent never initialises a GPU.  SDQLPY_AMD_BENCH_CHILD names another program to run as a
    rank (tests/: the same control flow on gloo with the CPU implementation injected)."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = str(sock.getsockname()[1])
    child = os.environ.get("SDQLPY_AMD_BENCH_CHILD") or os.path.abspath(__file__)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs between processes on this driver
        env.pop("SDQLPY_AMD_BENCH_CHILD", None)
        procs.append(subprocess.Popen([sys.executable, child] + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None))
    import threading
    chunks, failed = [], 0
    # rank 0 prints one line at its end; its pipe is drained by a thread while the ranks run (the loop below must stay free to
    # notice a rank that died: its peers would wait for it in a collective for ever)
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    try:
        pending = list(enumerate(procs))
        while pending:
            for r, p in list(pending):
                try:
                    rc = p.wait(timeout=0.2)
                except subprocess.TimeoutExpired:
                    continue
                pending.remove((r, p))
                if rc != 0:
                    failed = failed or rc or 1
                    sys.stderr.write("bench.py: rank %d exited with code %d\n" % (r, rc))
            if failed and pending:
                for _, p in pending:                              # exactly the processes started here
                    p.terminate()
                for _, p in pending:
                    try:
                        p.wait(timeout=30)
                    except subprocess.TimeoutExpired:
                        p.kill()
                pending = []
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if failed:
        return failed
    reader.join(timeout=30)
    out = [ln for ln in b"".join(chunks).decode(errors="replace").splitlines() if ln.strip()]
    if len(out) != 1:
        sys.stderr.write("bench.py: rank 0 printed %d lines, expected one JSON line\n" % len(out))
        return 1
    sys.stdout.write(out[0] + "\n")
    sys.stdout.flush()
    return 0


def main(argv=None, hooks=None):
    """hooks: injection points for the CPU test of the multi-process control flow
    (tests/bench_gloo_worker.py): {"backend": "gloo", "device": "cpu", "engine": Engine}.  The
    product run passes none: RCCL, cuda:LOCAL_RANK, the HIP engine."""
    args = parse(argv)
    hooks = hooks or {}
    device = hooks.get("device", "cuda")
    # RCCL prints a version banner on stdout when a communicator is created; the contract is ONE
    # JSON line on stdout, so everything else (including C-level writes) is sent to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    queries = [q for q in args.queries.split(",") if q]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    extra = args.extra_queries if args.extra_queries is not None else ("q6,q9" if world == 1 else "")
    extra = [q for q in extra.split(",") if q and q not in queries]
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local_rank = 0                                   # every rank on device 0 (the runner's hybrid mode: gloo + pinned host buffers)
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and not hooks:
            # plain `python bench.py --gpus N`: this process becomes the launcher — N ranks started as child processes BEFORE anything
            # here touches a GPU (no exec from a process that has), rank 0's JSON line relayed
            os.dup2(json_fd, 1)
            os.close(json_fd)
            raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv)))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one process per GPU (or none: bench.py starts them itself)" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    if device == "cuda":
        torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    coll_device = "cpu" if args.share_gpu else device    # (the bench's own small collectives: gloo takes host tensors)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if device == "cuda" and not args.share_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        elif args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group(hooks.get("backend", "gloo"))

    if args.lanes > 0:
        os.environ["SDQLPY_AMD_LANES"] = str(args.lanes)
    from sdqlpy_amd import engine, tpch
    from sdqlpy_amd import tpch_queries as Q
    from sdqlpy_amd.sdql_lib import sdqlpy_init

    t0 = time.time()
    need = tpch.columns_for(queries + extra)
    tables = sorted(need)
    # the distributed plan runs on tables that SAY they are row shards (dist.DistributedRunner._whole_params):
    # also on a group of one, where the shard is the whole table and every collective still executes
    shard = (rank, world) if use_dist else None
    sf_global = args.global_sf if args.global_sf > 0 else args.sf * world
    sf_per_gpu = sf_global / world
    db = tpch.generate(sf_global, tables=tables, columns=need, shard=shard)
    rows = {t: len(db[t].getContainer()["data"][0]) for t in tables}
    gen_s = time.time() - t0

    if "engine" in hooks:
        eng = engine.use_engine(hooks["engine"])
    else:
        sdqlpy_init(3, 1, device=local_rank)
        eng = engine.default_engine(device=local_rank)
    runner = runner_range = None
    if use_dist:
        from sdqlpy_amd import dist as sdist
        trivial = None if world > 1 else (not args.trivial_collectives)
        runner = sdist.DistributedRunner(eng, rank, world, partition=args.partition, skip_trivial=trivial)
        runner_range = sdist.DistributedRunner(eng, rank, world, partition="auto", skip_trivial=trivial)      # the clustered-shard shortcut, timed beside
        run_query = lambda q: runner.run(q, db)           # noqa: E731
    else:
        run_query = lambda q: Q.run(q, db)                # noqa: E731

    def barrier():
        if use_dist:
            dist.barrier()
        eng.ctx.synchronize()
        if device == "cuda":
            torch.cuda.synchronize()

    # first pass uploads the columns (pinned-staged H2D) — the PCIe-inclusive number
    barrier()
    t0 = time.time()
    first_run_ms, first_run_detail = {}, {}

    def first_run(q):
        """A query's first run, every launch of it between HIP events: what it pays and later runs do not — pinned-staged H2D of the
        columns no earlier query uploaded, narrow / byte twins, dictionaries, delta twins, run indexes, row packs (in row order or
        clustered), min / max and ordering facts, plan lowering, hiprtc — and the HBM it leaves resident beside the columns."""
        def in_use():
            if device != "cuda":
                return 0
            free_b, total_b = torch.cuda.mem_get_info(local_rank)
            return total_b - free_b
        use0, res0 = in_use(), int(eng.resident_bytes)
        prof = device == "cuda" and not use_dist
        if prof:
            eng.ctx.set_profiling(2, only=None)
        tq = time.perf_counter()
        r = run_query(q)
        r.wait() if hasattr(r, "wait") else None
        eng.ctx.synchronize()
        first_run_ms[q] = round((time.perf_counter() - tq) * 1e3, 2)
        launches = eng.ctx.profile() if prof else []
        if prof:
            eng.ctx.set_profiling(0)
        per = {}
        for name, ms in launches:
            per[name] = per.get(name, 0.0) + ms
        first_run_detail[q] = {"kernels_ms": per, "hbm_added": in_use() - use0, "columns_added": int(eng.resident_bytes) - res0}

    for q in queries:
        first_run(q)
    barrier()
    first_pass_s = time.time() - t0
    uploaded_bytes = int(eng.resident_bytes)              # host columns copied to HBM by that pass (pinned-staged H2D)

    # W warm-up steps of the timed kind (launched, then finished — result blocks get their sizes here) — at least three untimed steps in
    # all, whatever W: a plan settles in its first two runs and may be RECORDED in its third (engine.PlanGraph: milliseconds, once)
    for _ in range(max(args.warmup, 3)):
        for r in [run_query(q) for q in queries]:
            r.wait() if hasattr(r, "wait") else None

    # Timed region.  HIP events are recorded on the stream the kernels run on, around every launch
    # of the dominant kernel (profiling mode 2: record only, nothing synchronises; events around
    # all ~45 launches of a step would add ~0.1 ms of event packets per query); read afterwards.
    # the step's DOMINANT kernel: the longest single launch over ALL queries of the step, found by one fully evented run of each query
    # before the timed region (its name depends on the route the planner took: a kernel specialised at run time on the loop — xk_* —
    # or a fixed-shape one — k_*).  Q1's streaming kernel — the north-star's "Q1 at >= 60 % of the roofline" — is priced the same way
    # beside it as `roofline.q1` when it is not the dominant one.
    longest = {}
    for q in queries:
        eng.ctx.set_profiling(2, only=None)
        r = run_query(q)
        r.wait() if hasattr(r, "wait") else None
        barrier()
        launches_q = eng.ctx.profile()
        eng.ctx.set_profiling(0)
        longest[q] = max(launches_q, key=lambda kv: kv[1]) if launches_q else (DOMINANT[q][0], 0.0)
    dom_q = max(queries, key=lambda q: longest[q][1])
    dom_kernel = longest[dom_q][0]

    def step_order(qs):
        if args.launch_order == "given":
            return list(qs)
        if args.launch_order == "rows-first":
            rows_out = [q for q in qs if q in ROW_RESULTS]
            return rows_out + [q for q in qs if q not in rows_out]
        want = [q for q in args.launch_order.split(",") if q]
        return want if sorted(want) == sorted(qs) else list(qs)

    def finish(r):
        """A result is complete — every row on the host, the plan's host-side steps done — once wait() returns (results of plans
        whose last device call is launched without being waited for: sdqlpy_amd/result.py DeferredResultSet)."""
        return r.wait() if hasattr(r, "wait") else r

    def run_steps(nsteps, only, qs=None, run=None, each_waited_for=False):
        """One step = every query of `qs` LAUNCHED, then every result finished (a step ends with all its results complete on the
        host).  each_waited_for: every query's result finished before the next query is launched (per-query wall times)."""
        qs = queries if qs is None else qs
        run = run or run_query
        per_q = {q: 0.0 for q in qs}
        # only = "-": nothing is evented — profiling stays off and settled plans are launched as recorded graphs (engine.PlanGraph: one
        # call per query); a kernel name / None: that kernel's / every launch is evented, which needs the calls issued one by one
        if only != "-":
            eng.ctx.set_profiling(2, only=only)
        barrier()
        t_begin = time.perf_counter()
        marks = []                                       # (query, launches recorded so far on every context of the engine's family)
        family = [eng.ctx] + list(getattr(eng.ctx, "forks", []))
        order = step_order(qs) if not each_waited_for else qs
        for _ in range(nsteps):
            results = []
            for q in order:
                tq = time.perf_counter()
                r = run(q)
                if each_waited_for:
                    finish(r)
                else:
                    results.append(r)
                per_q[q] += (time.perf_counter() - tq) * 1e3
                marks.append((q, [c.lib.sdqh_profile_count(c.handle) for c in family]))
            for r in results:
                finish(r)
        barrier()
        took = time.perf_counter() - t_begin
        # [(kernel, ms, modelled HBM bytes)] of every recorded launch (sdqh_profile_entry_bytes: what the library's own choice of
        # encodings makes that launch stream; 0 where a kernel has no model)
        # (a plan runs on one lane of the engine — a context with a stream of its own: the launches are read per context)
        launches = [_context_launches(c) for c in family]
        eng.ctx.set_profiling(0)
        log, at = [], [0] * len(family)
        for q, upto in marks:
            for i, n in enumerate(upto[:len(launches)]):
                log += [(q, name, ms, nbytes) for name, ms, nbytes in launches[i][at[i]:n]]
                at[i] = max(at[i], n)
        return took, per_q, log

    if runner is not None:
        runner.reset_collectives()
    # THE TIMED REGION: args.steps steps, nothing evented — settled plans are launched as recorded graphs (one call per query) where the
    # engine records them (single-GPU plans; SDQLPY_AMD_PLAN_GRAPHS=0 switches that off)
    elapsed, launch_host_ms, _ = run_steps(args.steps, "-")
    graph_stats = dict(getattr(eng, "graph_stats", {}) or {})
    # the same steps once more with the dominant kernel's launches evented (HIP events on the stream it runs on): evented launches are
    # issued call by call — a recorded graph has no place for an event pair — so this region also says what the step costs without graphs
    elapsed_calls, launch_host_calls_ms, timed_log = run_steps(args.steps, dom_kernel)
    # the same steps with every query's result finished before the next query starts: per-query wall times, and the step as a caller
    # who reads each result at once sees it; then evented (in that region a kernel has the chip to itself — in the overlapped steps the
    # queries' kernels share it, lane by lane, and a kernel's duration there is not a statement about the kernel)
    run_steps(max(args.warmup, 3), "-", each_waited_for=True)      # (untimed: a query that is waited for alone is launched as a recorded plan — recorded here, not in the timed steps)
    elapsed_waited, per_query_ms, _ = run_steps(args.steps, "-", each_waited_for=True)
    _, _, waited_log = run_steps(args.steps, dom_kernel, each_waited_for=True)
    q1_log = None
    if "q1" in queries and dom_q != "q1":
        # Q1's streaming kernel evented in steps of its own (the filter takes one kernel name), outside `value`
        _, _, q1_log = run_steps(args.steps, longest["q1"][0], ["q1"], each_waited_for=True)
    timed_collectives = None
    if runner is not None:
        timed_collectives = {k: {"calls": v[0], "on_device_tensors": v[1], "bytes": v[2]} for k, v in sorted(runner.collectives.items())}
        if "q3" in queries:
            # a "distributed" step that took the single-GPU plan measures nothing (round 2's world-1 profile did)
            assert runner.last_partitioning == args.partition or (args.partition == "auto" and runner.last_partitioning in ("range", "hash")), \
                "q3 did not run the partitioned join (partitioning %r)" % (runner.last_partitioning,)
            assert runner.collectives.get("all_to_all", [0])[0] > 0 or runner.last_partitioning == "range" or (world == 1 and runner.skip_trivial), "no all-to-all ran in the timed step"
            if args.partition == "hash":
                assert runner.exchanged_rows.get("probe_sent", 0) > 0, runner.exchanged_rows
    nlanes = int(getattr(eng, "nlanes", 1))
    shared_launches = [ms for q, name, ms, _ in timed_log if q == dom_q and name == dom_kernel]
    alone_launches = [ms for q, name, ms, _ in waited_log if q == dom_q and name == dom_kernel]
    # one lane: the queries of a step run one behind the other on one stream, and the first region's launches are as alone as the second's
    dom_launches = alone_launches if (nlanes > 1 and alone_launches) else shared_launches
    dom_model = [nb for q, name, _, nb in (waited_log if dom_launches is alone_launches else timed_log) if q == dom_q and name == dom_kernel and nb > 0]
    # per-kernel table: a separate pass with events around every launch, after the timed region
    profile_steps = max(1, min(args.steps, 10))
    _, _, launch_log = run_steps(profile_steps, None, each_waited_for=True)      # (one query at a time: a kernel's duration is its own)
    exchange = None
    if use_dist and "q3" in queries:
        # q3's redistribution step, both ways, outside `value`: what moved and what it cost
        exchange = {}
        for label, rn in ((args.partition, runner), ("auto", runner_range)):
            rn.run("q3", db)
            took, _, _ = run_steps(max(1, min(args.steps, 10)), "-", ["q3"], lambda q, rn=rn: rn.run(q, db))
            ms = took / max(1, min(args.steps, 10)) * 1e3
            if world > 1:
                t = torch.tensor([ms], dtype=torch.float64, device=coll_device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms = float(t.item())
                moved = torch.tensor([rn.exchanged_bytes], dtype=torch.float64, device=coll_device)
                dist.all_reduce(moved)
                total_bytes = float(moved.item())
            else:
                total_bytes = float(rn.exchanged_bytes)
            exchange[label if label != "auto" else "auto(" + str(rn.last_partitioning) + ")"] = {
                "ms_q3": round(ms, 4), "partitioning": rn.last_partitioning, "exchanged_rows_rank0": rn.exchanged_rows,
                "exchanged_bytes_all_ranks": int(total_bytes),
                "all_to_all_GBs_all_ranks": round(total_bytes / (ms * 1e-3) / 1e9, 1) if ms > 0 else None}
    extra_ms, extra_log, extra_steps = {}, [], max(1, min(args.steps, 10))
    if extra:
        for q in extra:                                   # upload + warm-up, then the same protocol per query
            first_run(q)
            for _ in range(args.warmup):
                run_query(q)
        run_steps(3, "-", extra, each_waited_for=True)       # (untimed: plans settle / are recorded here)
        took_x, extra_ms, _ = run_steps(extra_steps, "-", extra, each_waited_for=True)
        _, _, extra_log = run_steps(extra_steps, None, extra, each_waited_for=True)
    # the timed region is args.steps steps (the driver fixes 20: 15 ms); the same step for ~1 s of back-to-back work, outside `value`:
    # is the figure steady state or a burst out of warm caches?
    steady = None
    if world == 1 and args.steady_steps > 0:
        took_long, _, _ = run_steps(args.steady_steps, "-")
        steady = {"steps": args.steady_steps, "ms_per_step": round(took_long / args.steady_steps * 1e3, 4), "seconds": round(took_long, 3)}
    # one finished result per query of the step, kept for the comparison with the CPU implementation's results on the same tables
    hip_results = {q: finish(run_query(q)) for q in queries} if (world == 1 and not args.no_cpu_baseline) else {}
    hbm = None
    try:                                                       # what the process holds of the GPU's HBM behind the timed region: columns at the reference's widths,
        import torch                                           # their twins and dictionaries, the pools' table memory, recorded plans, the runtime's own
        free_b, total_b = torch.cuda.mem_get_info(local_rank if use_dist else 0)
        hbm = {"in_use_GB": round((total_b - free_b) / 1e9, 2), "of_GB": round(total_b / 1e9, 1), "columns_at_reference_width_GB": round(int(eng.resident_bytes) / 1e9, 2)}
        try:
            # where it is (sdqh_memory_stats over the engine's lanes): what the pools have handed out — the columns, everything attached to them
            # (twins, codes, dictionaries, delta twins, run indexes, row packs) and tables still alive —, what they keep cached for the next
            # run's tables, what recorded plan graphs reserve; the rest is the runtime's (code objects, RCCL / torch, queues)
            ms = eng.ctx.memory_stats()
            cols = int(eng.resident_bytes)
            hbm["pools"] = {"handed_out_GB": round(ms["used"] / 1e9, 2), "of_which_columns_GB": round(cols / 1e9, 2),
                            "of_which_attached_to_columns_and_live_tables_GB": round(max(0, ms["used"] - cols - ms["graphs"]) / 1e9, 2),
                            "cached_free_GB": round(ms["cached_free"] / 1e9, 2), "reserved_by_recorded_plans_GB": round(ms["graphs"] / 1e9, 2), "blocks": ms["blocks"]}
            hbm["outside_the_pools_GB"] = round(((total_b - free_b) - ms["used"] - ms["cached_free"]) / 1e9, 2)
        except Exception as exc:                               # (reporting only)
            hbm["pools"] = {"error": str(exc)[:100]}
    except Exception as exc:                                   # (reporting only)
        hbm = {"error": str(exc)[:100]}
    reference_width = None
    if world == 1 and not use_dist and "engine" not in hooks and not args.no_reference_width:
        reference_width = reference_width_leg(args, eng, db, rows, queries + extra, run_query, run_steps, finish)
    scan_form = None
    if world == 1 and not use_dist and "engine" not in hooks and not args.no_scan_form:
        scan_form = scan_form_leg(args, eng, queries, queries + extra, run_query, run_steps, finish)
    hash_path = None
    if world == 1 and not use_dist and "engine" not in hooks and not args.no_hash_path:
        hash_path = hash_path_leg(args, eng, db, rows, queries + extra, run_steps, finish)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        rows_t = torch.tensor([sum(scanned_rows(q, rows) for q in queries)], dtype=torch.float64, device=coll_device)
        dist.all_reduce(rows_t)
        total_rows_per_step = float(rows_t.item())
    else:
        total_rows_per_step = float(sum(scanned_rows(q, rows) for q in queries))
    ms_per_step = elapsed / args.steps * 1e3
    value = total_rows_per_step * args.steps / elapsed

    kstat = {}
    for q, name, ms, nb in launch_log + extra_log:
        name = "%s:%s" % (q, name)
        tot, n, b = kstat.get(name, (0.0, 0, 0))
        kstat[name] = (tot + ms, n + 1, b + nb)
    kernels = {name: {"launches_per_step": n / profile_steps, "avg_launch_ms": tot / n, "ms_per_step": tot / profile_steps}
               for name, (tot, n, _) in kstat.items()}
    for name, (tot, n, b) in kstat.items():
        if b:                                            # streamed bytes by the library's own model, and the rate they imply
            kernels[name]["model_bytes_per_launch"] = b / n
            kernels[name]["model_GBs"] = b / n / (tot / n * 1e-3) / 1e9 if tot > 0 else None
    # kernels are attributed to the query that was running when they were launched
    device_ms = {q: sum(ms for qq, _, ms, _ in launch_log if qq == q) / profile_steps for q in queries}
    device_ms.update({q: sum(ms for qq, _, ms, _ in extra_log if qq == q) / extra_steps for q in extra})

    out = None
    if rank == 0:
        dom_name = dom_q + ":" + dom_kernel
        roofline = None
        def price(q, kernel, launches_ms, model):
            """One kernel against the HBM roofline.  `frac` is the PHYSICAL fraction: HBM bytes the kernel really moves / its average
            launch time (HIP events on the stream it runs on, inside the timed steps) / peak; the bytes are the committed PMC figure
            (`traffic`) when its row counts and kernel names match this run, else the library's own model of THIS run
            (`traffic_model`, sdqh_profile_entry_bytes: rows x the bytes per row of the encodings the engine chose + what the launch
            stores by construction), so the fraction can be recomputed from the line alone.  SURVEY.md 8(d)'s algorithmic figure (the
            reference's 8-byte / UCS-4 widths) is carried beside it as `frac_algorithmic`: the kernels read exact narrow encodings, so
            that one can exceed 1 and is no roofline fraction; `reference_width` times the same queries on the 8-byte columns."""
            ms = sum(launches_ms) / len(launches_ms)
            per_launch_bytes = DOMINANT[q][1](rows)
            achieved = per_launch_bytes / (ms * 1e-3) / 1e9
            traffic, traffic_source = pmc_traffic(q, kernel, rows)
            traffic_model = int(sum(model) / len(model)) if model else None
            bytes_used, frac_source = (traffic, "pmc") if traffic else (traffic_model, "model")
            phys_gbs = bytes_used / (ms * 1e-3) / 1e9 if bytes_used else None
            return {"bound": "hbm", "kernel": q + ":" + kernel, "achieved": round(phys_gbs, 1) if phys_gbs else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(phys_gbs / HBM_PEAK_GBS, 4) if phys_gbs else None, "frac_source": frac_source,
                    "traffic": traffic, "traffic_source": traffic_source, "traffic_stale": ("STALE" in traffic_source) if traffic_source else None,
                    "traffic_model": traffic_model, "traffic_model_over_pmc": round(traffic_model / traffic, 4) if traffic and traffic_model else None,
                    "frac_model": round(traffic_model / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic_model else None,
                    "achieved_algorithmic": round(achieved, 1), "frac_algorithmic": round(achieved / HBM_PEAK_GBS, 4),
                    "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_ms": round(ms, 4),
                    "launches_timed": len(launches_ms)}
        if dom_launches:
            roofline = price(dom_q, dom_kernel, dom_launches, dom_model)
            roofline["dominant"] = "the longest single launch over all queries of the step (one evented run of each query before the timed region): " + \
                                   ", ".join("%s %s %.4f ms" % (q, longest[q][0], longest[q][1]) for q in queries)
            roofline["achievable_peak_note"] = "MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured for a float4 copy"
            if q1_log is not None:
                k1 = longest["q1"][0]
                ms1 = [ms for q, name, ms, _ in q1_log if q == "q1" and name == k1]
                if ms1:
                    roofline["q1"] = price("q1", k1, ms1, [nb for q, name, _, nb in q1_log if q == "q1" and name == k1 and nb > 0])
                    roofline["q1"]["measured_in"] = "steps of q1 alone, waited for, after the timed region"
            if dom_launches is alone_launches:
                # which of the two timed regions the launches above are from, and the same kernel's launches in the other
                shared_ms = sum(shared_launches) / len(shared_launches) if shared_launches else None
                roofline["measured_in"] = ("%d steps whose queries are waited for one by one, every launch of this kernel between a pair of HIP events on the stream it "
                                           "runs on (calls issued one by one: a recorded plan has no place for events) — there the kernel has the chip to itself; in the "
                                           "overlapped steps the engine's %d lanes run the queries' kernels side by side" % (args.steps, nlanes))
                roofline["in_overlapped_steps"] = {"avg_launch_ms": round(shared_ms, 4) if shared_ms else None, "launches_timed": len(shared_launches),
                                                   "note": "the same kernel sharing the chip with the other lanes' kernels: the step, not the kernel, is the unit there (roofline.step)"}
            # the step as a whole against the same roofline: HBM bytes of its queries (PMC, per query) / the step's wall time
            step_bytes = [pmc_traffic(q, None, rows)[0] for q in queries]
            if all(step_bytes):
                roofline["step"] = {"physical_bytes": int(sum(step_bytes)), "ms_per_step": round(ms_per_step, 4),
                                    "frac": round(sum(step_bytes) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "frac_each_query_waited_for": round(sum(step_bytes) / (elapsed_waited / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                                    "traffic_source": "per-query PMC bytes (profiles/), wall time of this run"}
            if reference_width and dom_q in reference_width.get("queries", {}):
                rw = reference_width["queries"][dom_q]
                roofline["reference_width"] = {"kernel": rw["dominant_kernel"], "avg_launch_ms": rw["dominant_avg_launch_ms"], "achieved": rw["achieved"], "frac": rw["frac"]}
            if reference_width and "q1" in roofline and "q1" in reference_width.get("queries", {}):
                rw = reference_width["queries"]["q1"]
                roofline["q1"]["reference_width"] = {"kernel": rw["dominant_kernel"], "avg_launch_ms": rw["dominant_avg_launch_ms"], "achieved": rw["achieved"], "frac": rw["frac"]}
        per_query = {}
        for q in queries + extra:
            ab = algorithmic_bytes(q, rows)
            wall = (per_query_ms[q] / args.steps) if q in per_query_ms else (extra_ms[q] / extra_steps)
            phys, phys_source = pmc_traffic(q, None, rows, ran={name.split(":", 1)[1] for name in kernels if name.startswith(q + ":")})
            per_query[q] = {"ms_wall": round(wall, 4), "ms_kernels": round(device_ms[q], 4),
                            "in_timed_step": q in queries,
                            "rows_per_s_wall": round(scanned_rows(q, rows) / (wall * 1e-3), 1),
                            "algorithmic_bytes": ab,
                            "algorithmic_GBs_wall": round(ab / (wall * 1e-3) / 1e9, 1),
                            "algorithmic_GBs_kernels": round(ab / (device_ms[q] * 1e-3) / 1e9, 1) if device_ms[q] else None,
                            "algorithmic_frac_kernels": round(ab / (device_ms[q] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if device_ms[q] else None,
                            "physical_frac_kernels": round(phys / (device_ms[q] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if phys and device_ms[q] else None,
                            "first_run_ms": first_run_ms.get(q),
                            "built_once": built_once(q, first_run_detail.get(q), {name.split(":", 1)[1]: v["ms_per_step"] for name, v in kernels.items() if name.startswith(q + ":")},
                                                     device_ms[q], scan_form),
                            # physical bytes moved beside the algorithmic figure
                            "physical_bytes": phys,
                            "physical_GBs_kernels": round(phys / (device_ms[q] * 1e-3) / 1e9, 1) if phys and device_ms[q] else None,
                            "traffic_source": phys_source,
                            # streamed bytes of the query's kernels by the library's model of this run (lower bound: no gathers, no survivor-dependent stores)
                            "model_streamed_bytes": int(sum(v.get("model_bytes_per_launch", 0) * v["launches_per_step"] for k, v in kernels.items() if k.startswith(q + ":"))) or None}
        out = {
            "metric": "tpch_" + "_".join(queries) + "_sf%g_rows_per_sec" % sf_per_gpu, "value": round(value, 1), "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "TPCH " + "+".join(q.upper() for q in queries) + " SF=%g per GPU (BASELINE metric: Q1/Q3/Q5 at SF=10; q1 = configs[1], q3 = configs[2])" % sf_per_gpu,
                       "sf_per_gpu": sf_per_gpu, "sf_global": sf_global, "rows_per_gpu": rows,
                       "partitioning": "none" if not use_dist else "q1 row-sharded; q5 small builds replicated, orders-lineitem join co-partitioned; q3 partitioned on o_orderkey (%s), RCCL all-to-all; exchanged rows %s"
                                       % (runner.last_partitioning, runner.exchanged_rows)},
            # what a step is: its queries launched one after the other, then their results finished — a plan's last device call is
            # queued without being waited for (Engine.deferred_results) and every result is complete on the host before the step ends.
            # Beside it, the same step with every query's result finished before the next query is launched.
            # the same two figures under explicit names: `value` is the overlapped one
            "lanes": nlanes,
            "value_overlapped": round(value, 1), "value_sequential": round(total_rows_per_step * args.steps / elapsed_waited, 1),
            "step": {"launch_then_finish": True, "launch_order": step_order(queries), "deferred_results": bool(getattr(eng, "deferred_results", False)),
                     # host time to LAUNCH each query of an overlapped step (plan closures, ctypes calls, kernel launches; nothing waited for)
                     "host_launch_ms": {q: round(launch_host_ms[q] / args.steps, 4) for q in queries},
                     # ... and with the calls issued one by one (the evented repeat of the timed steps: no graphs there)
                     "host_launch_ms_calls_issued": {q: round(launch_host_calls_ms[q] / args.steps, 4) for q in queries},
                     "ms_per_step_calls_issued": round(elapsed_calls / args.steps * 1e3, 4),
                     "plan_graphs": graph_stats or None,
                     "ms_per_step_each_query_waited_for": round(elapsed_waited / args.steps * 1e3, 4),
                     "value_each_query_waited_for": round(total_rows_per_step * args.steps / elapsed_waited, 1)},
            "ms_per_query": per_query,
            "kernels_pass": "separate pass of %d steps after the timed region, every query waited for before the next starts, HIP events around every launch" % profile_steps,
            "kernels": {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in sorted(kernels.items())},
            "roofline": roofline,
            "first_pass_with_upload_s": round(first_pass_s, 3), "generate_s": round(gen_s, 2),
            # the boundary hands over host buffers: the PCIe-inclusive first pass (never the reported value)
            "first_pass_upload": {"bytes": uploaded_bytes, "GBs_including_plan_lowering": round(uploaded_bytes / first_pass_s / 1e9, 2) if first_pass_s > 0 else None},
        }
        out["engine_stats"] = eng.stats() if hasattr(eng, "stats") else None      # loops that ran on the host (none in the configured queries), plan graphs, resident bytes
        out["hbm"] = hbm
        if steady is not None:
            out["steady_state"] = steady
        if reference_width is not None:
            out["reference_width"] = reference_width
        if scan_form is not None:
            out["scan_form"] = scan_form
        if hash_path is not None:
            out["hash_path"] = hash_path
        try:
            c, d = eng.ctx.jit_stats()
            out["specialised_kernels"] = {"compiled_by_hiprtc_in_this_process": int(c), "loaded_from_jit_cache": int(d)}
        except Exception:
            pass
        if args.share_gpu:
            out["share_gpu"] = "the %d ranks of this run SHARED one GPU (gloo between the processes, pinned host buffers): every plan and kernel of the N > 1 path, none of its transport; not a scaling figure" % world
        if world > 1 or use_dist:
            out["n_gpus_note"] = "ranks in this run: %d%s" % (world, "" if world > 1 else " (the distributed plan on a group of one; N > 1 was not run here)")
        if exchange is not None:
            out["q3_exchange"] = exchange
        if timed_collectives is not None:
            out["collectives_in_timed_region_rank0"] = timed_collectives
            out["distributed_plan"] = {"device_sized_join_runs": runner.fast_runs, "repeated_with_exact_sizes": runner.fast_retries,
                                       "collectives_over_a_group_of_one": "skipped" if (world == 1 and runner.skip_trivial) else "issued"}
        if hip_results:
            ref = reference_at_bench_size(hip_results, sf_per_gpu, rows)
            if ref is not None:
                out["reference_at_bench_size"] = ref
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, queries, db, rows, hip_results)
            out["cpu_baseline"]["configs0_q6_sf1"] = q6_sf1_leg(eng)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        for rn in (runner, runner_range):                # collective buffers released before the process group and the engine's stream go
            if rn is not None:
                rn.close()
        if device == "cuda":
            torch.cuda.synchronize()
        dist.destroy_process_group()
    return out


# The options that make a loop a SCAN of its table's columns again (what the reference's loops are, and what `cpu_baseline` runs): no walk
# of a prebuilt run index instead of the column ("x_driven"), no row pack clustered by the first lookup's key ("cluster_pack"), no delta
# twin ("delta8").  Everything else about the timed step stays (twins, codes, row packs in row order).
SCAN_FORM_OPTIONS = {"x_driven": 0, "cluster_pack": 0, "delta8": 0}
SCAN_FORM_DEFAULTS = {"x_driven": 64, "cluster_pack": 1, "delta8": 1}
# ... and the reference-width leg on top of that: every numeric column streamed in the reference's own 8-byte form, gathered columns read
# where they lie (no row pack), loops on the fixed-shape kernels
REFERENCE_WIDTH_OPTIONS = dict(SCAN_FORM_OPTIONS, narrow=0, row_pack=0)
REFERENCE_WIDTH_DEFAULTS = dict(SCAN_FORM_DEFAULTS, narrow=1, row_pack=1)


def set_reference_width(eng, on):
    """Switch an engine to / from the reference-width configuration (bench.py `reference_width`, tools/run_queries.py --reference-width:
    the counter passes must run what the leg runs).  Returns what `on=False` needs to restore the routes."""
    for k, v in (REFERENCE_WIDTH_OPTIONS if on else REFERENCE_WIDTH_DEFAULTS).items():
        eng.ctx.set_option(k, v)
    if on:
        saved = (eng.stream_programs, eng.program_routes)
        eng.stream_programs, eng.program_routes = False, set()
        return saved


def source_digest():
    """sha256 (16 hex digits) over the sources that decide what a kernel moves: csrc/*.hip, *.hpp and the planner's *.py.  The committed
    PMC collections carry the digest of the tree they were counted on; a run on other sources says `stale: true` beside the bytes it
    quotes from them (the GPU box has no .git to ask for a commit distance)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "sdqlpy_amd")
    for path in sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.hpp")) + glob.glob(os.path.join(pkg, "*.py"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


WALK_STRUCTURES = ("k_run_index", "k_interleave", "k_rs_", "k_lower_bounds", "k_delta8")      # what the scan form does without: run indexes, clustered packs, delta twins


def built_once(q, detail, steady_kernels_ms, steady_ms, scan_form):
    """What a query's FIRST run built and later runs reuse (round-5 review: "built_once_ms / built_once_bytes and the number of runs that
    repay them"): device time of the first run's launches beyond a steady run's; of that, the structures only the timed step's walks and
    delta-twin streams need (WALK_STRUCTURES: the scan form runs without them); HBM left resident beside the columns; and how many runs
    repay the walk structures, priced at what a run saves over the scan form (kernels, one query at a time)."""
    if not detail or not detail.get("kernels_ms"):
        return None
    first = detail["kernels_ms"]
    walk = {k: round(v, 4) for k, v in first.items() if k.startswith(WALK_STRUCTURES)}
    rec = {"first_run_kernels_ms": round(sum(first.values()), 3), "steady_run_kernels_ms": round(steady_ms, 4),
           "built_once_ms": round(max(0.0, sum(first.values()) - steady_ms), 3),
           "built_once_bytes": max(0, int(detail["hbm_added"]) - int(detail["columns_added"])),
           "built_once_bytes_what": "HBM the process holds more after the first run than before, minus the columns uploaded at the reference's widths: twins, codes, "
                                    "dictionaries, delta twins, run indexes, row packs, pooled table memory",
           "walk_structures_ms": walk or None}
    if scan_form and q in scan_form.get("ms_per_query", {}):
        saved = scan_form["ms_per_query"][q]["ms_kernels"] - steady_ms
        rec["scan_form_kernels_ms"] = scan_form["ms_per_query"][q]["ms_kernels"]
        rec["saved_per_run_ms"] = round(saved, 4)
        if walk:
            rec["runs_to_repay_walk_structures"] = int(-(-sum(walk.values()) // saved)) if saved > 1e-4 else None
    return rec


HASH_LAYOUT_OPTIONS = {"direct_index": 0, "row_index": 0, "grouped_index": 0}      # every built table an open-addressing table (+ its hashed filter)


def hash_path_leg(args, eng, db, rows, ran, run_steps, finish):
    """north_star's own data structure — the open-addressing table — priced beside the direct layouts the timed step uses (round-5 review):
    (1) `hash_layouts`: the same queries with the direct layouts off (HASH_LAYOUT_OPTIONS): every build an open-addressing table (64-bit CAS
    insert, packed 32-byte slots) behind a hashed filter that the loops test on streamed registers; (2) `unordered_keys`: Q3 on the same
    data with every orders / customer key moved beyond 2^40 and the rows of orders and lineitem shuffled — no dense key range, no storage
    order: the data itself leaves nothing but the hash path — checked against the timed step's Q3 rows (same groups, keys shifted back).
    Outside `value`."""
    import numpy as np
    from sdqlpy_amd import tpch
    from sdqlpy_amd import tpch_queries as Q
    qs = [q for q in ("q3", "q5", "q9") if q in ran]
    out = {}
    n = max(1, min(args.steps, 10))

    def timed(queries, run):
        for _ in range(3):
            for q in queries:
                finish(run(q))
        took, per_q, _ = run_steps(n, "-", queries, run, each_waited_for=True)
        _, _, log = run_steps(n, None, queries, run, each_waited_for=True)
        res = {}
        for q in queries:
            per = {}
            for qq, name, ms, _ in log:
                if qq == q:
                    per[name] = per.get(name, 0.0) + ms / n
            res[q] = {"ms_wall": round(per_q[q] / n, 4), "ms_kernels": round(sum(per.values()), 4), "kernels_ms": {k: round(v, 4) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:6]}}
        return res

    want3 = finish(Q.run("q3", db)) if "q3" in qs else None
    for k, v in HASH_LAYOUT_OPTIONS.items():
        eng.ctx.set_option(k, v)
    try:
        eng.clear()
        out["hash_layouts"] = {"options": dict(HASH_LAYOUT_OPTIONS), "queries": timed(qs, lambda q: Q.run(q, db))}
        if want3 is not None:
            got = finish(Q.run("q3", db))
            cmp = compare_results(got, want3)
            out["hash_layouts"]["q3_rows_equal_direct_layouts"] = bool(cmp["rows_equal"]) and cmp["max_rel"] is not None and cmp["max_rel"] <= 1e-10
    finally:
        for k in HASH_LAYOUT_OPTIONS:
            eng.ctx.set_option(k, 1)
        eng.clear()
    if want3 is not None:
        # order keys spread over 2^46 (k * 1000003 + 2^40), customer keys over 2^35 + ..., the rows of orders and lineitem in random order: no
        # dense range for a bitmap, no storage order for a twin or a walk (tests/golden/make_golden.py's big_keys variant keeps the keys dense)
        off_o, mul_o, off_c, mul_c = np.int64(1) << 40, np.int64(1000003), np.int64(1) << 35, np.int64(1009)
        rng = np.random.default_rng(17)
        need = tpch.columns_for(("q3",))

        def remake(name, changes, shuffle):
            c = db[name].getContainer()
            keep = [h for h in c["headers"] if h in need[name]]
            cols = {h: a for h, a in zip(c["headers"], c["data"]) if h in keep}
            for col, fn in changes.items():
                cols[col] = fn(cols[col])
            if shuffle:
                perm = rng.permutation(len(c["data"][0]))
                cols = {h: (np.ascontiguousarray(a[perm]) if len(a) else a) for h, a in cols.items()}
            return tpch.table_from_columns(keep, [cols[h] for h in keep])
        t0 = time.perf_counter()
        big = dict(db)
        big["customer"] = remake("customer", {"c_custkey": lambda a: a * mul_c + off_c}, False)
        big["orders"] = remake("orders", {"o_orderkey": lambda a: a * mul_o + off_o, "o_custkey": lambda a: a * mul_c + off_c}, True)
        big["lineitem"] = remake("lineitem", {"l_orderkey": lambda a: a * mul_o + off_o}, True)
        prep_s = time.perf_counter() - t0
        try:
            res = timed(["q3"], lambda q: Q.run(q, big))
            got = finish(Q.run("q3", big))
            same = got.size() == want3.size()
            if same:
                gk, wk = (np.asarray(got.column("l_orderkey")) - off_o) // mul_o, np.asarray(want3.column("l_orderkey"))
                go, wo = np.argsort(gk, kind="stable"), np.argsort(wk, kind="stable")
                gr, wr = np.asarray(got.column("revenue"))[go], np.asarray(want3.column("revenue"))[wo]
                same = bool(np.array_equal(gk[go], wk[wo])) and bool(np.all(np.abs(gr - wr) <= 1e-10 * np.maximum(np.abs(wr), 1e-300)))
            out["unordered_keys"] = {"what": "Q3 at the bench's size with the order keys spread over 2^46 (k * 1000003 + 2^40), the customer keys over 2^35 + k * 1009, the rows of orders and lineitem "
                                             "shuffled: no dense key range, no storage order — open addressing (behind its hashed filter) is what is left", "q3": res["q3"], "rows": int(got.size()), "rows_equal_timed_step": bool(same),
                                     "prepared_on_host_s": round(prep_s, 2)}
        finally:
            for t in ("customer", "orders", "lineitem"):
                eng.invalidate(big[t])
    return out


def scan_form_leg(args, eng, step_queries, queries, run_query, run_steps, finish):
    """Identical work beside `value` (round-5 review): the timed step's final loops of Q5 / Q9 WALK resident orders of the data (a run index
    of l_orderkey, a row pack clustered by l_partkey) and Q3's loops stream l_orderkey / o_orderkey through delta twins; the CPU baseline
    rescans its columns every run.  This leg switches those three off (SCAN_FORM_OPTIONS): every loop streams every row of its table
    again, still through the exact narrow encodings.  Outside `value`."""
    for k, v in SCAN_FORM_OPTIONS.items():
        eng.ctx.set_option(k, v)
    try:
        eng.clear()
        for _ in range(3):
            for q in queries:
                finish(run_query(q))
        n = max(1, min(args.steps, 10))
        run_steps(3, "-", step_queries)
        took, _, _ = run_steps(n, "-", step_queries)                   # (the timed step's queries; the extra ones are priced per query below)
        run_steps(3, "-", queries, each_waited_for=True)
        took_w, per_q, _ = run_steps(n, "-", queries, each_waited_for=True)
        _, _, log = run_steps(n, None, queries, each_waited_for=True)
        kernels_ms = {q: round(sum(ms for qq, _, ms, _ in log if qq == q) / n, 4) for q in queries}
        return {"options": dict(SCAN_FORM_OPTIONS), "step": list(step_queries), "ms_per_step": round(took / n * 1e3, 4),
                "ms_per_step_each_query_waited_for": round(sum(per_q[q] for q in step_queries) / n, 4),
                "ms_per_query": {q: {"ms_wall": round(per_q[q] / n, 4), "ms_kernels": kernels_ms[q]} for q in queries},
                "what": "the same step with every loop streaming every row of its table (no driven walk, no clustered pack, no delta twins); outside `value`"}
    finally:
        for k, v in SCAN_FORM_DEFAULTS.items():
            eng.ctx.set_option(k, v)
        eng.clear()


def reference_width_leg(args, eng, db, rows, ran, run_query, run_steps, finish):
    """"the headline fraction stays on the algorithmic figure so CPU and GPU are compared on identical work".  The
    timed step streams exact narrow encodings of the columns (4-byte twins, 1- / 2-byte dictionary codes), so algorithmic bytes
    over its time exceed the HBM peak and are no roofline fraction.  This leg switches the encodings off (`narrow` 0: every numeric
    column is streamed in the reference's own 8-byte form; text columns that are only compared / grouped on still travel as int64
    dictionary codes, 8 bytes instead of 4 per code unit), drops everything resident, uploads again and times q1 / q3 / q6 with the
    protocol of the timed region.  frac = algorithmic bytes of the dominant kernel / its average launch time / 8 TB/s: at most 1,
    and the figure that is comparable with `cpu_baseline` (which reads the same 8-byte columns).  Outside `value`."""
    qs = [q for q in ("q1", "q3", "q6", "q5", "q9") if q in ran]
    if not qs:
        return None
    out = {"option": "narrow=0 (no 4-byte twins, no dictionary codes of numeric columns), row_pack=0 (gathered columns read where they lie), no driven walk / "
                     "clustered pack / delta twins; loops on the fixed-shape kernels, which are the ones tuned for 8-byte columns (two rows per lane: "
                     "k_groupby_reg, k_scan_sum, k_stage, k_probe_agg, k_build_lookup, k_lookup_agg)", "queries": {}}
    saved = set_reference_width(eng, True)
    try:
        eng.clear()
        # the pinned-staged upload alone (g1): lineitem's columns, nothing else running, no plan lowering, no twins
        li = db["lineitem"].getContainer()
        t0 = time.perf_counter()
        nbytes = 0
        for arr in li["data"]:
            if len(arr):
                eng.column(arr)
                nbytes += arr.nbytes
        eng.ctx.synchronize()
        up = time.perf_counter() - t0
        out["upload_only"] = {"table": "lineitem", "bytes": int(nbytes), "seconds": round(up, 4), "GBs": round(nbytes / up / 1e9, 2) if up > 0 else None,
                              "what": "sdqh_column_upload of every resident lineitem column back to back (pinned 2 x 32 MiB ring, async H2D), then one synchronise"}
        n = max(1, min(args.steps, 10))
        for q in qs:
            finish(run_query(q))
        eng.ctx.synchronize()
        for _ in range(2):
            for q in qs:
                finish(run_query(q))
        took, per_q, _ = run_steps(n, "-", qs, each_waited_for=True)
        _, _, log = run_steps(n, None, qs, each_waited_for=True)
        for q in qs:
            stat = {}
            for qq, name, ms, nb in log:
                if qq == q:
                    tot, cnt, b = stat.get(name, (0.0, 0, 0))
                    stat[name] = (tot + ms, cnt + 1, b + nb)
            if not stat:
                continue
            ms_kernels = sum(t for t, _, _ in stat.values()) / n
            # the kernel that runs the query's big loop (its name is known for the fixed-shape route this leg takes), else the longest launch
            dom = DOMINANT[q][0] if DOMINANT[q][0] in stat and q not in ("q1", "q3", "q6") else max(stat, key=lambda k: stat[k][0] / stat[k][1])
            tot, cnt, b = stat[dom]
            dom_ms = tot / cnt
            ab_q, ab_dom = algorithmic_bytes(q, rows), DOMINANT[q][1](rows)
            if q == "q3":
                # the probe loop reads l_extendedprice / l_discount on a hit only — the reference's own loop short-circuits the same way
                # (`if l_shipdate > d: if contains(l_orderkey): ... ep * (1.0 - disc)`) — so what every row costs the
                # kernel is the two tested columns: 16 bytes
                ab_dom = 16 * rows["lineitem"]
            elif q == "q5":
                ab_dom = 8 * rows["lineitem"]                 # (the same short circuit: l_orderkey on every row, the rest where the order passed — test_all.py:266-277)
            elif q == "q9":
                ab_dom = 16 * rows["lineitem"]                # (l_partkey, l_suppkey on every row, the other four columns where the part is green — test_all.py:474-487)
            rec = {"ms_wall": round(per_q[q] / n, 4), "ms_kernels": round(ms_kernels, 4), "algorithmic_bytes": ab_q,
                   "frac_kernels": round(ab_q / (ms_kernels * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "dominant_kernel": dom, "dominant_avg_launch_ms": round(dom_ms, 4), "dominant_algorithmic_bytes": ab_dom,
                   "achieved": round(ab_dom / (dom_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                   "frac": round(ab_dom / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if b:
                rec["dominant_model_bytes_per_launch"] = b // cnt
            # counter evidence for this leg: HBM bytes of the dominant kernel's launch and of the query's run from the committed PMC passes over
            # THIS configuration (tools/collect_profiles.sh: run_queries.py --reference-width) — the algorithmic bytes are a claim, these are counted
            t_dom, t_src = pmc_traffic(q, dom, rows, which="pmc_traffic_reference_width")
            t_run, _ = pmc_traffic(q, None, rows, ran=set(stat), which="pmc_traffic_reference_width")
            if t_dom:
                rec["dominant_traffic"] = t_dom
                rec["dominant_traffic_over_algorithmic"] = round(t_dom / ab_dom, 4)
                rec["dominant_physical_frac"] = round(t_dom / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                rec["traffic_source"] = t_src
            if t_run:
                rec["traffic"] = t_run
                rec["traffic_over_algorithmic"] = round(t_run / ab_q, 4)
            out["queries"][q] = rec
    finally:
        set_reference_width(eng, False)
        eng.stream_programs, eng.program_routes = saved
        eng.clear()
    return out


def pmc_traffic(q, kernel, rows, ran=None, which="pmc_traffic"):
    """(HBM bytes, source tag) from the committed rocprofv3 PMC summary profiles/rNN_pmc_traffic.json
    (tools/pmc_per_query.py: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes per query;
    bytes = 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950): with `kernel`,
    bytes per launch of that kernel inside query `q`; without, bytes of one whole run of `q`.  The
    numbers are constants of that committed run, valid only for the same row counts; else (None, None)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s.json" % which)))      # (which = "pmc_traffic_reference_width": the passes over the reference-width leg)
    if not found:
        return None, None
    path = found[-1]                                      # the latest round's collection
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None, None
    entry = rec.get("queries", {}).get(q)
    if not entry or any(rec.get("rows", {}).get(t) != rows.get(t) for t in entry.get("tables", ["lineitem"])):
        return None, None
    stale = rec.get("source_digest") != source_digest()      # counted on other kernel / planner sources than the ones running now?
    source = "committed rocprofv3 PMC run (profiles/%s%s), not this run%s" % (os.path.basename(path), ", collected at commit %s" % rec["commit"] if rec.get("commit") else "",
                                                                             "; STALE: counted on other sources than this run's (source_digest %s, here %s)" % (rec.get("source_digest"), source_digest()) if stale else "; sources unchanged since")
    if ran is not None and not set(ran) <= set(entry.get("kernels", {})):
        return None, None                                 # this run launched kernels the committed collection never saw: other code, other bytes
    if kernel is None:
        return entry.get("hbm_bytes_per_run"), source
    k = entry.get("kernels", {}).get(kernel)
    return (k.get("hbm_bytes_per_launch") if k else None), source


def reference_at_bench_size(hip_results, sf, rows):
    """The timed step's results against the REFERENCE's own, where the reference was run on exactly these inputs: tests/golden/
    tpch_golden_sf10.json.gz holds what its Python-mode interpreter returned for q1 / q3 / q5 / q6 / q9 on this generator's SF=10
    tables (tests/golden/make_golden.py --sf10, run once in the build container; a fixture — the reference itself never travels).
    Rows, keys and counts must be equal, sums within 1e-6 relative (north_star); the largest relative difference is printed."""
    import gzip
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "tpch_golden_sf10.json.gz")
    if abs(sf - 10.0) > 1e-9 or not os.path.exists(path):
        return None
    with gzip.open(path, "rt") as fh:
        case = json.load(fh)["cases"][0]
    if any(case["rows"].get(t) != rows.get(t) for t in case["rows"]):
        return {"skipped": "the generated tables are not the golden case's (row counts differ)"}
    out = {"source": "edin-dal/sdqlpy Python mode on the same generated SF=10 tables (tests/golden/tpch_golden_sf10.json.gz)", "queries": {}}
    dec = lambda v: float.fromhex(v["f"]) if isinstance(v, dict) else v      # noqa: E731
    for q, res in hip_results.items():
        gold = case["results"].get(q)
        if gold is None:
            continue
        if gold["kind"] == "scalar":
            want = dec(gold["value"])
            out["queries"][q] = {"rows_equal": True, "max_rel": abs(float(res) - want) / abs(want) if want else abs(float(res) - want)}
            continue
        cols = gold["columns"]
        want = [tuple(dec(x) for x in row) for row in gold["rows"]]
        got = sorted(zip(*[np.asarray(res.column(c)).tolist() for c in cols]))
        want.sort()
        ok, max_rel = len(got) == len(want), 0.0
        if ok:
            for a, b in zip(got, want):
                for x, y in zip(a, b):
                    if isinstance(y, float):
                        max_rel = max(max_rel, abs(x - y) / max(abs(x), abs(y), 1e-300))
                    elif x != y:
                        ok = False
        out["queries"][q] = {"rows": len(got), "rows_equal": bool(ok), "max_rel": max_rel, "within_1e-6": bool(ok and max_rel <= 1e-6)}
    return out


def physical_cores():
    """(physical cores, logical CPUs) of this host."""
    logical = os.cpu_count() or 1
    try:
        seen, phys, core = set(), None, None
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("physical id"):
                    phys = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    core = ln.split(":")[1].strip()
                elif not ln.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        if seen:
            return len(seen), logical
    except OSError:
        pass
    return logical, logical


def compare_results(got, want):
    """HIP result against the CPU implementation's on the same tables: row sets equal on every integer / text column (counts are
    integer columns), largest relative difference over the float columns."""
    import numpy as np
    if isinstance(want, float) or isinstance(got, float):
        return {"rows_equal": True, "counts_equal": True, "max_rel": abs(got - want) / abs(want) if want else abs(got - want)}
    if got.size() != want.size() or list(got.columns) != list(want.columns):
        return {"rows_equal": False, "counts_equal": False, "max_rel": None, "rows": [got.size(), want.size()]}
    ga, wa = [np.asarray(a) for a in got.arrays], [np.asarray(a) for a in want.arrays]
    exact = [i for i, a in enumerate(wa) if a.dtype.kind != "f"]
    order = lambda arrs: np.lexsort([arrs[i] for i in reversed(exact)]) if exact else np.arange(len(arrs[0]))      # noqa: E731
    go, wo = order(ga), order(wa)
    rows_equal = all(np.array_equal(ga[i][go], wa[i][wo]) for i in exact)
    counts = [i for i in exact if "count" in got.columns[i]]
    max_rel = 0.0
    for i, a in enumerate(wa):
        if a.dtype.kind == "f" and len(a):
            g, w = ga[i][go], a[wo]
            den = np.maximum(np.abs(g), np.abs(w))
            rel = np.where(den > 0, np.abs(g - w) / np.where(den > 0, den, 1.0), 0.0)
            max_rel = max(max_rel, float(rel.max()))
    return {"rows": int(want.size()), "rows_equal": bool(rows_equal), "counts_equal": bool(all(np.array_equal(ga[i][go], wa[i][wo]) for i in counts)), "max_rel": max_rel}


def cpu_baseline(args, queries, db, rows, hip_results=None):
    """The CPU port (oracle/, same plan as the reference's TBB code: per-thread partials + merge)
    timed on this box's host cores on a bounded sample of the same workload."""
    import numpy as np
    from sdqlpy_amd import abi, engine, frontend, tpch
    from sdqlpy_amd import tpch_queries as Q
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    cores_phys, cores = physical_cores()
    lib = abi.Library(os.path.join(ROOT, "oracle", "libsdqloracle.so"))
    eng = engine.Engine(lib.context(threads=cores))
    # the sample: by default the WHOLE workload (a pass over SF=10 costs the CPU port under a second on
    # a many-core host); --cpu-sample-sf bounds it to a prefix of the same generated tables (lineitem cut
    # on an order boundary) for hosts where that would take minutes
    frac = min(1.0, args.cpu_sample_sf / args.sf) if args.cpu_sample_sf > 0 else 1.0
    sample = {}
    n_ord = int(rows.get("orders", 0) * frac)
    for t, table in db.items():
        c = table.getContainer()
        if t == "lineitem":
            ok = c["data"][c["headers"].index("l_orderkey")] if "l_orderkey" in c["headers"] else None
            if ok is not None and "orders" in db:
                last_key = tpch.column(db["orders"], "o_orderkey")[max(0, n_ord - 1)]
                n = int(np.searchsorted(ok, last_key, side="right"))
            else:
                n = int(len(c["data"][0]) * frac)
        elif t in ("orders",):
            n = n_ord
        else:
            n = len(c["data"][0])               # dimension-side tables stay whole so key references resolve
        sample[t] = table if n == len(c["data"][0]) else tpch.table_from_columns(c["headers"], [np.ascontiguousarray(a[:n]) for a in c["data"]])
    srows = {t: len(sample[t].getContainer()["data"][0]) for t in sample}
    plans = {q: frontend.lower_function(Q.QUERIES[q]) for q in queries}
    run = lambda q: engine.execute_plan(eng, plans[q], [sample[t] for t in Q.QUERY_TABLES[q]])   # noqa: E731
    cpu_results = {}
    for q in queries:
        cpu_results[q] = run(q)                  # warm-up (also copies the columns into the oracle's memory); the results are the checker's
    iters, t0 = 0, time.perf_counter()
    per_q = {q: 0.0 for q in queries}
    while iters < 5 and (time.perf_counter() - t0) < 20.0:
        for q in queries:
            tq = time.perf_counter()
            run(q)
            per_q[q] += time.perf_counter() - tq
        iters += 1
    # parity at the size the line is quoted on: the HIP results of this run against the CPU implementation's, whole workload only
    parity = None
    if hip_results and frac >= 1.0:
        parity = {q: compare_results(hip_results[q], cpu_results[q]) for q in queries if q in hip_results}
    elapsed = time.perf_counter() - t0
    total_rows = sum(scanned_rows(q, srows) for q in queries)
    eng.close()
    # one-thread figure: the same sample, one pass, one worker
    eng1 = engine.Engine(lib.context(threads=1))
    run1 = lambda q: engine.execute_plan(eng1, plans[q], [sample[t] for t in Q.QUERY_TABLES[q]])   # noqa: E731
    for q in queries:
        run1(q)
    t1 = time.perf_counter()
    for q in queries:
        run1(q)
    one_thread = total_rows / (time.perf_counter() - t1)
    eng1.close()
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": round(total_rows * iters / elapsed, 1), "unit": "rows/s", "cores": cores, "threads": cores, "cores_physical": cores_phys, "kind": "port",
            "cpu_model": model, "value_one_thread": round(one_thread, 1),
            "sample": ("the whole workload (%d lineitem rows)" % srows.get("lineitem", 0) if frac >= 1.0 else
                       "same generator, first %.0f%% of orders + their lineitems (%d lineitem rows), dimension tables whole" % (100 * frac, srows.get("lineitem", 0)))
                      + "; %d passes of %s, data resident in host RAM" % (iters, "+".join(queries)),
            "ms_per_query": {q: round(per_q[q] / iters * 1e3, 2) for q in queries},
            "note": "a faithful port of the reference's plan (per-thread partial tables, SERIAL insert(range) / AddMap merges: sdql_ir_cpp_generator_par.py:331-369, map_helper.h:1-23): "
                    "the merges do not scale with threads, so many cores buy little (value vs value_one_thread); a stated baseline, not a target — the roofline fraction is the measure",
            "parity_at_bench_size": parity}


def q6_sf1_leg(hip_eng):
    """BASELINE configs[0]: TPCH Q6 at SF=1 on the CPU path (the oracle port, all hardware threads and one),
    with the HIP path on the same 6 M rows beside it."""
    from sdqlpy_amd import abi, engine, frontend, tpch
    from sdqlpy_amd import tpch_queries as Q
    cols = tpch.columns_for(["q6"])
    db1 = tpch.generate(1.0, tables=["lineitem"], columns=cols)
    n = len(db1["lineitem"].getContainer()["data"][0])
    plan = frontend.lower_function(Q.QUERIES["q6"])
    lib = abi.Library(os.path.join(ROOT, "oracle", "libsdqloracle.so"))
    out = {"rows": n, "kind": "port"}
    want = None
    for label, threads in (("ms_all_threads", os.cpu_count() or 1), ("ms_one_thread", 1)):
        ceng = engine.Engine(lib.context(threads=threads))
        want = engine.execute_plan(ceng, plan, [db1["lineitem"]])         # warm-up: copies the columns
        t0, it = time.perf_counter(), 0
        while it < 5 and time.perf_counter() - t0 < 5.0:
            engine.execute_plan(ceng, plan, [db1["lineitem"]])
            it += 1
        out[label] = round((time.perf_counter() - t0) / it * 1e3, 3)
        ceng.close()
    out["threads"] = os.cpu_count() or 1
    got = engine.execute_plan(hip_eng, plan, [db1["lineitem"]])
    for _ in range(3):
        engine.execute_plan(hip_eng, plan, [db1["lineitem"]])
    t0 = time.perf_counter()
    for _ in range(20):
        engine.execute_plan(hip_eng, plan, [db1["lineitem"]])
    out["hip_ms_wall"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
    out["rel_diff_hip_vs_cpu"] = abs(got - want) / abs(want) if want else None
    return out


if __name__ == "__main__":
    main()
