"""The N > 1 path on CPU: world_size = 2 over gloo, the CPU oracle behind the ABI, the same
distributed plan as on GPUs (sdqlpy_amd/dist.py).  Checked against the single-process oracle on the
whole database: integers and group keys exactly, sums to 1e-12 (rank-order folding)."""
import json
import os
import socket
import subprocess
import sys

import pytest

import helpers
from sdqlpy_amd import engine, tpch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SF = 0.01


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(mode, tmp_path, world=2, env=None):
    port = free_port()
    out = str(tmp_path / ("dist_%s.json" % mode))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), str(port), str(SF), mode, out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **(env or {}))) for r in range(world)]
    logs = [p.communicate(timeout=300)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    with open(out) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def single(oracle_lib):
    eng = engine.Engine(oracle_lib.context(threads=2))
    qs = ["q1", "q3", "q5", "q6", "q9", "q4", "q14", "q18"]
    db = tpch.generate(SF, tables=sorted(tpch.columns_for(qs)), columns=tpch.columns_for(qs), threads=2)
    res = {q: helpers.run_query(eng, q, db) for q in qs}
    import dist_queries
    dist_queries.register()
    res["q18"] = helpers.run_query(eng, "q18_low", db)          # q18 with a threshold that leaves rows at this scale
    assert res["q18"].size() > 3
    eng.close()
    return res


def as_rows(rows):
    return [tuple(r) for r in rows]


def _check_chain_overflow(got, single, what):
    """A settled chain whose chunk bounds were cut to two rows: one collective re-run with exact sizes, the right rows, bounds measured
    again, and the run after it settled again without a retry.  (Where q5 replicates no table with payload there is nothing to outgrow.)"""
    ov = got.get("chain_overflow")
    if ov is None:
        # (q5's chain has no settled form on these shards — e.g. the probe-aggregate's table is a replica — or nothing with payload to replicate)
        assert not what.startswith("range/") and what != "range", what
        return False
    helpers.assert_rows_match(sorted(as_rows(ov["rows"])), helpers.result_rows(single["q5"], ov["columns"]), 1e-12, what + "/q5 after an overflow")
    helpers.assert_rows_match(sorted(as_rows(ov["next"]["rows"])), helpers.result_rows(single["q5"], ov["columns"]), 1e-12, what + "/q5 after the re-run")
    assert ov["retries"] == 1 and ov["next"]["retries"] == 1, ov
    assert all(v > 2 for v in ov["caps_after"].values()), ov["caps_after"]
    return True


@pytest.mark.parametrize("mode", ["range", "range_foreign", "hash", "shuffled"])
def test_two_ranks_match_single_process(mode, tmp_path, single, oracle_lib):
    got = run_world(mode, tmp_path)
    assert abs(got["q6"] - single["q6"]) <= 1e-12 * abs(single["q6"])
    w1 = single["q1"]
    helpers.assert_rows_match(sorted(as_rows(got["q1"]["rows"])), helpers.result_rows(w1, got["q1"]["columns"]), 1e-12, mode + "/q1")
    for q in ("q5", "q9", "q4"):
        helpers.assert_rows_match(sorted(as_rows(got[q]["rows"])), helpers.result_rows(single[q], got[q]["columns"]), 1e-12, mode + "/" + q)
    assert abs(got["q14"] - single["q14"]) <= 1e-12 * abs(single["q14"])
    # the settled chains (device-sized replication, groups folded behind one all-gather): the same rows, and they really took that path
    for tag, res in got["again"].items():
        q = tag.split("/")[0]
        helpers.assert_rows_match(sorted(as_rows(res["rows"])), helpers.result_rows(single[q], res["columns"]), 1e-12, mode + "/again/" + tag)
        if q == "q9" and res["seams"]:
            # the fixed-shape lookup loop (sdqh_lookup_aggregate_block, ABI 7): this rank's partial groups into the all-gather's send
            # buffer, the ranks' blocks folded by packed key — its keys decode through `nation`, whole on every rank
            assert res["seams"]["folded"] == ["profit"] and not res["seams"]["merged_on_host"], (mode, tag, res["seams"])
    assert got["chain_fast_runs"] >= 6 and got["chain_fast_retries"] == 0, (got["chain_fast_runs"], got["chain_fast_retries"])
    _check_chain_overflow(got, single, mode)
    w3 = single["q3"]
    helpers.assert_rows_match(sorted(as_rows(got["q3"]["rows"])), helpers.result_rows(w3, got["q3"]["columns"]), 1e-12, mode + "/q3")
    # ORDER BY ... LIMIT k: the same first rows, in the same order, as ordering the single-process result
    from sdqlpy_amd import tpch_queries as Q
    for q, top in (("q3", Q.TPCH_ORDER["q3"]), ("q5", (3, [("revenue", "desc")]))):
        want = single[q].top(*top)
        cols = got[q + "_top"]["columns"]
        want_rows = list(zip(*[want.column(c).tolist() for c in cols]))
        assert len(got[q + "_top"]["rows"]) == len(want_rows) == min(top[0], single[q].size())
        assert [r[0] for r in got[q + "_top"]["rows"]] == [r[0] for r in want_rows], mode + "/" + q + "_top order"
        helpers.assert_rows_match(as_rows(got[q + "_top"]["rows"]), want_rows, 1e-12, mode + "/" + q + "_top")
    # q18 (row-keyed group-by, HAVING, joins): local when lineitem and orders are co-partitioned on the order
    # key, each rank returning its partition's groups; refused — by every rank alike — otherwise
    if mode in ("range", "hash"):                               # the same co-partitioned shards ("hash" only forces q3's exchange)
        cols = got["q18"]["columns"]
        helpers.assert_rows_match(sorted(as_rows(got["q18"]["rows"])), helpers.result_rows(single["q18"], cols), 1e-12, mode + "/q18")
        assert 0 <= got["q18"]["local_rows"] <= len(got["q18"]["rows"]) and len(got["q18"]["rows"]) > 0
        want = single["q18"].top(*Q.TPCH_ORDER["q18"])
        tcols = got["q18_top"]["columns"]
        helpers.assert_rows_match(as_rows(got["q18_top"]["rows"]), list(zip(*[want.column(c).tolist() for c in tcols])), 1e-12, mode + "/q18_top")
    else:
        assert "unsupported" in got["q18"], got["q18"]
    assert "may span ranks" in got["q10"].get("unsupported", "") or "unsupported" in got["q10"], got["q10"]
    if mode == "range":
        assert "may span ranks" in got["q10"]["unsupported"]
        # dbgen-shaped shards are co-clustered on o_orderkey: nothing has to move
        assert got["q3"]["partitioning"] == "range" and got["q3"]["exchanged"]["probe_sent"] == 0
    elif mode == "range_foreign":
        assert got["q3"]["partitioning"] == "range" and got["q3"]["exchanged"]["probe_sent"] > 0
    else:
        assert got["q3"]["partitioning"] == "hash" and got["q3"]["exchanged"]["build"] > 0
        # the prefilter (replicated bitmap of all build keys): only probe rows that hit travel — a few hundred at this scale, not tens of thousands
        assert 0 < got["q3"]["exchanged"]["probe_sent"] < 3000, got["q3"]["exchanged"]
    assert 0 < got["q3"]["local_rows"] < len(got["q3"]["rows"])
    # customer whole on every rank: same result as with the sharded customer; orders whole: refused
    helpers.assert_rows_match(sorted(as_rows(got["q3_customer_whole"]["rows"])), helpers.result_rows(w3, got["q3_customer_whole"]["columns"]), 1e-12, mode + "/q3 customer whole")
    assert "row-sharded" in got["q3_orders_whole"].get("unsupported", ""), got["q3_orders_whole"]
    assert sorted(as_rows(got["q1_by_function"]["rows"])) == sorted(as_rows(got["q1"]["rows"]))
    # decorated functions called directly with a runner installed (the sdqlpy_init(3, devices=N) route)
    assert got["q6_decorated"] == got["q6"]
    assert sorted(as_rows(got["q3_decorated"]["rows"])) == sorted(as_rows(got["q3"]["rows"]))
    assert as_rows(got["q3_decorated_top"]) == as_rows(got["q3_top"]["rows"])
    # tables that lost their shard mark: refused (never a silently partial answer); a "whole" table that differs
    # between the ranks: caught by the cross-rank row-count / checksum test; the mark carried on: same answer
    for q in ("q6", "q3"):
        assert "row-shard mark" in got["unmarked"][q], got["unmarked"]
    assert "differ between ranks" in got["whole_mismatch"], got["whole_mismatch"]
    assert got["q6_remarked"] == got["q6"]
    assert got["collectives"]["all_gather"][0] > 0
    assert (got["collectives"].get("all_to_all", [0])[0] > 0) == (mode != "range")        # co-clustered shards: nothing to exchange


@pytest.mark.parametrize("mode,world", [("range", 2), ("range", 3)])
def test_settled_chains_fold_their_groups_behind_one_all_gather(mode, world, tmp_path, single, oracle_lib):
    """Round 6: with every loop a row program (as on the GPU) a settled chain's last group-by writes this rank's partial groups into
    the send buffer of ONE all-gather and the ranks' blocks are folded by packed key on the device (sdqh_xgroupby_partial / _fold; here
    the CPU implementation of the same calls), its replicated tables travel in fixed-capacity chunks (sdqh_table_partition_pack with
    one part / sdqh_unpack_chunks) — nothing is read back between the first call and the result.  Same rows as one process; and the
    seams say that this is the path that ran."""
    got = run_world(mode, tmp_path, world=world, env={"SDQLPY_AMD_FORCE_PROGRAMS": "1"})
    for tag, res in got["again"].items():
        q = tag.split("/")[0]
        helpers.assert_rows_match(sorted(as_rows(res["rows"])), helpers.result_rows(single[q], res["columns"]), 1e-12, "%s/programs/%s" % (mode, tag))
        seams = res["seams"]
        assert seams["plan"] == q, (tag, seams)
        if q in ("q1", "q5", "q4"):
            assert seams["folded"] and not seams["merged_on_host"], (tag, seams)
        if q == "q5":
            assert seams["replicated"] == ["asian_customers", "supplier_nations"], (tag, seams)
    assert got["chain_fast_runs"] == 8 and got["chain_fast_retries"] == 0
    assert _check_chain_overflow(got, single, mode + "/programs")
    helpers.assert_rows_match(sorted(as_rows(got["q3"]["rows"])), helpers.result_rows(single["q3"], got["q3"]["columns"]), 1e-12, mode + "/programs/q3")


def test_four_ranks_match_single_process(tmp_path, single, oracle_lib):
    """world_size 4 (an odd row split per rank, three peers per exchange) on the shuffled layout:
    hash partitioning with real traffic between every pair of ranks."""
    got = run_world("shuffled", tmp_path, world=4)
    assert abs(got["q6"] - single["q6"]) <= 1e-12 * abs(single["q6"])
    for q in ("q1", "q5", "q9", "q4", "q3"):
        helpers.assert_rows_match(sorted(as_rows(got[q]["rows"])), helpers.result_rows(single[q], got[q]["columns"]), 1e-12, "world4/" + q)
    assert abs(got["q14"] - single["q14"]) <= 1e-12 * abs(single["q14"])
    assert got["q3"]["partitioning"] == "hash" and got["q3"]["exchanged"]["probe_sent"] > 0
    assert "unsupported" in got["q18"]                         # every rank refuses alike: groups and joins would span ranks


def test_bench_contract_under_a_two_rank_launch(tmp_path):
    """bench.py as the driver launches it for N > 1 (RANK / WORLD_SIZE / MASTER_* in the
    environment), on CPU: gloo + the CPU implementation of the ABI injected through bench.main's
    hooks.  Rank 0 prints exactly one JSON line with the contract's fields; the other rank prints none."""
    port = free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bench_gloo_worker.py"), str(r), "2", str(port), "0.01"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    lines = [ln for ln in outs[0][0].splitlines() if ln.strip()]
    assert len(lines) == 1 and not outs[1][0].strip(), (outs[0][0][-500:], outs[1][0][-500:])
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in rec, key
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak" and rec["vs_baseline"] is None
    assert rec["unit"] == "rows/s" and rec["higher_is_better"] is True and rec["value"] > 0 and rec["ms_per_step"] > 0
    assert "workload" in rec["config"] and "q3 partitioned" in rec["config"]["partitioning"]
    # the timed step hash-partitions q3: build and probe rows really cross ranks; the clustered-shard
    # shortcut is reported beside it and moves nothing
    ex = rec["q3_exchange"]
    assert ex["hash"]["partitioning"] == "hash" and ex["hash"]["exchanged_rows_rank0"]["probe_sent"] > 0 and ex["hash"]["exchanged_bytes_all_ranks"] > 0
    assert ex["auto(range)"]["exchanged_bytes_all_ranks"] == 0 and ex["auto(range)"]["exchanged_rows_rank0"]["probe_sent"] == 0
    assert "hash" in rec["config"]["partitioning"]
    # whole-job aggregate: rows of BOTH ranks per step over the max-over-ranks time
    per_rank = sum(rec["config"]["rows_per_gpu"][t] for t in ("lineitem",)) * 3 + rec["config"]["rows_per_gpu"]["customer"] * 2 + rec["config"]["rows_per_gpu"]["orders"] * 2
    assert rec["value"] * rec["ms_per_step"] * 1e-3 > 1.5 * per_rank


def test_bench_starts_its_own_ranks():
    """Plain `python bench.py --gpus 2` — no torchrun, no WORLD_SIZE: bench.py is its own launcher (spawn_ranks: child processes with
    the rendezvous environment, started before the parent touches a GPU), relays rank 0's single JSON line and exits 0; a rank that
    dies makes the launcher stop its peers and exit non-zero instead of hanging in a collective."""
    env = dict(os.environ, SDQLPY_AMD_BENCH_CHILD=os.path.join(ROOT, "tests", "bench_spawn_worker.py"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--sf", "0.01", "--no-cpu-baseline"]
    done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, done.stdout[-1000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["value"] > 0 and rec["q3_exchange"]["hash"]["exchanged_bytes_all_ranks"] > 0
    bad = subprocess.run(cmd, env=dict(env, SDQLPY_TEST_FAIL_RANK="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert bad.returncode != 0 and not bad.stdout.strip(), (bad.returncode, bad.stdout[-500:])
    assert "rank 1 exited with code 7" in bad.stderr


def test_device_sized_exchanges_and_what_happens_when_a_bound_is_too_small(tmp_path, oracle_lib):
    """The hash-partitioned join from its second run on moves fixed-capacity chunks with their counts in the chunk headers — no count
    visits the host (dist.DistributedRunner._hash_join_device_sized; include/sdqh.h ABI 5) — and returns a result that is launched,
    not waited for.  Same rows as the first, exact-size run, every time; bounds forced too small are found out from the all-reduced
    status when the result is collected and the join is repeated COLLECTIVELY with exact sizes; a K-F result block forced too small —
    on both ranks, then on rank 0 alone — is repeated on that rank's kept tables without any collective (a rank re-running the plan
    alone would build from its shard only, or hang its peers: the round-4 advice).  The range-partitioned form the same."""
    sf = 0.3                                                        # ~1 700 result rows per rank: more than the smallest result block (1024 rows)
    port, out = free_port(), str(tmp_path / "fast.json")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_fast_worker.py"), str(r), "2", str(port), str(sf), out],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    with open(out) as fh:
        got = json.load(fh)
    eng = engine.Engine(oracle_lib.context(threads=2))
    cols = tpch.columns_for(["q3"])
    want = helpers.run_query(eng, "q3", tpch.generate(sf, tables=sorted(cols), columns=cols, threads=2))
    eng.close()
    for mode in ("hash", "range"):
        runs = {r["label"]: r for r in got[mode]}
        for label, r in runs.items():
            helpers.assert_rows_match(sorted(as_rows(r["rows"])), helpers.result_rows(want, want.columns), 1e-12, "%s / %s" % (mode, label))
            assert r["local_rows"] > 1024 and r["partitioning"] == mode, (mode, label, r["local_rows"], r["partitioning"])
        assert not runs["first"]["deferred"] or mode == "range"
        assert runs["second"]["deferred"] and runs["third"]["deferred"] and runs["last"]["deferred"], mode
    h = {r["label"]: r for r in got["hash"]}
    assert h["first"]["fast_runs"] == 0 and h["second"]["fast_runs"] == 1 and h["third"]["fast_runs"] == 2
    assert h["second"]["exchanged"] == h["first"]["exchanged"] and h["first"]["exchanged"]["probe_sent"] > 0, (h["first"]["exchanged"], h["second"]["exchanged"])
    assert h["third"]["retries"] == 0 and h["chunks too small"]["retries"] == 1 and h["after the collective re-run"]["retries"] == 1
    assert h["after the collective re-run"]["fast_runs"] == h["chunks too small"]["fast_runs"] + 1
    # K-F blocks too small never repeat the join: the device-sized runs go on, no collective re-run
    assert h["last"]["retries"] == 1 and h["last"]["fast_runs"] == h["after the collective re-run"]["fast_runs"] + 3
