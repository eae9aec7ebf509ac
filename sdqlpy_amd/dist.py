"""Multi-GPU execution of the hot-path queries: one process per GPU, torch.distributed over RCCL.

The reference has no distributed execution at all; this is the sharding scheme of
SURVEY.md §8(e), written against the same C ABI as the single-GPU path:

  q6 / q1   rows are independent: every rank runs the query on its row shard; the partial scalar /
            the <= 64 partial groups are all-gathered (a few hundred bytes) and folded in rank order.

  q3        build 1 (customer keys that pass the filter) is small: every rank compacts the keys of
            its shard (sdqh_scan_compact), the keys are all-gathered, and each rank builds the
            replicated set.  The orders <-> lineitem join is PARTITIONED ON THE BUILD-SIDE KEY
            (o_orderkey):
              * range partitioning when the ranks' build shards cover disjoint, ascending key ranges
                (dbgen data is clustered on o_orderkey, so both tables are already where they
                belong): the build side stays put, and only the probe rows whose key falls into
                another rank's range are exchanged;
              * hash partitioning (mix64(key) mod G) otherwise: build survivors and filtered probe
                rows are redistributed.
            Either way the redistribution step is a count exchange followed by one all-to-all per
            column (torch.distributed.all_to_all_single = RCCL all-to-all over xGMI), and every
            rank then probes / aggregates its own key partition.  Group keys are disjoint across
            ranks, so the result is the concatenation of the ranks' results.

  q5 / q9 / q4 / q14 / q18   the chain executor: builds over a sharded table that another shard's rows look
            up are replicated (entries all-gathered on device memory, rebuilt), joins whose two sides
            are co-partitioned on the key stay local (an all-gathered min / max test decides, so every
            rank decides alike), partial groups over a small domain and partial scalars are folded in
            rank order.  A row-keyed group-by (q18's sum per l_orderkey), its HAVING key set and a
            probe-aggregate into a co-partitioned build are local too when the shards are partitioned
            by that key: each rank finishes the groups of its key partition and the result is the
            concatenation, as for q3.  Shards that are not partitioned that way are refused (by every
            rank), as is replicating a build whose text payload refers to one rank's rows.

The collectives carry exactly the bytes that have to move; there is no collective in the data path
of q1 / q6 beyond the final few hundred bytes.  `backend` only needs all_gather, all_to_all_single
(or point-to-point for gloo) and barrier, so the same code runs under gloo on CPU tensors with the
CPU oracle behind the ABI — that is how tests/test_dist_cpu.py covers the N > 1 path without GPUs.
"""
import contextlib
import os
import sys
import time
import weakref

import numpy as np
import torch
import torch.distributed as dist

from . import abi, engine, frontend, xplan
from . import tpch_queries as Q
from .frontend import Col, FinalizeOp, PayloadField, RecordCons, ScanOp
from .result import ResultSet


# (an experiment switch, never set in production: eager collectives on the recorded streams themselves — what round 6 first shipped, and what
#  tests/dist_record_race_worker.py shows to end in a core dump when torch's watchdog polls during a recording)
_COLL_ON_RECORDED_STREAM = os.environ.get("SDQLPY_AMD_DIST_COLL_ON_RECORDED_STREAM") == "1"


class DistributedRunner:
    def __init__(self, eng, rank, world, group=None, device=None, partition="auto", prefilter=True, skip_trivial=None, device_sized=None):
        self.eng, self.ctx = eng, eng.ctx
        # a collective over a group of ONE moves nothing: skipped (every rank knows the world size: all decide alike).  False makes a
        # group of one issue them all the same (tests / bench.py --force-dist: the RCCL calls themselves on one GPU)
        self.skip_trivial = (os.environ.get("SDQLPY_AMD_DIST_TRIVIAL_COLLECTIVES", "0") != "1") if skip_trivial is None else bool(skip_trivial)
        # the hash-partitioned join's exchanges sized on the device (fixed-capacity chunks, counts in their headers, no host wait) once a
        # first run has measured them; False: every run exchanges exact sizes through the host
        self.device_sized = (os.environ.get("SDQLPY_AMD_DIST_DEVICE_SIZED", "1") != "0") if device_sized is None else bool(device_sized)
        self.fast_runs = 0                  # partitioned joins that ran with device-sized exchanges / that had to be repeated with exact sizes
        self.fast_retries = 0
        self._stat_ring, self._stat_next = [], 0
        self.rank, self.world, self.group = rank, world, group
        self.backend = dist.get_backend(group)
        # LANES (round 6): a SETTLED plan — device-sized exchanges, nothing read back — runs on one of the engine's lanes (a context of the
        # family: a stream, a pool and result blocks of its own), torch's current stream being that lane's for the duration of the launch:
        # a collective is ordered behind the kernels of ITS plan (torch makes RCCL's stream wait for, and be waited for by, the current
        # stream), and the plans of a step share the chip as on one GPU.  Every rank issues its collectives in program order, each
        # lane on a communicator of its own (below).  First (exact) runs, range
        # partitioning and everything on CPU tensors stay on lane 0.  Measured on a group of one with its collectives issued (q1+q3+q5 at
        # SF=10): with the calls issued one by one the step is the HOST's — 1.2 ms of Python and torch launching ~70 calls and 8
        # collectives (tools/dist_call_times2.py) — and lanes change nothing (1.37 against 1.21-1.38); with the settled plans RECORDED
        # (below: 0.06 ms of host time per query) the step is the device's, 1.01 ms on one stream and 0.80-0.82 on three lanes.
        # SDQLPY_AMD_DIST_LANES=0: one stream.
        self.lanes = self.backend == "nccl" and os.environ.get("SDQLPY_AMD_DIST_LANES", "1") != "0"
        if not (world == 1 and self.skip_trivial):
            if not self.lanes:
                eng.nlanes = 1                                  # collectives and kernels ordered on ONE stream
            eng.plan_graphs = 0                                 # no plan is recorded on a stream RCCL's collectives are ordered on (a stream in
                                                                # capture mode beside torch's / RCCL's use of it has never run anywhere)
        self._plan_lane, self._lane_next, self._ext_streams = {}, 0, {}
        # settled chains recorded WITH their collectives and launched by one call (_chain_device_sized); SDQLPY_AMD_DIST_GRAPHS=0: the calls are issued
        self.graphs = self.backend == "nccl" and os.environ.get("SDQLPY_AMD_DIST_GRAPHS", "1") != "0"
        self.graph_recordings = self.graph_launches = 0
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")
        self.device = device
        # ONE COMMUNICATOR PER LANE.  The recordings of different plans are launched on different streams; their collectives must not share a
        # communicator — RCCL's kernels of one communicator use the same channel buffers and flags and rely on being run one after the other,
        # in the same order on every rank, which nothing orders across streams (and across graphs only RCCL's own graph-mixing events, never
        # run here).  Every rank creates the lanes' groups here, in the same order; a plan keeps its lane (and so its communicator) for good;
        # a communicator's first collectives are the eager ones of its plans' first settled runs, before anything is recorded on it.
        self._lane_groups = {}
        if self.lanes and eng.nlanes > 1 and not (world == 1 and self.skip_trivial):
            ranks = None if group is None or group is dist.group.WORLD else dist.get_process_group_ranks(group)
            for k in range(1, eng.nlanes):
                self._lane_groups[k] = dist.new_group(ranks=ranks, backend="nccl", use_local_synchronization=ranks is not None)
            warm = torch.zeros(1, dtype=torch.int64, device=device)           # (the communicators are made by their first collective: now, on every rank alike, not inside a timed run)
            for k in sorted(self._lane_groups):
                dist.all_reduce(warm, group=self._lane_groups[k])
            torch.cuda.synchronize(device)
        # HYBRID (round 6): the HIP library behind a process group that is NOT RCCL (gloo) — several ranks sharing ONE GPU, which RCCL refuses.
        # Collective buffers are then PINNED host tensors: the library's kernels read and write them through their device-visible addresses,
        # gloo moves them between the processes, and the stream is waited for before the host (a collective, a torch op) touches what
        # kernels wrote.  Slow (PCIe) and only for tests: it is how the N > 1 kernels — multi-part packs, chunks of several sources, folds of
        # several blocks — run on the hardware where a second GPU is not to be had (tests/test_hip_parity.py).
        self.hybrid = self.backend != "nccl" and eng.ctx.library.backend_name() == "hip-gfx950"
        self.partition = partition          # "auto" | "range" | "hash"
        self.prefilter = prefilter          # hash partitioning: probe rows are tested against a replicated bitmap of ALL build keys before they travel
        self.last_partitioning = None
        self.exchanged_rows = {}
        self.exchanged_bytes = 0            # bytes this rank sent to OTHER ranks through the all-to-all of its last partitioned join
        self._plans = {}
        self._gather_bufs = {}
        self._inflight = []                 # collective buffers that queued copy_ins still read (dropped at the next run)
        self._top = None                    # (k, order) while a run carries ORDER BY ... LIMIT k
        self._partitioned_result = False    # the last run returned this rank's key partition (else the global result)
        self._local_text = set()            # ids of text arrays / dictionaries that only mean something on this rank
        self._whole_checked = {}            # ids of the "whole" argument tables already verified across the ranks
        # every collective this runner issued since the last reset_collectives(): name -> [calls, calls on device tensors, bytes]
        self.collectives = {}
        self._ext_stream = None             # RCCL: torch's view of the engine's HIP stream (see _device_order)
        self._coll_side = None              # RCCL: the stream eager collectives run on (see _coll)
        self.last_chain = None              # what the seams of the last device-sized chain did (_chain_device_sized)

    def _device_order(self):
        """RCCL: torch's current stream becomes the ENGINE's stream for the duration of a run.  A collective is then ordered on the
        device with the library's kernels before and behind it (torch makes its communication stream wait for, and be waited for by,
        the current stream): the host waits only where it needs a value — row counts that size an exchange, facts, final results —
        and no longer after every collective.  gloo / CPU tensors: everything is synchronous anyway."""
        if self.backend != "nccl":
            return contextlib.nullcontext()
        if self._ext_stream is None:
            if os.environ.get("TORCH_NCCL_AVOID_RECORD_STREAMS", "0") == "1":
                raise RuntimeError("TORCH_NCCL_AVOID_RECORD_STREAMS=1: the runner releases collective buffers in stream order and relies on torch "
                                   "recording the communication stream's use of them")
            self._ext_stream = torch.cuda.ExternalStream(int(self.ctx.stream()), device=self.device)
        return torch.cuda.stream(self._ext_stream)

    def _coll(self, fn, *args, **kwargs):
        """One collective.  RCCL, not recording: on a stream of the runner's own, forked from torch's current stream (the engine's, or a
        lane's) and joined to it again.  torch's watchdog thread polls the end event of every eager collective until it is reaped, and on
        this runtime hipEventQuery of an event that was recorded on a stream BEFORE that stream's capture began invalidates the capture —
        and the stream with it, for good (tools/exp/capture_abort.hip case 4; tools/exp_watchdog_capture.py: a core dump when the
        collectives before a recording ran on the stream that is recorded).  So no eager collective ever leaves an event on a stream that
        plans are recorded on; inside a recording the collective joins the capture on the recorded stream itself (torch does not hand
        captured work to its watchdog)."""
        if self.backend != "nccl" or self.ctx.capturing() or not self.graphs or _COLL_ON_RECORDED_STREAM:
            return fn(*args, **kwargs)                               # (not self.graphs: this runner records nothing, so no stream of its plans is ever captured)
        cur = torch.cuda.current_stream(self.device)
        side = self._coll_side
        if side is None:
            side = self._coll_side = torch.cuda.Stream(device=self.device)
        side.wait_stream(cur)
        try:
            with torch.cuda.stream(side):
                return fn(*args, **kwargs)
        finally:
            cur.wait_stream(side)

    def _lane_of(self, plan):
        """The lane a settled plan runs on (assigned round robin, kept)."""
        if not self.lanes or self.eng.nlanes <= 1:
            return 0
        k = self._plan_lane.get(id(plan))
        if k is None:
            k = self._plan_lane[id(plan)] = self._lane_next % self.eng.nlanes
            self._lane_next += 1
        return k

    @contextlib.contextmanager
    def _on_lane(self, k):
        """Run a settled plan's launch on lane k: self.ctx is that lane's context and torch's current stream its stream for the duration
        (everything the runner and the plan's seams call goes to that context; buffers are allocated — and reused — on that stream)."""
        if k == 0:
            yield
            return
        lane_ctx = self.eng.lane(k).ctx
        ext = self._ext_streams.get(k)
        if ext is None:
            ext = self._ext_streams[k] = torch.cuda.ExternalStream(int(lane_ctx.stream()), device=self.device)
        saved = (self.ctx, self._ext_stream, self.group)
        self.ctx, self._ext_stream, self.group = lane_ctx, ext, self._lane_groups.get(k, self.group)      # (the lane's own communicator: __init__)
        lane_ctx.set_option("async_copies", 1)
        try:
            with torch.cuda.stream(ext):
                yield
        finally:
            lane_ctx.set_option("async_copies", 0)
            self.ctx, self._ext_stream, self.group = saved

    def _on_engine_stream(self):
        """Collective buffers must be allocated with the engine's stream current (see _run): asserted where they are made."""
        if self.backend == "nccl":
            assert self._ext_stream is not None and torch.cuda.current_stream(self.device) == self._ext_stream, "collective buffer allocated outside _device_order()"

    def reset_collectives(self):
        self.collectives = {}

    def close(self):
        """Release what the runner holds on the device (collective buffers of the last run, cached gather buffers, torch's view of the
        engine's stream).  Call it before the engine is closed: torch must not be left with tensors that were allocated on a stream the
        library has destroyed."""
        if self.backend == "nccl" and self.ctx.handle is not None:
            self.ctx.synchronize()
        self._inflight.clear()
        self._gather_bufs.clear()
        for _, plan in self._plans.values():                   # prepared joins / chains of THIS runner (their buffers are torch tensors made on the engine's stream)
            for cache in ("_dist_prepared", "_dist_chain"):
                d = plan.__dict__.get(cache)
                if d:
                    for k in [k for k in d if k[0] == id(self)]:
                        for r in d[k].__dict__.pop("recordings", []):      # (a recording owns pool memory and names the collective buffers: it goes first)
                            r["pg"].free()
                        del d[k]
        self._plans.clear()
        self._stat_ring = []
        self._ext_stream = None
        self._ext_streams = {}
        if self._lane_groups and self.backend == "nccl":
            torch.cuda.synchronize(self.device)
        for g in self._lane_groups.values():                      # (every rank closes its runner at the same point: the groups go in the same order)
            try:
                dist.destroy_process_group(g)
            except Exception:                                     # noqa: BLE001  (the default group went first: its sub-groups went with it)
                pass
        self._lane_groups = {}

    def _empty(self, n, dtype=torch.int64):
        """A collective buffer of n elements: device memory under RCCL, pinned host memory in the hybrid mode, plain host memory on CPU."""
        if self.hybrid:
            return torch.empty(int(n), dtype=dtype).pin_memory()
        return torch.empty(int(n), dtype=dtype, device=self.device)

    def _zeros(self, n, dtype=torch.int64):
        t = self._empty(n, dtype)
        t.zero_()
        return t

    def _host_touch(self):
        """Hybrid mode: the host is about to read or write memory that queued kernels write / read — wait for the stream."""
        if self.hybrid:
            self.ctx.synchronize()

    def _note(self, name, tensor):
        self._host_touch()                                      # (every collective is announced here, right in front of it)
        rec = self.collectives.setdefault(name, [0, 0, 0])
        rec[0] += 1
        rec[1] += 1 if tensor.is_cuda else 0
        rec[2] += tensor.numel() * tensor.element_size()

    # ---- small collectives ---------------------------------------------------------------------
    def _all_gather_array(self, arr):
        """Fixed-shape numpy array -> list of the ranks' arrays (rank order).  On GPUs the staging
        tensors (pinned host in / out, device in / out) are cached per shape, the copies are
        asynchronous on torch's stream and there is exactly one stream synchronisation."""
        arr = np.ascontiguousarray(arr)
        if self.world == 1 and self.skip_trivial:
            return [arr]
        if self.backend != "nccl":
            t = torch.from_numpy(arr)
            out = [torch.empty_like(t) for _ in range(self.world)]
            self._note("all_gather", t)
            self._coll(dist.all_gather, out, t, group=self.group)
            return [o.numpy() for o in out]
        key = (arr.shape, arr.dtype.str)
        bufs = self._gather_bufs.get(key)
        if bufs is None:
            tdt = torch.from_numpy(np.zeros(1, arr.dtype)).dtype
            n = arr.size
            bufs = (torch.empty(n, dtype=tdt).pin_memory(), torch.empty(n, dtype=tdt, device=self.device),
                    torch.empty(n * self.world, dtype=tdt, device=self.device), torch.empty(n * self.world, dtype=tdt).pin_memory())
            self._gather_bufs[key] = bufs
        h_in, d_in, d_out, h_out = bufs
        h_in.numpy()[:] = arr.reshape(-1)
        d_in.copy_(h_in, non_blocking=True)
        self._note("all_gather", d_in)
        self._coll(dist.all_gather_into_tensor, d_out, d_in, group=self.group)
        h_out.copy_(d_out, non_blocking=True)
        self.ctx.synchronize()                   # torch's current stream IS the engine's (_device_order): its wait spins on a word the stream writes, where
                                                 # torch's stream wait sleeps on an interrupt — tens of microseconds per collective, and a step has about ten
        flat = h_out.numpy().copy()
        return [flat[r * arr.size:(r + 1) * arr.size].reshape(arr.shape) for r in range(self.world)]

    def _all_gather_varlen(self, arr):
        """1-d int64 array of any length per rank -> concatenation in rank order."""
        n = np.array([len(arr)], np.int64)
        sizes = [int(x[0]) for x in self._all_gather_array(n)]
        m = max(sizes + [1])
        pad = np.zeros(m, np.int64)
        pad[: len(arr)] = arr
        parts = self._all_gather_array(pad)
        return np.concatenate([p[:s] for p, s in zip(parts, sizes)]) if sum(sizes) else np.zeros(0, np.int64)

    def _all_gather_columns(self, cols, n):
        """Concatenate int64 / f64 Columns of n rows per rank over all ranks (rank order), staying in
        device memory: the sizes are exchanged once, then ONE padded all_gather carries every column
        (column-major inside each rank's slot)."""
        sizes = [int(x[0]) for x in self._all_gather_array(np.array([n], np.int64))]
        total, m = sum(sizes), max(sizes + [1])
        self._last_gather_max = max(sizes + [0])                             # (every rank gathered the same sizes: the bound of the next run's chunk)
        k = len(cols)
        self._on_engine_stream()
        send = self._empty(m * k)     # padding rows are never read back; no fill kernel
        if n:                                                                # on torch's stream to race the library's copies
            for j, col in enumerate(cols):
                self.ctx.copy_out(col, 0, n, send.data_ptr() + j * m * 8)      # queued ("async_copies"): the collective is ordered behind them on the stream
        recv = self._empty(m * k * self.world)
        self._note("all_gather", send)
        if self.backend == "nccl":
            self._coll(dist.all_gather_into_tensor, recv, send, group=self.group)
        else:
            self._coll(dist.all_gather, list(recv.view(self.world, m * k).unbind(0)), send, group=self.group)
        outs = []
        for j, col in enumerate(cols):
            out = self.ctx.alloc(total, col.dtype).mark_transient()
            at = 0
            for r, sz in enumerate(sizes):
                if sz:
                    self.ctx.copy_in(out, at, sz, recv.data_ptr() + (r * k + j) * m * 8)
                    at += sz
            outs.append(out)
        self._inflight.extend([send, recv])                                  # queued copies read / wrote them; released at the next run
        return outs, total

    def _all_gather_column(self, col, n):
        outs, total = self._all_gather_columns([col], n)
        return outs[0], total

    def _all_to_all_counts(self, counts):
        send = torch.from_numpy(np.ascontiguousarray(counts, np.int64)).to(self.device)
        recv = torch.empty_like(send)
        self._a2a(recv, send, [1] * self.world, [1] * self.world)
        return recv.cpu().numpy()

    def _a2a(self, recv, send, out_splits, in_splits):
        """all_to_all_single; gloo has no all-to-all, so there it is spelled as isend / irecv pairs."""
        self._note("all_to_all", send)
        if self.backend == "nccl":
            self._coll(dist.all_to_all_single, recv, send, out_splits, in_splits, group=self.group)
            return
        so = np.concatenate([[0], np.cumsum(in_splits)]).astype(int)
        ro = np.concatenate([[0], np.cumsum(out_splits)]).astype(int)
        recv[ro[self.rank]:ro[self.rank + 1]] = send[so[self.rank]:so[self.rank + 1]]
        ops = []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            if in_splits[peer]:
                ops.append(dist.P2POp(dist.isend, send[so[peer]:so[peer + 1]], peer, group=self.group))
            if out_splits[peer]:
                ops.append(dist.P2POp(dist.irecv, recv[ro[peer]:ro[peer + 1]], peer, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def _probe_program(self, st, probes):
        """The probe side's filter + semi-join as a row program whose key is the probe key and whose values are the value operands (8-byte
        bit patterns), or None when the loop has no such program (text / column-vs-column conditions, too many operands)."""
        if len(st.ops_c) > abi.MAX_PAYLOAD or st.flt_c._keep[2] or st.flt_c._keep[3]:
            return None
        P = abi.Program()
        gates = []
        for f in _ipreds(st.flt_c):
            c = P.op(abi.X_COL, abi.T_I64, col=f.col_obj)
            if f.lo > abi.INT64_MIN:
                gates.append(P.op(abi.X_GE, abi.T_BOOL, a=c, b=P.op(abi.X_CONST, abi.T_I64, imm_i=f.lo)))
            if f.hi < abi.INT64_MAX:
                gates.append(P.op(abi.X_LE, abi.T_BOOL, a=c, b=P.op(abi.X_CONST, abi.T_I64, imm_i=f.hi)))
        for col, lo, hi in _fpreds(st.flt_c):
            c = P.op(abi.X_COL, abi.T_F64, col=col)
            if lo > -np.inf:
                gates.append(P.op(abi.X_GE, abi.T_BOOL, a=c, b=P.op(abi.X_CONST, abi.T_F64, imm_f=lo)))
            if hi < np.inf:
                gates.append(P.op(abi.X_LE, abi.T_BOOL, a=c, b=P.op(abi.X_CONST, abi.T_F64, imm_f=hi)))
        key = P.op(abi.X_COL, abi.T_I64, col=st.key_c)
        for tbl, kcol in probes:
            k = key if kcol is st.key_c else P.op(abi.X_COL, abi.T_I64, col=kcol)
            gates.append(P.op(abi.X_LOOKUP, abi.T_BOOL, a=k, table=tbl))
        P.gates, P.key = gates, key
        P.vals = [P.op(abi.X_COL, abi.T_F64 if c.dtype == abi.F64 else abi.T_I64, col=c) for c in st.ops_c]
        return P

    def _compact_probe_rows(self, st, probes):
        """The probe side's rows that pass its filter and whose key some rank holds (`probes`: the replicated key set), as columns
        [key, operands ...] of 8-byte bit patterns.  As a row program (sdqh_xcompact: on the GPU the value-queue stage kernel — every
        column streamed at its tightest encoding, the key set tested on streamed keys, survivors' values queued and stored whole;
        the stage is never indexed, so equal keys all stay).  Shapes the program route refuses: sdqh_scan_compact."""
        ctx = self.ctx
        P = self._probe_program(st, probes) if probes else None
        if P is not None:
            try:
                return ctx.xcompact(st.nc, P)
            except abi.SdqhError as exc:
                if exc.code != abi.ERR_UNSUPPORTED:
                    raise
        return ctx.scan_compact(st.nc, st.flt_c, probes, [st.key_c] + st.ops_c)

    def _exchange(self, nrows, key, cols, range_upper=None, dtypes=None):
        """Partition `nrows` rows of `cols` by part(key) (hash, or range_upper's ranges) and move every part to its rank.  The
        partitioning pass writes straight into the collective's send buffer in the all-to-all's own layout (sdqh_partition_pack: the
        chunk for rank d holds every column's rows for d), so a redistribution step is ONE all_to_all_single whatever the number of
        columns, and the received buffer is taken apart into columns by one kernel (sdqh_unpack_parts).  The G x G count matrix is
        gathered first (G * 8 bytes per rank: every rank learns what it will receive, and all ranks agree when nothing at all has to
        move and skip the data collective).  Returns (received Columns, their row count, rows sent to every rank)."""
        k = len(cols)
        self._on_engine_stream()
        send = self._empty(max(nrows * k, 1))
        counts = self.ctx.partition_pack(nrows, key, self.world, cols, send.data_ptr(), range_upper)      # (waits for the counts: they size the exchange)
        matrix = np.stack(self._all_gather_array(counts))                                              # [source, dest]
        recv_counts = matrix[:, self.rank]
        self._last_matrix_max = int(matrix.max()) if matrix.size else 0          # (every rank sees the same matrix: the bound of the next run's chunks)
        n_recv = int(recv_counts.sum())
        self.exchanged_bytes += 8 * k * (nrows - int(counts[self.rank]))
        if self.world == 1 and self.skip_trivial:
            recv = send                                                       # a group of one: what was packed for rank 0 is what rank 0 receives
        else:
            recv = self._empty(max(n_recv * k, 1))
        if matrix.sum() > 0 and recv is not send:
            self._a2a(recv[:n_recv * k], send[:nrows * k], [int(c) * k for c in recv_counts], [int(c) * k for c in counts])
        out, n = self.ctx.unpack_parts(recv.data_ptr(), recv_counts, dtypes or [c.dtype for c in cols])      # (dtypes: entries of a staged table come back as raw 8-byte columns)
        self._inflight.extend([send, recv])                                  # queued kernels read them; released at the next run
        return out, n, counts

    # ---- queries ---------------------------------------------------------------------------------
    def _resolve(self, query, db):
        """(decorated function, plan, positional table arguments) of a query given by registry name or
        as the decorated function itself; `db` is a {table name: table} dict or the positional list."""
        fn = Q.QUERIES[query] if isinstance(query, str) else query
        key = id(fn)
        if key not in self._plans:
            self._plans[key] = (fn, frontend.lower_function(fn, getattr(fn, "__sdql_in_type__", None)))
        plan = self._plans[key][1]
        args = [db[t] for t in Q.tables_of(fn)] if isinstance(db, dict) else list(db)
        if len(args) != len(plan.params):
            raise TypeError("%s expects %d tables, got %d" % (plan.name, len(plan.params), len(args)))
        return fn, plan, args

    def _whole_params(self, fn, plan, args, whole_tables):
        """Parameters whose table every rank holds completely.  Explicit `whole_tables` (database table
        names) wins; otherwise the tables say it themselves: a table cut by tpch.generate(shard=...) /
        sdql_lib.shard_rows carries `.shard = (rank, world)`, every other table is whole."""
        if whole_tables is not None:
            whole = {p for p, t in zip(plan.params, Q.tables_of(fn)) if t in whole_tables}
        else:
            whole = {p for p, a in zip(plan.params, args) if getattr(a, "shard", None) is None}
            if self.world > 1 and len(whole) == len(plan.params):
                # the mark is an attribute of the table object: a table rebuilt from a shard's columns has lost it, and
                # every rank would then answer for its slice alone with no collective — a silently partial result
                raise ValueError("%s: with %d ranks and no `whole_tables`, at least one argument must carry a row-shard mark "
                                 "(sdql_lib.shard_rows / tpch.generate(shard=...)); pass whole_tables=[...] to say which "
                                 "tables every rank holds completely" % (plan.name, self.world))
        self._verify_whole(plan, args, whole)
        return whole

    def _verify_whole(self, plan, args, whole):
        """A table taken as whole must be the same table on every rank: row count and a checksum of its first and
        last rows are all-gathered once per table object, and a disagreement (a shard that lost its mark) raises
        on every rank alike instead of producing a partial answer."""
        if self.world == 1:
            return
        todo = [(p, a) for p, a in zip(plan.params, args) if p in whole and self._whole_checked.get(id(a)) is not a]
        if not todo:
            return
        facts = np.zeros((len(todo), 2), np.int64)
        for i, (_, a) in enumerate(todo):
            data = a.getContainer()["data"]
            n = len(data[0]) if data else 0
            h = 1469598103934665603                      # FNV-style fold of the edge rows' bytes, as a Python int
            for col in data:
                if len(col):
                    edge = np.concatenate([np.ascontiguousarray(col[:64]).view(np.uint8), np.ascontiguousarray(col[-64:]).view(np.uint8)])
                    h = ((h ^ int(edge.astype(np.uint64).sum())) * 1099511628211) & 0x7FFFFFFFFFFFFFFF
            facts[i] = (n, h)
        parts = self._all_gather_array(facts)
        bad = [p for i, (p, _) in enumerate(todo) if any((part[i] != parts[0][i]).any() for part in parts)]
        if bad:
            raise ValueError("%s: table(s) %s are taken as whole on every rank but differ between ranks — mark row shards with "
                             "sdql_lib.shard_rows(table, rank, world) or name the whole tables explicitly" % (plan.name, ", ".join(sorted(bad))))
        for _, a in todo:
            self._whole_checked[id(a)] = a

    @staticmethod
    def _shape(plan):
        """Which distributed plan a query gets, decided from its loops alone:
        "scalar"   one scalar sum over one table                      (q6)
        "groups"   one small-domain group-by + its reshape            (q1)
        "join"     build A -> build B (semi-join A) -> probe-aggregate C into B -> reshape   (q3)
        "chain"    everything else the chain executor understands     (q5, q9, q4, q14, q18)"""
        ops = plan.ops
        if len(ops) == 1 and isinstance(ops[0], ScanOp) and ops[0].kind == "scalar" and not any(isinstance(c, frontend.Contains) for c in ops[0].conds):
            return "scalar"
        if len(ops) == 2 and isinstance(ops[0], ScanOp) and isinstance(ops[1], FinalizeOp) and ops[0].kind == "dict" and not ops[0].unique \
                and ops[0].probe is None and not any(isinstance(c, frontend.Contains) for c in ops[0].conds):
            found = []
            engine._walk_lookups(ops[0].key, found); engine._walk_lookups(ops[0].val, found)
            if not found:
                return "groups"
        if _is_join_shape(ops):
            return "join"
        return "chain"

    def run(self, query, db, whole_tables=None, top=None):
        """See _run.  Column copies to / from collective buffers are only queued while a run is in
        progress (option "async_copies"); the runner synchronises once per batch."""
        return self._guarded(lambda: self._run(query, db, whole_tables, top))

    def _guarded(self, fn):
        self.ctx.set_option("async_copies", 1)
        ok = False
        try:
            with self._device_order():
                res = fn()
            ok = True
            return res
        finally:
            # RCCL: nothing to wait for — what the host needed it has waited for, a result launched and not looked at yet finishes when
            # it is (result.DeferredResultSet), and the collective buffers of this run are released at the start of the next one, in
            # stream order (torch's allocator reuses a block on the stream it was allocated on: the engine's).  The wait stays for
            # CPU tensors (the library's queued copies read them) and after an error.
            if self.backend != "nccl" or not ok:
                self.ctx.synchronize()
            self.ctx.set_option("async_copies", 0)

    def _run(self, query, db, whole_tables=None, top=None):
        """Run a query (registry name or decorated function) on this rank's shard `db`; returns this
        rank's share of the result (a scalar: the global value on every rank; group-bys over a small
        domain (q1, q5, q9): the global groups on every rank; the partitioned join (q3) and chains that
        end in a local aggregation (q18): the groups of this rank's key partition).  `whole_tables`
        names the tables every rank holds completely; by default the tables' own `.shard` marks decide.
        top = (k, [(column, "asc" | "desc")]) adds ORDER BY ... LIMIT k: every rank returns the same
        global first k rows (partitioned results: each rank's device top-k, k rows per rank gathered
        and ordered again)."""
        # the collective buffers of the previous run: released now.  gloo: that run ended synchronised.  RCCL: it did not — the tensors
        # were allocated with the ENGINE's stream current (_device_order), torch's caching allocator hands a freed block out again on
        # that same stream only, behind the kernels queued there that still read it, and a tensor a collective used on RCCL's own
        # stream carries torch's record_stream mark (TORCH_NCCL_AVOID_RECORD_STREAMS=1 would void that: refused in _device_order)
        self._inflight.clear()
        fn, plan, args = self._resolve(query, db)
        whole = self._whole_params(fn, plan, args, whole_tables)
        shape = self._shape(plan)
        if top is not None:
            k, order = int(top[0]), [(str(n), str(d)) for n, d in top[1]]
            if shape == "scalar":
                raise frontend.UnsupportedQuery("top(k) applies to queries that end in a result set")
            # a result partitioned over the ranks: top-k per rank first, k rows per rank gathered and
            # ordered again.  Global results are ordered as they are.
            self._top, self._partitioned_result = (k, order), False
            try:
                local = self._run(query, db, whole_tables)
            finally:
                self._top = None
            if not self._partitioned_result:
                return local.top(k, order)
            return self._gather_result(local).top(k, order)
        self._partitioned_result = False
        if len(whole) == len(plan.params):
            # nothing is sharded: every rank holds the whole database and computes the whole answer
            return engine.execute_plan(self.eng, plan, args, self._top, lane=0)
        if self.world == 1 and self.skip_trivial and shape != "join":
            # a group of ONE: this rank's shards are the tables — nothing to replicate, nothing to merge; the engine's own plan, on one of
            # its lanes, its result launched and not waited for (the partitioned join keeps its exchange: that is the step the metric names)
            return engine.execute_plan(self.eng, plan, args, self._top)
        if shape == "scalar":
            local = engine.execute_plan(self.eng, plan, args, lane=0)
            parts = self._all_gather_array(np.array([local], np.float64))
            total = 0.0
            for p in parts:                                   # fixed (rank) order
                total += float(p[0])
            return total
        if shape == "groups":
            if self.device_sized and self._top is None and os.environ.get("SDQLPY_AMD_DIST_GROUPS_AS_CHAIN", "1") != "0":
                # a chain with nothing to replicate: its first run merges the partial groups on the host and compares the ranks' key
                # encodings; from the second on they are folded on the device behind one all-gather, nothing waited for
                return self._sharded_chain(plan, args, whole, None)
            return self._row_sharded_groupby(plan, args)
        if shape == "join":
            a_op, b_op, c_op = plan.ops[0], plan.ops[1], plan.ops[2]
            if b_op.table in whole or c_op.table in whole:
                # decided from the arguments' marks, the same on every rank: all raise alike
                raise frontend.UnsupportedQuery("the partitioned join needs its build side '%s' and probe side '%s' row-sharded; a table held "
                                                "whole on every rank would put every group on every rank" % (b_op.table, c_op.table))
            self._partitioned_result = True
            return self._partitioned_join(plan, args, a_whole=a_op.table in whole)
        return self._sharded_chain(plan, args, whole, self._top)

    def _gather_result(self, local):
        """The ranks' ResultSet rows concatenated on every rank (rank order, then each rank's own
        order): one size exchange and one padded fixed-shape all_gather per column of raw bytes — text
        travels as its UCS-4 code units, nothing is pickled."""
        n = local.size()
        sizes = [int(x[0]) for x in self._all_gather_array(np.array([n], np.int64))]
        m = max(sizes + [1])
        arrays = []
        for a in local.arrays:
            a = np.ascontiguousarray(a)
            width = np.array([a.dtype.itemsize], np.int64)
            widths = [int(x[0]) for x in self._all_gather_array(width)]
            w = max(widths)
            if a.dtype.kind == "U" and a.dtype.itemsize != w:   # text columns decoded from differently sized dictionaries
                a = a.astype("<U%d" % (w // 4))
            elif a.dtype.kind != "U" and len(set(widths)) != 1:
                raise RuntimeError("ranks disagree on a result column's type")
            words = (w + 7) // 8
            buf = np.zeros((m, words), np.int64)
            if n:
                buf.view(np.uint8).reshape(m, words * 8)[:n, :w] = a.view(np.uint8).reshape(n, w)
            parts = self._all_gather_array(buf)
            rows = [p.view(np.uint8).reshape(m, words * 8)[:sz, :w] for p, sz in zip(parts, sizes)]
            flat = np.ascontiguousarray(np.concatenate(rows, axis=0))
            arrays.append(flat.view(a.dtype).reshape(sum(sizes)))
        return ResultSet(local.columns, arrays)

    # ---- multi-join chains (q5, q9): replicate what is probed across shards, keep co-partitioned joins local ----
    def _prepare_chain(self, plan, args, whole):
        eng = self.eng
        tabs = {p: engine.HostTable(p, a) for p, a in zip(plan.params, args)}
        st = type("ChainState", (), {})()
        st.args, st.generation = tuple(args), eng.generation
        st.steps, st.replicate = [], {}
        scan_ops = [o for o in plan.ops if isinstance(o, ScanOp)]
        built_by = {o.out: o for o in scan_ops if o.kind == "dict" and o.unique}
        # which scans look a built table up, and by which key expression
        consumers = {name: [] for name in built_by}
        for o in scan_ops:
            found = []
            if o.probe is not None:
                engine._walk_lookups(o.probe, found)
            for c in list(o.conds) + [c for _, _, fc in (o.fields or []) for c in fc]:
                if isinstance(c, frontend.Contains):
                    engine._walk_lookups(c.lookup, found)
            if o.kind == "dict":
                engine._walk_lookups(o.key, found)
            if o.val is not None and not isinstance(o.val, frontend.Const):
                engine._walk_lookups(o.val, found)
            for _, e, _ in (o.fields or []):
                engine._walk_lookups(e, found)
            for lk in found:
                if lk.dict_name in consumers:
                    consumers[lk.dict_name].append((o, lk))
        # HAVING key sets (SelectKeysOp over a row-keyed group-by): usable where they are computed only if every
        # consumer's keys live on the same rank as the groups — the co-partitioning test below, with the
        # group-by's key column as the build side; a key set cannot be replicated
        st.local_only = {}                                       # name -> why it must stay local
        group_ops = {o.out: o for o in scan_ops if o.kind == "dict" and not o.unique and o.probe is None
                     and isinstance(o.key, Col) and o.table not in whole}
        select_src = {o.out: group_ops[o.source] for o in plan.ops if isinstance(o, frontend.SelectKeysOp) and o.source in group_ops}
        facts_req = []                                           # (table name, build col array, probe col array)
        for name, gop in list(group_ops.items()) + list(select_src.items()):
            garr = tabs[gop.table].array(gop.key.name, gop)
            st.local_only[name] = "the groups of '%s' are not partitioned by key over the ranks" % name
            facts_req.append((name, garr, garr))
        for o in scan_ops:
            found = []
            for c in list(o.conds) + [c for _, _, fc in (o.fields or []) for c in fc]:
                if isinstance(c, frontend.Contains):
                    engine._walk_lookups(c.lookup, found)
            if o.probe is not None:
                engine._walk_lookups(o.probe, found)
            for lk in found:
                if lk.dict_name in select_src:
                    gop = select_src[lk.dict_name]
                    if o.table in whole or not isinstance(lk.key, Col):
                        st.local_only[lk.dict_name] = "'%s' is looked up from a table every rank holds whole" % lk.dict_name
                        facts_req.append((lk.dict_name, np.zeros(0, np.int64), np.zeros(0, np.int64)))
                    else:
                        facts_req.append((lk.dict_name, tabs[gop.table].array(gop.key.name, gop), tabs[o.table].array(lk.key.name, o)))
        for name, bop in built_by.items():
            if bop.table in whole:
                st.replicate[name] = False                       # identical on every rank already
                continue
            need = False
            for cop, lk in consumers[name]:
                local_ok = (cop.table not in whole and isinstance(bop.key, Col) and isinstance(lk.key, Col)
                            and tabs[bop.table].cols.get(bop.key.name) is not None and tabs[cop.table].cols.get(lk.key.name) is not None)
                if local_ok:
                    facts_req.append((name, tabs[bop.table].array(bop.key.name, bop), tabs[cop.table].array(lk.key.name, cop)))
                else:
                    need = True
            st.replicate[name] = need
        # co-partitioning test: every rank's probe keys must lie inside its own build-key range,
        # and the build ranges must be disjoint and ascending (so no other rank holds a matching key)
        for name, barr, carr in facts_req:
            b = (int(barr.min()), int(barr.max())) if len(barr) else (abi.INT64_MAX, abi.INT64_MIN)
            c = (int(carr.min()), int(carr.max())) if len(carr) else b
            facts = self._all_gather_array(np.array([b[0], b[1], c[0], c[1], len(barr), len(carr)], np.int64))
            ok = all(int(f[4]) > 0 for f in facts) and all(int(facts[i][1]) < int(facts[i + 1][0]) for i in range(self.world - 1)) \
                and all(int(f[5]) == 0 or (int(f[2]) >= int(f[0]) and int(f[3]) <= int(f[1])) for f in facts)
            if not ok:
                st.replicate[name] = True
        st.unsupported = None
        checked = {}
        for name, _, _ in facts_req:
            if name in st.local_only:
                checked[name] = checked.get(name, True) and not st.replicate.get(name, False)
        for name, ok in checked.items():
            if not ok:
                st.unsupported = st.local_only[name]
            st.replicate.pop(name, None)
        # a probe-aggregate folds its groups into the probed table: that table carries accumulators, and the
        # join must be local (co-partitioned) — each rank then holds the finished groups of its key partition
        accumulate_into = {op.probe.dict_name for op in scan_ops
                           if op.kind == "dict" and not op.unique and op.probe is not None and op.probe.dict_name in built_by
                           and engine._is_simple(op, tabs[op.table], [c.lookup for c in op.conds if isinstance(c, frontend.Contains)])}
        for name in accumulate_into:
            if st.replicate.get(name) or built_by[name].table in whole:
                st.unsupported = "the probe-aggregate into '%s' is not local to the ranks' shards: it needs the partitioned-join plan" % name
        # a loop whose groups are fields of the matched entry but which looks other tables up as well (Q5's lineitem loop) runs as the
        # program it is on one GPU — a handful of groups, merged across ranks like any group-by — where its join is local; where the probed
        # table is a replica the fixed lookup-aggregate call does it, as before
        accumulate_into |= {op.probe.dict_name for op in scan_ops
                            if op.kind == "dict" and not op.unique and op.probe is not None and op.probe.dict_name in built_by
                            and xplan.groups_by_entry(op) and not st.replicate.get(op.probe.dict_name) and built_by[op.probe.dict_name].table not in whole}
        # builds that only answer `tbl[k] != None` are key sets, as on one GPU; one that other shards' rows look up is replicated
        # through its exact bitmap over the global key range (gathered here, once) — unless that range does not suit a bitmap: then
        # it stays an ordinary table whose entries travel
        member_only, st.key_range = set(), {}
        for name in engine._membership_only(plan):
            if not st.replicate.get(name, False):
                member_only.add(name)
                continue
            bop = built_by.get(name)
            if bop is None or not isinstance(bop.key, Col) or tabs[bop.table].cols.get(bop.key.name) is None:
                continue
            karr = tabs[bop.table].array(bop.key.name, bop)
            mine = eng.column(karr).minmax() if len(karr) else (abi.INT64_MAX, abi.INT64_MIN)
            facts = self._all_gather_array(np.array(mine, np.int64))
            lo, hi = min(int(f[0]) for f in facts), max(int(f[1]) for f in facts)
            if lo <= hi and hi - lo + 1 <= (1 << 31):
                member_only.add(name)
                st.key_range[name] = (lo, hi)
        st.member_only = member_only
        # replicated tables WITH payload, keyed by one column: the global key range too (gathered once) — the replica is then rebuilt by the
        # value-queue build, told its bounds, instead of the fixed-shape unique build with its minimum / maximum pass
        st.table_range = {}
        for name, need in st.replicate.items():
            bop = built_by.get(name)
            if not need or name in member_only or bop is None or not isinstance(bop.key, Col) or tabs[bop.table].cols.get(bop.key.name) is None:
                continue
            karr = tabs[bop.table].array(bop.key.name, bop)
            if karr.dtype != np.int64:
                continue
            mine = eng.column(karr).minmax() if len(karr) else (abi.INT64_MAX, abi.INT64_MIN)
            facts = self._all_gather_array(np.array(mine, np.int64))
            lo, hi = min(int(f[0]) for f in facts), max(int(f[1]) for f in facts)
            if lo <= hi:
                st.table_range[name] = (lo, hi)
        # ... and those keyed by two columns (travelling packed): the global range of each part, gathered once — the rebuilt replica is
        # told them (sdqh_column_set_bounds) instead of running a minimum / maximum pass over each part, and reading it back, every run
        st.part_ranges = {}
        for name, need in st.replicate.items():
            bop = built_by.get(name)
            if not need or name in member_only or bop is None or not isinstance(bop.key, RecordCons) or len(bop.key.fields) != 2:
                continue
            parts = [e for _, e in bop.key.fields]
            if not all(isinstance(e, Col) and tabs[bop.table].cols.get(e.name) is not None and tabs[bop.table].cols[e.name].dtype == np.int64 for e in parts):
                continue
            mine = []
            for e in parts:
                arr = tabs[bop.table].array(e.name, bop)
                mine += list(eng.column(arr).minmax()) if len(arr) else [abi.INT64_MAX, abi.INT64_MIN]
            facts = self._all_gather_array(np.array(mine, np.int64))
            rng = [(min(int(f[2 * j]) for f in facts), max(int(f[2 * j + 1]) for f in facts)) for j in range(2)]
            if all(lo <= hi for lo, hi in rng):
                st.part_ranges[name] = rng
        for op in plan.ops:
            if isinstance(op, ScanOp):
                st.steps.append((op, engine._prepare_scan(eng, op, tabs[op.table], accumulate_into, op.out in member_only)))
            else:
                st.steps.append((op, None))
        st.sharded = {op.out: (op.table not in whole) for op in scan_ops}
        # ---- the device-sized form of this chain (_chain_device_sized): which exchanges it has, whether the engine's own prepared plan
        # can carry them at its seams.  Everything here follows from the plan and from facts every rank gathered: all decide alike.
        st.whole = set(whole)
        st.caps, st.measured, st.fold_fp, st.partitioned_result = {}, False, {}, False
        st.fast_tables = [op.out for op in scan_ops if st.replicate.get(op.out) and op.out not in st.key_range]
        engine_acc = {op.probe.dict_name for op in scan_ops
                      if op.kind == "dict" and not op.unique and op.probe is not None
                      and (engine._is_simple(op, tabs[op.table], [c.lookup for c in op.conds if isinstance(c, frontend.Contains)]) or xplan.groups_by_entry(op))}
        st.fast_ok = (len(st.fast_tables) <= 4
                      and all(name in st.table_range or name in st.part_ranges for name in st.fast_tables)
                      and all(name in st.key_range for name in member_only if st.replicate.get(name))
                      and engine_acc == set(accumulate_into)
                      and not any(isinstance(op, (frontend.SelectKeysOp,)) for op in plan.ops)
                      and all(isinstance(op, (ScanOp, FinalizeOp, frontend.ScalarExprOp)) for op in plan.ops))
        # text that only means something on this rank: string columns of sharded tables and their dictionaries
        st.local_text = set()
        for p_, t_ in tabs.items():
            if p_ not in whole:
                for arr in t_.cols.values():
                    if arr is not None and arr.dtype.kind == "U":
                        st.local_text.add(id(arr))
                        hit = eng._dicts.get(id(arr))
                        if hit is not None and hit[0] is arr and hit[2] is not None:
                            st.local_text.add(id(hit[2]))
        return st

    def _replicate_table(self, bt, key_range=None, table_range=None, part_ranges=None, measure=None):
        """All ranks' entries of a built table on every rank, without leaving device memory.  A key set (key_range given): its exact
        bitmap over the global key range, one collective, the replica a key set again — the layout the loops that test it are
        specialised on.  A table with payload: entries -> all-gather -> rebuild, a composite key (travelling packed) from its two
        parts through sdqh_build, which lays it out as the single-GPU plan's build would (linearised rectangle / high-part bitmap in
        front of the hash: what the final loops of Q5 / Q9 are tuned for)."""
        ctx = self.ctx
        if key_range is not None:
            table, words = self._replicated_key_set(bt.table, key_range, disjoint=False)
            bt.table.free()
            bt.table = table
            bt._keep = words
            return bt
        local = self._local_text          # text arrays / dictionaries of this rank's shards (see _prepare_chain)
        if any(id(d) in local for d in list(bt.decoders.values()) + list(bt.field_decoders.values()) if d is not None):
            raise frontend.UnsupportedQuery("'%s' carries text as references to this rank's rows: it cannot be replicated yet "
                                            "(hold its table whole on every rank)" % bt.key_name)
        cols, n = ctx.table_entries(bt.table)
        gathered, total = self._all_gather_columns(cols, n)
        if measure is not None:
            measure(self._last_gather_max)
        for c in cols:
            c.free()
        if bt.key_parts is not None and total:
            hi, lo = ctx.unpack2(gathered[0], total)
            if part_ranges is not None:                                     # (a filtered build's keys are a subset of its columns' values: bounds all the same)
                hi.set_bounds(*part_ranges[0]); lo.set_bounds(*part_ranges[1])
            table = ctx.build(total, abi.make_filter(), [], [abi.src_col(hi), abi.src_col(lo)], [abi.src_col(c) for c in gathered[1:]])
            gathered = list(gathered) + [hi, lo]
        elif table_range is not None and total:
            table = self._build_from_columns(total, gathered, table_range, accumulate=False)
        else:
            table = ctx.hash_build_unique(total, abi.make_filter(), [], gathered[0], gathered[1:])
        new = engine.BuiltTable(table, bt.key_name, bt.key_is_record, bt.val_fields, bt.val_is_record, bt.payload_dtypes)
        new.decoders, new.key_parts, new.field_decoders = bt.decoders, bt.key_parts, bt.field_decoders
        new.key_decoder = bt.key_decoder
        new._keep = gathered
        bt.table.free()
        return new

    def _sharded_chain(self, plan, args, whole, top=None, waited=False):
        cache = plan.__dict__.setdefault("_dist_chain", {})
        key = (id(self), tuple(sorted(whole))) + tuple(id(a) for a in args)
        st = cache.get(key)
        if st is None or st.generation != self.eng.generation or any(x is not y for x, y in zip(st.args, args)):
            st = cache[key] = self._prepare_chain(plan, args, whole)
        if st.unsupported:
            raise frontend.UnsupportedQuery(st.unsupported)          # decided from all-gathered facts: every rank raises
        self._local_text = st.local_text
        if self.device_sized and not waited and top is None and st.fast_ok and st.measured and not (self.world == 1 and self.skip_trivial):
            try:
                lane = self._lane_of(plan)
                with self._on_lane(lane):
                    return self._chain_device_sized(st, plan, args, lane)
            except abi.SdqhError as exc:
                if exc.code != abi.ERR_UNSUPPORTED:
                    raise
                st.fast_ok = False                   # (decided by the tables' layouts, the same on every rank: all fall back alike)
        env = {}
        caps = {}
        try:
            for op, step in st.steps:
                if isinstance(op, ScanOp):
                    res = step(env)
                    if isinstance(res, engine.BuiltTable) and st.replicate.get(op.out) and not (self.world == 1 and self.skip_trivial):
                        res = self._replicate_table(res, st.key_range.get(op.out), st.table_range.get(op.out), st.part_ranges.get(op.out),
                                                    measure=lambda most, name=op.out: caps.__setitem__(name, _chunk_bound(most, st.caps.get(name))))
                    elif isinstance(res, engine.DictResult) and st.sharded[op.out]:
                        res = self._merge_groups(res, st, op.out)
                    elif isinstance(res, float) and st.sharded[op.out]:
                        res = sum(float(p[0]) for p in self._all_gather_array(np.array([res], np.float64)))
                    elif isinstance(res, dict) and st.sharded[op.out]:       # scalar record: one partial sum per field, folded in rank order
                        names = sorted(res)
                        parts = self._all_gather_array(np.array([res[k] for k in names], np.float64))
                        res = {k: sum(float(p[j]) for p in parts) for j, k in enumerate(names)}
                    elif isinstance(res, tuple) and st.sharded[op.out] and op.probe is None and op.out not in st.local_only:
                        raise frontend.UnsupportedQuery("a group-by over a large key domain must be keyed by one column of a sharded table")
                    env[op.out] = res                            # ("aggregated", table): the finished groups of this rank's key partition
                elif isinstance(op, frontend.SelectKeysOp):
                    env[op.out] = engine._select_keys(self.eng, op, env)
                elif isinstance(op, FinalizeOp):
                    local_groups = isinstance(env.get(op.source), tuple)           # finished groups of this rank's key partition
                    if local_groups:
                        agg_table = env[env[op.source][1]]
                        if not any(src == "key" for _, src in agg_table.agg[0]):
                            # Q10: groups named by fields of the matched entry (a customer's orders) are not
                            # partitioned with the build key — the same group can finish on several ranks
                            raise frontend.UnsupportedQuery("groups keyed by fields of the matched entry may span ranks: "
                                                            "their partial sums would have to be merged by key (not distributed yet)")
                    if op.out == plan.result:                                      # (aggregating a table every rank holds whole gives every rank all groups)
                        self._partitioned_result = local_groups and st.sharded.get(op.source, True)
                    env[op.out] = engine._finalize(self.eng, op, env, top if op.out == plan.result and local_groups else None)
                elif isinstance(op, frontend.ScalarExprOp):
                    env[op.out] = engine._eval_scalar_expr(op.expr, env, op.lineno)
                else:
                    raise frontend.UnsupportedQuery("%s is not part of the distributed chain plan yet" % type(op).__name__)
            st.caps.update(caps)
            st.partitioned_result = self._partitioned_result
            st.measured = True                       # (a complete run with exact sizes: the next one may move fixed-capacity chunks)
            return env[plan.result]
        finally:
            for v in env.values():
                if isinstance(v, engine.BuiltTable):
                    v.table.free()

    # ---- the chain with NO host wait between its first kernel and its result (round 6) ---------------------------------------------
    def _chain_device_sized(self, st, plan, args, lane=0):
        """A settled chain — a first run has exchanged exact sizes and remembered them — as the ENGINE's own prepared plan (the
        single-GPU plan's kernels, its last loop launched and not waited for) with the exchanges at its seams, every one sized on the
        device:
          a replicated table with payload   its entries packed in build order into ONE fixed-capacity chunk (sdqh_table_partition_pack,
                                            one part), one all-gather, the ranks' chunks taken apart into columns padded to the capacity
                                            (sdqh_unpack_chunks) and rebuilt — the rows beyond the true counts carry a key a gate drops;
          a replicated key set              its exact bitmap, one collective (as before: nothing was waited for there);
          partial groups (<= 256)           this rank's group table straight into the send buffer of ONE all-gather, the ranks' blocks
                                            folded by key in rank order on the device (sdqh_xgroupby_partial / _fold) into the result
                                            block the engine's deferred result reads — where the first run found the packed keys to mean
                                            the same on every rank; else the groups are merged on the host as in the first run.
        The largest chunk counts travel in the spare words of the group block (no collective of their own) or, for a plan without one,
        in an all-reduce of four words; they land in pinned memory in front of the result and are read when it is collected: a chunk
        that overflowed anywhere makes every rank repeat the chain with exact sizes (collective_rerun), otherwise they bound the next
        run.  Collectives per run: one per replicated table + one for the groups.
        RECORDED (round 6, RCCL only; `self.graphs`): a chain that has run this way twice is recorded — the plan's calls AND its
        collectives, torch's current stream being the lane's stream in capture mode, so RCCL's work joins the capture
        (tools/exp_rccl_in_graph.py: a graph with an all-gather and an all-reduce inside replays right) — and from then on launched by
        ONE call; nothing in the recorded region allocates from torch (persistent buffers, a status block of the recording's own)."""
        ctx, eng = self.ctx, self.eng
        G = self.world
        pp = engine.prepared_plan(eng, plan, args, lane=lane, member_only=st.member_only)      # (the chain's own choice of key sets: a set whose entries travel is a table)
        self._on_engine_stream()
        dev = st.__dict__.get("dev_bufs")
        if dev is None:
            dev = st.dev_bufs = {"stat": self._zeros(abi.EXCHANGE_STAT_WORDS), "bufs": {}, "caps": {}, "ranks_words": self._zeros(4 * self.world)}
            dev["stat_col"] = ctx.wrap(dev["stat"].data_ptr(), abi.EXCHANGE_STAT_WORDS, abi.I64, keepalive=dev["stat"])
        stat_t, stat = dev["stat"], dev["stat_col"]
        names = [n for n in st.fast_tables]                          # replicated tables with payload, in plan order: their status slots
        SW = abi.EXCHANGE_STAT_WORDS

        def collective_rerun():
            self.fast_retries += 1
            for r in st.__dict__.pop("recordings", []):             # (the bounds they were recorded with are the ones that just failed)
                r["pg"].free()
            st.settled_runs = 0
            return self._guarded(lambda: self._sharded_chain(plan, args, st.whole, None, waited=True))

        def make_precheck(host, busy, state):
            def precheck():
                try:
                    if names and state["status_sent"]:
                        if host[SW - 1] == -1:
                            t0 = time.perf_counter()
                            while host[SW - 1] == -1:
                                if time.perf_counter() - t0 > 2.0:
                                    ctx.synchronize()
                                    if self.backend == "nccl":
                                        torch.cuda.synchronize(self.device)
                                    break
                        if state.get("status_from") == "ranks":      # every rank's counts came with its group block: the largest of each
                            most = [max(int(host[SW + 4 * r + i]) for r in range(G)) for i in range(len(names))]
                        else:
                            most = [int(host[abi.STAT_MAX_COUNT + i]) for i in range(len(names))]
                    else:
                        most = []
                finally:
                    busy[0] = False
                over = [(n, m, st.caps[n]) for n, m in zip(names, most) if m > st.caps[n]]
                for n, m in zip(names, most):
                    st.caps[n] = _chunk_bound(m, st.caps[n])
                if over:
                    raise engine.RetryPlan("a replicated table outgrew its chunk: %s" % ", ".join("%s %d > %d" % o for o in over))
            return precheck

        # a recording of this chain that is free to be launched again?
        recs = st.__dict__.setdefault("recordings", [])
        use_graphs = self.graphs and st.__dict__.get("graph_ok", True) and not ctx._profiling and not ctx._prof_mode and not st.key_range
        if use_graphs:
            if recs and any(r["caps"] != st.caps for r in recs):    # (recorded with other chunk bounds than the chain has now: stale)
                for r in recs:
                    r["pg"].free()
                del recs[:]
                st.settled_runs = 0                                  # (two runs with the calls issued first: the buffers of the new bounds are made OUTSIDE a recording)
            for r in recs:
                pg = r["pg"]
                if pg.state == "flying" and pg.result() is None:    # launched, its result dropped unread: wait for the lane, then it is free
                    ctx.synchronize()
                    pg.state = "free"
                if pg.state == "free" and not pg.rows_out():
                    r["host"][SW - 1] = -1
                    r["busy"][0] = True
                    pg.graph.launch()
                    pg.state = "flying"
                    self.fast_runs += 1
                    self.graph_launches += 1
                    self.last_chain = dict(r["info"], recorded=True)
                    self._partitioned_result = st.partitioned_result
                    rs = pp._graph_result(pg, precheck=make_precheck(r["host"], r["busy"], r["state"]), on_retry=collective_rerun)
                    pg.result = weakref.ref(rs)
                    return rs
        record = use_graphs and len(recs) < 2 and st.__dict__.get("settled_runs", 0) >= 2
        if record:
            host_t = torch.zeros(SW + 4 * G, dtype=torch.int64).pin_memory()
            host, busy = host_t.numpy(), [True]
        else:
            host_t, host, busy = self._stat_buffer()
        host[SW - 1] = -1
        keep = []
        state = {"status_sent": False}

        def buffers(name, words):
            pair = dev["bufs"].get(name)
            if pair is None or pair[0].numel() != words:
                if record:                                           # (torch's allocator knows nothing of the capture: nothing is allocated inside it)
                    raise abi.SdqhError(abi.ERR_UNSUPPORTED, "a collective buffer would be allocated inside the recording")
                send = self._empty(words)
                pair = dev["bufs"][name] = (send, self._empty(words * G))
            return pair

        def gather(recv, send):
            self._note("all_gather", send)
            if self.backend == "nccl":
                self._coll(dist.all_gather_into_tensor, recv, send, group=self.group)
            else:
                self._coll(dist.all_gather, list(recv.view(G, -1).unbind(0)), send, group=self.group)

        def status_to_host(head=None, ranks=None):
            if ranks is not None:                                      # (recv, words, spare): every rank's four words as they came — no reduction, nothing allocated
                recv_t, words, spare = ranks
                # ONE strided device copy gathers the ranks' words, ONE copy takes them to the host: two launches whatever the group's size
                # (a copy per rank was 5 us each on the chain's critical path — eight ranks, eight copies)
                stage = dev["ranks_words"]
                stage.view(G, 4).copy_(recv_t.view(G, words)[:, spare:spare + 4])
                host_t[SW:SW + 4 * G].copy_(stage, non_blocking=True)
                state["status_from"] = "ranks"
            else:
                host_t[:4].copy_(head, non_blocking=True)
                state["status_from"] = "head"
            host_t[SW - 1:SW].copy_(stat_t[SW - 1:], non_blocking=True)      # (the sentinel's word: 0 on the device)
            state["status_sent"] = True

        def replicate(name):
            def seam(env):
                bt = env[name]
                if not isinstance(bt, engine.BuiltTable):
                    return None
                if name in st.key_range:
                    info["bitmaps"].append(name)
                    return self._replicate_table(bt, st.key_range[name])            # (a bitmap through one collective: no host wait in it)
                info["replicated"].append(name)
                slot = names.index(name)
                cap = st.caps[name]
                ncols = 1 + bt.table.npayload
                cw = ctx.chunk_words(ncols, cap)
                send, recv = buffers(name, cw)
                ctx.table_partition_pack(bt.table, 1, cap, send.data_ptr())
                gather(recv, send)
                if bt.key_parts is not None:
                    (lo0, hi0), (lo1, hi1) = st.part_ranges[name]
                    pad = (hi0 + 1) << 32
                    cols = ctx.unpack_chunks(recv.data_ptr(), G, [abi.I64] * ncols, cap, pad, stat, slot, sent_ptr=None, self_part=self.rank)
                    hi, lo = ctx.unpack2(cols[0], G * cap)
                    hi.set_bounds(lo0, hi0 + 1); lo.set_bounds(min(lo1, 0), hi1)
                    table = ctx.build(G * cap, abi.make_filter([(hi, lo0, hi0)], [], []), [], [abi.src_col(hi), abi.src_col(lo)], [abi.src_col(c) for c in cols[1:]])
                    cols = list(cols) + [hi, lo]
                else:
                    lo_k, hi_k = st.table_range[name]
                    pad = lo_k - 1
                    cols = ctx.unpack_chunks(recv.data_ptr(), G, [abi.I64] * ncols, cap, pad, stat, slot, sent_ptr=None, self_part=self.rank)
                    cols[0].set_bounds(pad, hi_k)
                    table = self._build_from_columns(G * cap, cols, (pad, hi_k), accumulate=False, first_key=lo_k)
                new = engine.BuiltTable(table, bt.key_name, bt.key_is_record, bt.val_fields, bt.val_is_record, bt.payload_dtypes)
                new.decoders, new.key_parts, new.field_decoders = bt.decoders, bt.key_parts, bt.field_decoders
                new.key_decoder = bt.key_decoder
                for attr in ("slot_rng", "key_part_decoders", "key_bounds", "slot_roots", "slot_plain"):
                    if hasattr(bt, attr):
                        setattr(new, attr, getattr(bt, attr))
                new._keep = cols
                keep.extend(cols)
                bt.table.free()
                if name == names[-1] and not st.fold_fp:
                    # no group block will carry the chunks' counts: four words all-reduced behind the last exchange, in front of whatever
                    # the plan ends in (a K-F launched and not waited for: _partitioned results)
                    head = stat_t[:4]
                    self._note("all_reduce", head)
                    self._coll(dist.all_reduce, head, op=dist.ReduceOp.MAX, group=self.group)
                    status_to_host(head)
                return new
            return seam

        def fold_hook(name, fp):
            if name not in st.fold_fp or name not in pp.defer_names:
                return None                                            # not foldable by packed key: the loop is waited for, its groups merged on the host (the seam below)
            if fp is None or st.fold_fp.get(name) != fp:
                # (the digest the first run compared across the ranks belongs to this prepared chain's compiled loop: a different one
                #  means the loop was compiled again behind the runner's back)
                raise RuntimeError("%s: the group keys of '%s' were encoded anew since the ranks compared their encodings" % (plan.name, name))

            def exchange(nbytes):
                words = nbytes // 8
                send, recv = buffers(("groups", name), words)
                info["folded"].append(name)

                def issue():
                    # the chunks' largest counts ride in the block's spare words (behind its completion word, which a device block does
                    # not use): no collective of their own
                    self._host_touch()                                 # (hybrid mode: the torch ops below run on the host)
                    spare = words - 6
                    send[spare:spare + 4].copy_(stat_t[:4])
                    gather(recv, send)
                    status_to_host(ranks=(recv, words, spare))
                    return recv.data_ptr(), G
                return send.data_ptr(), issue
            return exchange

        def merged(name):
            def seam(env):
                res = env[name]
                if isinstance(res, engine.DictResult):
                    info["merged_on_host"].append(name)
                    return self._merge_groups(res, st, name)
                if isinstance(res, float):
                    return sum(float(p[0]) for p in self._all_gather_array(np.array([res], np.float64)))
                if isinstance(res, dict):
                    keys = sorted(res)
                    parts = self._all_gather_array(np.array([res[k] for k in keys], np.float64))
                    return {k: sum(float(p[j]) for p in parts) for j, k in enumerate(keys)}
                return None                                            # (a Pending — the folded groups —, a local build, finished local groups)
            return seam

        after = {}
        for op, _ in st.steps:
            if not isinstance(op, ScanOp):
                continue
            if st.replicate.get(op.out):
                after[op.out] = replicate(op.out)
            elif st.sharded[op.out]:
                after[op.out] = merged(op.out)                         # (a Pending — the folded groups — passes through)

        precheck = make_precheck(host, busy, state)

        self.fast_runs += 1
        self.last_chain = info = {"plan": plan.name, "replicated": [], "bitmaps": [], "folded": [], "merged_on_host": []}      # (what this run's seams did: tests, bench.py)
        self._partitioned_result = st.partitioned_result
        if record:
            # the chain's calls and collectives into a graph (nothing executes), then its first launch
            pg = None
            try:
                pg = pp._record(pp._graph_epoch(), after=after, env_extra={"__group_fold__": fold_hook})
            finally:
                self._inflight.extend(keep)
            if pg is not None and (not names or state["status_sent"]):
                rec = {"pg": pg, "host_t": host_t, "host": host, "busy": busy, "state": state, "info": dict(info), "keep": keep, "caps": dict(st.caps)}
                recs.append(rec)
                self.graph_recordings += 1
                host[SW - 1] = -1
                pg.graph.launch()
                pg.state = "flying"
                self.graph_launches += 1
                self.last_chain = dict(info, recorded=True)
                rs = pp._graph_result(pg, precheck=precheck, on_retry=collective_rerun)
                pg.result = weakref.ref(rs)
                return rs
            if pg is not None:
                pg.free()
            st.graph_ok = False                                      # (refused — a call that waits, a loop that is not deferred —: the calls are issued, as before)
            st.graph_refused = pp._graph_refused or "the exchange status never left the device inside the recording"
            if os.environ.get("SDQLPY_AMD_DIST_DEBUG") == "1":
                print("[dist] %s: not recorded: %s" % (plan.name, st.graph_refused), file=sys.stderr, flush=True)
            pp._graph_refused = None                                 # (the engine's own note: this plan is not recorded WITHOUT its seams either, by the guard in run)
            busy[0] = False
            return self._chain_device_sized(st, plan, args, lane)
        try:
            res = pp.run(None, after=after, keep_tables=True, on_retry=collective_rerun, precheck=precheck, env_extra={"__group_fold__": fold_hook})
        finally:
            self._inflight.extend(keep)
        st.settled_runs = st.__dict__.get("settled_runs", 0) + 1
        if use_graphs and (info["merged_on_host"] or not isinstance(res, engine.DeferredResultSet)):
            # (groups merged on the host, a last loop that is waited for: a recording would refuse at that call — known now, not tried)
            st.graph_ok = False
            st.graph_refused = "the chain waits for the device: %s" % (", ".join("'%s' merged on the host" % n for n in info["merged_on_host"]) or "its last call is not deferred")
        if isinstance(res, engine.DeferredResultSet):
            if names and not state["status_sent"]:
                raise RuntimeError("%s: a deferred chain whose exchange status never left the device" % plan.name)
            return res
        # every call was waited for (a final loop without a deferred form): the status — where no group block carried it, four words
        # all-reduced now — and the verdict, the same on every rank
        if names:
            if not state["status_sent"]:
                head = stat_t[:4]
                self._note("all_reduce", head)
                self._coll(dist.all_reduce, head, op=dist.ReduceOp.MAX, group=self.group)
                status_to_host(head)
            ctx.synchronize()
        try:
            precheck()
        except engine.RetryPlan:
            return collective_rerun()
        return res

    def _merge_groups(self, d, st=None, name=None):
        """Partial groups of every rank -> the global groups on every rank (folded in rank order).
        The rows travel as raw bytes in ONE fixed-size all_gather (<= 256 groups per rank: the
        group-by kernels' own limit), not as pickled objects."""
        if self.world == 1 and self.skip_trivial:
            return d                                # a group of one: the partial groups are the groups
        fields = list(d.key_fields) + list(d.val_fields)
        dtypes = [np.dtype(a.dtype) for _, a in fields]
        rowbytes = sum(dt.itemsize for dt in dtypes)
        cap = abi.MAX_LOOKUP_GROUPS
        n = d.size()
        if n > cap:
            raise frontend.UnsupportedQuery("more than %d partial groups on one rank" % cap)
        body_words = (cap * rowbytes + 7) // 8
        buf = np.zeros(3 + body_words, np.int64)
        buf[0], buf[1] = n, rowbytes
        fp = getattr(d, "encoding_fp", None)
        buf[2] = int(fp[:15], 16) if fp else -1              # what this rank's packed group keys mean, as a digest (xplan._encoding_fingerprint)
        body = buf[3:].view(np.uint8)[: cap * rowbytes].reshape(cap, rowbytes)
        at = 0
        for (_, a), dt in zip(fields, dtypes):
            if n:
                body[:n, at:at + dt.itemsize] = np.ascontiguousarray(a).view(np.uint8).reshape(n, dt.itemsize)
            at += dt.itemsize
        parts = self._all_gather_array(buf)
        if st is not None and name is not None:
            # every rank packs its group keys alike (same dictionaries, same value ranges): later runs may fold the partial groups by
            # packed key on the device (_chain_device_sized); else they go on being merged here, by their decoded values
            same = fp is not None and all(int(p[2]) == int(buf[2]) for p in parts)
            if same:
                st.fold_fp[name] = fp
            else:
                st.fold_fp.pop(name, None)
        merged, order = {}, []
        nk = len(d.key_fields)
        for p in parts:
            pn = int(p[0])
            if not pn:
                continue
            if int(p[1]) != rowbytes:
                raise RuntimeError("ranks disagree on the group row layout")
            pbody = p[3:].view(np.uint8)[: cap * rowbytes].reshape(cap, rowbytes)
            cols, at = [], 0
            for dt in dtypes:
                cols.append(np.ascontiguousarray(pbody[:pn, at:at + dt.itemsize]).view(dt).reshape(pn).tolist())
                at += dt.itemsize
            for row in zip(*cols):
                k, v = row[:nk], row[nk:]
                if k not in merged:
                    merged[k] = list(v); order.append(k)
                else:
                    merged[k] = [x + y for x, y in zip(merged[k], v)]
        if not order:
            return d
        order.sort()
        kf = [(nm, np.array([k[j] for k in order], dtype=dtypes[j])) for j, (nm, _) in enumerate(d.key_fields)]
        vf = [(nm, np.array([merged[k][j] for k in order], dtype=dtypes[nk + j])) for j, (nm, _) in enumerate(d.val_fields)]
        return engine.DictResult(kf, vf, d.key_is_record, d.val_is_record)

    def _row_sharded_groupby(self, plan, args):
        """Every rank aggregates its shard; the <= 64 partial groups per rank travel in ONE small
        all_gather (header + schema + rows, 8 KiB) and are folded in rank order."""
        op = plan.ops[0]
        nkey = len(op.key.fields) if isinstance(op.key, RecordCons) else 1
        local = engine.execute_plan(self.eng, plan, args, lane=0)      # ResultSet of this shard's groups
        if self.world == 1 and self.skip_trivial:
            return local                            # a group of one: launched, not waited for — as on one GPU
        cap, maxc = abi.MAX_SMALL_GROUPS, 16
        n = local.size() if local is not None else 0
        buf = np.zeros(2 + maxc + cap * maxc, np.int64)
        if n:
            ncols = len(local.columns)
            buf[0], buf[1] = n, ncols
            body = buf[2 + maxc:].reshape(cap, maxc)
            for j, a in enumerate(local.arrays):
                if a.dtype.kind == "U":
                    body[:n, j] = a.astype("<U1").view(np.uint32).astype(np.int64); buf[2 + j] = ord("U")
                elif a.dtype.kind == "f":
                    body[:n, j] = a.astype(np.float64).view(np.int64); buf[2 + j] = ord("f")
                else:
                    body[:n, j] = a.astype(np.int64); buf[2 + j] = ord("i")
        parts = self._all_gather_array(buf)
        src = next((p for p in parts if p[0] > 0), None)
        if src is None:
            return local
        ncols = int(src[1])
        kinds = [chr(int(x)) for x in src[2:2 + ncols]]
        cols = local.columns if n else ([f for f, _ in op.key.fields] + [f for f, _ in op.val.fields])
        merged = {}
        for p in parts:                                          # rank order: deterministic sums
            body = p[2 + maxc:].reshape(cap, maxc)
            for g in range(int(p[0])):
                key = tuple(int(x) for x in body[g, :nkey])
                vals = [body[g, nkey + j:nkey + j + 1].view(np.float64)[0] if kinds[nkey + j] == "f" else int(body[g, nkey + j]) for j in range(ncols - nkey)]
                acc = merged.get(key)
                if acc is None:
                    merged[key] = vals
                else:
                    for j, v in enumerate(vals):
                        acc[j] += v
        keys = sorted(merged)
        arrays = []
        for j in range(nkey):
            col = np.array([k[j] for k in keys], np.int64)
            arrays.append(col.astype(np.uint32).view("<U1") if kinds[j] == "U" else col)
        for j in range(ncols - nkey):
            arrays.append(np.array([merged[k][j] for k in keys], np.float64 if kinds[nkey + j] == "f" else np.int64))
        return ResultSet(cols, arrays)

    # ---- q3-shaped plans: build(A) -> build(B, semi-join A) -> probe-aggregate(C into B) -> finalise --
    def _prepare_join(self, plan, args, a_whole=False):
        """Everything about the distributed join that depends only on the plan and on which tables
        it is bound to: lowered filters / tuples, resident columns, and the (static) facts gathered
        once from all ranks — global key range of A, per-rank key ranges of B, whether every rank's
        probe keys already lie in its own range."""
        eng, ctx = self.eng, self.ctx
        ops = plan.ops
        if not _is_join_shape(ops):
            raise frontend.UnsupportedQuery("%s does not have the build / build / probe-aggregate shape" % plan.name)
        tabs = {p: engine.HostTable(p, a) for p, a in zip(plan.params, args)}
        a_op, b_op, c_op, _ = ops
        ta, tb, tc = tabs[a_op.table], tabs[b_op.table], tabs[c_op.table]
        st = type("JoinState", (), {})()
        st.args = tuple(args)
        st.generation = eng.generation
        st.a_whole = a_whole
        st.na, st.nb, st.nc = ta.nrows, tb.nrows, tc.nrows

        st.flt_a, look_a = engine._build_filter(eng, a_op, ta, a_op.conds)
        if look_a or not isinstance(a_op.key, Col):
            raise frontend.UnsupportedQuery("unsupported first build in the distributed join")
        st.key_a = eng.column(ta.array(a_op.key.name, a_op))

        st.flt_b, look_b = engine._build_filter(eng, b_op, tb, b_op.conds)
        if look_b or not isinstance(b_op.key, Col) or not isinstance(b_op.probe.key, Col):
            raise frontend.UnsupportedQuery("unsupported second build in the distributed join")
        st.key_b = eng.column(tb.array(b_op.key.name, b_op))
        st.probe_b = eng.column(tb.array(b_op.probe.key.name, b_op))
        st.pay_names = [e.name for _, e in b_op.val.fields] if isinstance(b_op.val, RecordCons) else []
        st.pay_b = [eng.column(tb.array(nm, b_op)) for nm in st.pay_names]
        st.pay_dtypes = [tb.array(nm, b_op).dtype for nm in st.pay_names]

        st.flt_c, look_c = engine._build_filter(eng, c_op, tc, c_op.conds)
        if look_c or not isinstance(c_op.probe.key, Col):
            raise frontend.UnsupportedQuery("unsupported probe side in the distributed join")
        st.tup_c, st.vnames, st.count_idx = engine._build_tuple(eng, c_op, tc, c_op.val)
        st.ckey_name = c_op.probe.key.name
        st.key_c = eng.column(tc.array(st.ckey_name, c_op))
        slots = []
        for _, e in (c_op.val.fields if isinstance(c_op.val, RecordCons) else [(None, c_op.val)]):
            e.shape(slots)
        st.ops_c = [eng.column(tc.array(s[1], c_op)) for s in slots]
        st.key_fields = c_op.key.fields if isinstance(c_op.key, RecordCons) else [(None, c_op.key)]
        # the probe loop's own closure (fixed-shape route): run on rows that reached this rank through the exchange, it aggregates them
        # into the table and leaves the table marked as the engine's own step would, so the engine's K-F finishes the plan
        st.step_c = engine._prepare_scan_fixed(eng, c_op, tc, {b_op.out: None})
        st.step_c_args = (c_op, tc, {b_op.out: None})           # (a settled join on another lane prepares the same closure on that lane's context)

        # ---- static facts, gathered once ---------------------------------------------------------
        a_lo, a_hi = st.key_a.minmax() if st.na else (abi.INT64_MAX, abi.INT64_MIN)
        b_lo, b_hi = st.key_b.minmax() if st.nb else (abi.INT64_MAX, abi.INT64_MIN)
        c_lo, c_hi = st.key_c.minmax() if st.nc else (b_lo, b_hi)
        facts = self._all_gather_array(np.array([a_lo, a_hi, b_lo, b_hi, st.nb, c_lo, c_hi, st.nc], np.int64))
        st.a_range = (min(int(f[0]) for f in facts), max(int(f[1]) for f in facts))
        a_bits = st.a_range[1] - st.a_range[0] + 1
        st.a_bitmap = 0 < a_bits <= (1 << 31)
        st.b_ranges = [(int(f[2]), int(f[3])) for f in facts]
        nonempty = all(int(f[4]) > 0 for f in facts)
        disjoint = nonempty and all(st.b_ranges[i][1] < st.b_ranges[i + 1][0] for i in range(self.world - 1))
        mode = self.partition
        if mode == "auto":
            mode = "range" if disjoint else "hash"
        if mode == "range" and not disjoint:
            raise frontend.UnsupportedQuery("range partitioning needs disjoint ascending build-key ranges per rank")
        st.mode = mode
        st.upper = np.array([r[1] for r in st.b_ranges[:-1]], np.int64)
        # does every rank's probe shard only hold keys of its own build range?  (true for dbgen data:
        # lineitem is clustered on l_orderkey like orders on o_orderkey)
        st.all_local = mode == "range" and all(int(f[7]) == 0 or (int(f[5]) >= int(f[2]) and int(f[6]) <= int(f[3])) for f in facts)
        base_ip = [(f.col_obj, f.lo, f.hi) for f in _ipreds(st.flt_c)]
        my_lo, my_hi = st.b_ranges[self.rank]
        st.flt_own = abi.make_filter(base_ip + [(st.key_c, my_lo, my_hi)], _fpreds(st.flt_c), [])
        st.flt_foreign = [abi.make_filter(base_ip + [(st.key_c, lo_f, hi_f)], _fpreds(st.flt_c), [])
                          for lo_f, hi_f in ((abi.INT64_MIN, my_lo - 1), (my_hi + 1, abi.INT64_MAX)) if lo_f <= hi_f]
        st.empty = abi.make_filter()
        # device-sized exchanges (hash partitioning): the chunk capacities come from a first run that measured them (None until then);
        # the probe side must have a row program
        st.caps = None
        st.fast_ok = mode == "hash" and len(st.ops_c) <= abi.MAX_PAYLOAD and not st.flt_c._keep[2] and not st.flt_c._keep[3] \
            and 1 + len(st.pay_b) <= abi.MAX_COMPACT_COLS
        return st

    def _replicated_key_set(self, table, rng, disjoint=True, cache=None):
        """The union over the ranks of a table's keys as a key set on every rank, through its exact bitmap over the GLOBAL key range
        `rng`: exported straight into the collective's buffer (a torch tensor wrapped as a column; key sets and direct-layout tables
        copy their own bitmap words, shifted) and made global by ONE collective — an all-reduce when the ranks' key sets are
        disjoint (SUM = OR: the keys of a unique build over row shards), else an all-gather folded with bitwise OR on the device
        (RCCL has no bitwise reduction).  Returns (key-set Table, what it borrows: keep alive while the table is used)."""
        ctx = self.ctx
        lo, hi = rng
        n64 = ((hi - lo + 1 + 31) // 32 + 1) // 2
        # (cache: a prepared plan's own buffers, keyed by the caller — written and read in stream order run after run, and the same memory in
        #  every replay of a recording of the plan)
        buf = cache.get(("bits", n64)) if cache is not None else None
        if buf is None:
            if self.ctx.capturing():                                 # (nothing is allocated from torch inside a recording)
                raise abi.SdqhError(abi.ERR_UNSUPPORTED, "a collective buffer would be allocated inside the recording")
            buf = self._empty(max(n64, 1))
            if cache is not None:
                cache[("bits", n64)] = buf
        words = ctx.wrap(buf.data_ptr(), n64, abi.I64, keepalive=buf)
        ctx.table_export_bitmap(table, lo, hi, into=words)                    # queued under "async_copies": the collective is ordered behind it
        if self.world == 1 and self.skip_trivial:
            pass                                    # (a group of one: the exported bitmap is the union already)
        elif disjoint:
            self._note("all_reduce", buf)
            self._coll(dist.all_reduce, buf, group=self.group)
        else:
            parts = cache.get(("parts", n64)) if cache is not None else None
            if parts is None:
                parts = self._empty(max(n64, 1) * self.world)
                if cache is not None:
                    cache[("parts", n64)] = parts
            self._note("all_gather", buf)
            if self.backend == "nccl":
                self._coll(dist.all_gather_into_tensor, parts, buf, group=self.group)
            else:
                self._coll(dist.all_gather, list(parts.view(self.world, -1).unbind(0)), buf, group=self.group)
            rows = parts.view(self.world, -1)
            for r in range(self.world):
                if r == 0:
                    buf.copy_(rows[0])
                else:
                    torch.bitwise_or(buf, rows[r], out=buf)
            self._inflight.append(parts)
        return ctx.table_from_bitmap(words, lo, hi), words

    def _build_from_columns(self, n, cols, key_range, accumulate=True, first_key=None):
        """A table with accumulators from received entry columns [key, payload ...] (every row an entry; keys within key_range).  As a
        row program when the library takes it — the value-queue build streams the columns once and needs no minimum / maximum pass,
        having been told the bounds: 0.03 ms for Q3's 1.46 M received orders where the fixed-shape build took 0.27 — else the
        fixed-shape unique build.  first_key: rows whose key is below it are no entries (the padding rows of sdqh_unpack_chunks carry
        first_key - 1, which key_range then includes): a gate of the program / a predicate of the fixed-shape build drops them."""
        ctx = self.ctx
        lo, hi = key_range
        if n and len(cols) <= 1 + abi.MAX_PAYLOAD and lo <= hi:
            prog = abi.Program()
            prog.key = prog.op(abi.X_COL, abi.T_I64, col=cols[0])
            if first_key is not None:
                prog.gates = [prog.op(abi.X_GE, abi.T_BOOL, a=prog.key, b=prog.op(abi.X_CONST, abi.T_I64, imm_i=first_key))]
            prog.vals = [prog.op(abi.X_COL, abi.T_F64 if c.dtype == abi.F64 else abi.T_I64, col=c) for c in cols[1:]]
            try:
                return ctx.xbuild(n, prog, lo, hi, accumulate=accumulate)
            except abi.SdqhError as exc:
                if exc.code != abi.ERR_UNSUPPORTED:
                    raise
        flt = abi.make_filter() if first_key is None else abi.make_filter([(cols[0], first_key, abi.INT64_MAX)], [], [])
        return ctx.hash_build_unique(n, flt, [], cols[0], cols[1:], accumulate=accumulate)

    def _replicated_set(self, st, local, cache=None):
        """Table A (a BuiltTable built from this rank's shard) on every rank; see _replicated_key_set.  Key range too wide for a
        bitmap: the surviving keys are all-gathered and the set is built from them."""
        ctx = self.ctx
        if self.world == 1 and self.skip_trivial:
            return []                               # a group of one: this rank's set IS the global set
        if st.a_bitmap:
            table, words = self._replicated_key_set(local.table, st.a_range, cache=cache)
            keep = [words]
        else:
            (ka,), n_a = ctx.scan_compact(st.na, st.flt_a, [], [st.key_a])
            rep_keys, n_rep = self._all_gather_column(ka, n_a)
            ka.free()
            table, keep = ctx.hash_build_unique(n_rep, st.empty, [], rep_keys, []), [rep_keys]
        local.table.free()
        local.table = table
        return keep

    def _partitioned_join(self, plan, args, a_whole=False, waited=False):
        """The engine's own prepared steps (the kernels of the single-GPU plan: tight-encoded build and probe programs, K-F behind the
        call) with the exchange at their seams:
          A      built from this rank's shard, then replaced by the replicated set (one collective; none when A is whole everywhere);
          range  B and the probe-aggregate run in place; probe rows whose key another rank owns are compacted, exchanged (one packed
                 all-to-all) and aggregated into B before the engine's K-F;
          hash   B's survivors — the entries of the table the engine's build step made of this rank's shard — are hash-partitioned,
                 exchanged, and B is rebuilt from what arrived; its keys' bitmap over the global range is replicated (one all-reduce)
                 so that only probe rows that will hit travel; those are compacted, exchanged and aggregated; then the engine's K-F.
        Hash partitioning has two forms.  The FIRST run of a prepared join (and any run after a bound turned out too small: `waited`)
        exchanges exact sizes: every count visits the host (table entries, two count matrices) — and is remembered.  Later runs
        (_hash_join_device_sized) move fixed-capacity chunks sized from those counts, with the true counts in the chunk headers: no
        host wait between the first kernel and the result."""
        ctx, eng = self.ctx, self.eng
        cache = plan.__dict__.setdefault("_dist_prepared", {})
        key = (id(self), a_whole, self.partition) + tuple(id(a) for a in args)
        st = cache.get(key)
        if st is None or st.generation != eng.generation or any(x is not y for x, y in zip(st.args, args)):
            st = cache[key] = self._prepare_join(plan, args, a_whole)
        self.last_partitioning = st.mode
        self.exchanged_bytes = 0
        pp = engine.prepared_plan(eng, plan, args, lane=0)          # (the collectives are ordered on the first context's stream)
        a_op, b_op, c_op, f_op = plan.ops
        keep = []

        def replicate_a(env):
            if not st.a_whole:                   # (A whole on every rank: its set is built locally and no collective runs —
                keep.extend(self._replicated_set(st, env[a_op.out], cache=st.__dict__.setdefault("bits_a", {})))      #  summing the ranks' bitmaps of identical sets would carry bits into their neighbours)

        def collective_rerun():
            # what a deferred run of this join does when it has to be repeated: the whole join once more with every size exact — a
            # COLLECTIVE run, which every rank enters because every rank read the same all-reduced status (or, for a failure only this
            # rank saw, fails loudly on the others' side rather than hang: the collectives carry torch's own timeout)
            self.fast_retries += 1
            return self._guarded(lambda: self._partitioned_join(plan, args, a_whole, waited=True))

        try:
            if st.mode == "range":
                after = {a_op.out: replicate_a}
                self.exchanged_rows = {"build": 0, "probe_sent": 0, "probe_received": 0}
                if not st.all_local:
                    def foreign_rows(env):
                        # rows of this rank's probe shard whose key another rank's build range owns: compacted, sent there, and what
                        # arrives here aggregated into B (the step before aggregated the rows that were already where they belong —
                        # a foreign key finds nothing in this rank's B)
                        pieces = [ctx.scan_compact(st.nc, flt, [], [st.key_c] + st.ops_c) for flt in st.flt_foreign]
                        foreign_n = sum(n for _, n in pieces)
                        merged = _concat_columns(ctx, pieces, [abi.I64] + [abi.F64] * len(st.ops_c))
                        recv, n_recv, _ = self._exchange(foreign_n, merged[0], merged, range_upper=st.upper)
                        if n_recv:
                            ctx.hash_probe_aggregate(n_recv, st.empty, env[b_op.out].table, recv[0], abi.make_tuple(st.tup_c.shape, recv[1:]))
                        keep.extend(recv)
                        self.exchanged_rows = {"build": 0, "probe_sent": int(foreign_n), "probe_received": int(n_recv)}
                    after[c_op.out] = foreign_rows
                # (keep_tables: a K-F block that turns out too small is repeated on this rank's own tables — the plan is never re-run by
                #  one rank alone, which would build A from its shard only and skip the exchange)
                return pp.run(self._top, after=after, keep_tables=True, on_retry=collective_rerun)
            if self.device_sized and st.fast_ok and st.caps is not None and not waited:
                try:
                    lane = self._lane_of(plan)
                    with self._on_lane(lane):
                        pp_lane = pp if lane == 0 else engine.prepared_plan(eng, plan, args, lane=lane)
                        return self._hash_join_device_sized(st, pp_lane, plan, replicate_a, keep, collective_rerun)
                except abi.SdqhError as exc:
                    if exc.code != abi.ERR_UNSUPPORTED:
                        raise
                    st.fast_ok = False              # (decided by the tables' layouts, the same on every rank: all fall back alike)
            # ---- hash partitioning with exact sizes: the steps driven from here ---------------------------------------------------
            steps = dict(pp.steps)
            env = {}
            try:
                env[a_op.out] = steps[a_op.out](env)
                replicate_a(env)
                env[b_op.out] = bt_b = steps[b_op.out](env)                   # this rank's survivors, by the engine's own build kernel
                bcols, nb = ctx.table_entries(bt_b.table)                     # [key, payload ...]
                brecv, nb_recv, _ = self._exchange(nb, bcols[0], bcols)
                most_b = self._last_matrix_max
                for c in bcols:
                    c.free()
                table_b = self._build_from_columns(nb_recv, brecv, (min(r[0] for r in st.b_ranges), max(r[1] for r in st.b_ranges)))
                bt_b.table.free()
                bt_b.table = table_b
                keep.extend(brecv)
                # Only probe rows whose key some rank holds need to travel (Q3: half a per cent of the filtered lineitem rows): every rank
                # exports the exact bitmap of ITS partition's keys over the global key range, one all-reduce (SUM = OR: partitions are
                # disjoint) replicates the set of all build keys, and the filter + semi-join compaction drops the rest before the
                # partitioning pass and the all-to-all ever see them.  Skipped when the key range is too wide for a bitmap.
                probes_c = []
                lo_g, hi_g = min(r[0] for r in st.b_ranges), max(r[1] for r in st.b_ranges)
                if self.prefilter and lo_g <= hi_g and hi_g - lo_g + 1 <= (1 << 31):
                    all_keys, words = self._replicated_key_set(table_b, (lo_g, hi_g))
                    probes_c = [(all_keys, st.key_c)]
                    keep.append(words)
                ccols, nc = self._compact_probe_rows(st, probes_c)
                for tbl, _ in probes_c:
                    tbl.free()                                                # (the words it borrowed stay alive in `keep`)
                recv, n_recv, _ = self._exchange(nc, ccols[0], ccols, dtypes=[abi.I64] + [c.dtype for c in st.ops_c])
                most_c = self._last_matrix_max
                for c in ccols:
                    c.free()
                keep.extend(recv)
                self.exchanged_rows = {"build": int(nb), "probe_sent": int(nc), "probe_received": int(n_recv)}
                # the bounds of the next run's chunks: the largest (source, destination) count of each exchange — every rank gathered the
                # same matrices, every rank sets the same bounds
                st.caps = (_chunk_bound(most_b), _chunk_bound(most_c))
                env[c_op.out] = st.step_c(env, rows=(n_recv, recv[0], abi.make_tuple(st.step_c.tuple_shape, recv[1:])))
                return engine._finalize(eng, f_op, env, self._top)
            finally:
                for v in env.values():
                    if isinstance(v, engine.BuiltTable):
                        v.table.free()
        finally:
            self._inflight.extend(keep)          # queued kernels may still read them: released at the start of the next run (see _run)

    def _stat_buffer(self):
        """A pinned host landing block for one run's exchange status (EXCHANGE_STAT_WORDS int64), from a small ring: a block is taken
        again only after its run's result was collected (its last word is then no sentinel any more)."""
        for _ in range(len(self._stat_ring)):
            self._stat_next = (self._stat_next + 1) % len(self._stat_ring)
            t, arr, busy = self._stat_ring[self._stat_next]
            if not busy[0]:
                busy[0] = True
                return self._stat_ring[self._stat_next]
        t = torch.zeros(abi.EXCHANGE_STAT_WORDS + 4 * self.world, dtype=torch.int64)      # (+ every rank's four chunk counts: the chains' status as it comes with the group blocks)
        if self.backend == "nccl":
            t = t.pin_memory()
        self._stat_ring.append((t, t.numpy(), [True]))
        return self._stat_ring[-1]

    def _hash_join_device_sized(self, st, pp, plan, replicate_a, keep, collective_rerun):
        """The hash-partitioned join with NO host wait between its first kernel and its result (include/sdqh.h, ABI 5: device-sized
        redistribution).  Both exchanges move fixed-capacity chunks — capacity = the previous run's largest (source, destination) count
        plus an eighth — through ONE equal-split all-to-all each; the chunk headers carry the true counts, the receiver takes the chunks
        apart into columns padded to the capacity with a key that nothing holds, and the consumers (the rebuild of B, the probe loop)
        run on the capacity.  A status block records the largest count of each exchange; all-reduced (MAX) it tells every rank alike
        whether any chunk anywhere overflowed — then the join is repeated, collectively, with exact sizes — and bounds the next run.
        K-F is the engine's own deferred K-F; the status lands in pinned memory in front of it on the same stream."""
        ctx = self.ctx
        a_op, b_op, c_op, f_op = plan.ops
        G = self.world
        trivial = G == 1 and self.skip_trivial
        cap_b, cap_c = st.caps
        lo_g, hi_g = min(r[0] for r in st.b_ranges), max(r[1] for r in st.b_ranges)
        pad = lo_g - 1                                              # no entry, no probe row carries it; B's bounds are told to include it
        self._on_engine_stream()
        # what a prepared join keeps from run to run: the status block (every exchange overwrites its own words, the last word stays 0)
        # and the collective buffers — written and read in stream order, run after run, so nothing is allocated per run
        dev = st.__dict__.get("dev_bufs")
        if dev is None or dev["caps"] != (cap_b, cap_c):
            dev = st.dev_bufs = {"caps": (cap_b, cap_c), "stat": self._zeros(abi.EXCHANGE_STAT_WORDS), "bufs": {}}
            dev["stat_col"] = ctx.wrap(dev["stat"].data_ptr(), abi.EXCHANGE_STAT_WORDS, abi.I64, keepalive=dev["stat"])
        stat_t, stat = dev["stat"], dev["stat_col"]
        SW = abi.EXCHANGE_STAT_WORDS

        def make_precheck(host, busy):
            def precheck():
                # the result is being collected: the status first (it landed before K-F's rows: same stream, queued earlier)
                try:
                    if host[SW - 1] == -1:
                        t0 = time.perf_counter()
                        while host[SW - 1] == -1:
                            if time.perf_counter() - t0 > 2.0:
                                ctx.synchronize()
                                if self.backend == "nccl":
                                    torch.cuda.synchronize(self.device)
                                break
                    most_b, most_c = int(host[abi.STAT_MAX_COUNT + 0]), int(host[abi.STAT_MAX_COUNT + 1])
                    d = host[abi.STAT_DETAIL:abi.STAT_DETAIL + 8]
                    sent_b, self_b, recv_c, sent_c, self_c = int(d[1]), int(d[2]), int(d[4]), int(d[5]), int(d[6])
                finally:
                    busy[0] = False
                st.caps = (_chunk_bound(most_b, cap_b), _chunk_bound(most_c, cap_c))
                if most_b > cap_b or most_c > cap_c:
                    raise engine.RetryPlan("an exchange outgrew its chunks (%d > %d or %d > %d rows)" % (most_b, cap_b, most_c, cap_c))
                self.exchanged_rows = {"build": sent_b, "probe_sent": sent_c, "probe_received": recv_c}
                self.exchanged_bytes = 8 * ((1 + len(st.pay_b)) * (sent_b - self_b) + (1 + len(st.ops_c)) * (sent_c - self_c))
            return precheck

        def rerun():
            for r in st.__dict__.pop("recordings", []):             # (recorded with the bounds that just failed)
                r["pg"].free()
            st.settled_runs = 0
            return collective_rerun()

        # RECORDED (round 6): a join that has run this way twice is recorded with its collectives — torch's current stream is the engine's,
        # in capture mode, so the all-reduces and all-to-alls join the capture — and from then on launched by one call
        recs = st.__dict__.setdefault("recordings", [])
        use_graphs = self.graphs and self._top is None and st.__dict__.get("graph_ok", True) and not trivial and not ctx._profiling and not ctx._prof_mode
        if use_graphs:
            if recs and any(r["caps"] != (cap_b, cap_c) or r["ctx"] is not ctx for r in recs):
                for r in recs:
                    r["pg"].free()
                del recs[:]
                st.settled_runs = 0                                  # (the buffers of the new bounds are made by runs with the calls issued, outside a recording)
            for r in recs:
                pg = r["pg"]
                if pg.state == "flying" and pg.result() is None:
                    ctx.synchronize()
                    pg.state = "free"
                if pg.state == "free" and not pg.rows_out():
                    r["host"][SW - 1] = -1
                    r["busy"][0] = True
                    pg.graph.launch()
                    pg.state = "flying"
                    self.fast_runs += 1
                    self.graph_launches += 1
                    rs = pp._graph_result(pg, precheck=make_precheck(r["host"], r["busy"]), on_retry=rerun)
                    pg.result = weakref.ref(rs)
                    return rs
        record = use_graphs and len(recs) < 2 and st.__dict__.get("settled_runs", 0) >= 2
        if record:
            host_t = torch.zeros(SW + 4 * G, dtype=torch.int64).pin_memory()
            host, busy = host_t.numpy(), [True]
        else:
            host_t, host, busy = self._stat_buffer()
        host[SW - 1] = -1                                          # sentinel: overwritten (by 0) when the status has landed
        run = {"probes": []}

        def exchange(table, cap, dtypes, slot):
            pair = dev["bufs"].get(slot)
            if pair is None:
                if record:                                           # (nothing is allocated from torch inside a recording)
                    raise abi.SdqhError(abi.ERR_UNSUPPORTED, "a collective buffer would be allocated inside the recording")
                cw = ctx.chunk_words(len(dtypes), cap)
                send = self._empty(G * cw)
                pair = dev["bufs"][slot] = (send, send if trivial else self._empty(G * cw), cw)      # (a group of one: what was packed for rank 0 is what rank 0 receives)
            send, recv, cw = pair
            ctx.table_partition_pack(table, G, cap, send.data_ptr())
            if recv is not send:
                self._a2a(recv, send, [cw] * G, [cw] * G)
            return ctx.unpack_chunks(recv.data_ptr(), G, dtypes, cap, pad, stat, slot, sent_ptr=send.data_ptr(), self_part=self.rank)

        def rebuild_b(cols):
            # the rebuild's program is made once per prepared join; a run only names its columns (they are new every run)
            pb = st.__dict__.get("prog_b")
            if pb is None or len(pb[1]) != len(cols):
                prog = abi.Program()
                prog.key = prog.op(abi.X_COL, abi.T_I64, col=cols[0])
                prog.gates = [prog.op(abi.X_GE, abi.T_BOOL, a=prog.key, b=prog.op(abi.X_CONST, abi.T_I64, imm_i=lo_g))]
                prog.vals = [prog.op(abi.X_COL, abi.T_I64, col=c) for c in cols[1:]]
                pb = st.prog_b = (prog, [prog.key] + list(prog.vals))
            prog, slots = pb
            for i, c in zip(slots, cols):
                prog.bind_col(i, c)
            try:
                return ctx.xbuild(G * cap_b, prog, pad, hi_g, accumulate=True)
            except abi.SdqhError as exc:
                if exc.code != abi.ERR_UNSUPPORTED:
                    raise
                return self._build_from_columns(G * cap_b, cols, (pad, hi_g), first_key=lo_g)

        def exchange_b(env):
            bt_b = env[b_op.out]
            brecv = exchange(bt_b.table, cap_b, [abi.I64] * (1 + bt_b.table.npayload), 0)
            brecv[0].set_bounds(pad, hi_g)
            table_b = rebuild_b(brecv)
            bt_b.table.free()
            bt_b.table = table_b
            keep.extend(brecv)
            if trivial:
                run["probes"] = [(table_b, st.key_c, False)]          # a group of one: B itself answers "does any rank hold this key"
            elif self.prefilter and lo_g <= hi_g and hi_g - lo_g + 1 <= (1 << 31):
                all_keys, words = self._replicated_key_set(table_b, (lo_g, hi_g), cache=st.__dict__.setdefault("bits_b", {}))
                run["probes"] = [(all_keys, st.key_c, True)]
                keep.append(words)

        def probe_received(env):
            pc = st.__dict__.get("prog_c")
            if pc is None or pc[2] != len(run["probes"]):
                P = self._probe_program(st, [(t, k) for t, k, _ in run["probes"]])
                pc = st.prog_c = (P, [i for i, o in enumerate(P.ops) if o["code"] == abi.X_LOOKUP], len(run["probes"]))
            P = pc[0]
            for i, (t, _, _) in zip(pc[1], run["probes"]):
                P.bind_table(i, t)
            staged = ctx.xstage(st.nc, P)
            try:
                crecv = exchange(staged, cap_c, [abi.I64] + [c.dtype for c in st.ops_c], 1)
            finally:
                staged.free()
                for t, _, owned in run["probes"]:
                    if owned:
                        t.free()                                      # (the words it borrowed stay alive in `keep`)
            crecv[0].set_bounds(pad, hi_g)
            keep.extend(crecv)
            step_c = st.step_c
            if ctx is not self.eng.ctx:                          # (this join runs on a lane: its probe loop on that lane's context and stream)
                by_ctx = st.__dict__.setdefault("step_c_lane", {})
                step_c = by_ctx.get(id(ctx))
                if step_c is None:
                    lane_eng = next(self.eng.lane(k) for k in range(1, self.eng.nlanes) if self.eng.lane(k).ctx is ctx)
                    step_c = by_ctx[id(ctx)] = engine._prepare_scan_fixed(lane_eng, *st.step_c_args)
            res = step_c(env, rows=(G * cap_c, crecv[0], abi.make_tuple(step_c.tuple_shape, crecv[1:])))
            # the status: the largest counts made global, then to the host — queued, in front of the engine's K-F on the same stream
            if not trivial:
                head = stat_t[:4]
                self._note("all_reduce", head)
                self._coll(dist.all_reduce, head, op=dist.ReduceOp.MAX, group=self.group)
            host_t[:abi.EXCHANGE_STAT_WORDS].copy_(stat_t, non_blocking=True)
            return res

        precheck = make_precheck(host, busy)

        self.fast_runs += 1
        if record:
            pg = pp._record(pp._graph_epoch(), after={a_op.out: replicate_a, b_op.out: exchange_b}, replace={c_op.out: probe_received})
            if pg is not None:
                recs.append({"pg": pg, "host_t": host_t, "host": host, "busy": busy, "caps": (cap_b, cap_c), "ctx": ctx, "keep": list(keep)})
                self.graph_recordings += 1
                host[SW - 1] = -1
                pg.graph.launch()
                pg.state = "flying"
                self.graph_launches += 1
                rs = pp._graph_result(pg, precheck=precheck, on_retry=rerun)
                pg.result = weakref.ref(rs)
                return rs
            st.graph_ok = False
            st.graph_refused = pp._graph_refused
            if os.environ.get("SDQLPY_AMD_DIST_DEBUG") == "1":
                print("[dist] %s: the join is not recorded: %s" % (plan.name, st.graph_refused), file=sys.stderr, flush=True)
            pp._graph_refused = None
            busy[0] = False
            self.fast_runs -= 1
            return self._hash_join_device_sized(st, pp, plan, replicate_a, keep, collective_rerun)
        res = pp.run(self._top, after={a_op.out: replicate_a, b_op.out: exchange_b}, replace={c_op.out: probe_received},
                     keep_tables=True, on_retry=rerun, precheck=precheck)
        st.settled_runs = st.__dict__.get("settled_runs", 0) + 1
        if not isinstance(res, engine.DeferredResultSet):
            # every call was waited for (ORDER BY ... LIMIT, profiling): the status is here too
            try:
                precheck()
            except engine.RetryPlan:
                return rerun()
        return res

    def _join_sort_spec(self, st):
        """sdqh_table_topk sort keys for the partitioned join's result columns, or None (host ordering)."""
        spec = []
        for name, direction in self._top[1]:
            desc = direction == "desc"
            found = False
            for fname, e in st.key_fields:
                if isinstance(e, Col) and e.name == st.ckey_name and name == (fname or st.ckey_name):
                    spec.append((abi.SORT_KEY, 0, desc, False)); found = True
                elif isinstance(e, PayloadField) and name == (fname or e.field):
                    j = st.pay_names.index(e.field)
                    spec.append((abi.SORT_PAYLOAD, j, desc, np.dtype(st.pay_dtypes[j]).kind == "f")); found = True
            if not found and name in st.vnames:
                i = st.vnames.index(name)
                if st.count_idx is not None and i == st.count_idx:
                    spec.append((abi.SORT_HITS, 0, desc, False))
                else:
                    spec.append((abi.SORT_VALUE, i - (1 if st.count_idx is not None and st.count_idx < i else 0), desc, True))
                found = True
            if not found:
                raise KeyError("top: the result has no column %r" % name)
        return spec if len(spec) <= abi.MAX_SORT_KEYS else None

    # ---- helpers for tests / reporting ---------------------------------------------------------------
    def gather_rows(self, res):
        """Concatenate the ranks' ResultSet rows on every rank (validation only, not timed)."""
        rows = res.rows() if isinstance(res, ResultSet) else []
        out = [None] * self.world
        self._coll(dist.all_gather_object, out, rows, group=self.group)
        merged = []
        for r in out:
            merged += r
        return sorted(merged)


def default_runner(eng, devices, partition="auto"):
    """The runner behind sdqlpy_init(3, devices=N): joins (or creates, from the launcher's RANK /
    WORLD_SIZE / MASTER_* environment) the process group of this job — RCCL when the engine runs on a
    GPU — and checks that the job really has one process per requested device."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not dist.is_initialized():
        if world != devices:
            raise RuntimeError("sdqlpy_init(devices=%d) needs one process per GPU (python -m torch.distributed.run "
                               "--nproc-per-node %d ...); this process sees WORLD_SIZE=%d" % (devices, devices, world))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        on_gpu = eng.ctx.library.backend_name() == "hip-gfx950"
        if on_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
        else:
            dist.init_process_group("gloo")
    if dist.get_world_size() != devices:
        raise RuntimeError("sdqlpy_init(devices=%d): the process group has %d ranks" % (devices, dist.get_world_size()))
    return DistributedRunner(eng, dist.get_rank(), dist.get_world_size(), partition=partition)


def _chunk_bound(most, current=None):
    """Rows per chunk of the next device-sized exchange, from the largest (source, destination) count the last run saw: an eighth of
    head-room and 1024 rows; a bound that is still roomy (the count shrank by less than a quarter) is kept — every change is a new
    buffer size for the allocator."""
    want = int(most) + int(most) // 8 + 1024
    want += want & 1
    if current is not None and most <= current and current <= want + want // 4:
        return current
    return want


def _is_join_shape(ops):
    return (len(ops) == 4 and all(isinstance(o, ScanOp) for o in ops[:3]) and isinstance(ops[3], FinalizeOp)
            and ops[0].kind == "dict" and ops[0].unique and ops[0].probe is None
            and ops[1].kind == "dict" and ops[1].unique and ops[1].probe is not None and ops[1].probe.dict_name == ops[0].out
            and ops[2].kind == "dict" and not ops[2].unique and ops[2].probe is not None and ops[2].probe.dict_name == ops[1].out)


def _ipreds(flt):
    """(column object, lo, hi) triples kept by abi.make_filter."""
    class P:   # noqa: N801
        pass
    out = []
    for col, lo, hi in flt._keep[0]:
        p = P(); p.col_obj, p.lo, p.hi = col, lo, hi
        out.append(p)
    return out


def _fpreds(flt):
    return list(flt._keep[1])


def _concat_columns(ctx, pieces, dtypes):
    """Concatenate compacted column sets [(cols, n), ...] into one set of columns."""
    total = sum(n for _, n in pieces)
    if len(pieces) == 1:
        return pieces[0][0]
    out = []
    for c, dtype in enumerate(dtypes):
        new = ctx.alloc(total, dtype)
        at = 0
        for cols, n in pieces:
            if n:
                ctx.copy_in(new, at, n, cols[c].data_ptr())
                at += n
        out.append(new)
    return out
