"""Throughput mode of the POD for SEQUENCES of independent snapshot sets (the per-parameter time-level PODs of a tree
walk, rom.py:317-366 / deim.py:330-338; the steps of bench.py): the CUs of the MI355X are split between two streams
and the stages of consecutive PODs run side by side.

Why.  One POD is a chain  Gram (FP64 matrix cores, all CUs) -> n x n eigensolve (latency-bound: 32 cooperating
workgroups, most of the chip idle for ~2 ms at n = 512) -> back-projection (HBM-bound).  Nothing inside ONE chain can
overlap, but the Gram of the NEXT snapshot set needs nothing from this one.

How.  ``hipExtStreamCreateWithCUMask`` through ``rt_stream_create_cu_range``: stream E owns CUs [0, e) of every XCD,
stream G the other 32 - e (a queue's mask must leave no XCD empty, so a partition is symmetric over the XCDs;
tools/probes/probe_cumask.hip).  Kernels of the two streams occupy disjoint CUs, so they overlap whatever their launch
order, and the persistent Gram grid is sized for its share ("cu_limit").  Each stream has its own rt_ctx (own scratch
arenas).  Per snapshot set i:

    G:  Gram_i  [all-reduce over the row slabs]                                         (record g_i)
    E:  wait g_i;  scale, tridiagonalise, eigenvalues, k eigenvectors, D^-1 W S^-1      (record e_i)
        [row-sharded: on rank i mod P only; its results are broadcast on a third stream]
    G:  Gram_{i+1};  wait e_i;  Q_i = X_i (D^-1 W S^-1)                                  (record b_i)

so stream G runs  Gram_{i+1} | back-projection_i | Gram_{i+2} ...  without gaps while stream E works one set behind.
The host only enqueues; it reads the eigenvalues of set i (for sigma, the energy curve and the same acceptance checks
``pod_device`` applies) when it hands the result out, ``depth`` sets later.  A set that fails a check (deep spectrum,
clustered eigenvalues, hand-off timeout) is recomputed by ``pod.pod_device`` - the regular, latency-mode route.

Only ``num`` truncation can be enqueued ahead (the number of modes is known before the spectrum, pod.py:51-53);
``tol`` / default truncation take the regular route."""
from __future__ import annotations

import collections
import ctypes as C
import os
import weakref

import numpy as np
import torch

from . import _lib, ops, pod

_p = C.c_void_p


SMALL_SET = 150_000_000   # rows x columns below which a set takes the regular route (see run())
_STREAMS: dict = {}
_STREAMS_PID: list = []
_OPEN_RUNS = weakref.WeakSet()   # PodPipeline.run generators that have not finished (closed by shutdown())


def shutdown():
    """Destroy the process-wide masked streams; registered with ``atexit`` as well, because a process that ends with
    CU-masked queues alive was seen to crash in the profiler's finaliser (rocprofv3).  Safe whatever the caller still
    holds: every tensor a pipeline hands out (Q, colnorm) is allocated on the CALLER's stream before the masked stream
    is entered, so no block that outlives a pipeline call belongs to a masked stream's pool (torch's allocator tags a
    block with the stream it was allocated under and must not meet a destroyed stream at teardown); what was allocated
    under the masked streams are the pipeline's own work buffers, dropped when a set is handed out, and the open
    generators that could still hold some are closed here first."""
    import gc

    if not _STREAMS:
        return
    if _STREAMS_PID and _STREAMS_PID[0] != os.getpid():
        _STREAMS.clear()                      # a forked child: the queues belong to the parent
        return
    try:
        for gen in list(_OPEN_RUNS):
            gen.close()                       # drops the in-flight items (work buffers allocated under masked streams)
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        lib = _lib.load()
        _lib.Context.unbind_streams()         # no ctx keeps the handle of a stream that is about to go
        for st in list(_STREAMS.values()):
            lib.rt_stream_destroy(_p(st.cuda_stream))
    finally:
        _STREAMS.clear()


def _masked_stream(lib, dev, first, count):
    key = (dev, first, count)
    if key not in _STREAMS:
        h = _p()
        rc = lib.rt_stream_create_cu_range(dev, first, count, C.byref(h))
        if rc != 0:
            raise _lib.RomtimeHipError(f"rt_stream_create_cu_range({first}, {count}) failed ({rc})")
        _STREAMS[key] = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", dev))
        _STREAMS_PID[:] = [os.getpid()]
    return _STREAMS[key]


import atexit  # noqa: E402

atexit.register(shutdown)


class PodPipeline:
    def __init__(self, eig_cus_per_xcd: int = 4, device=None, group=None, eig_first_cu: int = 0, gram_range=None,
                 small_set=None):
        """``eig_cus_per_xcd``: CUs of every XCD given to the eigensolver stream (4 -> 32 CUs: a CU per cooperating
        workgroup; 4 and 8 keep the shader engines of an XCD evenly loaded, other values measured slower).
        ``group``: torch.distributed process group of a row-sharded run - the Gram matrices are summed over it on stream
        G (one all-reduce per snapshot set), the small eigenproblem is replicated on every rank (identical inputs,
        deterministic kernels: identical outputs).
        ``eig_first_cu`` / ``gram_range`` = (first, count): explicit per-XCD CU ranges of the two streams, for processes that
        share one GPU (their eigensolver teams must not share CUs: a team spins until all its workgroups are resident).
        ``small_set``: sets with fewer than this many entries (global rows x columns; default SMALL_SET) take the regular
        route when their turn comes - the overlap does not pay for them (see run())."""
        self.small_set = SMALL_SET if small_set is None else int(small_set)
        if not torch.cuda.is_available():
            raise _lib.RomtimeHipError("no MI355X visible: romtime_amd's hot path runs on the GPU only")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        dev = self.device.index
        lib = _lib.load()
        self.lib = lib
        per_xcd = torch.cuda.get_device_properties(dev).multi_processor_count // 8
        e = int(eig_cus_per_xcd)
        if not (1 <= e < per_xcd):
            raise ValueError(f"eig_cus_per_xcd must be in [1, {per_xcd})")
        g_first, g_count = gram_range if gram_range is not None else (eig_first_cu + e, per_xcd - e - eig_first_cu)
        if eig_first_cu < 0 or g_count < 1 or g_first + g_count > per_xcd or not (eig_first_cu + e <= g_first or g_first + g_count <= eig_first_cu):
            raise ValueError("the CU ranges of the two streams must be disjoint and inside the XCD")
        # The masked streams live as long as the process: torch's caching allocator remembers the stream every block
        # was allocated on, so a stream must outlive all tensors produced under it (destroying one made the allocator's
        # teardown abort).  They are cached per (device, CU range) and shared by all pipelines.
        self.sE = _masked_stream(lib, dev, eig_first_cu, e)
        self.sG = _masked_stream(lib, dev, g_first, g_count)
        self.ctxE, self.ctxG = _lib.Context(dev), _lib.Context(dev)
        self.ctxE.set_option("cu_limit", 8 * e)
        self.ctxE.set_option("eig_one_xcd", 0)       # the E CUs span all XCDs: write-through hand-off
        self.ctxG.set_option("cu_limit", 8 * g_count)
        # the paced Gram (L2 sharing between the tiles of an XCD) costs nothing on the whole chip but 2 % on this stream's
        # share of it (5.98 vs 5.87 ms per POD, tools/probes/pace_pipeline_ab.sh): off here unless asked for
        self.ctxG.set_option("gram_pace", int(os.environ.get("ROMTIME_PIPELINE_GRAM_PACE", "0")))
        self.group = group
        # Row-sharded run: the small collectives of stream E (the broadcast of a set's eigen-results from the rank that
        # solved it) go through a process group of their own, so that they never queue behind the Gram all-reduce of the
        # next snapshot set on stream G.
        self.group_e, self.world, self.rank = None, 1, 0
        if group is not None:
            import torch.distributed as dist

            # ROMTIME_FORCE_COLLECTIVES: a one-rank group takes the collective route too - the rehearsal of the RCCL
            # calls (all-reduce on the CU-masked stream G, broadcast on stream C) that a one-GPU box allows
            if dist.get_world_size(group) > 1 or os.environ.get("ROMTIME_FORCE_COLLECTIVES"):
                self.group_e = dist.new_group(ranks=dist.get_process_group_ranks(group), backend=dist.get_backend(group))
                self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.sC = torch.cuda.Stream(self.device) if self.group_e is not None else None   # broadcasts of eigen-results
        # The Gram all-reduce is issued from stream E: only the eigensolve needs the sum, the next Gram on stream G does
        # not.  Issued from stream G, the collective (ProcessGroupNCCL makes the calling stream wait for its own) put two
        # cross-stream hand-overs and the all-reduce on the critical stream of every set (one-rank RCCL rehearsal: 6.29 ms
        # per POD against 5.86 without collectives); a FIFTH stream for it was worse still (7.5 ms: the streams of a
        # process share a handful of hardware queues, and the Gram's queue picked up false dependencies).  Stream E has
        # 1.8 ms of slack per set.
        self.sR = self.sE if self.group_e is not None else None
        self.eig_cus = 8 * e
        self.recomputed = 0                          # sets that failed a check and took the regular route
        self.gram_kernel_ms = []                     # per set: Gram kernels + slab reduction, stream events on stream G
        self.last_stage_ms = {}

    def close(self):
        """Wait for everything in flight (the streams themselves are process-wide, see __init__)."""
        torch.cuda.synchronize(self.device)

    # ---- stages ----------------------------------------------------------------------------------------------
    def _gram(self, item):
        X = item["X"]
        n = X.shape[1]
        with self.ctxG.use(self.sG):
            item["t0"] = torch.cuda.Event(enable_timing=True)
            item["t0"].record()
            Gbuf = torch.empty(n * n + 1, dtype=torch.float64, device=X.device)   # G and the row count: one all-reduce
            item["G"] = ops.gram(X, out=Gbuf[: n * n].view(n, n))
            Gbuf[n * n:].fill_(float(X.shape[0]))
            item["k0"] = torch.cuda.Event(enable_timing=True)
            item["k0"].record()                                                    # Gram kernels + slab reduction end here
            item["Gbuf"] = Gbuf
            if self.sR is None:
                if self.group is not None:      # a one-rank group without forced collectives: nothing to sum
                    import torch.distributed as dist

                    dist.all_reduce(Gbuf, op=dist.ReduceOp.SUM, group=self.group)
                item["g"] = torch.cuda.Event(enable_timing=True)
                item["g"].record()
        if self.sR is not None:
            import torch.distributed as dist

            with torch.cuda.stream(self.sR):
                self.sR.wait_event(item["k0"])
                dist.all_reduce(Gbuf, op=dist.ReduceOp.SUM, group=self.group)
                item["g"] = torch.cuda.Event(enable_timing=True)
                item["g"].record()

    def _eig(self, item):
        """Stream E: the n x n eigenproblem of this set - on every rank of a single-GPU run, on ONE rank (set index mod
        P) of a row-sharded run.  After the all-reduce every rank holds the same G and the sets are independent, so the
        ranks take turns: each rank's eigensolver stream carries 1/P of the sets and the replicated, latency-bound
        tridiagonalisation stops being the Amdahl term of a sequence of PODs (it still is for a single POD:
        pod.pod_device can only split the multisection).  Enqueued right behind the set's Gram, so that every rank has
        its own eigenproblems under way before anybody waits for anybody else's results."""
        k, normalize = item["k"], item["normalize"]
        item["owner"] = item["index"] % self.world
        if self.rank != item["owner"]:
            return
        with self.ctxE.use(self.sE):
            self.sE.wait_event(item["g"])
            G = item["G"]
            n = G.shape[0]
            colnorm, flag = ops.gram_scale(G, normalize)
            lam_d, status = ops.sym_eig_values(G)
            Z = ops.sym_eig_vectors(lam_d, k)
            Zs = ops.backproject_weights(Z, lam_d, colnorm if normalize else None)
            # eigenvalues | column norms | hand-off status | zero-norm flag | D^-1 W S^-1: what the other ranks need
            item["payload"] = torch.cat([lam_d, colnorm, status.to(torch.float64), flag.to(torch.float64), Zs.reshape(-1)])
            item["_keep"] = (lam_d, status, flag, Z, Zs, colnorm)
            item["ec"] = torch.cuda.Event(enable_timing=True)
            item["ec"].record()

    def _share(self, item):
        """The set's eigen-results on every rank: a broadcast from the rank that solved it (stream C, its own process
        group), then the host copy of the spectrum."""
        n, k = item["G"].shape[0], item["k"]
        st = self.sE if self.group_e is None else self.sC
        with torch.cuda.stream(st):
            if self.group_e is not None:
                import torch.distributed as dist

                if self.rank == item["owner"]:
                    st.wait_event(item["ec"])
                else:
                    item["payload"] = torch.empty(n * (k + 2) + 2, dtype=torch.float64, device=item["G"].device)
                dist.broadcast(item["payload"], src=dist.get_global_rank(self.group_e, item["owner"]), group=self.group_e)
            payload = item["payload"]
            item["colnorm"].copy_(payload[n:2 * n])                # handed out: lives on the caller's stream (admit())
            item["Zs"] = payload[2 * n + 2:].view(n, k)
            head = torch.cat([payload[:n], payload[2 * n:2 * n + 2], item["Gbuf"][-1:]])
            item["head"] = torch.empty(head.numel(), dtype=torch.float64).pin_memory()
            item["head"].copy_(head, non_blocking=True)
            item["_keep2"] = head
            item["e"] = torch.cuda.Event(enable_timing=True)
            item["e"].record()

    def _backproject(self, item):
        # on stream G: the tall-skinny product wants the bandwidth of many CUs (on the 32 CUs of stream E it took 30 ms)
        ctx, st = self.ctxG, self.sG
        with ctx.use(st):
            st.wait_event(item["e"])
            ops.gemm_nn(item["X"], item["Zs"], out=item["Q"])      # Q: allocated on the caller's stream by admit()
            item["b"] = torch.cuda.Event(enable_timing=True)
            item["b"].record()

    def _regular(self, item):
        """The set on the caller's context and (unmasked) stream.  Its eigensolver team wants every CU of one XCD,
        stream E's CUs included, where teams of later sets may be spinning: two partly resident teams can starve each
        other into the hand-off's time-out, so stream E is drained first (PodLanes._regular does the same for its
        lane)."""
        self.recomputed += 1
        self.sE.synchronize()
        return pod.pod_device(item["X"], num=item["num"], normalize=item["normalize"], group=self.group)

    def _finish(self, item):
        if item["direct"]:
            return self._regular(item)
        item["b"].synchronize()
        n, k = item["X"].shape[1], item["k"]
        head = item["head"].numpy()
        lam, status, zero_norm, n_rows = head[:n], int(head[n]), int(head[n + 1]), int(round(head[n + 2]))
        if item["normalize"] and zero_norm:
            raise ValueError("array must not contain infs or NaNs (zero-norm snapshot with normalize=True)")
        s = np.sqrt(np.clip(lam, 0.0, None))
        energy = pod._energy(s)
        gaps = lam[:k] - lam[1:k + 1] if k < n else np.r_[lam[:k - 1] - lam[1:k], lam[k - 1]]
        ok = (status == 0 and s[0] > 0 and s[k - 1] >= pod.TWO_PASS_RATIO * s[0]
              and gaps.min() >= pod.RR_GAP * max(lam[0], 1e-300) and n_rows >= n)
        self.last_stage_ms = dict(gram_kernel_ms=item["t0"].elapsed_time(item["k0"]),
                                  gram_allreduce_ms=item["t0"].elapsed_time(item["g"]),
                                  eig_chain_ms=item["g"].elapsed_time(item["e"]),
                                  gram_start_to_basis_ms=item["t0"].elapsed_time(item["b"]))
        self.gram_kernel_ms.append(self.last_stage_ms["gram_kernel_ms"])
        if not ok:
            # what pod_device decides after the fact too: this spectrum needs deflated levels / a Rayleigh-Ritz step
            return self._regular(item)
        return dict(Q=item["Q"], s=s, energy=energy, VT=None, r=k, passes=1, colnorm=item["colnorm"])

    # ---- driver -----------------------------------------------------------------------------------------------
    def run(self, snapshot_sets, num, normalize=True, depth=2):
        """Generator over the results (dicts as ``pod.pod_device`` returns) of ``orth(X, num=num, normalize=normalize)``
        for every X of ``snapshot_sets`` (float64 CUDA tensors, N x n; any iterable, also a lazy one that produces each
        set on the caller's stream when asked for it), in order.  ``depth`` sets are in flight."""
        if not num:
            raise ValueError("PodPipeline enqueues ahead of the spectrum: it needs `num` (pod.py:51-53)")
        gen = self._run(snapshot_sets, num, normalize, depth)
        _OPEN_RUNS.add(gen)
        return gen

    def _run(self, snapshot_sets, num, normalize, depth):
        main = torch.cuda.current_stream(self.device)
        flight = collections.deque()
        it = iter(snapshot_sets)
        self._admitted = 0
        depth = max(depth, self.world + 1)       # enough sets in flight for every rank's eigensolver stream to have one

        def after_producer():
            ready = torch.cuda.Event()
            ready.record(main)
            for st in (self.sG, self.sE, self.sC):
                if st is not None:
                    st.wait_event(ready)

        # A list / tuple exists before the run: ONE event orders the pipeline's streams after whatever produced it.  Any
        # other iterable may produce a set on the caller's stream at the moment it is asked for it: one event per set.
        # (Per set on the LEGACY DEFAULT stream that costs the overlap - an event recorded on the null stream completes
        # only when every blocking stream has drained, the CU-masked ones included: 9.6 instead of 5.8 ms per POD of
        # 1e6 x 512 - so a lazy producer should run under a stream of its own, `with torch.cuda.stream(s): ...`.)
        materialised = isinstance(snapshot_sets, (list, tuple))
        if materialised:
            after_producer()

        def admit():
            try:
                X = next(it)
            except StopIteration:
                return None
            if X.dim() != 2 or not X.is_cuda or X.dtype != torch.float64:
                raise _lib.RomtimeHipError("PodPipeline takes 2-D float64 CUDA tensors")
            item = dict(X=X, num=num, k=int(min(num, X.shape[1])), normalize=bool(normalize), index=self._admitted)
            self._admitted += 1
            n = X.shape[1]
            # what the eigensolver's CU share cannot hold (the 128-workgroup team of n > 512) or the device eigensolver
            # does not take (n < 3) goes the regular route when its turn comes; the order of the results is kept
            # ... and so do sets too small for the overlap to pay: the eigensolve (about 3.5 us per column on the
            # 32-CU share, where its hand-offs cross XCDs) outlasts a Gram of N n^2 / 6e13 s when N n < 2e8, and the
            # whole chip runs such a POD faster on its own (16 sets of 1e5 x 256: 1.34 ms each one after the other,
            # 1.61 through the streams; tools/bench_configs.py c2pipe).  With row shards every rank sees the same
            # (global) shape rule, so the collectives stay matched.
            rows = X.shape[0] * self.world
            item["direct"] = not (3 <= n <= 512) or rows * n < self.small_set
            if not item["direct"]:
                # This set may have been produced on the caller's stream a moment ago (next(it) of a lazy iterable):
                # the pipeline's streams wait for THIS point of the caller's stream, per set.  What is handed out is
                # allocated here, on the caller's stream (see shutdown()).
                item["Q"] = torch.empty((X.shape[0], item["k"]), dtype=torch.float64, device=X.device)
                item["colnorm"] = torch.empty(n, dtype=torch.float64, device=X.device)
                if not materialised:
                    after_producer()
                self._gram(item)
                self._eig(item)
            return item

        pending = collections.deque()
        for _ in range(self.world):               # Grams (and, on their owners, eigensolves) of the first P sets
            item = admit()
            if item is None:
                break
            pending.append(item)
        while pending:
            cur = pending.popleft()
            if not cur["direct"]:
                self._share(cur)
            nxt = admit()                         # the Gram of a later set goes onto stream G BEFORE this set's back-projection
            if nxt is not None:
                pending.append(nxt)
            if not cur["direct"]:
                self._backproject(cur)
            flight.append(cur)
            while len(flight) > depth or (not pending and flight):
                yield self._finish(flight.popleft())
        done = torch.cuda.Event()
        done.record(self.sG)
        main.wait_event(done)                    # results are safe to use on the caller's stream
        done2 = torch.cuda.Event()
        done2.record(self.sE)
        main.wait_event(done2)
        if self.sC is not None:
            done3 = torch.cuda.Event()
            done3.record(self.sC)
            main.wait_event(done3)

    def map(self, snapshot_sets, num, normalize=True, depth=2):
        return list(self.run(snapshot_sets, num, normalize=normalize, depth=depth))


class PodLanes:
    """Many SMALL independent PODs (the tree walks' inner loops: rom.py:317-406, deim.py:279-397): eight at a time.

    A small set's time is its n x n eigensolve - 32 workgroups on ONE XCD handing columns to each other - while the
    other seven XCDs idle.  A lane is a Context whose eigensolver works on XCD ``lane`` (option "eig_xcd") plus a
    stream of its own; set i runs its whole chain (Gram, scaling, eigenvalues, vectors, back-projection) on lane
    i mod 8, so eight chains are on the chip together and no two eigensolver teams ever want the same CUs.
    Everything is enqueued ahead of the spectrum (see run() for how `tol` is served); a set whose spectrum turns out
    to need deflated levels or a Rayleigh-Ritz step is recomputed by ``pod.pod_device``.  Single GPU.  While a ``run()``
    generator is open, PODs of OTHER contexts on the same device may meet a lane's team on their XCD (``map()`` has no
    such window).
    """

    def __init__(self, lanes: int = 8, device=None):
        if not torch.cuda.is_available():
            raise _lib.RomtimeHipError("no MI355X visible: romtime_amd's hot path runs on the GPU only")
        if not (1 <= int(lanes) <= 8):
            raise ValueError("lanes must be 1 .. 8 (one per XCD)")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.ctx, self.streams = [], []
        for lane in range(int(lanes)):
            c = _lib.Context(self.device.index)
            c.set_option("eig_xcd", lane)
            self.ctx.append(c)
            self.streams.append(torch.cuda.Stream(self.device))
        self.recomputed = 0

    def close(self):
        """Wait for everything in flight on the lanes' streams."""
        torch.cuda.synchronize(self.device)

    def _enqueue(self, item, lane):
        X, k, normalize = item["X"], item["k"], item["normalize"]
        n = X.shape[1]
        with self.ctx[lane].use(self.streams[lane]):
            # one host call for the whole chain (rt_pod_enqueue): with a Python call per kernel the host, not the chip,
            # decided how many chains were in flight
            Q, lam_d, status2, colnorm, keep = ops.pod_enqueue(X, k, normalize)
            head = torch.cat([lam_d, status2.to(torch.float64)])
            item["head"] = torch.empty(n + 2, dtype=torch.float64).pin_memory()
            item["head"].copy_(head, non_blocking=True)
            item["Q"], item["colnorm"] = Q, colnorm
            item["_keep"] = (keep, head, lam_d, status2)
            item["done"] = torch.cuda.Event()
            item["done"].record()

    def _regular(self, item):
        """The set on the caller's context and stream.  Its eigensolver team goes to the XCD of THAT context (0 unless
        set otherwise), where a lane may have a team in flight: two teams that each want every CU of one XCD can
        starve each other into the hand-off's time-out, so that lane is drained first."""
        self.recomputed += 1
        xcd = int(_lib.Context.current().options.get("eig_xcd", 0))
        if xcd < len(self.streams):
            self.streams[xcd].synchronize()
        return pod.pod_device(item["X"], num=item["num"], tol=item["tol"], normalize=item["normalize"])

    def _finish(self, item):
        if item["direct"]:
            return self._regular(item)
        item["done"].synchronize()
        n, k = item["X"].shape[1], item["k"]
        head = item["head"].numpy()
        lam, status, zero_norm = head[:n], int(head[n]), int(head[n + 1])
        if item["normalize"] and zero_norm:
            raise ValueError("array must not contain infs or NaNs (zero-norm snapshot with normalize=True)")
        s = np.sqrt(np.clip(lam, 0.0, None))
        energy = pod._energy(s)
        r = pod.truncation_rank(s, energy, num=item["num"], tol=item["tol"])     # orth's own rule, after the fact
        if status != 0 or not (1 <= r <= k) or not s[0] > 0 or item["X"].shape[0] < n:
            return self._regular(item)
        gaps = lam[:r] - lam[1:r + 1] if r < n else np.r_[lam[:r - 1] - lam[1:r], lam[r - 1]]
        if s[r - 1] < pod.TWO_PASS_RATIO * s[0] or gaps.min() < pod.RR_GAP * max(lam[0], 1e-300):
            return self._regular(item)           # deep or clustered among the kept modes: deflated levels / Rayleigh-Ritz
        Q = item["Q"] if r == k else item["Q"][:, :r]
        for t in (item["Q"], item["colnorm"]):                      # allocated under the lane's stream, used by the caller's
            t.record_stream(torch.cuda.current_stream(self.device))
        return dict(Q=Q, s=s, energy=energy, VT=None, r=r, passes=1, colnorm=item["colnorm"])

    def run(self, snapshot_sets, num=None, normalize=True, tol=None, cap=64):
        """Generator over the results (dicts as ``pod.pod_device`` returns) of ``orth(X, num=num, tol=tol,
        normalize=normalize)`` for every X of ``snapshot_sets`` (float64 CUDA tensors, N x n), in order; one set per lane
        is in flight.  Everything is enqueued ahead of the spectrum: with ``num`` alone that many modes are computed;
        with ``tol`` (or neither: the 1e-7 rule) ``cap`` modes are, orth's truncation rule is applied to the spectrum
        afterwards and the basis cut to it - a set that wants more than ``cap`` modes takes the regular route."""
        main = torch.cuda.current_stream(self.device)
        it = iter(snapshot_sets)
        pending = collections.deque()
        admitted = 0
        materialised = isinstance(snapshot_sets, (list, tuple))     # see PodPipeline._run
        if materialised:
            ready = torch.cuda.Event()
            ready.record(main)
            for st in self.streams:
                st.wait_event(ready)

        def admit():
            nonlocal admitted
            try:
                X = next(it)
            except StopIteration:
                return False
            if X.dim() != 2 or not X.is_cuda or X.dtype != torch.float64:
                raise _lib.RomtimeHipError("PodLanes takes 2-D float64 CUDA tensors")
            n = X.shape[1]
            k = int(min(num, n)) if (num and not tol) else int(min(cap, n))
            item = dict(X=X, num=num, tol=tol, k=k, normalize=bool(normalize), direct=not (3 <= n <= 512))
            if not item["direct"]:
                lane = admitted % len(self.streams)
                if not materialised:
                    ready = torch.cuda.Event()   # a lazy iterable produces this set on the caller's stream just now
                    ready.record(main)
                    self.streams[lane].wait_event(ready)
                self._enqueue(item, lane)
            admitted += 1
            pending.append(item)
            return True

        for _ in range(len(self.streams)):
            if not admit():
                break
        while pending:
            out = self._finish(pending.popleft())
            admit()                                # the lane just freed takes the next set
            yield out
        done = torch.cuda.Event()
        for st in self.streams:                    # later work on the caller's stream sees every Q
            done.record(st)
            main.wait_event(done)

    def map(self, snapshot_sets, num=None, normalize=True, tol=None, cap=64):
        return list(self.run(snapshot_sets, num=num, normalize=normalize, tol=tol, cap=cap))


class PodWorkers:
    """Many independent PODs of ANY kind, up to eight at a time: every snapshot set runs ``pod.pod_device`` - the regular
    route with everything it knows: ``tol`` / ``num`` / default truncation, deflated levels for deep spectra, the
    Rayleigh-Ritz step for clusters - in a worker thread, on a stream of its own, through a Context of its own whose
    eigensolver team works on an XCD of its own (option "eig_xcd").

    ``PodLanes`` enqueues whole chains ahead of the spectrum from ONE thread, which is the cheapest form when the number
    of modes is known beforehand and the spectrum is shallow; a set that turns out deep (sigma_r / sigma_1 < 1e-2 - every
    energy tolerance below ~1e-4, i.e. the tree walks' usual setting, rom.py:335) is recomputed there one after the
    other.  Here the host-side decisions of a deep POD (how many modes a level accepts) are taken by the set's own
    thread while the other threads' kernels run: the device sees up to eight dependent chains at once whatever their
    kind.  Sets of up to 1024 columns go through ``rt_pod_orth`` - one foreign call per POD, interpreter lock released;
    wider ones through ``pod.pod_device``, whose host work holds the lock between its waits.  Results come back in
    order."""

    def __init__(self, workers: int = 8, device=None):
        import concurrent.futures
        import queue

        if not torch.cuda.is_available():
            raise _lib.RomtimeHipError("no MI355X visible: romtime_amd's hot path runs on the GPU only")
        if not (1 <= int(workers) <= 8):
            raise ValueError("workers must be 1 .. 8 (one eigensolver team per XCD)")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.workers = int(workers)
        ids = queue.Queue()
        for i in range(self.workers):
            ids.put(i)
        self._tls = __import__("threading").local()
        dev = self.device

        def init():
            torch.cuda.set_device(dev)
            self._tls.index = ids.get()
            self._tls.stream = torch.cuda.Stream(dev)
            with torch.cuda.stream(self._tls.stream):
                ctx = _lib.Context.current()           # this thread's own rt_ctx (own arenas, own counters)
                ctx.set_option("eig_xcd", self._tls.index)

        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=self.workers, initializer=init,
                                                          thread_name_prefix="romtime-pod")

    def _job(self, X, ready, kwargs):
        st = self._tls.stream
        with torch.cuda.stream(st):
            st.wait_event(ready)                       # the set was produced on the caller's stream
            n = X.shape[1]
            if 3 <= n <= 1024 and X.numel() < SMALL_SET:
                # the composite C entry point: the whole POD (truncation rule, deflated levels, Rayleigh-Ritz) in ONE
                # foreign call, during which the interpreter lock is released - the worker threads then really run side
                # by side (with pod.pod_device the Python between a deep POD's kernels serialised them: 1.3x for eight)
                Q, s, energy, levels = ops.pod_orth(X, num=kwargs["num"], tol=kwargs["tol"], normalize=kwargs["normalize"])
                if 2 * Q.shape[1] < Q.stride(0):
                    # with `tol` / default truncation the composite was given room for every column; the kept ones are a
                    # view of that buffer, and a walk holds one result per set until the level is done: compact copy
                    Q = Q.contiguous()
                out = dict(Q=Q, s=s, energy=energy, VT=None, r=int(Q.shape[1]), passes=1 if levels <= 1 else "deflate",
                           levels=levels, colnorm=None)
            else:
                # wider than the composite takes, or a LARGE set: pod_device allocates the basis only once the spectrum
                # has fixed its width (the composite wants room for every column up front - as much again as the set -
                # and eight of those in flight is not what 4 GB sets should cost); such sets are device-bound anyway
                out = pod.pod_device(X, **kwargs)
            done = torch.cuda.Event()
            done.record(st)
        return out, done

    def run(self, snapshot_sets, num=None, tol=None, normalize=True):
        """Generator over the results of ``pod.pod_device(X, num=num, tol=tol, normalize=normalize)`` for every X of
        ``snapshot_sets`` (any iterable, also a lazy one), in order; up to ``workers`` sets are in flight."""
        main = torch.cuda.current_stream(self.device)
        kwargs = dict(num=num, tol=tol, normalize=bool(normalize))
        pending = collections.deque()
        it = iter(snapshot_sets)

        def admit():
            try:
                X = next(it)
            except StopIteration:
                return False
            if X.dim() != 2 or not X.is_cuda or X.dtype != torch.float64:
                raise _lib.RomtimeHipError("PodWorkers takes 2-D float64 CUDA tensors")
            ready = torch.cuda.Event()
            ready.record(main)
            pending.append((X, self.pool.submit(self._job, X, ready, kwargs)))
            return True

        for _ in range(self.workers):
            if not admit():
                break
        while pending:
            X, fut = pending.popleft()
            out, done = fut.result()                   # re-raises what the POD raised (ValueError on a zero-norm set, ...)
            main.wait_event(done)
            for t in (out["Q"], out.get("colnorm")):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(main)              # allocated under the worker's stream, used by the caller's
            admit()
            yield out

    def map(self, snapshot_sets, num=None, tol=None, normalize=True):
        return list(self.run(snapshot_sets, num=num, tol=tol, normalize=normalize))

    def close(self):
        self.pool.shutdown(wait=True)
