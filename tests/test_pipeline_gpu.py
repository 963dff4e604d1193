"""PodPipeline (romtime_amd/pipeline.py): sequences of independent PODs with the eigensolve of one snapshot set running
beside the Gram kernel of the next on CU-partitioned streams.  Every set's result must be what ``pod.pod_device`` (one
POD at a time) and the oracle give; sets whose spectrum fails the single-pass checks take the regular route."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import romtime_oracle as oracle

pytestmark = pytest.mark.gpu
EPS = 2.2e-16


def _matrix(rng, N, n, decay):
    U, _ = np.linalg.qr(rng.standard_normal((N, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (U * 10.0 ** (-decay * np.arange(n) / (n - 1))) @ V.T


def _same_columns(Q, Qr, tol):
    for i in range(Q.shape[1]):
        assert min(np.linalg.norm(Q[:, i] - Qr[:, i]), np.linalg.norm(Q[:, i] + Qr[:, i])) <= tol, i


def test_pipeline_matches_single_pods_and_oracle():
    from romtime_amd import ops, pod
    from romtime_amd.pipeline import PodPipeline

    rng = np.random.RandomState(7)
    k = 6
    mats = [_matrix(rng, 9000, 96, 1.5), _matrix(rng, 20000, 130, 2.0), _matrix(rng, 9000, 96, 60.0),   # deep: regular route
            np.asfortranarray(_matrix(rng, 7001, 200, 1.0)), _matrix(rng, 4096, 512, 3.0), _matrix(rng, 9000, 96, 1.5) * 7.0,
            _matrix(rng, 3000, 600, 2.0)]                                                                   # n > 512: regular route
    dev = [ops.to_device(m) for m in mats]
    pipe = PodPipeline(small_set=0)      # the test's sets are small: keep them on the streams
    outs = pipe.map(dev, num=k, normalize=True)
    assert len(outs) == len(mats) and pipe.recomputed == 2
    for X, Xd, out in zip(mats, dev, outs):
        single = pod.pod_device(Xd, num=k, normalize=True)
        assert out["r"] == single["r"] == k and out["passes"] == single["passes"]
        Qo, so, eo = oracle.orth(X, num=k, normalize=True)
        bar = 2e-13 * so[0] + 8 * EPS * so[0] ** 2 / np.maximum(so, 1e-300)
        assert np.all(np.abs(out["s"] - so) <= bar) and np.all(np.abs(single["s"] - so) <= bar)
        np.testing.assert_allclose(out["energy"], eo, rtol=1e-10)
        Q = out["Q"].cpu().numpy()
        assert np.abs(Q.T @ Q - np.eye(k)).max() < 1e-10
        _same_columns(Q, single["Q"].cpu().numpy(), 1e-11)
        _same_columns(Q, Qo, 1e-9)
    # the generator form hands results out in order while later sets are in flight, and works again after a run
    again = [o["s"] for o in pipe.run(dev[:2], num=k, normalize=False)]
    for Xd, s in zip(dev[:2], again):
        np.testing.assert_allclose(s[:k], pod.pod_device(Xd, num=k, normalize=False)["s"][:k], rtol=1e-11)
    with pytest.raises(ValueError):
        list(pipe.run(dev[:1], num=None))
    Z = ops.to_device(np.c_[mats[0][:, :5], np.zeros((9000, 1))])
    with pytest.raises(ValueError):                                   # zero-norm column + normalize, as orth does
        pipe.map([Z], num=3, normalize=True)
    pipe.close()


def test_lanes_match_single_pods_and_oracle():
    """PodLanes: eight small PODs on the chip at a time (one eigensolver team per XCD, option "eig_xcd"; one host call per
    chain, rt_pod_enqueue).  Same answers as pod_device and the oracle, in order, for more sets than lanes, both
    memory orders, a deep spectrum (recomputed on the regular route) and n > 512 (regular route)."""
    from romtime_amd import ops, pod
    from romtime_amd._lib import Context
    from romtime_amd.pipeline import PodLanes

    rng = np.random.RandomState(11)
    k = 5
    mats = [_matrix(rng, 6000 + 500 * i, 64 + 16 * (i % 5), 1.0 + 0.2 * i) for i in range(19)]
    mats[3] = np.asfortranarray(mats[3])
    mats[7] = _matrix(rng, 9000, 96, 60.0)          # deep: regular route after the fact
    mats[12] = _matrix(rng, 3000, 600, 2.0)         # n > 512: regular route
    dev = [ops.to_device(m) for m in mats]
    lanes = PodLanes()
    outs = lanes.map(dev, num=k, normalize=True)
    assert len(outs) == len(mats) and lanes.recomputed == 2
    timeouts = sum(c.counter("eig_timeouts") for c in lanes.ctx)
    assert timeouts == 0
    assert sum(c.counter("eig_one_xcd") for c in lanes.ctx) >= len(mats) - 2     # every team found its XCD
    for X, Xd, out in zip(mats, dev, outs):
        single = pod.pod_device(Xd, num=k, normalize=True)
        Qo, so, eo = oracle.orth(X, num=k, normalize=True)
        bar = 2e-13 * so[0] + 8 * EPS * so[0] ** 2 / np.maximum(so, 1e-300)
        assert out["r"] == k and np.all(np.abs(out["s"] - so) <= bar)
        np.testing.assert_allclose(out["energy"], eo, rtol=1e-10)
        Q = out["Q"].cpu().numpy()
        assert np.abs(Q.T @ Q - np.eye(k)).max() < 1e-10
        _same_columns(Q, single["Q"].cpu().numpy(), 1e-11)
        _same_columns(Q, Qo, 1e-9)
    # energy truncation (what the tree walks ask for, rom.py:165-183): `cap` modes are computed ahead of the spectrum and
    # orth's rule cuts them afterwards; a set that wants more than `cap` takes the regular route
    before = lanes.recomputed
    tol = 1.0 - 1e-3
    outs_t = lanes.map(dev[:10], tol=tol, normalize=False, cap=8)
    for X, out in zip(mats[:10], outs_t):
        Qo, so, eo = oracle.orth(X, tol=tol, normalize=False)
        assert out["r"] == Qo.shape[1] and out["Q"].shape[1] == Qo.shape[1], (out["r"], Qo.shape)
        np.testing.assert_allclose(out["energy"], eo, rtol=1e-10)
        if Qo.shape[1]:
            _same_columns(out["Q"].cpu().numpy(), Qo, 1e-9)
    assert lanes.recomputed - before >= 1          # mats[7] is deep
    Z = ops.to_device(np.c_[mats[0][:, :5], np.zeros((mats[0].shape[0], 1))])
    with pytest.raises(ValueError):                                   # zero-norm column + normalize, as orth does
        lanes.map([Z], num=3, normalize=True)


def test_cu_partition_abi():
    """rt_stream_create_cu_range / "cu_limit": argument checks, and kernels on a masked stream give the same numbers."""
    import ctypes as C

    from romtime_amd import _lib, ops

    lib = _lib.load()
    h = C.c_void_p()
    assert lib.rt_stream_create_cu_range(0, 0, 0, C.byref(h)) < 0          # empty range
    assert lib.rt_stream_create_cu_range(0, 30, 4, C.byref(h)) < 0         # beyond the 32 CUs of an XCD
    assert lib.rt_stream_create_cu_range(0, 8, 8, C.byref(h)) == 0 and h.value
    ctx = _lib.Context(0)
    with pytest.raises(_lib.RomtimeHipError):
        ctx.set_option("cu_limit", 12)                                    # not a multiple of 8
    ctx.set_option("cu_limit", 64)
    st = torch.cuda.ExternalStream(h.value)
    X = torch.randn((60000, 256), dtype=torch.float64, device="cuda")
    ref = ops.gram(X)
    with ctx.use(st):
        G = ops.gram(X)
        assert ctx.launch_info()["grid"] <= 2 * 64                        # the persistent grid follows the CU share
    st.synchronize()
    assert float((G - ref).abs().max().item()) <= 1e-12 * float(ref.abs().max().item())
    ctx.set_option("cu_limit", 0)
    # (this stream stays: torch's allocator remembers the stream of every block handed out under it)
    h2 = C.c_void_p()
    assert lib.rt_stream_create_cu_range(0, 0, 4, C.byref(h2)) == 0 and lib.rt_stream_destroy(h2) == 0   # an unused one


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mats, k, ret):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from romtime_amd.pipeline import PodPipeline

        # two processes on ONE GPU: eigensolver teams on disjoint CUs (a team spins until it is resident), Gram streams shared
        pipe = PodPipeline(group=dist.group.WORLD, eig_first_cu=4 * rank, gram_range=(8, 24), small_set=0)
        local = [torch.from_numpy(np.array_split(m, world)[rank]).cuda() for m in mats]
        outs = pipe.map(local, num=k, normalize=True)
        ret[rank] = [dict(Q=o["Q"].cpu().numpy(), s=o["s"], r=o["r"]) for o in outs]
        pipe.close()
    finally:
        dist.destroy_process_group()


def test_row_sharded_pipeline_two_ranks_one_gpu():
    import torch.multiprocessing as mp

    from romtime_amd import ops, pod

    rng = np.random.RandomState(3)
    k, world = 5, 2
    mats = [_matrix(rng, 12000, 64, 1.2), _matrix(rng, 12000, 140, 1.8), _matrix(rng, 8000, 64, 1.2)]
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), mats, k, ret), nprocs=world, join=True)
    for j, X in enumerate(mats):
        single = pod.pod_device(ops.to_device(X), num=k, normalize=True)
        Qs = single["Q"].cpu().numpy()
        np.testing.assert_array_equal(ret[0][j]["s"], ret[1][j]["s"])          # replicated eigenproblem: ranks agree bitwise
        np.testing.assert_allclose(ret[0][j]["s"][:k], single["s"][:k], rtol=1e-11)
        Q = np.concatenate([ret[r][j]["Q"] for r in range(world)], axis=0)      # row slabs stack back into the basis
        _same_columns(Q, Qs, 1e-10)


def test_lazy_snapshot_sets_are_ordered_after_their_producer():
    """A lazy iterable produces each snapshot set on the caller's stream when the pipeline asks for it (the tree-walk
    use: one set per parameter).  The pipeline's streams must wait for THAT point of the caller's stream, per set
    (the `ready` event used to be recorded once, before any set was pulled).  The producer here is slow on
    purpose - a long chain of kernels that fills the buffer last - so a Gram that does not wait sees the wrong data."""
    from romtime_amd import ops, pod
    from romtime_amd.pipeline import PodLanes, PodPipeline

    rng = np.random.RandomState(11)
    k = 5
    host = [_matrix(rng, 6000, 64, 1.0 + 0.2 * i) for i in range(5)]
    final = [ops.to_device(m) for m in host]
    burn = torch.randn(2048, 2048, device="cuda")

    def lazy():
        for Xf in final:
            X = torch.zeros_like(Xf)           # what a Gram that runs too early would read
            for _ in range(30):
                burn @ burn                    # ~ms of work on the caller's stream before the set exists
            X.copy_(Xf)
            yield X

    want = [pod.pod_device(Xf, num=k, normalize=True)["s"] for Xf in final]
    pipe = PodPipeline(small_set=0)
    for out, s in zip(pipe.run(lazy(), num=k, normalize=True), want):
        np.testing.assert_allclose(out["s"][:k], s[:k], rtol=1e-11)
    pipe.close()
    lanes = PodLanes()
    for out, s in zip(lanes.run(lazy(), num=k, normalize=True), want):
        np.testing.assert_allclose(out["s"][:k], s[:k], rtol=1e-11)


_EXIT_CHILD = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, {root!r})
from romtime_amd import ops
from romtime_amd.pipeline import PodPipeline
rng = np.random.RandomState(3)
sets = [ops.to_device(rng.standard_normal((5000, 48))) for _ in range(4)]
pipe = PodPipeline(small_set=0)
KEPT = pipe.map(sets, num=4, normalize=True)          # module global: alive when the interpreter exits
OPEN = pipe.run(sets, num=4, normalize=True)          # and a generator that was never finished
FIRST = next(OPEN)
assert KEPT[0]["Q"].shape == (5000, 4) and float(KEPT[0]["Q"].abs().sum()) > 0
print("child done", flush=True)
"""


def test_process_exits_cleanly_with_pipeline_results_alive():
    """Results of PodPipeline held in module globals at interpreter exit, plus an unfinished generator: the process must
    end with status 0.  (Round 2: blocks tagged with a CU-masked stream met torch's allocator teardown after the stream
    had been destroyed.  Now everything handed out is allocated on the caller's stream; run ONCE, in a child.)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", _EXIT_CHILD.format(root=root)], capture_output=True, text=True, timeout=300)
    assert "child done" in res.stdout, res.stderr[-3000:]
    assert res.returncode == 0, (res.returncode, res.stderr[-3000:])


def test_workers_match_single_pods_any_truncation():
    """PodWorkers: every snapshot set's whole POD in a thread, stream and context of its own (rt_pod_orth, interpreter
    lock released), eight at a time - `tol`, `num` and default truncation, shallow and deep spectra, a set wider than the
    composite takes (pod.pod_device in the worker), errors re-raised in order."""
    from romtime_amd import ops, pod
    from romtime_amd.pipeline import PodWorkers

    rng = np.random.RandomState(21)
    mats = [_matrix(rng, 6000, 64, 1.0), _matrix(rng, 6000, 64, 9.0), _matrix(rng, 9000, 130, 5.0), _matrix(rng, 4000, 48, 14.0),
            np.asfortranarray(_matrix(rng, 7001, 96, 3.0)), _matrix(rng, 2500, 1100, 2.0), _matrix(rng, 6000, 64, 1.0) * 3.0,
            _matrix(rng, 5000, 200, 11.0), _matrix(rng, 6000, 64, 6.0), _matrix(rng, 6000, 80, 2.0)]
    dev = [ops.to_device(m) for m in mats]
    workers = PodWorkers()
    for kw in (dict(tol=1.0 - 1e-9), dict(num=7), dict()):
        outs = workers.map(dev, normalize=True, **kw)
        assert len(outs) == len(mats)
        for X, Xd, out in zip(mats, dev, outs):
            single = pod.pod_device(Xd, normalize=True, **kw)
            Qo, so, eo = oracle.orth(X, normalize=True, **kw)
            assert out["r"] == single["r"] == Qo.shape[1], kw
            bar = 2e-13 * so[0] + 8 * EPS * so[0] ** 2 / np.maximum(so, 1e-300)
            assert np.all(np.abs(out["s"] - so) <= bar)
            Q = out["Q"].cpu().numpy()
            assert np.abs(Q.T @ Q - np.eye(out["r"])).max() < 1e-9
            assert np.linalg.norm(Q - Qo @ (Qo.T @ Q), 2) <= 1e-7          # same subspace as dgesvd's columns
    Z = ops.to_device(np.c_[mats[0][:, :5], np.zeros((6000, 1))])
    with pytest.raises(ValueError):                                   # zero-norm column + normalize, as orth does
        workers.map([dev[0], Z, dev[1]], num=3, normalize=True)
    again = workers.map(dev[:3], num=5, normalize=False)              # the pool works on after an error
    assert [o["r"] for o in again] == [5, 5, 5]
    workers.close()


def test_fork_after_worker_threads_is_safe():
    """A child forked after PodWorkers has run (multiprocessing.Manager, a DataLoader ...) drops the worker threads'
    thread-local contexts; their finalisers must not call into HIP there (round 3: segmentation fault in the child)."""
    import multiprocessing as mp

    from romtime_amd import ops
    from romtime_amd.pipeline import PodWorkers

    rng = np.random.RandomState(5)
    workers = PodWorkers(workers=3)
    outs = workers.map([ops.to_device(_matrix(rng, 3000, 32, 1.0)) for _ in range(4)], num=3)
    assert len(outs) == 4
    m = mp.get_context("fork").Manager()       # forks a server process and waits for its address
    d = m.dict()
    d["ok"] = 1
    assert d["ok"] == 1
    m.shutdown()
    workers.close()


def test_walk_sequences_of_large_sets_take_the_pipeline():
    """walks.pod_sequence routes LARGE snapshot sets with `num` truncation through PodPipeline (the form bench.py
    measures) and everything else through the worker threads; same results either way."""
    from romtime_amd import ops, pod, walks

    rng = np.random.RandomState(41)
    big = [ops.to_device(_matrix(rng, 400_000, 384, 2.0)) for _ in range(3)]          # 1.5e8 entries each
    outs = list(walks.pod_sequence(big, num=12, normalize=True))
    assert any(k[1] == "pipeline" for k in walks._tls.runners)
    for Xd, out in zip(big, outs):
        single = pod.pod_device(Xd, num=12, normalize=True)
        np.testing.assert_allclose(out["s"][:12], single["s"][:12], rtol=1e-11)
        Q, Qs = out["Q"].cpu().numpy(), single["Q"].cpu().numpy()
        _same_columns(Q, Qs, 1e-10)
    small = [ops.to_device(_matrix(rng, 5000, 64, 2.0)) for _ in range(3)]
    outs = list(walks.pod_sequence(small, num=5, normalize=True))
    assert [o["r"] for o in outs] == [5, 5, 5]


def test_package_shutdown_gives_everything_back_and_the_package_works_on():
    """romtime_amd.shutdown(): runners closed (worker threads joined), masked streams and every context destroyed, in
    that order; the next call makes new ones and computes the same bits."""
    import threading

    import romtime_amd
    from romtime_amd import ops, pod, walks
    from romtime_amd._lib import Context

    rng = np.random.RandomState(77)
    sets = [ops.to_device(_matrix(rng, 5000, 48, 2.0)) for _ in range(4)]
    before = list(walks.pod_sequence(sets, tol=1.0 - 1e-9, normalize=True))
    big = pod.pod_device(sets[0], num=5, normalize=True)
    assert any(t.name.startswith("romtime-pod") for t in threading.enumerate())
    romtime_amd.shutdown()
    assert not any(t.name.startswith("romtime-pod") for t in threading.enumerate())
    assert len(Context._live) == 0 and not walks._tls.runners
    after = list(walks.pod_sequence(sets, tol=1.0 - 1e-9, normalize=True))
    for a, b in zip(before, after):
        assert a["r"] == b["r"] and torch.equal(a["Q"], b["Q"])
        np.testing.assert_array_equal(a["s"], b["s"])
    again = pod.pod_device(sets[0], num=5, normalize=True)
    assert torch.equal(big["Q"], again["Q"])
