"""Device-resident POD tree walks (rom/rom.py:317-406, deim/deim.py:279-397, deim/nonlinear.py:320-468).

A tree walk is a SEQUENCE of independent small PODs (one per parameter at the time level, one per time step at the
N-MDEIM basis level) followed by the POD of their concatenated bases.  The reference runs them one after the other on
the host and ``np.hstack``s the bases.  Here

  * every snapshot set is uploaded once, when its turn comes (the FOM callbacks that produce it stay on the host),
  * the PODs of a level run eight at a time through ``pipeline.PodWorkers``: one thread, stream and context per set,
    its eigensolver team on an XCD of its own, the whole POD - truncation rule, deflated levels for the deep spectra
    that the walks' energy tolerances ask for, Rayleigh-Ritz for clusters - in one foreign call (``rt_pod_orth``) with
    the interpreter lock released, so the results are those of ``orth`` set by set (``pipeline.PodLanes``, which
    enqueues whole chains ahead of the spectrum from one thread, remains for ``num`` truncation of shallow spectra),
  * the bases never leave the device: the next level's snapshot matrix is ``torch.cat`` of them, and only the final
    basis (the classes keep NumPy attributes, as the reference's pickles do) and the spectra of the reports come back.
"""
from __future__ import annotations

import threading

import torch

from . import ops, pod

_tls = threading.local()


def _runner(device, kind):
    from .pipeline import PodLanes, PodWorkers

    cache = getattr(_tls, "runners", None)
    if cache is None:
        cache = _tls.runners = {}
    key = (device.index, kind)
    if key not in cache:
        if kind == "lanes":
            cache[key] = PodLanes(device=device.index)
        elif kind == "pipeline":
            from .pipeline import PodPipeline

            cache[key] = PodPipeline(device=device.index)
        else:
            cache[key] = PodWorkers(device=device.index)
    return cache[key]


def close_runners():
    """Close the sequence runners this thread has cached (joins the worker threads); the next walk makes new ones."""
    cache = getattr(_tls, "runners", None)
    if not cache:
        return
    for runner in list(cache.values()):
        try:
            runner.close()
        except Exception:
            pass
    cache.clear()


# "workers": every set runs the whole POD in a thread of its own (any truncation rule, deep spectra included);
# "lanes": whole chains enqueued ahead of the spectrum from one thread (`num` truncation of shallow spectra only: deep
# sets are recomputed one after the other).  Measured, 16 device-resident sets of 1e5 x 256, 40 modes + the POD of their
# concatenation: one by one 27.5 ms, lanes 21.1, workers 18.1; 8 deep sets (tol = 1 - 1e-8): 33.4 / 39.8 / 20.9 ms
# (tools/bench_configs.py walk16, walkdeep).  "auto" = workers.
MODE = "auto"


def pod_sequence(snapshot_sets, num=None, tol=None, normalize=True, cap=64):
    """Generator over ``pod.pod_device``-style results (``Q`` on the device, ``s`` / ``energy`` host arrays, ``r``) of
    ``orth(X, num=num, tol=tol, normalize=normalize)`` for every X of ``snapshot_sets``, in order.  ``snapshot_sets``
    may be lazy (each set produced - assembled on the host, uploaded - when asked for): up to eight are asked for
    before the first result is handed out."""
    it = iter(snapshot_sets)
    try:
        first = next(it)
    except StopIteration:
        return

    def chain():
        yield first
        yield from it

    if not first.is_cuda:
        # only reachable with the operators replaced by host stand-ins (tests/cpu_stub.py: the host-logic tests);
        # the real ``ops.to_device`` has produced a CUDA tensor or raised
        for X in chain():
            yield pod.pod_device(X, num=num, tol=tol, normalize=normalize)
        return
    kind = MODE if MODE != "auto" else "workers"
    if MODE == "auto" and num and not tol:
        from .pipeline import SMALL_SET

        if first.numel() >= SMALL_SET:
            # LARGE sets with the number of modes known beforehand: the CU-partitioned pipeline (eigensolve of one set
            # beside the Gram of the next: 5.8 instead of 8.2 ms per 1e6 x 512 set) - what bench.py's headline measures
            kind = "pipeline"
    if kind == "pipeline":
        yield from _runner(first.device, "pipeline").run(chain(), num=num, normalize=normalize)
    elif kind == "lanes":
        yield from _runner(first.device, "lanes").run(chain(), num=num, tol=tol, normalize=normalize, cap=cap)
    else:
        yield from _runner(first.device, "workers").run(chain(), num=num, tol=tol, normalize=normalize)


def pod_of_stack(bases, num=None, tol=None, normalize=True):
    """POD of the column-wise concatenation of device bases (the mu level / time level of a walk): the stacked matrix
    is built on the device (``np.hstack`` in the reference, rom.py:368, deim.py:340) and never visits the host."""
    stacked = torch.cat(list(bases), dim=1)
    out = pod.pod_device(stacked, num=num, tol=tol, normalize=normalize)
    out["stacked_columns"] = int(stacked.shape[1])
    return out


def upload(snapshots):
    """Host snapshot matrix -> device, keeping its memory order (no host transpose)."""
    return ops.to_device(snapshots)


class _Staging:
    """Two pinned host buffers used in turn: the next set is stacked into one while the other's DMA may still be in
    flight (an event per buffer says when it is free again)."""

    def __init__(self):
        self.buf = [None, None]
        self.free = [None, None]
        self.turn = 0

    def take(self, count):
        i = self.turn
        self.turn ^= 1
        if self.free[i] is not None:
            self.free[i].synchronize()
        if self.buf[i] is None or self.buf[i].numel() < count:
            self.buf[i] = torch.empty(count, dtype=torch.float64).pin_memory()
        return i, self.buf[i][:count]


def upload_columns(vectors, zero_first_row=False):
    """The N x n snapshot matrix whose columns are the n host vectors (what ``np.array(list).T`` builds in the
    reference, deim.py:384), on the device, column-major.  The vectors are stacked straight into a reusable pinned
    buffer - ``np.array(list)`` of a 205 MB set spends 21 ms mostly on the page faults of its fresh allocation, the
    stack into resident pinned memory 9 ms - and go over PCIe asynchronously (3.6 ms) while the caller assembles the
    next set (tools/probes/upload_probe.py).  ``zero_first_row``: the MDEIM convention (deim.py:388-389)."""
    if getattr(ops.to_device, "__module__", ops.__name__) != ops.__name__ or not torch.cuda.is_available():
        # the operators have been replaced by host stand-ins (tests/cpu_stub.py: the host-logic tests), or there is no
        # GPU, in which case the real ops.to_device raises
        import numpy as np

        snapshots = np.array(vectors).T
        if zero_first_row:
            snapshots[0, :] = 0.0
        return ops.to_device(snapshots)
    import numpy as np

    staging = getattr(_tls, "staging", None)
    if staging is None:
        staging = _tls.staging = _Staging()
    n, N = len(vectors), int(np.asarray(vectors[0]).size)
    slot, flat = staging.take(n * N)
    host = flat.view(n, N)
    np.stack([np.asarray(v, dtype=np.float64).reshape(-1) for v in vectors], out=host.numpy())
    if zero_first_row:
        host.numpy()[:, 0] = 0.0
    dev = torch.empty((n, N), dtype=torch.float64, device="cuda")
    dev.copy_(host, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    staging.free[slot] = done
    return dev.T
