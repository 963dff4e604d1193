"""The N>1 path of the POD on the real device operators: three ranks sharing cuda:0 over gloo (RCCL refuses ranks
that share a device; the collectives are the same torch.distributed calls).  Covers what the CPU rehearsal cannot:
the split of the device eigensolver over the ranks (rt_sym_eig_values_part, per-rank eigenvector shares, all-gather)
behind the Gram all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, X, kwargs, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from romtime_amd import pod
        from romtime_amd._lib import Context

        Context.current().set_option("eig_one_xcd", 0)   # the ranks share one GPU: none of them can have a whole XCD
        rows = np.array_split(np.arange(X.shape[0]), world)[rank]
        out = pod.pod_device(torch.from_numpy(X[rows]).cuda(), group=dist.group.WORLD, **kwargs)
        ret[rank] = dict(rows=rows, Q=out["Q"].cpu().numpy(), s=out["s"], energy=out["energy"], r=out["r"],
                         passes=out["passes"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kwargs", [dict(num=7, normalize=True), dict(tol=1 - 1e-9, normalize=False),
                                    dict(normalize=True, passes="deflate", num=20)])
def test_row_sharded_pod_on_device(kwargs):
    from romtime_amd import pod

    rng = np.random.RandomState(11)
    N, n = 20_000, 45
    U0, _ = np.linalg.qr(rng.standard_normal((N, n)))
    V0, _ = np.linalg.qr(rng.standard_normal((n, n)))
    X = (U0 * 10.0 ** (-np.arange(n) * 0.22)) @ V0.T
    single = pod.pod_device(torch.from_numpy(X).cuda(), **kwargs)
    world = 3          # 45 eigenvalues -> 15 per rank; 7 vectors -> shares of 3 with an overlapping last slice
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), X, kwargs, ret), nprocs=world, join=True)
    assert set(ret.keys()) == set(range(world))
    Qs = single["Q"].cpu().numpy()
    for rank in range(world):
        out = ret[rank]
        assert out["r"] == single["r"] and out["passes"] == single["passes"]
        np.testing.assert_allclose(out["s"], ret[0]["s"], rtol=0, atol=0)            # ranks agree bit for bit
        np.testing.assert_allclose(out["s"][: out["r"]], single["s"][: out["r"]], rtol=1e-11)
        tol = 2e-13 * single["s"][0] + 8 * 2.3e-16 * single["s"][0] ** 2 / np.maximum(single["s"], 1e-300)
        assert np.all(np.abs(out["s"] - single["s"]) <= tol)                      # the one-pass Gram bar of test_surface
        # same subspace rows (signs / rotations inside clusters may differ): compare projectors on the local rows
        Ql, Qr = out["Q"], Qs[out["rows"]]
        cos = np.abs(np.sum(Ql * Qr, axis=0)) / (np.linalg.norm(Ql, axis=0) * np.linalg.norm(Qr, axis=0))
        assert cos.min() > 1 - 1e-8, cos.min()                                           # columns agree up to sign
