"""The N>1 path of the POD (row-sharded snapshots, Gram all-reduce) with gloo on the CPU, world size 2.

The device operators are stubbed with the oracle's arithmetic (tests/cpu_stub.py); what is under test
is the distributed logic of ``pod.pod_device``: where the all-reduce sits, that every rank derives the
same spectrum / truncation, and that the row-sharded Q equals the unsharded one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, X, kwargs, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from romtime_amd import ops, pod
        from tests import cpu_stub

        for name in ("to_device", "to_device_index", "gram", "gram_scale", "gemm_tn", "gemm_nn", "rank_update"):
            setattr(ops, name, getattr(cpu_stub, name))
        rows = np.array_split(np.arange(X.shape[0]), world)[rank]
        out = pod.pod_device(torch.from_numpy(X[rows]), group=dist.group.WORLD, **kwargs)
        ret[rank] = dict(rows=rows, Q=out["Q"].numpy(), s=out["s"], energy=out["energy"], r=out["r"],
                         passes=out["passes"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kwargs", [dict(num=6, normalize=True), dict(tol=1 - 1e-9, normalize=False),
                                    dict(normalize=True, passes=2), dict(normalize=False), dict(normalize=True, passes="deflate", num=9)])
def test_row_sharded_pod_matches_single_process(cpu_ops, kwargs):
    rng = np.random.RandomState(7)
    U0, _ = np.linalg.qr(rng.standard_normal((301, 12)))
    V0, _ = np.linalg.qr(rng.standard_normal((12, 12)))
    X = (U0 * 10.0 ** (-np.arange(12) * 0.6)) @ V0.T
    from romtime_amd import pod

    single = pod.pod_device(torch.from_numpy(X), **kwargs)
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), X, kwargs, ret), nprocs=world, join=True)
    assert set(ret.keys()) == {0, 1}
    for rank in range(world):
        out = ret[rank]
        assert out["r"] == single["r"] and out["passes"] == single["passes"]
        np.testing.assert_allclose(out["s"], single["s"], rtol=0, atol=1e-13 * single["s"][0])
        np.testing.assert_allclose(out["energy"], single["energy"], rtol=1e-12)
        Qs = single["Q"].numpy()[out["rows"]]
        for i in range(out["r"]):
            err = min(np.linalg.norm(out["Q"][:, i] - Qs[:, i]), np.linalg.norm(out["Q"][:, i] + Qs[:, i]))
            assert err < 1e-9, (rank, i, err)
    # both ranks computed bit-identical spectra (replicated eigensolve on an identical all-reduced G)
    np.testing.assert_array_equal(ret[0]["s"], ret[1]["s"])
