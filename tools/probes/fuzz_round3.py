"""Round-3 fuzz: random snapshot sets through (1) pod.orth, (2) the composite rt_pod_orth, (3) the worker threads, against
the oracle's orth (dgesvd) - small column counts included (the single-workgroup tridiagonalisation for n <= 64, the
n <= 128 / 256 instantiations).   python3 tools/probes/fuzz_round3.py [cases] [seed]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import ops, pod
from romtime_amd.pipeline import PodWorkers
from oracle import romtime_oracle as oracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
workers = PodWorkers()
bad = 0
batch = []
def check(tag, Q, s, e, Qo, so, eo, meta):
    global bad
    r = Q.shape[1]
    ok = Q.shape == Qo.shape and (r == 0 or np.abs(s[:r] - so[:r]).max() <= 1e-10 * so[0]) and np.allclose(e, eo, rtol=1e-9, atol=1e-12)
    if ok and r:
        ok = np.linalg.norm(Q @ (Q.T @ Qo) - Qo, 2) < 1e-7 and np.abs(Q.T @ Q - np.eye(r)).max() < 1e-9
    if not ok:
        bad += 1
        print("MISMATCH", tag, meta, Q.shape, Qo.shape, flush=True)
for it in range(cases):
    n = int(rng.choice([3, 5, 16, 31, 33, 48, 64, 65, 100, 127, 128, 129, 200, 256, 257, 400]))
    N = int(rng.randint(max(n, 40), 60000))
    decay = rng.uniform(0.3, 6.0)
    U, _ = np.linalg.qr(rng.standard_normal((N, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    X = (U * 10.0 ** (-decay * np.arange(n) / max(n - 1, 1))) @ V.T * rng.uniform(0.1, 100.0)
    kw = dict(num=int(rng.randint(1, min(n, 40) + 1))) if rng.rand() < 0.5 else dict(tol=1.0 - 10.0 ** (-rng.uniform(2, 8)))
    normalize = bool(rng.rand() < 0.5)
    meta = (N, n, kw, normalize)
    Qo, so, eo = oracle.orth(X, normalize=normalize, **kw)
    Q, s, e = pod.orth(X, normalize=normalize, **kw)
    check("orth", Q, s, e, Qo, so, eo, meta)
    Xd = ops.to_device(X)
    Qc, sc, ec, _ = ops.pod_orth(Xd, normalize=normalize, **kw)
    check("composite", Qc.cpu().numpy(), sc, ec, Qo, so, eo, meta)
    batch.append((Xd, kw, normalize, Qo, so, eo, meta))
    if len(batch) == 8 or it == cases - 1:
        # the worker threads take one truncation rule per run: group by rule
        for Xd, kw, normalize, Qo, so, eo, meta in batch:
            pass
        for key in set((tuple(sorted(b[1].items())), b[2]) for b in batch):
            group = [b for b in batch if (tuple(sorted(b[1].items())), b[2]) == key]
            outs = workers.map([b[0] for b in group], normalize=key[1], **dict(key[0]))
            for b, out in zip(group, outs):
                check("workers", out["Q"].cpu().numpy(), out["s"], out["energy"], b[3], b[4], b[5], b[6])
        batch = []
    if it % 10 == 9:
        print("case", it + 1, "mismatches", bad, flush=True)
workers.close()
print("done:", cases, "cases x 3 routes,", bad, "mismatches")
sys.exit(1 if bad else 0)
