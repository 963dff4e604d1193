"""Random snapshot sets through romtime_amd.pod.orth against the oracle's orth (dgesvd): sigma, energy, spans.
   python3 tools/probes/fuzz_orth.py [cases] [seed]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import pod
from oracle import romtime_oracle as oracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(cases):
    n = int(rng.choice([3, 7, 33, 64, 100, 128, 200, 256, 300, 512, 600]))
    N = int(rng.randint(max(n, 50), 120000))
    decay = rng.uniform(0.3, 5.0)
    U, _ = np.linalg.qr(rng.standard_normal((N, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    X = (U * 10.0 ** (-decay * np.arange(n) / max(n - 1, 1))) @ V.T * rng.uniform(0.1, 100.0)
    if rng.rand() < 0.5:
        X = np.asfortranarray(X)
    kw = dict(num=int(rng.randint(1, min(n, 40) + 1))) if rng.rand() < 0.5 else dict(tol=1.0 - 10.0 ** (-rng.uniform(2, 8)))
    normalize = bool(rng.rand() < 0.5)
    Q, s, e = pod.orth(X, normalize=normalize, **kw)
    Qo, so, eo = oracle.orth(X, normalize=normalize, **kw)
    r = Q.shape[1]
    ok = Q.shape == Qo.shape and (r == 0 or np.abs(s[:r] - so[:r]).max() <= 1e-10 * so[0]) and np.allclose(e, eo, rtol=1e-9, atol=1e-12)
    if ok and Q.shape[1]:
        ok = np.linalg.norm(Q @ (Q.T @ Qo) - Qo, 2) < 1e-7 and np.abs(Q.T @ Q - np.eye(Q.shape[1])).max() < 1e-9
    if not ok:
        bad += 1
        print("MISMATCH", N, n, kw, normalize, Q.shape, Qo.shape, flush=True)
    if it % 10 == 9:
        print("case", it + 1, "mismatches", bad, flush=True)
print("done:", cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
