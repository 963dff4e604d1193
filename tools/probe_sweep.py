import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from romtime_amd import ops
from romtime_amd.sweep import rom_bdf_sweep
from romtime_amd.testing.mock import AffineBurgers
nt, n_mu, N, r = 100, 32, 100_000, 80
fom = AffineBurgers(N=N, nt=nt, dt=1e-4, bdf2=True, seed=5)
xs = (np.arange(N) + 0.5) / N
V, _ = np.linalg.qr(np.stack([np.sin((k + 1) * np.pi * xs) for k in range(r)], axis=1) + 1e-3 * np.random.RandomState(1).standard_normal((N, r)))
mus = [dict(alpha=0.5 + 0.02 * i, beta=1.0 - 0.01 * i, delta=0.3 + 0.005 * i, omega=7.0 + 0.1 * i) for i in range(n_mu)]
d = fom.descriptor(mus)
args = [ops.to_device(V), d["indptr"], d["indices"], ops.to_device(d["mass"]), ops.to_device(d["terms"]), ops.to_device(d["term_coef"]), ops.to_device(d["tril"]), ops.to_device(d["rhs_terms"]), ops.to_device(d["rhs_coef"]), d["dt"]]
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); rom_bdf_sweep(*args, bdf2=True); torch.cuda.synchronize(); print("sweep s", time.perf_counter() - t0)
