"""Many random small snapshot sets through PodLanes: answers against pod_device, eigensolver hand-off time-outs counted.
   python3 tools/probes/lanes_stress.py [sets] [seed]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import ops, pod, pipeline
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
Xs = []
for i in range(sets):
    N, n = int(rng.randint(600, 60000)), int(rng.choice([8, 24, 64, 100, 200, 256, 384, 512]))
    N = max(N, n + 5)
    U, _ = np.linalg.qr(rng.standard_normal((N, n)))
    V, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Xs.append(ops.to_device((U * 10.0 ** (-rng.uniform(0.5, 4.0) * np.arange(n) / max(n - 1, 1))) @ V.T))
lanes = pipeline.PodLanes()
t0 = time.time()
outs = lanes.map(Xs, num=6, normalize=True)
torch.cuda.synchronize()
t1 = time.time()
bad = 0
for X, out in zip(Xs, outs):
    ref = pod.pod_device(X, num=6, normalize=True)
    k = min(6, X.shape[1])
    if out["r"] != ref["r"] or np.abs(out["s"][:k] - ref["s"][:k]).max() > 1e-11 * ref["s"][0]:
        bad += 1
        continue
    Q, Qr = out["Q"], ref["Q"]
    if float(torch.linalg.matrix_norm(Q @ (Q.T @ Qr) - Qr, 2)) > 1e-9:
        bad += 1
print("sets", sets, "lanes time %.1f ms per set" % (1e3 * (t1 - t0) / sets), "mismatches", bad, "recomputed", lanes.recomputed,
      "timeouts", sum(c.counter("eig_timeouts") for c in lanes.ctx), "one_xcd", sum(c.counter("eig_one_xcd") for c in lanes.ctx),
      "general_form", sum(c.counter("eig_general_form") for c in lanes.ctx))
sys.exit(1 if bad else 0)
