import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from romtime_amd import ops
from oracle import romtime_oracle as oracle
rng = np.random.RandomState(5)
bad = 0
for it in range(14):
    m = int(rng.choice([97, 128, 129, 200, 257, 400, 640, 1000, 1024]))
    N = int(rng.randint(m, m + 3000)) if it % 3 else m
    Phi, _ = np.linalg.qr(rng.standard_normal((N, m)))
    idx, PT_U, margin = ops.deim_greedy(ops.to_device(Phi))
    ref, _, _ = oracle.deim_greedy(Phi)
    same = list(idx.cpu().numpy()) == list(ref)
    uniq = len(set(idx.cpu().numpy().tolist())) == m
    print(N, m, "same" if same else "DIFFERENT", "unique" if uniq else "DUPLICATES", "min margin %.1e" % float(margin.min()), flush=True)
    bad += (not same) or (not uniq)
print("bad", bad)
