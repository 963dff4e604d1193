"""Randomised shapes through the round's new kernel paths, against torch on the device:
  * rt_gram: one-launch / two-launch plans, shifted last panel (even n), predicated loader (odd n), both memory orders,
    padded leading dimensions, ragged K tails;
  * rt_gemm_tn with m <= 16 modes (streaming kernel) and above (generic), slices of wider buffers.
python3 tools/probes/fuzz_gram_tn.py [cases] [seed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import ops  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def rnd(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


bad = 0
for c in range(cases):
    n = rnd(97, 1024) if c % 3 else 128 * rnd(1, 8) - rnd(0, 1) * rnd(1, 127)
    N = rnd(8192, 400_000) if n > 400 else rnd(8192, 1_200_000)
    pad = rnd(0, 3)
    order = "F" if rnd(0, 2) == 0 else "C"
    if order == "C":
        X = torch.randn((N, n + pad), dtype=torch.float64, device="cuda")[:, :n]
    else:
        X = torch.randn((n, N + pad), dtype=torch.float64, device="cuda")[:, :N].T
    G = ops.gram(X)
    ref = X.T @ X
    err = float((G - ref).abs().max() / ref.abs().max())
    ok = err < 1e-12 and bool(torch.equal(G, G.T))
    m = rnd(1, 24)
    Qw = torch.randn((N, m + 4), dtype=torch.float64, device="cuda")
    C = ops.gemm_tn(Qw[:, 1:1 + m], X)
    cref = Qw[:, 1:1 + m].T @ X
    cerr = float((C - cref).abs().max() / cref.abs().max())
    ok = ok and cerr < 1e-12
    bad += not ok
    print(f"{c:3d} N={N:8d} n={n:5d} pad={pad} {order} gram err {err:.1e} | m={m:2d} tn err {cerr:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
    del X, G, ref, Qw, C, cref
print("mismatches:", bad)
sys.exit(1 if bad else 0)
