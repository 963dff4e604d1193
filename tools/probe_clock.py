"""Does the Gram kernel run faster inside the orth pipeline when the chip is kept busy during the latency-bound
eigensolver phase?  Variants: plain pipeline; a rocBLAS dgemm on a side stream during the eigenvalue stage."""
import sys

import torch

from romtime_amd import ops

dev = torch.device("cuda", 0)
N, n, r = 1_000_000, 512, 40
X = torch.randn(N, n, dtype=torch.float64, device=dev)
fs = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
A = torch.randn(fs, fs, dtype=torch.float64, device=dev)
B = torch.randn(fs, fs, dtype=torch.float64, device=dev)
C = torch.empty_like(A)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
G = torch.empty(n, n, dtype=torch.float64, device=dev)


def step(filler, ev):
    ev[0].record()
    ops.gram(X, out=G)
    ev[1].record()
    Gs = G.clone()
    ops.gram_scale(Gs, True)
    lam, st = ops.sym_eig_values(Gs)
    if filler:
        side.wait_event(ev[1])
        with torch.cuda.stream(side):
            torch.mm(A, B, out=C)
    ev[2].record()
    W = ops.sym_eig_vectors(lam, r)
    Q = ops.gemm_nn(X, W)
    if filler:
        main.wait_stream(side)
    ev[3].record()
    return Q


for filler in (0, 1, 0, 1):
    rows = []
    for it in range(8):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        step(filler, ev)
        torch.cuda.synchronize()
        rows.append((ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3]), ev[0].elapsed_time(ev[3])))
    rows = rows[3:]
    m = [sum(c) / len(c) for c in zip(*rows)]
    print(f"filler={filler} (dgemm {fs}): gram {m[0]:.3f}  eigvals {m[1]:.3f}  vectors+backproj {m[2]:.3f}  total {m[3]:.3f} ms", flush=True)
