"""What does the Gram kernel's duration inside the orth pipeline depend on?  Times every Gram of the sequence
G G G E G B G G E B G (E = eigenvalues + eigenvectors, B = back-projection) and, with an argument, the same with a
rocBLAS dgemm of that size on a side stream during E."""
import sys

import torch

from romtime_amd import ops

dev = torch.device("cuda", 0)
N, n, r = 1_000_000, 512, 40
X = torch.randn(N, n, dtype=torch.float64, device=dev)
fs = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if fs:
    A = torch.randn(fs, fs, dtype=torch.float64, device=dev)
    B = torch.randn(fs, fs, dtype=torch.float64, device=dev)
    C = torch.empty_like(A)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
G = torch.empty(n, n, dtype=torch.float64, device=dev)
state = {}


def do(op, times):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if op == "G":
        ops.gram(X, out=G)
    elif op == "E":
        Gs = G.clone()
        ops.gram_scale(Gs, True)
        lam, st = ops.sym_eig_values(Gs)
        if fs:
            side.wait_event(e0)
            with torch.cuda.stream(side):
                torch.mm(A, B, out=C)
        state["W"] = ops.sym_eig_vectors(lam, r)
        if fs:
            main.wait_stream(side)
    elif op == "B":
        state["Q"] = ops.gemm_nn(X, state["W"])
    e1.record()
    times.append((op, e0, e1))


seq = "GGGEGBGGEBG"
acc = None
for it in range(6):
    times = []
    for op in seq:
        do(op, times)
    torch.cuda.synchronize()
    ms = [e0.elapsed_time(e1) for _, e0, e1 in times]
    if it >= 2:
        acc = ms if acc is None else [a + b for a, b in zip(acc, ms)]
print("filler dgemm", fs, " ".join(f"{op}:{a / 4:.2f}" for op, a in zip(seq, acc)), flush=True)
