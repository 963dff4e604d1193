import torch, numpy as np, time
from romtime_amd import ops
from romtime_amd._lib import Context
ctx = Context.current()
import ctypes as C
def ptr(t): return C.c_void_p(t.data_ptr())
for r in (16, 32, 80, 128):
    for B in (1, 32, 256):
        K = torch.randn(B, r, r, dtype=torch.float64, device="cuda") + 4 * torch.eye(r, dtype=torch.float64, device="cuda")
        b = torch.randn(B, r, dtype=torch.float64, device="cuda")
        info = torch.zeros(B, dtype=torch.int32, device="cuda")
        bw = b.clone()
        def run():
            ctx.check(ctx.lib.rt_dense_solve_batched(ctx.handle, ptr(K), ptr(bw), r, B, ptr(info)), "solve")
        for _ in range(3): bw.copy_(b); run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        bw.copy_(b); run()
        res = (torch.einsum("bij,bj->bi", K, bw) - b).abs().max().item()
        print(f"r={r} B={B}: {us:.1f} us/call, {us*2.35e3/r:.0f} cycles/column, residual {res:.1e}", flush=True)
