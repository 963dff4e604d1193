import time, numpy as np, torch
n=512
rng=np.random.RandomState(0)
s=10.0**(-8*np.arange(n)/(n-1)); V,_=np.linalg.qr(rng.standard_normal((n,n))); G=(V*s**2)@V.T; G=(G+G.T)/2
Gd=torch.from_numpy(G).cuda()
for name,fn in [("eigh",lambda: torch.linalg.eigh(Gd)),("eigvalsh",lambda: torch.linalg.eigvalsh(Gd))]:
    for _ in range(2): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(5): out=fn()
    torch.cuda.synchronize(); print(name, n, (time.time()-t)/5*1e3, "ms")
lam,W=torch.linalg.eigh(Gd)
print("resid", float((Gd@W-W*lam).abs().max()), "orth", float((W.T@W-torch.eye(n,device='cuda',dtype=torch.float64)).abs().max()))
import os
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
from threadpoolctl import threadpool_limits, threadpool_info
print([ (d['internal_api'], d['num_threads']) for d in threadpool_info()])
for nt in (4,8,16,32):
    with threadpool_limits(limits=nt):
        np.linalg.eigh(G); t=time.time(); np.linalg.eigh(G); print("host eigh threads",nt,(time.time()-t)*1e3)
