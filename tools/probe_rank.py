import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from romtime_amd import orth
from oracle import romtime_oracle as oracle
rng=np.random.RandomState(0)
for (N,n,rk) in [(200,12,5),(3000,64,10),(50,50,50),(40,60,40)]:
    X=rng.standard_normal((N,rk))@rng.standard_normal((rk,n))
    for kw in (dict(num=n), dict(), dict(tol=1-1e-15), dict(num=n, normalize=False)):
        try:
            Q,s,e=orth(X,**kw); Qr,sr,er=oracle.orth(X.copy(),**kw)
            k=min(rk, Q.shape[1])
            sub=np.linalg.norm(Q[:,:k]@(Q[:,:k].T@Qr[:,:k])-Qr[:,:k],2) if k else 0
            print(N,n,rk,kw,"shape",Q.shape,Qr.shape,"subspace err(lead rk)",f"{sub:.1e}","finite",np.isfinite(Q).all(), "s err", f"{np.abs(s[:rk]-sr[:rk]).max()/sr[0]:.1e}")
        except Exception as ex:
            print(N,n,rk,kw,"EXC",type(ex).__name__,str(ex)[:100])
