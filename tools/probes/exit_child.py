"""One short-lived process that ends the way a test session does: a cached PodWorkers with live threads, a cached
PodPipeline (CU-masked streams) and PodLanes, results alive at interpreter exit.  Driven by exit_stress.py."""
import ctypes
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))

if __name__ == "__main__":
    native = os.open(sys.argv[1], os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    ctypes.CDLL(os.path.join(here, "btsig.so")).btsig_install(native)
    import numpy as np
    import torch

    from romtime_amd import ops, pipeline, walks

    rng = np.random.RandomState(int(sys.argv[2]))
    small = [ops.to_device(rng.standard_normal((4000, 48))) for _ in range(4)]
    KEPT = list(walks.pod_sequence(small, tol=1.0 - 1e-9, normalize=True))          # worker threads (cached runner)
    pipe = walks._runner(small[0].device, "pipeline")                                # cached PodPipeline, masked streams
    pipe.small_set = 0
    KEPT2 = pipe.map(small, num=4, normalize=True)
    OPEN = pipe.run(small, num=4, normalize=True)
    FIRST = next(OPEN)
    lanes = walks._runner(small[0].device, "lanes")
    KEPT3 = lanes.map(small, num=4, normalize=True)
    if int(sys.argv[2]) % 2:
        import multiprocessing as mp

        m = mp.get_context("fork").Manager()
        d = m.dict()
        d["ok"] = 1
    print("child done", sys.argv[2], flush=True)
