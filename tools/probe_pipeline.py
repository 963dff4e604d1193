#!/usr/bin/env python3
"""Per-step time of the CU-partitioned POD pipeline on the bench workload for several partitions."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from romtime_amd import pod  # noqa: E402
from romtime_amd.pipeline import PodPipeline  # noqa: E402

dev = torch.device("cuda", 0)
X = bench.build_local_matrix(0, 1, bench.N_H, bench.N_S, dev)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for _ in range(3):
    ref = pod.pod_device(X, num=40, normalize=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    ref = pod.pod_device(X, num=40, normalize=True)
torch.cuda.synchronize()
print(json.dumps(dict(mode="latency (pod_device)", ms_per_step=1e3 * (time.perf_counter() - t0) / K)), flush=True)
for e in (4, 8, 3):
    try:
        pipe = PodPipeline(eig_cus_per_xcd=e)
        pipe.map([X] * 3, num=40)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = pipe.map([X] * K, num=40)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / K
        Q, Qr = outs[-1]["Q"], ref["Q"]
        sgn = torch.sign((Q * Qr).sum(dim=0))
        err = float((Q * sgn[None, :] - Qr).abs().max().item())
        print(json.dumps(dict(mode=f"pipeline e={e} ({8 * e} CUs of every 256 for the eigensolver)", ms_per_step=ms,
                              recomputed=pipe.recomputed, max_abs_diff_Q=err,
                              s_equal=bool(np.allclose(outs[-1]["s"], ref["s"], rtol=0, atol=1e-13)), **pipe.last_stage_ms)),
              flush=True)
        pipe.close()
    except Exception as ex:  # noqa: BLE001
        print(json.dumps(dict(mode=f"pipeline e={e}", error=repr(ex))), flush=True)
