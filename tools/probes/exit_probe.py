import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from romtime_amd.pipeline import PodPipeline, _STREAMS
from romtime_amd import _lib
mode = sys.argv[1]
X = torch.randn((200000, 256), dtype=torch.float64, device="cuda")
pipe = PodPipeline()
outs = pipe.map([X] * 4, num=10)
pipe.close()
print("done", mode, flush=True)
if mode == "destroy":
    del outs, pipe, X
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    lib = _lib.load()
    for st in list(_STREAMS.values()):
        lib.rt_stream_destroy(_lib._p(st.cuda_stream))
    _STREAMS.clear()
elif mode == "osexit":
    sys.stdout.flush()
    os._exit(0)
