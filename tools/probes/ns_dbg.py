import os, sys, ctypes, numpy as np
sys.path.insert(0, os.getcwd())
sys.argv = ["bench_configs.py", "c5h"]
import runpy
try:
    runpy.run_path("tools/bench_configs.py", run_name="__main__")
except SystemExit:
    pass
from romtime_amd._lib import Context
buf = (ctypes.c_ulonglong * 16)()
Context.current().lib.rt_ns_debug_dump(buf)
t = np.array(list(buf)[:11], dtype=np.int64)
print("stamps (cycles from kernel start):", (t - t[0]).tolist())
