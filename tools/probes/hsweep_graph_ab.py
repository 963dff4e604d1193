"""Hyper-reduced sweep (tools/bench_configs.py c5h) with and without the step graph (option sweep_graph), one process.
   python3 tools/probes/hsweep_graph_ab.py"""
import os, sys, json, importlib.util
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
spec = importlib.util.spec_from_file_location("bench_configs", os.path.join(root, "tools", "bench_configs.py"))
bc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bc)
from romtime_amd._lib import Context
for graph in (0, 1, 0, 1):
    Context.current().set_option("sweep_graph", graph)
    out = bc.c5h()
    print("sweep_graph", graph, "us per step %.1f" % (1e3 * out["ms_per_step_all_mu"]), "err", out.get("first5_rel_err_vs_oracle"), flush=True)
