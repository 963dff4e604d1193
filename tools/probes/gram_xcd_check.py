"""Do the Gram kernel's workgroups run on the XCD their K range was laid out for (blockIdx % 8)?  Unmasked stream, then
the CU-masked streams of the POD pipeline.  Counter "gram_off_xcd"."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import ops, pipeline
from romtime_amd._lib import Context
X = torch.randn((1000000, 512), dtype=torch.float64, device="cuda")
ctx = Context.current()
for _ in range(3): ops.gram(X)
print("unmasked: workgroups off their XCD after 3 Grams:", ctx.counter("gram_off_xcd"))
pipe = pipeline.PodPipeline()
outs = pipe.map([X] * 4, num=40, normalize=True)
tot = 0
for c in {id(c): c for c in (getattr(pipe, "ctxG", None), getattr(pipe, "ctxE", None)) if c is not None}.values():
    tot += c.counter("gram_off_xcd")
print("pipeline (224-CU masked stream): off their XCD after 4 PODs:", tot, " attrs:", [a for a in dir(pipe) if "ctx" in a.lower()])
pipeline.shutdown()
