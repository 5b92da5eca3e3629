#!/usr/bin/env python3
"""Where do torch's own copy / conversion kernels inside one eager forward come from?  (torch.profiler with stacks)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import motion324_amd as m
from motion324_amd import synth
dev = torch.device("cuda", 0)
w = bench.WORKLOAD
model, _ = bench.build_model(dev, w["frames"])
s = synth.synth_inputs(w["B"], w["T"], w["N"], w["S"], w["HW"], seed=1)
sample = {k: torch.from_numpy(v).to(dev) for k, v in s.items()}
m.set_precision("bf16")
with torch.no_grad():
    for _ in range(2):
        model(sample)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        model(sample)
        torch.cuda.synchronize()
import collections
seen = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::repeat", "aten::cat", "aten::contiguous"):
        st = [f for f in (e.stack or []) if "motion324_amd" in f or "bench.py" in f][:2]
        seen[(e.name, str(getattr(e, "input_shapes", "")), " <- ".join(st))] += 1
for k, n in seen.most_common(40):
    print(n, k)
