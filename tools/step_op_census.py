#!/usr/bin/env python3
"""Which framework-level ops (fills, copies, elementwise kernels) does one eager step launch beside the library's own
kernels, and from where?  torch.profiler over one step of a bench workload, grouped by op name and Python call site.
    python tools/step_op_census.py [c2|c3]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dclip_amd import optim

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda:0")
spec = ("c2", "ViT-B/32", None, 256, 8, "fp32", "fp32") if which == "c2" else ("c3", "ViT-B/32", "ViT-B/32", 256, 8, "bf16", "bf16")
module, cfg, tcfg, batch = bench.build_workload(*spec, dev, None, 0, fast_teacher_init=True)
trainable = [p for p in module.parameters() if p.requires_grad]
opt = optim.FusedAdamW(trainable, lr=1e-6, max_grad_norm=0.5)


def step():
    loss = module.training_step(batch)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
names = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::add", "aten::mul", "aten::clone",
         "aten::contiguous", "aten::to", "aten::_to_copy", "aten::sum", "aten::div", "aten::ones_like", "aten::add_", "aten::mul_")
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in names and ev.device_time_total > 0 or ev.name in ("aten::fill_", "aten::copy_"):
        site = next((s for s in ev.stack if "/dclip_amd/" in s or "bench.py" in s or "step_op_census" in s), ev.stack[0] if ev.stack else "?")
        cnt[(ev.name, site.strip()[-110:])] += 1
for (n, site), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{c:4d}  {n:18s} {site}")
