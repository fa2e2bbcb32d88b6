#!/usr/bin/env python3
"""Multi-tensor AdamW + clip on a parameter set of the student's size (87.8 M fp32): time per optimizer step.
   adamw_bench.py [lib]  — another build of the library for A/B on one box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import _lib, optim
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda:0")
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)] * 12 + [(768,)] * 100 + [(768, 3072)]
params = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.02) for s in shapes]
for p in params:
    p.grad = torch.randn_like(p) * 0.01
n = sum(p.numel() for p in params)
opt = optim.FusedAdamW(params, lr=1e-4, max_grad_norm=0.5)
for _ in range(3):
    opt.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    opt.step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"{n / 1e6:.1f} M params: {ms * 1e3:.1f} us per step (sumsq + clip + AdamW), {n * 32 / ms / 1e6:.0f} GB/s of 28 + 4 B/param")
