#!/usr/bin/env python3
"""How long is ONE round of resident 128x128 tiles (512 = 2 per CU), and how do rounds add up?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DCLIP_GEMM_TILE"] = os.environ.get("DCLIP_GEMM_TILE", "128x128")
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")
def t(m, n, k, layout=3, iters=20):
    a = torch.randn((m, k) if layout & 1 else (k, m), device=dev); b = torch.randn((n, k) if layout & 2 else (k, n), device=dev)
    out = torch.empty(m, n, device=dev)
    for _ in range(3): ops.gemm(a, b, layout, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(a, b, layout, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for K in (768, 3072):
    for tiles_m, label in ((8, "0.5 round (256 tiles)"), (16, "1 round"), (24, "1.5"), (32, "2 rounds"), (48, "3"), (64, "4"), (128, "8")):
        m, n = tiles_m * 128, 32 * 128
        us = t(m, n, K)
        print(f"K={K:5d} tiles={tiles_m*32:5d} {label:22s} {us:8.1f} us  {2.0*m*n*K/us/1e6:6.1f} TF  per-round {us/(tiles_m*32/512):7.1f} us", flush=True)
print("fixed-cost probe: K sweep at 256 tiles (1 per CU) and 512 tiles")
for tiles_m in (8, 16):
    for K in (32, 64, 128, 256, 512):
        m, n = tiles_m * 128, 32 * 128
        us = t(m, n, K)
        print(f"tiles={tiles_m*32:4d} K={K:4d} {us:7.1f} us", flush=True)
