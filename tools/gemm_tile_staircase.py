#!/usr/bin/env python3
"""How does the fp32 GEMM's time grow with the tile count around whole multiples of the CU count?  M is varied in steps of
one 128-row tile band at N = 768 / 2304 (12 / 36 tile columns of 64): a staircase (time jumps when tiles/CU crosses an
integer) is what stream-K / tail splitting could recover; a straight line means the dispatcher already balances."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")


def t(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for N, K in [(768, 768), (768, 3072), (2304, 768)]:
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    tn = N // 64
    print(f"--- N={N} K={K} ({tn} tile columns)")
    bands = sorted(set([int(256 * k / tn) + d for k in (3, 4, 5) for d in (-2, -1, 0, 1, 2, 3)] + [100, 96, 90, 85, 80]))
    for tm in bands:
        M = tm * 128
        a = torch.randn(M, K, device=dev)
        us = t(lambda: ops.gemm(a, w, ops.LAYOUT_NT, bias=b))
        tiles = tm * tn
        print(f"M={M:6d} tiles={tiles:5d} tiles/CU={tiles / 256:6.3f}  {us:8.1f} us  {us / tiles * 256:7.2f} us per tile-per-CU  "
              f"{2.0 * M * N * K / us / 1e6:6.1f} TF/s", flush=True)
