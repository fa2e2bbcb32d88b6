#!/usr/bin/env python3
"""Experiment: full rounds with 128x128 tiles + remainder rows with 64x64 tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")
def run(a, b, out, m1):
    os.environ["DCLIP_GEMM_TILE"] = "128x128"
    ops.gemm(a[:m1], b, 3, out=out[:m1])
    if m1 < a.shape[0]:
        os.environ["DCLIP_GEMM_TILE"] = "64x64"
        ops.gemm(a[m1:], b, 3, out=out[m1:])
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for (M, N, K) in ((12800, 2304, 768), (12800, 3072, 768), (12800, 768, 3072), (19712, 2048, 512)):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
    tn = (N + 127) // 128
    tiles = ((M + 127) // 128) * tn
    full = tiles // 512
    res = []
    for m1_tiles in sorted(set([(M + 127) // 128, (full * 512) // tn, (full * 512) // tn + 1, ((full * 512) // tn) - 2])):
        m1 = min(M, m1_tiles * 128)
        us = t(lambda: run(a, b, out, m1))
        res.append(f"m1={m1_tiles:3d}tiles:{us:7.1f}us({2.0*M*N*K/us/1e6:5.1f}TF)")
    print(M, N, K, f"tiles={tiles} full_rounds={full} | " + "  ".join(res), flush=True)
