#!/usr/bin/env python3
"""Do two independent GEMMs on two streams pack better than back to back on one?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")
M = 12800
dy = torch.randn(M, 3072, device=dev); w = torch.randn(3072, 768, device=dev); x = torch.randn(M, 768, device=dev)
dx = torch.empty(M, 768, device=dev); dw = torch.empty(3072, 768, device=dev)
dy2 = torch.randn(M, 768, device=dev); w2 = torch.randn(768, 3072, device=dev); g = torch.randn(M, 3072, device=dev)
dh = torch.empty(M, 3072, device=dev); dw2 = torch.empty(768, 3072, device=dev)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
def seq():
    ops.gemm(dy, w, 1, out=dx); ops.gemm(dy, x, 0, out=dw)
    ops.gemm(dy2, w2, 1, out=dh); ops.gemm(dy2, g, 0, out=dw2)
def par():
    side.wait_stream(main)
    ops.gemm(dy, w, 1, out=dx)
    with torch.cuda.stream(side):
        ops.gemm(dy, x, 0, out=dw)
    ops.gemm(dy2, w2, 1, out=dh)
    with torch.cuda.stream(side):
        ops.gemm(dy2, g, 0, out=dw2)
    main.wait_stream(side)
def t(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / iters
for r in range(3):
    print(f"round {r}: sequential {t(seq):8.1f} us   two streams {t(par):8.1f} us", flush=True)
fl = 4 * 2.0 * M * 768 * 3072
print("flops per call", fl)
