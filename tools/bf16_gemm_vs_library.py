#!/usr/bin/env python3
"""The bf16 GEMM shapes of configs c3 / c5 on this library's ping-pong kernel and on the vendor library (torch.mm ->
hipBLASLt / rocBLAS), plain C = A W^T with bf16 output, no epilogue: how far is the hand-written kernel from what the
library reaches on the same shape?  (Measurement only: the product never calls the library GEMM.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")


def t(f, n=20):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K in [(102400, 2304, 768), (102400, 768, 768), (102400, 3072, 768), (102400, 768, 3072), (12800, 2304, 768),
                (12800, 768, 768), (12800, 3072, 768), (12800, 768, 3072), (19712, 2048, 512), (19712, 512, 2048),
                (526336, 1024, 1024), (526336, 4096, 1024), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    wt = w.t().contiguous()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    mine = t(lambda: ops.gemm_bf16(a, w, out_bf16=True))
    lib_nt = t(lambda: torch.mm(a, w.t(), out=out))
    lib_nn = t(lambda: torch.mm(a, wt, out=out))
    fl = 2.0 * M * N * K / 1e9
    print(f"{M}x{N}x{K}: this kernel {mine * 1e3:7.1f} us {fl / mine:6.0f} TF/s | library A W^T {lib_nt * 1e3:7.1f} us {fl / lib_nt:6.0f} TF/s | "
          f"library A B {lib_nn * 1e3:7.1f} us {fl / lib_nn:6.0f} TF/s", flush=True)
