#!/usr/bin/env python3
"""wgrad GEMM with and without the fused bias gradient (DCLIP_EPI_A_ROWSUM) vs wgrad + separate colsum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")


def t(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M, N, K in [(3072, 768, 12800), (768, 3072, 12800), (2304, 768, 12800), (768, 768, 12800), (2048, 512, 19712)]:
    dy, x = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    db, out = torch.empty(M, device=dev), torch.empty(M, N, device=dev)
    for rep in range(2):
        a = t(lambda: ops.gemm(dy, x, ops.LAYOUT_TN, out=out))
        b = t(lambda: ops.gemm(dy, x, ops.LAYOUT_TN, out=out, a_rowsum=db))
        c = t(lambda: (ops.gemm(dy, x, ops.LAYOUT_TN, out=out), ops.colsum(dy, out=db)))
        print(f"M={M} N={N} K={K}: plain {a:.1f} us | fused rowsum {b:.1f} us | plain + colsum {c:.1f} us", flush=True)
