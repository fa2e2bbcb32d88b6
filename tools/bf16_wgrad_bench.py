#!/usr/bin/env python3
"""bf16 weight-gradient GEMMs (split-K over the token axis): time per shape.  DCLIP_BF16_PP=0 -> the 128x128 split-K kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()


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


for M, N, K in [(3072, 768, 12800), (768, 3072, 12800), (2304, 768, 12800), (768, 768, 12800), (3072, 768, 3200), (768, 768, 3200),
                (3072, 768, 25600)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    ms = t(lambda: ops.gemm_bf16_wgrad(a, w, K))
    print(f"{M:5d}x{N:5d}x{K:6d}: splits {lib.dclip_gemm_bf16_splitk_plan(M, N, K):3d} {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:5.0f} TF/s", flush=True)
