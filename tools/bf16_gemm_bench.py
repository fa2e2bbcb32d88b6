#!/usr/bin/env python3
"""bf16 GEMM shapes of the region encoder (M = 2048 crops x 50 tokens, or L/14) — big-tile kernel timing."""
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


for M, N, K in [(102400, 2304, 768), (102400, 768, 768), (102400, 3072, 768), (102400, 768, 3072), (65792, 3072, 1024),
                (65792, 1024, 4096), (526336, 1024, 1024), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    ms = t(lambda: ops.gemm_bf16(a, w, bias=b))
    ms16 = t(lambda: ops.gemm_bf16(a, w, bias=b, out_bf16=True))
    print(f"{M}x{N}x{K}: fp32 out {ms:.3f} ms {2.0 * M * N * K / ms / 1e9:.0f} TF/s | bf16 out {ms16:.3f} ms "
          f"{2.0 * M * N * K / ms16 / 1e9:.0f} TF/s", flush=True)
