#!/usr/bin/env python3
"""bf16 GEMM: the 256x256 LDS-DMA kernel vs the 128x128 register-staged kernel per shape (DCLIP_BF16_BIG_MIN is read once
per process, so each variant runs in its own interpreter: usage  bf16_plan_sweep.py [big_min])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["DCLIP_BF16_BIG_MIN"] = sys.argv[1]
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


SHAPES = [(12800, 2304, 768), (12800, 768, 768), (12800, 3072, 768), (12800, 768, 3072), (12800, 768, 2304),
          (25600, 2304, 768), (25600, 768, 768), (25600, 3072, 768), (25600, 768, 3072),
          (19712, 1536, 512), (19712, 512, 512), (19712, 2048, 512), (19712, 512, 2048),
          (102400, 2304, 768), (102400, 768, 768), (102400, 3072, 768), (102400, 768, 3072),
          (131584, 3072, 1024), (131584, 1024, 1024), (131584, 4096, 1024), (131584, 1024, 4096)]
for M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    ms = t(lambda: ops.gemm_bf16(a, w, bias=b))
    ms16 = t(lambda: ops.gemm_bf16(a, w, bias=b, out_bf16=True))
    print(f"big_min={os.environ.get('DCLIP_BF16_BIG_MIN', 'default'):>7s} {M:7d}x{N:5d}x{K:5d}: fp32 out {ms * 1e3:8.1f} us "
          f"{2.0 * M * N * K / ms / 1e9:5.0f} TF/s | bf16 out {ms16 * 1e3:8.1f} us {2.0 * M * N * K / ms16 / 1e9:5.0f} TF/s", flush=True)
    del a, w
