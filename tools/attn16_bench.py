#!/usr/bin/env python3
"""bf16 attention forward (frozen towers): time per tower shape.  DCLIP_ATTN16_TILED=1 -> the 64-query tiled kernel."""
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


for name, B, S, H, causal in [("B/32 2048 crops", 2048, 50, 12, False), ("text 256", 256, 77, 8, True), ("text L/14 64", 64, 77, 12, True),
                              ("B/16 1024 crops", 1024, 197, 12, False), ("L/14 512 crops", 512, 257, 16, False),
                              ("L/14 64 crops", 64, 257, 16, False)]:
    qkv = torch.randn(B * S, 3 * H * 64, device=dev).to(torch.bfloat16)
    ms = t(lambda: ops.attention_fwd_bf16(qkv, B, S, H, causal))
    fl = 4.0 * S * S * 64 * H * B * (0.5 if causal else 1.0)
    by = qkv.numel() * 2 + B * S * H * 64 * 2
    print(f"{name:18s} B={B:5d} S={S:3d} H={H:2d}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.0f} TF/s (useful)  {by / ms / 1e6:6.0f} GB/s", flush=True)
