#!/usr/bin/env python3
"""The bf16 GEMMs of the TRAINING student (M = 12,800 tokens: 256 images x 50) with the epilogues the step gives them, timed
one by one: forward qkv / out / fc1 / fc2, data gradients fc2 / fc1 / out / qkv.  Kernel choice is read from the environment
once per process (DCLIP_BF16_BIG_MIN, DCLIP_BF16_MID_DMA, DCLIP_BF16_PP): run once per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12800


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
    return e0.elapsed_time(e1) / n


def mk(n, k):
    return torch.randn(M, k, device=dev).to(torch.bfloat16), torch.randn(n, k, device=dev).to(torch.bfloat16)


D, I = 768, 3072
tot = 0.0
rows = []
for name, n, k, kind in [("qkv fwd  (bias, bf16 out)", 3 * D, D, "b16"), ("out fwd  (bias+residual)", D, D, "res"),
                         ("fc1 fwd  (gelu+save bf16)", I, D, "gelu"), ("fc2 fwd  (bias+residual)", D, I, "res"),
                         ("fc2 dgrad (dgelu, bf16 out)", I, D, "dgelu"), ("fc1 dgrad (fp32 out)", D, I, "f32"),
                         ("out dgrad (bf16 out)", D, D, "o16"), ("qkv dgrad (fp32 out)", D, 3 * D, "f32")]:
    a, w = mk(n, k)
    b = torch.randn(n, device=dev)
    res = torch.randn(M, n, device=dev)
    h = torch.randn(M, n, device=dev).to(torch.bfloat16)
    f = {"b16": lambda: ops.gemm_bf16(a, w, bias=b, out_bf16=True), "res": lambda: ops.gemm_bf16(a, w, bias=b, residual=res),
         "gelu": lambda: ops.gemm_bf16(a, w, bias=b, gelu=True, out_bf16=True, save_preact=True),
         "dgelu": lambda: ops.gemm_bf16(a, w, dgelu_of=h, out_bf16=True), "f32": lambda: ops.gemm_bf16(a, w),
         "o16": lambda: ops.gemm_bf16(a, w, out_bf16=True)}[kind]
    ms = t(f)
    tot += ms
    rows.append(f"{name:30s} {M}x{n}x{k}: {ms * 1e3:7.1f} us {2.0 * M * n * k / ms / 1e9:6.0f} TF/s  tiles256 {((M + 255) // 256) * ((n + 255) // 256)}")
print("\n".join(rows))
print(f"sum per layer {tot * 1e3:.1f} us  (env BIG_MIN={os.environ.get('DCLIP_BF16_BIG_MIN')} MID_DMA={os.environ.get('DCLIP_BF16_MID_DMA')})")
