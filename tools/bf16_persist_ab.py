#!/usr/bin/env python3
"""Persistent ping-pong GEMM (gemm_bf16_ppp_kernel) against the one-tile-per-workgroup kernel on the frozen towers' shapes
(M = 2048 crops x 50 tokens for ViT-B/32, 512 x 257 for ViT-L/14) and the student's, with the epilogues the step gives them.
The kernel choice is read from the environment per call, so both run in one process on one box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")


def t(f, n=20):
    for _ in range(4):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


SHAPES = [(102400, 2304, 768, "b16"), (102400, 768, 768, "res"), (102400, 3072, 768, "gelu"), (102400, 768, 3072, "res"),
          (131584, 3072, 1024, "b16"), (131584, 1024, 1024, "res"), (131584, 4096, 1024, "gelu"), (131584, 1024, 4096, "res"),
          (19712, 1536, 512, "b16"), (19712, 512, 2048, "res"),
          (12800, 2304, 768, "b16"), (12800, 768, 768, "res"), (12800, 3072, 768, "gelusave"), (12800, 768, 3072, "res"),
          (12800, 768, 3072, "f32"), (12800, 768, 2304, "f32"), (8192, 8192, 8192, "f32")]
tot = {"0": 0.0, "1": 0.0}
for M, N, K, kind in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev) if kind == "res" else None
    f = {"b16": lambda: ops.gemm_bf16(a, w, bias=b, out_bf16=True), "res": lambda: ops.gemm_bf16(a, w, bias=b, residual=res),
         "gelu": lambda: ops.gemm_bf16(a, w, bias=b, gelu=True, out_bf16=True),
         "gelusave": lambda: ops.gemm_bf16(a, w, bias=b, gelu=True, out_bf16=True, save_preact=True),
         "f32": lambda: ops.gemm_bf16(a, w)}[kind]
    line = f"{M:7d}x{N:5d}x{K:5d} {kind:8s} tiles {((M + 255) // 256) * ((N + 255) // 256):5d}:"
    for mode, env in (("one-tile", "0"), ("persistent", "1")):
        os.environ["DCLIP_BF16_PERSIST"] = env
        os.environ["DCLIP_BF16_PERSIST_MIN"] = os.environ.get("AB_PERSIST_MIN", "1")
        ms = t(f)
        tot[env] += ms
        line += f"  {mode} {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:6.0f} TF/s"
    print(line, flush=True)
    del a, w, res
print(f"sum: one-tile {tot['0']:.3f} ms, persistent {tot['1']:.3f} ms")
