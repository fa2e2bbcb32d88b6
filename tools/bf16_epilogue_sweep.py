#!/usr/bin/env python3
"""bf16 GEMM: time per epilogue kind on the tower shapes (bias / bias+residual fp32 / GELU bf16 / GELU+saved pre-activation /
dGELU), to see what the epilogue costs beside the K loop.  DCLIP_BF16_PP=0 selects the lock-step 256x256 kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops, _lib
if len(sys.argv) > 1:                      # another build of the library (A/B on one box)
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
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


SHAPES = [(12800, 2304, 768), (12800, 3072, 768), (12800, 768, 3072), (102400, 2304, 768), (102400, 768, 768),
          (102400, 3072, 768), (102400, 768, 3072), (131584, 3072, 1024), (131584, 1024, 1024), (131584, 4096, 1024),
          (131584, 1024, 4096), (8192, 8192, 8192)]
for M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    h = torch.randn(M, N, device=dev).to(torch.bfloat16)
    o32 = torch.empty(M, N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    cases = [("bias f32", lambda: ops.gemm_bf16(a, w, bias=b, out=o32)),
             ("bias bf16", lambda: ops.gemm_bf16(a, w, bias=b, out_bf16=True, out=o16)),
             ("bias+res f32", lambda: ops.gemm_bf16(a, w, bias=b, residual=res, out=o32)),
             ("gelu bf16", lambda: ops.gemm_bf16(a, w, bias=b, gelu=True, out_bf16=True, out=o16)),
             ("gelu+save bf16", lambda: ops.gemm_bf16(a, w, bias=b, gelu=True, save_preact=True, out_bf16=True, out=o16)),
             ("dgelu bf16", lambda: ops.gemm_bf16(a, w, dgelu_of=h, out_bf16=True, out=o16))]
    line = f"{M:7d}x{N:5d}x{K:5d}:"
    for name, f in cases:
        ms = t(f)
        line += f" {name} {ms * 1e3:7.1f} us {2.0 * M * N * K / ms / 1e9:5.0f} TF |"
    print(line, flush=True)
    del a, w, res, h, o32, o16
