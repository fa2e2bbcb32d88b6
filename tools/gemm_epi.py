#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for tile in ("128x128", "64x64"):
    os.environ["DCLIP_GEMM_TILE"] = tile
    for (M, N, K) in ((12800, 2304, 768), (12800, 768, 3072)):
        a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); out = torch.empty(M, N, device=dev)
        bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev); aux = torch.empty(M, N, device=dev)
        r = {}
        r["plain"] = t(lambda: ops.gemm(a, b, 3, out=out))
        r["bias"] = t(lambda: ops.gemm(a, b, 3, out=out, bias=bias))
        r["bias+res"] = t(lambda: ops.gemm(a, b, 3, out=out, bias=bias, residual=res))
        r["bias+gelu+aux"] = t(lambda: ops.gemm(a, b, 3, out=out, bias=bias, aux=aux, epilogue=ops.EPI_GELU))
        r["dgelu"] = t(lambda: ops.gemm(a, b, 3, out=out, aux=res, epilogue=ops.EPI_DGELU))
        r["plain2"] = t(lambda: ops.gemm(a, b, 3, out=out))
        print(tile, M, N, K, " ".join(f"{k}:{v:.1f}" for k, v in r.items()), flush=True)
