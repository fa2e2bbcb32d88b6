#!/usr/bin/env python3
"""Same GEMM shape through the three operand layouts, interleaved rounds (medians)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")
def t(fn, iters=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for (M, N, K) in ((12800, 2304, 768), (12800, 768, 768), (12800, 3072, 768), (12800, 768, 3072)):
    a = torch.randn(M, K, device=dev); at = a.t().contiguous()
    w = torch.randn(N, K, device=dev); wt = w.t().contiguous()
    out = torch.empty(M, N, device=dev)
    fns = {"NT (A[M,K] W[N,K])": lambda: ops.gemm(a, w, 3, out=out), "NN (A[M,K] Wt[K,N])": lambda: ops.gemm(a, wt, 1, out=out),
           "TN (At[K,M] Wt[K,N])": lambda: ops.gemm(at, wt, 0, out=out), "TT (At[K,M] W[N,K])": lambda: ops.gemm(at, w, 2, out=out)}
    for f in fns.values():
        for _ in range(3): f()
    res = {k: [] for k in fns}
    for r in range(5):
        for k, f in fns.items():
            res[k].append(t(f))
    print(M, N, K, " | ".join(f"{k}: {statistics.median(v):.1f}us {2.0*M*N*K/statistics.median(v)/1e6:.1f}TF" for k, v in res.items()), flush=True)
