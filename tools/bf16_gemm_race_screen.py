#!/usr/bin/env python3
"""Race screen for the ping-pong bf16 GEMM (counted vmcnt / raw barriers: an early LDS read passes a reference check
whenever the DMA happens to land first).  Every shape is launched many times, alone and beside a bandwidth-heavy copy on
another stream (which moves the DMA latencies), and every result must be bit-identical to the first one, which is checked
against an fp64 product of the rounded inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
side = torch.cuda.Stream()
big_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
big_b = torch.empty_like(big_a)
bad = 0
for M, N, K, kind in [(4096, 4096, 64, "bias16"), (4096, 4096, 128, "res"), (8192, 4096, 192, "gelu16"), (12800, 2304, 768, "bias16"),
                      (12800, 768, 3072, "res"), (25700, 1028, 1024, "gelu16"), (8192, 8192, 2048, "bias16"), (3072, 768, 12800, "wgrad"),
                      (768, 768, 12800, "wgrad")]:
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev) if kind == "res" else None
    run = {"bias16": lambda: ops.gemm_bf16(a, w, bias=b, out_bf16=True), "res": lambda: ops.gemm_bf16(a, w, bias=b, residual=res),
           "gelu16": lambda: ops.gemm_bf16(a, w, bias=b, gelu=True, out_bf16=True), "wgrad": lambda: ops.gemm_bf16_wgrad(a, w, K)}[kind]
    first = run().clone()
    if kind in ("bias16", "res", "wgrad") and M * N <= 40e6:
        want = a.double() @ w.double().t()
        if kind != "wgrad":
            want = want + b.double()
        if kind == "res":
            want = want + res.double()
        err = float((first.double() - want).abs().max() / want.abs().max())
        assert err < (1e-2 if kind == "bias16" else 1e-4), (M, N, K, kind, err)
    mism = 0
    for i in range(reps):
        if i % 2:
            with torch.cuda.stream(side):
                big_b.copy_(big_a, non_blocking=True)
        out = run()
        if not torch.equal(out, first):
            mism += 1
    torch.cuda.synchronize()
    bad += mism
    print(f"{M}x{N}x{K} {kind}: {reps} launches, {mism} differ from the first", flush=True)
print("RACE SCREEN", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
