#!/usr/bin/env python3
"""Does launching the fp32 GEMM of an M = 12,800 shape as TWO row ranges — whole multiples of the CU count first, the
remainder as its own launch — beat the single launch?  (gemm_tile_staircase.py: a single launch loses ~5 % when
tiles / CU is not an integer.)  Same stream, back to back; every candidate split in 128-row steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")


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
    return e0.elapsed_time(e1) * 1e3 / n


M = 12800
for N, K, layout, name in [(768, 768, ops.LAYOUT_NT, "out fwd"), (768, 3072, ops.LAYOUT_NT, "fc2 fwd"), (2304, 768, ops.LAYOUT_NT, "qkv fwd"),
                           (3072, 768, ops.LAYOUT_NT, "fc1 fwd"), (768, 2304, ops.LAYOUT_NN, "qkv dgrad"), (3072, 768, ops.LAYOUT_NN, "fc2 dgrad")]:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) if layout == ops.LAYOUT_NT else torch.randn(K, N, device=dev)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    one = t(lambda: ops.gemm(a, w, layout, bias=b, out=out))
    ref = out.clone()
    tn = N // 64
    best = None
    rows = []
    for k in range(1, 40):
        tm1 = (256 * k) // tn                      # tile rows that fit k tiles per CU
        if tm1 <= 0 or tm1 >= 100:
            continue
        for d in (0, -1):
            m1 = (tm1 + d) * 128
            if m1 <= 0 or m1 >= M or (m1, ) in rows:
                continue
            rows.append((m1, ))

            def two(m1=m1):
                ops.gemm(a[:m1], w, layout, bias=b, out=out[:m1])
                ops.gemm(a[m1:], w, layout, bias=b, out=out[m1:])
            us = t(two)
            same = torch.equal(out, ref)
            if best is None or us < best[0]:
                best = (us, m1, same)
    print(f"{name:10s} M={M} N={N} K={K}: one launch {one:7.1f} us ({2.0 * M * N * K / one / 1e6:6.1f} TF/s) | best two-launch split at M1={best[1]} "
          f"({best[1] // 128 * tn} + {(M - best[1]) // 128 * tn} tiles): {best[0]:7.1f} us ({2.0 * M * N * K / best[0] / 1e6:6.1f} TF/s), "
          f"bit-identical {best[2]}", flush=True)
