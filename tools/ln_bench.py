#!/usr/bin/env python3
"""LayerNorm backward (+ its partial reduction) and forward on the step's shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops, _lib
if len(sys.argv) > 1:                      # another build of the library (A/B on one box)
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
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


for rows, D in [(12800, 768), (19712, 512), (256, 768), (102400, 768), (131584, 1024)]:
    x, dy, dres = (torch.randn(rows, D, device=dev) for _ in range(3))
    g, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
    y, m, r = ops.layernorm_fwd(x, g, b, 1e-5)
    f = t(lambda: ops.layernorm_fwd(x, g, b, 1e-5))
    bw = t(lambda: ops.layernorm_bwd(dy, x, g, m, r, dresidual=dres, need_param_grads=True))
    bn = t(lambda: ops.layernorm_bwd(dy, x, g, m, r, dresidual=dres, need_param_grads=False))
    f16 = t(lambda: ops.layernorm_fwd_bf16(x, g, b, 1e-5))
    print(f"rows={rows} D={D}: fwd {f:.1f} us ({rows * D * 8 / f / 1e3:.0f} GB/s) | fwd bf16 out {f16:.1f} us ({rows * D * 6 / f16 / 1e3:.0f} GB/s) | "
          f"bwd with param grads {bw:.1f} us ({rows * D * 16 / bw / 1e3:.0f} GB/s) | bwd dx only {bn:.1f} us", flush=True)
