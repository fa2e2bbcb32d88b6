#!/usr/bin/env python3
"""GEMM microbenchmark on the step's shapes (tuning aid; run on the GPU box).
   python tools/gemm_bench.py [--tile 128x128] """
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--tile", default=None)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--shapes", default="step")
ap.add_argument("--lib", default=None, help="alternative libdclip_hip.so (A/B runs)")
args = ap.parse_args()
if args.tile:
    os.environ["DCLIP_GEMM_TILE"] = args.tile
import torch
from dclip_amd import ops, _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)

dev = torch.device("cuda:0")
M = 12800
SHAPES = {
    "step": [("qkv_fwd NT", 3, M, 2304, 768, "bias"), ("out_fwd NT+res", 3, M, 768, 768, "res"),
             ("fc1_fwd NT+gelu", 3, M, 3072, 768, "gelu"), ("fc2_fwd NT+res", 3, M, 768, 3072, "res"),
             ("fc2_dgrad NN+dgelu", 1, M, 3072, 768, "dgelu"), ("fc1_dgrad NN", 1, M, 768, 3072, ""),
             ("out_dgrad NN", 1, M, 768, 768, ""), ("qkv_dgrad NN", 1, M, 768, 2304, ""),
             ("fc1_wgrad TN", 0, 3072, 768, M, ""), ("qkv_wgrad TN", 0, 2304, 768, M, ""), ("out_wgrad TN", 0, 768, 768, M, ""),
             ("txt_qkv NT", 3, 19712, 1536, 512, "bias"), ("txt_fc1 NT", 3, 19712, 2048, 512, "gelu"),
             ("txt_fc2 NT", 3, 19712, 512, 2048, "res")],
    "ksweep": [(f"NT K={k}", 3, M, 768, k, "") for k in (256, 512, 1024, 2048, 4096)] +
              [(f"NT N=3072 K={k}", 3, M, 3072, k, "") for k in (256, 512, 1024, 2048)],
    "big": [("4096^3 NT", 3, 4096, 4096, 4096, ""), ("8192^3 NT", 3, 8192, 8192, 8192, "")],
}
for name, layout, m, n, k, epi in SHAPES[args.shapes]:
    a = torch.randn((m, k) if layout & 1 else (k, m), device=dev)
    b = torch.randn((n, k) if layout & 2 else (k, n), device=dev)
    kw = {}
    if epi == "bias":
        kw["bias"] = torch.randn(n, device=dev)
    if epi == "res":
        kw["bias"] = torch.randn(n, device=dev); kw["residual"] = torch.randn(m, n, device=dev)
    if epi == "gelu":
        kw["bias"] = torch.randn(n, device=dev); kw["aux"] = torch.empty(m, n, device=dev); kw["epilogue"] = ops.EPI_GELU
    if epi == "dgelu":
        kw["aux"] = torch.randn(m, n, device=dev); kw["epilogue"] = ops.EPI_DGELU
    out = torch.empty(m, n, device=dev)
    for _ in range(3):
        ops.gemm(a, b, layout, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        ops.gemm(a, b, layout, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.iters
    # the library path PyTorch would take for the same product (rocBLAS / hipBLASLt fp32), bare matmul without epilogue
    ta = a if layout & 1 else a.t()
    tb = b.t() if layout & 2 else b
    for _ in range(3):
        torch.matmul(ta, tb, out=out)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.iters):
        torch.matmul(ta, tb, out=out)
    e1.record()
    torch.cuda.synchronize()
    us_t = e0.elapsed_time(e1) * 1e3 / args.iters
    print(f"{name:24s} M={m:6d} N={n:5d} K={k:6d}  {us:8.1f} us  {2.0*m*n*k/us/1e6:7.1f} TF/s | torch.matmul (no epilogue) "
          f"{us_t:8.1f} us {2.0*m*n*k/us_t/1e6:7.1f} TF/s", flush=True)
