#!/usr/bin/env python3
"""Ablation of the fp32 GEMM's phases (diagnostic library only; results are WRONG by construction for dbg != 0):
   dbg bit 1 = no epilogue, 2 = no LDS-DMA in the K loop, 4 = no K-loop barrier.  Prints steady-state TF/s per variant."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops, _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "libdclip_hip_stamps.so")
_lib.load()
dev = torch.device("cuda:0")
M = 12800
SHAPES = [("fc2_dgrad NN", 1, M, 3072, 768), ("fc1_fwd NT", 3, M, 3072, 768), ("fc2_fwd NT", 3, M, 768, 3072),
          ("out_fwd NT", 3, M, 768, 768), ("8192x8192x2048 NT", 3, 8192, 8192, 2048)]
tiles = sys.argv[1:] or [""]
for tile in tiles:
    if tile:
        os.environ["DCLIP_GEMM_TILE"] = tile
    for name, layout, m, n, k in SHAPES:
        a = torch.randn((m, k) if layout & 1 else (k, m), device=dev)
        b = torch.randn((n, k) if layout & 2 else (k, n), device=dev)
        out = torch.empty(m, n, device=dev)
        row = []
        for dbg in [int(x) for x in os.environ.get('ABLATE', '0,1,2,4,3,7,0').split(',')]:
            os.environ["DCLIP_GEMM_DBG"] = str(dbg)
            for _ in range(10):
                ops.gemm(a, b, layout, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(a, b, layout, out=out)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            row.append(f"dbg{dbg}: {us:7.1f} us {2.0 * m * n * k / us / 1e6:6.1f} TF")
        print(f"{tile or 'plan':8s} {name:20s} " + " | ".join(row), flush=True)
        del a, b, out
