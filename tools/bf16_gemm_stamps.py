#!/usr/bin/env python3
"""Where does a launch of the bf16 ping-pong GEMM spend its time?  Diagnostic (GPU box; `make stamps` first).
Per shape: launch span against the MFMA time at the held clock, and per workgroup (median / p10 / p90, shader cycles):
prologue (entry -> first K-tile landed), K loop (per K-tile against the 2,048-cycle MFMA floor), epilogue split into
first pass / second pass / wait for the stores' acknowledgement; workgroups per CU and CU idle time inside the span."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab", "libdclip_hip_stamps.so"))
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
import numpy as np
import torch
from dclip_amd import ops, _lib
_lib.LIB_PATH = os.path.abspath(args.lib)
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
raw.dclip_debug_set_bf16_stamps.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
SHAPES = [("qkv L/14", 131584, 3072, 1024, "bias16"), ("out L/14", 131584, 1024, 1024, "res"), ("fc1 L/14", 131584, 4096, 1024, "gelu16"),
          ("fc2 L/14", 131584, 1024, 4096, "res"), ("qkv B/32 x8", 102400, 2304, 768, "bias16"), ("out B/32 x8", 102400, 768, 768, "res"),
          ("fc2 B/32 x8", 102400, 768, 3072, "res"), ("8192^3", 8192, 8192, 8192, "bias16")]
stamps = torch.zeros((1 << 16, 16), dtype=torch.int64, device=dev)
for name, M, N, K, kind in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev) if kind == "res" else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if kind == "res" else torch.bfloat16)
    run = {"bias16": lambda: ops.gemm_bf16(a, w, bias=b, out_bf16=True, out=out),
           "res": lambda: ops.gemm_bf16(a, w, bias=b, residual=res, out=out),
           "gelu16": lambda: ops.gemm_bf16(a, w, bias=b, gelu=True, out_bf16=True, out=out)}[kind]
    raw.dclip_debug_set_bf16_stamps(None)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    stamps.zero_()
    raw.dclip_debug_set_bf16_stamps(stamps.data_ptr())
    for _ in range(args.reps - 1):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    raw.dclip_debug_set_bf16_stamps(None)
    nwg = -(-M // 256) * -(-N // 256)
    s = stamps[:nwg].cpu().numpy().astype(np.int64)
    t0, t1, t2, t3, t4, r0, r1, hw, t8 = [s[:, i] for i in range(9)]
    nk = K // 64
    span_us = (r1.max() - r0.min()) / 100.0
    clk = np.median((t4 - t0) / np.maximum(r1 - r0, 1)) / 10.0
    flops = 2.0 * M * N * K
    cu_key = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 0x7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xF)
    cus = np.unique(cu_key)
    per_cu = np.array([(cu_key == c).sum() for c in cus])
    idle = []
    for c in cus:
        sel = cu_key == c
        iv = sorted(zip(r0[sel], r1[sel]))
        cov, cs, ce = 0, iv[0][0], iv[0][1]
        for a_, b_ in iv[1:]:
            if a_ > ce:
                cov += ce - cs
                cs, ce = a_, b_
            else:
                ce = max(ce, b_)
        cov += ce - cs
        idle.append((r1.max() - r0.min()) - cov)
    idle = np.array(idle) / 100.0
    q = lambda x: f"{np.median(x):7.0f} ({np.percentile(x, 10):6.0f}..{np.percentile(x, 90):6.0f})"
    print(f"{name:12s} {M}x{N}x{K} {kind}: event {e0.elapsed_time(e1) * 1e3:7.1f} us span {span_us:7.1f} us = {flops / span_us / 1e6:5.0f} TF/s | clock {clk:.2f} GHz "
          f"(MFMA peak at that clock {2500 * clk / 2.4:5.0f} TF/s) | {nwg} WGs, {nwg / 256:.2f}/CU (CUs seen {len(cus)}, max {per_cu.max()})", flush=True)
    print(f"    cycles median (p10..p90): prologue {q(t1 - t0)} | K loop {q(t2 - t1)} = {np.median(t2 - t1) / nk:6.0f}/K-tile (floor 2048) | "
          f"epilogue pass 1 {q(t3 - t2)} | pass 2 {q(t8 - t3)} | store ack {q(t4 - t8)} | whole {q(t4 - t0)}")
    # how many workgroups are in their epilogue at the same time?  (epilogue interval on the real-time clock, from the
    # cycle stamps' share of the workgroup's life)
    life = np.maximum(t4 - t0, 1).astype(np.float64)
    e_beg = r0 + (t2 - t0) / life * (r1 - r0)
    e_end = r0 + (t8 - t0) / life * (r1 - r0)
    grid = np.linspace(r0.min(), r1.max(), 2000)
    conc = np.array([np.count_nonzero((e_beg <= g) & (g < e_end)) for g in grid])
    running = np.array([np.count_nonzero((r0 <= g) & (g < r1)) for g in grid])
    mid = slice(200, 1800)
    print(f"    workgroups in their epilogue at the same time (middle 80 % of the span): mean {conc[mid].mean():5.1f} of {running[mid].mean():5.1f} running, "
          f"p10 {np.percentile(conc[mid], 10):4.0f} p90 {np.percentile(conc[mid], 90):4.0f} max {conc[mid].max()}")
    print(f"    CU idle inside span: median {np.median(idle):6.1f} us max {idle.max():6.1f} us mean {idle.mean():6.1f} us ({idle.mean() / span_us * 100:4.1f} %)", flush=True)
    del a, w, out, res
