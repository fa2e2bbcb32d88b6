#!/usr/bin/env python3
"""fp32 attention forward / backward on the step's shapes: time per launch.  --lib selects another build of the library
(A/B of kernel changes on one box: run once per library)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
args = ap.parse_args()
import torch
from dclip_amd import ops, _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
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
    return e0.elapsed_time(e1) / n * 1e3


for name, B, S, H, causal in [("vision B/32 bs256", 256, 50, 12, False), ("text bs256", 256, 77, 8, True),
                              ("vision B/16 bs128", 128, 197, 12, False)]:
    D = H * 64
    qkv = torch.randn(B * S, 3 * D, device=dev)
    out, lse = ops.attention_fwd(qkv, B, S, H, causal)
    dout = torch.randn_like(out)
    fwd = t(lambda: ops.attention_fwd(qkv, B, S, H, causal))
    bwd = t(lambda: ops.attention_bwd(qkv, out, dout, lse, B, S, H, causal))
    by_f = (qkv.numel() + out.numel()) * 4
    by_b = (2 * qkv.numel() + 2 * out.numel()) * 4
    print(f"{name:20s} fwd {fwd:7.1f} us ({by_f / fwd / 1e3:5.0f} GB/s)   bwd {bwd:7.1f} us ({by_b / bwd / 1e3:5.0f} GB/s)", flush=True)
