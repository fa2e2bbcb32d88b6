#!/usr/bin/env python3
"""Attention forward: whole-row kernel vs the 64x64 tiled kernel (DCLIP_ATTN_TILED=1) on the step's shapes."""
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


for name, B, S, H, causal in [("vision B/32", 256, 50, 12, False), ("text", 256, 77, 8, True), ("text 12 heads", 256, 77, 12, True),
                              ("regions", 2048, 50, 12, False)]:
    qkv = torch.randn(B * S, 3 * H * 64, device=dev)
    for rep in range(2):
        os.environ.pop("DCLIP_ATTN_TILED", None)
        a = t(lambda: ops.attention_fwd(qkv, B, S, H, causal))
        o1, l1 = ops.attention_fwd(qkv, B, S, H, causal)
        os.environ["DCLIP_ATTN_TILED"] = "1"
        b = t(lambda: ops.attention_fwd(qkv, B, S, H, causal))
        o2, l2 = ops.attention_fwd(qkv, B, S, H, causal)
        err = float((o1 - o2).abs().max()), float((l1 - l2).abs().max())
        mb = (4 * B * S * H * 64 * 4) / 1e6
        print(f"{name}: rows {a:.1f} us ({mb / a * 1e3 / 1e3:.2f} TB/s) | tiled {b:.1f} us | max diff out {err[0]:.2e} lse {err[1]:.2e}", flush=True)

print("--- backward")
for name, B, S, H, causal in [("vision B/32", 256, 50, 12, False), ("text", 256, 77, 8, True), ("tiny", 8, 17, 2, True)]:
    qkv = torch.randn(B * S, 3 * H * 64, device=dev)
    os.environ.pop("DCLIP_ATTN_TILED", None)
    o, l = ops.attention_fwd(qkv, B, S, H, causal)
    do = torch.randn_like(o)
    for rep in range(2):
        os.environ.pop("DCLIP_ATTN_TILED", None)
        a = t(lambda: ops.attention_bwd(qkv, o, do, l, B, S, H, causal))
        g1 = ops.attention_bwd(qkv, o, do, l, B, S, H, causal)
        os.environ["DCLIP_ATTN_TILED"] = "1"
        os.environ["DCLIP_ATTN_FUSED"] = "1"
        b = t(lambda: ops.attention_bwd(qkv, o, do, l, B, S, H, causal))
        g2 = ops.attention_bwd(qkv, o, do, l, B, S, H, causal)
        os.environ.pop("DCLIP_ATTN_FUSED", None)
        err = float((g1 - g2).abs().max() / g2.abs().max())
        mb = (8 * B * S * H * 64 * 4) / 1e6
        print(f"{name}: rows {a:.1f} us ({mb / a / 1e3:.2f} TB/s) | previous {b:.1f} us | max rel diff {err:.2e}", flush=True)

print("--- bf16 forward (frozen towers)")
for name, B, S, H, causal in [("regions B/32", 2048, 50, 12, False), ("L/14", 64, 257, 16, False), ("B/16", 128, 197, 12, False),
                              ("text", 256, 77, 8, True)]:
    os.environ.pop("DCLIP_ATTN_TILED", None)
    qkv = torch.randn(B * S, 3 * H * 64, device=dev)
    q16 = qkv.to(torch.bfloat16)
    a = t(lambda: ops.attention_fwd(qkv, B, S, H, causal))
    b = t(lambda: ops.attention_fwd_bf16(q16, B, S, H, causal))
    fl = 4.0 * S * S * 64 * B * H
    print(f"{name}: fp32 {a:.1f} us ({fl / a / 1e6:.0f} TF/s) | bf16 {b:.1f} us ({fl / b / 1e6:.0f} TF/s)", flush=True)

print("--- fp32 forward, long sequences: streamed kernel vs tiled")
for name, B, S, H, causal in [("B/16", 128, 197, 12, False), ("L/14", 64, 257, 16, False), ("causal 130", 64, 130, 8, True)]:
    qkv = torch.randn(B * S, 3 * H * 64, device=dev)
    os.environ.pop("DCLIP_ATTN_TILED", None)
    a = t(lambda: ops.attention_fwd(qkv, B, S, H, causal))
    o1, l1 = ops.attention_fwd(qkv, B, S, H, causal)
    os.environ["DCLIP_ATTN_TILED"] = "1"
    b = t(lambda: ops.attention_fwd(qkv, B, S, H, causal))
    o2, l2 = ops.attention_fwd(qkv, B, S, H, causal)
    os.environ.pop("DCLIP_ATTN_TILED", None)
    fl = 4.0 * S * S * 64 * B * H
    print(f"{name}: streamed {a:.1f} us ({fl / a / 1e6:.0f} TF/s) | tiled {b:.1f} us | max diff {float((o1 - o2).abs().max()):.2e} "
          f"lse {float((l1 - l2).abs().max()):.2e}", flush=True)

print("--- fp32 backward, long sequences: streamed kernels vs tiled")
for name, B, S, H, causal in [("B/16", 128, 197, 12, False), ("L/14", 64, 257, 16, False), ("causal 130", 64, 130, 8, True)]:
    qkv = torch.randn(B * S, 3 * H * 64, device=dev)
    os.environ.pop("DCLIP_ATTN_TILED", None)
    o, l = ops.attention_fwd(qkv, B, S, H, causal)
    do = torch.randn_like(o)
    a = t(lambda: ops.attention_bwd(qkv, o, do, l, B, S, H, causal))
    g1 = ops.attention_bwd(qkv, o, do, l, B, S, H, causal)
    os.environ["DCLIP_ATTN_TILED"] = "1"
    b = t(lambda: ops.attention_bwd(qkv, o, do, l, B, S, H, causal))
    g2 = ops.attention_bwd(qkv, o, do, l, B, S, H, causal)
    os.environ.pop("DCLIP_ATTN_TILED", None)
    fl = 10.0 * S * S * 64 * B * H
    print(f"{name}: streamed {a:.1f} us ({fl / a / 1e6:.0f} TF/s algorithmic) | tiled {b:.1f} us | max rel diff "
          f"{float((g1 - g2).abs().max() / g2.abs().max()):.2e}", flush=True)
