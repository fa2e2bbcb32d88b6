#!/usr/bin/env python3
"""Long-sequence fp32 attention backward (ViT-B/16: 197 tokens, ViT-L/14: 257): dS passed through memory (default; the
dQ kernel reads the dS blocks the dK/dV kernel formed: 5 MFMA products) against the two-kernel recompute split
(DCLIP_ATTN_NO_DS=1: 7 products).  Same process, alternating; gradients compared with each other and with fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")


def t(f, n=20):
    for _ in range(4):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def ref64(qkv, dout, B, S, H):
    x = qkv.double().view(B, S, 3, H, 64).requires_grad_(True)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B * S, H * 64)
    o.backward(dout.double())
    return x.grad.view(B * S, 3 * H * 64)


for name, B, S, H in [("B/16", 128, 197, 12), ("L/14", 64, 257, 16), ("S=81", 32, 81, 4), ("S=224", 16, 224, 12)]:
    qkv = torch.randn(B * S, 3 * H * 64, device=dev)
    o, l = ops.attention_fwd(qkv, B, S, H, False)
    do = torch.randn_like(o)
    for rep in range(2):
        os.environ.pop("DCLIP_ATTN_NO_DS", None)
        a = t(lambda: ops.attention_bwd(qkv, o, do, l, B, S, H, False))
        g1 = ops.attention_bwd(qkv, o, do, l, B, S, H, False)
        os.environ["DCLIP_ATTN_NO_DS"] = "1"
        b = t(lambda: ops.attention_bwd(qkv, o, do, l, B, S, H, False))
        g2 = ops.attention_bwd(qkv, o, do, l, B, S, H, False)
        os.environ.pop("DCLIP_ATTN_NO_DS", None)
    fl = 10.0 * S * S * 64 * B * H
    nb = min(B, 8)
    r = ref64(qkv[:nb * S], do[:nb * S], nb, S, H)
    e1 = float((g1[:nb * S].double() - r).abs().max() / r.abs().max())
    e2 = float((g2[:nb * S].double() - r).abs().max() / r.abs().max())
    print(f"{name} (B={B}, S={S}, H={H}): dS passed {a:.1f} us ({fl / a / 1e6:.0f} TF/s algorithmic = {fl / a / 1e6 / 157.3:.2f} of the "
          f"fp32 MFMA peak) | recompute {b:.1f} us ({fl / b / 1e6 / 157.3:.2f}) | max rel diff between them "
          f"{float((g1 - g2).abs().max() / g2.abs().max()):.2e} | vs fp64: {e1:.2e} / {e2:.2e}", flush=True)
