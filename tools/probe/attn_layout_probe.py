import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dclip_amd import ops
dev = torch.device("cuda:0")
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
S = 50
for B, H in [(2048, 12), (2048 * 12, 1), (2048 * 3, 4), (256, 12), (256 * 12, 1)]:
    qkv = torch.randn(B * S, 3 * 64 * H, device=dev).to(torch.bfloat16)
    print(f"fwd bf16 head kernel: B={B} H={H} (row stride {3*64*H*2} B): {t(lambda: ops.attention_fwd_bf16(qkv, B, S, H, False)):.1f} us")
    if B * H <= 3072 * 8:
        o, l = ops.attention_fwd_bf16_lse(qkv, B, S, H, False)
        do = torch.randn_like(o)
        print(f"   bwd bf16 kernel: {t(lambda: ops.attention_bwd_bf16(qkv, o, do, l, B, S, H, False)):.1f} us")
