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
B, S, H = 256, 50, 12
D = 64 * H
qkv = torch.randn(B * S, 3 * D, device=dev).to(torch.bfloat16)
do = torch.randn(B * S, D, device=dev).to(torch.bfloat16)
o1, l1 = ops.attention_fwd_io16(qkv, B, S, H, False)
o2, l2 = ops.attention_fwd_bf16_lse(qkv, B, S, H, False)
print("fwd io16 %.1f us | fwd bf16 mfma %.1f us" % (t(lambda: ops.attention_fwd_io16(qkv, B, S, H, False)), t(lambda: ops.attention_fwd_bf16_lse(qkv, B, S, H, False))))
print("bwd io16 %.1f us | bwd bf16 mfma %.1f us" % (t(lambda: ops.attention_bwd_io16(qkv, o1, do, l1, B, S, H, False)), t(lambda: ops.attention_bwd_bf16(qkv, o2, do, l2, B, S, H, False))))
