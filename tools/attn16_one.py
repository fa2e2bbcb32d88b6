#!/usr/bin/env python3
"""One shape of the bf16 attention forward, a few launches (for rocprofv3 counter passes): attn16_one.py B S H causal"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops
B, S, H, causal = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4]))
dev = torch.device("cuda:0")
qkv = torch.randn(B * S, 3 * H * 64, device=dev).to(torch.bfloat16)
for _ in range(5):
    out = ops.attention_fwd_bf16(qkv, B, S, H, causal)
torch.cuda.synchronize()
