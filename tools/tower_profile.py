#!/usr/bin/env python3
"""Frozen-tower forward in bf16 for rocprofv3 kernel breakdowns:  tower_profile.py {b32|l14} [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import config as dcfg, synth
from dclip_amd.clip_model import from_hf_state_dict
which = sys.argv[1] if len(sys.argv) > 1 else "l14"
cfg = dcfg.vit_l14() if which == "l14" else dcfg.vit_b32()
B = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if which == "l14" else 2048)
dev = torch.device("cuda:0")
m = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0), device=dev)
pix = torch.randn(B, 3, cfg.vision.image_size, cfg.vision.image_size, device=dev)
with torch.no_grad():
    for _ in range(5):
        m.get_image_features(pixel_values=pix, precision="bf16")
torch.cuda.synchronize()
