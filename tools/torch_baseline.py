#!/usr/bin/env python3
"""The reference's step as the reference runs it — HF transformers CLIPModel + torch ops, PyTorch-ROCm eager, fp32 —
timed on the same MI355X as bench.py, same workload (BASELINE config c2: ViT-B/32, bs 256, vision trainable, text
frozen, teacher image embedding given, cosine + contrastive losses, clip-norm 0.5, AdamW).  Context for bench.py's
number only: nothing here is part of the product, and nothing is imported from oracle/ or from the reference.

    python tools/torch_baseline.py [--batch 256] [--steps 10] [--warmup 3] [--sdpa]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from transformers import CLIPConfig, CLIPModel
from dclip_amd import config as dcfg, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--sdpa", action="store_true", help="attn_implementation='sdpa' (default: eager, as the golden vectors)")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = dcfg.vit_b32()
hf = CLIPConfig(projection_dim=cfg.projection_dim,
                vision_config=dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                                   image_size=224, patch_size=32),
                text_config=dict(hidden_size=512, intermediate_size=2048, num_hidden_layers=12, num_attention_heads=8,
                                 max_position_embeddings=77, vocab_size=49408))
hf._attn_implementation = "sdpa" if args.sdpa else "eager"
model = CLIPModel(hf).to(dev).float()
for p in model.text_model.parameters():
    p.requires_grad = False
model.text_projection.weight.requires_grad = False
model.logit_scale.requires_grad = False
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-6, fused=True)
B = args.batch
pix = synth.synth_pixel_values(B, cfg.vision, seed=0).to(dev)
ids = synth.synth_input_ids(B, cfg.text, seed=100).to(dev)
t_img = synth.synth_embeddings(B, cfg.projection_dim, seed=1000).to(dev)


def feats(x):
    return x.pooler_output if hasattr(x, "pooler_output") else x


def step():
    img = feats(model.get_image_features(pixel_values=pix)).float()
    with torch.no_grad():
        txt = feats(model.get_text_features(input_ids=ids)).float()
    l_img = (1 - (F.normalize(img, dim=1) * F.normalize(t_img, dim=1)).sum(1)).mean()
    l_txt = (1 - (F.normalize(txt, dim=1) * F.normalize(txt, dim=1)).sum(1)).mean()
    z = F.normalize(img, dim=1) @ F.normalize(txt, dim=1).t() / 0.05
    lab = torch.arange(B, device=dev)
    loss = l_img + l_txt + 0.5 * (F.cross_entropy(z, lab) + F.cross_entropy(z.t(), lab))
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 0.5)
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss


for _ in range(args.warmup):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    last = step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"what": "HF CLIPModel + torch eager fp32 on this GPU (the reference's own software path)",
                  "attention": "sdpa" if args.sdpa else "eager", "batch": B, "ms_per_step": round(dt / args.steps * 1e3, 2),
                  "images_per_s": round(B * args.steps / dt, 1), "loss": float(last), "torch": torch.__version__}))
