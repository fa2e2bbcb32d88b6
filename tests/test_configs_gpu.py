"""BASELINE.json configs c3 and c4 as parity-test cases at real model size (small batch): the meta-teacher with 8
regions per image in front of the ViT-B/32 student, and a ViT-B/16 distill step — against the CPU oracle."""
import argparse

import pytest
import torch

from dclip_amd import config as dcfg, synth
from oracle import dclip_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_c3_meta_teacher_8_regions_real_size():
    """c3: ViT-B/32 + meta-teacher (8 regions/img in [0,1], ragged counts, 77-token captions)."""
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    dev = torch.device("cuda:0")
    cfg = dcfg.vit_b32()
    sd = synth.synth_clip_state_dict(cfg, seed=0, gain=3.0)
    cm = synth.synth_cross_modal_state_dict(cfg.projection_dim, seed=31)
    student = from_hf_state_dict(cfg, sd, device=dev)
    teacher = PatchTextAggregation(embed_dim=512, num_heads=8, clip_model=student)
    teacher.load_state_dict({f"cross_modal_attention.{k}": v for k, v in cm.items()})
    hp = argparse.Namespace(learning_rate=1e-5, warmup_steps=0, total_steps=10, train_batch_size=2, eval_batch_size=2)
    mod = CLIPImageDistillation(hp, student, None, teacher=teacher.to(dev), freeze_mode="north_star").to(dev)
    B, R = 2, 8
    pix = synth.synth_pixel_values(B, cfg.vision, seed=0)
    regions = synth.synth_regions(B, R, cfg.vision, seed=2)
    counts = torch.tensor([8, 5])
    ids = synth.synth_input_ids(B, cfg.text, seed=3, ragged=True, min_len=20)
    loss = mod.training_step({"pixel_values": pix, "input_ids": ids, "regions": regions, "region_counts": counts})
    loss.backward()
    # oracle
    with torch.no_grad():
        toks, n_tok, sent = O.teacher_token_embeddings(sd, ids, cfg.text)
        embs = [O.vision_tower(sd, regions[b, :int(counts[b])], cfg.vision) for b in range(B)]
        t_img = O.global_embedding(cm, toks, O.pad_regions(embs, 512), heads=8)
    with torch.no_grad():
        got_t = mod.teacher.compute_global_embedding_tensors(regions.to(dev), ids.to(dev), counts)
    assert _rel(got_t, t_img) < 1e-3
    p = {k: v.clone().requires_grad_(v.is_floating_point() and k.startswith("vision") or k == "visual_projection.weight")
         for k, v in sd.items()}
    ref = O.distill_step(p, cfg, pix, ids, t_img)
    ref["loss"].backward()
    assert abs(float(loss.detach()) - float(ref["loss"])) < 1e-3 * abs(float(ref["loss"]))
    g = mod.student.visual_projection.weight.grad
    assert _rel(g, p["visual_projection.weight"].grad) < 2e-3


def test_c4_vit_b16_step():
    """c4's per-GPU work at small batch: ViT-B/16 (S = 197: multi-tile attention) distill step."""
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.CLIP_image_distillation import distill_losses
    dev = torch.device("cuda:0")
    cfg = dcfg.vit_b16()
    sd = synth.synth_clip_state_dict(cfg, seed=1, gain=3.0)
    m = from_hf_state_dict(cfg, sd, device=dev)
    B = 2
    pix = synth.synth_pixel_values(B, cfg.vision, seed=0)
    ids = synth.synth_input_ids(B, cfg.text, seed=3, ragged=True, min_len=8)
    t_img, t_txt = synth.synth_embeddings(B, 512, seed=1), synth.synth_embeddings(B, 512, seed=5)
    img = m.get_image_features(pixel_values=pix.to(dev))
    txt = m.get_text_features(input_ids=ids.to(dev))
    out = distill_losses(img, txt, t_img.to(dev), t_txt.to(dev))
    out["loss"].backward()
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref = O.distill_step(p, cfg, pix, ids, t_img, t_txt)
    ref["loss"].backward()
    assert _rel(img, ref["image_emb"]) < 1e-3 and _rel(txt, ref["text_emb"]) < 1e-3
    assert abs(float(out["loss"].detach()) - float(ref["loss"])) < 1e-3 * abs(float(ref["loss"]))
    for name, mine in (("vision_model.embeddings.patch_embedding.weight", m.vision_model.embeddings.patch_embedding.weight),
                       ("vision_model.encoder.layers.0.mlp.fc1.weight", m.vision_model.encoder.layers[0].mlp.fc1.weight),
                       ("vision_model.encoder.layers.11.self_attn.out_proj.weight",
                        m.vision_model.encoder.layers[11].self_attn.out_proj.weight),
                       ("text_model.encoder.layers.5.mlp.fc2.weight", m.text_model.encoder.layers[5].mlp.fc2.weight)):
        a, b = mine.grad.double().cpu().reshape(-1), p[name].grad.double().reshape(-1)
        assert float(a @ b / (a.norm() * b.norm())) > 0.9999, name
