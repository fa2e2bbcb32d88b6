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


def test_c3_as_quoted_bf16_student_bf16_towers_8_regions():
    """c3 AS BASELINE STATES IT: bf16 student + bf16 teacher towers + 8 ragged regions per image + the meta-teacher
    inside the step (reference step: training/CLIP_image_distillation.py:594-628, fp32 there).  Gate: loss within 1e-3
    relative of the fp32 CPU oracle run on the same inputs; printed beside it: cosine of the teacher target against the
    oracle's, per-tensor gradient cosine of the student's trainable set against the oracle's fp32 gradients."""
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    dev = torch.device("cuda:0")
    cfg = dcfg.vit_b32()
    sd = synth.synth_clip_state_dict(cfg, seed=0, gain=3.0)
    tsd = synth.synth_clip_state_dict(cfg, seed=7, gain=3.0)                     # a separate frozen teacher CLIP, as in bench c3
    cm = synth.synth_cross_modal_state_dict(cfg.projection_dim, seed=31)
    student = from_hf_state_dict(cfg, sd, device=dev)
    teacher_clip = from_hf_state_dict(cfg, tsd, device=dev)
    for p_ in teacher_clip.parameters():
        p_.requires_grad = False
    teacher = PatchTextAggregation(embed_dim=512, num_heads=8, clip_model=teacher_clip, tower_precision="bf16")
    teacher.load_state_dict({f"cross_modal_attention.{k}": v for k, v in cm.items()})
    hp = argparse.Namespace(learning_rate=1e-5, warmup_steps=0, total_steps=10, train_batch_size=4, eval_batch_size=4)
    mod = CLIPImageDistillation(hp, student, None, teacher=teacher.to(dev), freeze_mode="north_star",
                                student_precision="bf16").to(dev)
    B, R = 4, 8
    pix = synth.synth_pixel_values(B, cfg.vision, seed=0)
    regions = synth.synth_regions(B, R, cfg.vision, seed=2)
    counts = torch.tensor([8, 5, 1, 7])
    ids = synth.synth_input_ids(B, cfg.text, seed=3, ragged=True, min_len=12)
    loss = mod.training_step({"pixel_values": pix, "input_ids": ids, "regions": regions, "region_counts": counts})
    loss.backward()
    with torch.no_grad():                                                        # the fp32 oracle of the same step
        toks, n_tok, sent = O.teacher_token_embeddings(tsd, ids, cfg.text)
        embs = [O.vision_tower(tsd, regions[b, :int(counts[b])], cfg.vision) for b in range(B)]
        t_img = O.global_embedding(cm, toks, O.pad_regions(embs, 512), heads=8)
        got_t = mod.teacher.compute_global_embedding_tensors(regions.to(dev), ids.to(dev), counts).float().cpu()
    t_cos = float(torch.nn.functional.cosine_similarity(got_t, t_img, dim=1).min())
    p = {k: v.clone().requires_grad_(v.is_floating_point() and k.startswith("vision") or k == "visual_projection.weight")
         for k, v in sd.items()}
    ref = O.distill_step(p, cfg, pix, ids, t_img, sent)
    ref["loss"].backward()
    got, want = float(loss.detach()), float(ref["loss"])
    named = dict(mod.student.state_dict(keep_vars=True))
    cos = {}
    for k, v in p.items():
        if v.grad is None or "self_attn" in k and ("q_proj" in k or "k_proj" in k or "v_proj" in k):
            continue
        g = named[k].grad
        if g is None:
            continue
        a, b = g.double().cpu().reshape(-1), v.grad.double().reshape(-1)
        cos[k] = float(a @ b / (a.norm() * b.norm()).clamp_min(1e-30))
    for i in (0, 11):                                                            # fused q/k/v in memory: compare the fusion
        qkv = mod.student.vision_model.encoder.layers[i].self_attn.qkv_proj.weight.grad.double().cpu().reshape(-1)
        refq = torch.cat([p[f"vision_model.encoder.layers.{i}.self_attn.{n}.weight"].grad for n in ("q_proj", "k_proj", "v_proj")])
        refq = refq.double().reshape(-1)
        cos[f"layers.{i}.qkv_proj.weight"] = float(qkv @ refq / (qkv.norm() * refq.norm()).clamp_min(1e-30))
    worst = sorted(cos.items(), key=lambda kv: kv[1])[:3]
    print(f"c3 as quoted (bf16 student, bf16 teacher towers, 8 ragged regions, B={B}): loss {got:.6f} vs fp32 oracle {want:.6f} "
          f"(rel {abs(got - want) / abs(want):.2e}); teacher-target min cosine {t_cos:.6f}; gradient cosine over {len(cos)} tensors "
          f"min {min(cos.values()):.5f} median {sorted(cos.values())[len(cos) // 2]:.5f}; worst {worst}")
    assert abs(got - want) <= 1e-3 * abs(want)
    assert t_cos > 0.9995
    assert min(cos.values()) > 0.99


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


# ------------------------------------------------------------------------------------------ config c5
# ViT-L/14 teacher (P = 768) -> ViT-B/32 student (P = 512).  Goldens: tests/golden/towers_l14.npz (HF CLIPModel towers)
# and tests/golden/step_c5.npz (the REFERENCE's compute_global_embedding_batch / aggregate_text over those towers,
# the build's declared 768->512 bridge, the reference's losses, an HF student) — oracle/make_golden.py::gen_c5.

def _T(a):
    import numpy as np
    return torch.from_numpy(np.ascontiguousarray(a))


def test_c5_vit_l14_towers_fp32(golden):
    from dclip_amd.clip_model import from_hf_state_dict
    import numpy as np
    g = golden("towers_l14.npz")
    dev = torch.device("cuda:0")
    cfg = dcfg.vit_l14()
    m = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=2, gain=3.0), device=dev)
    pix = synth.synth_pixel_values(2, cfg.vision, seed=0).to(dev)
    ids = _T(g["l14.input_ids"]).to(dev)
    with torch.no_grad():
        img = m.get_image_features(pixel_values=pix)
        txt = m.get_text_features(input_ids=ids)
        vh = m.hidden_states(pixel_values=pix)
    assert _rel(img, _T(g["l14.image_emb"])) < 1e-3            # BASELINE parity gate: 1e-3 relative on embeddings
    assert _rel(txt, _T(g["l14.text_emb"])) < 1e-3
    stats = np.array([[float(h.double().mean()), float(h.double().std()), float(h.double().abs().max())] for h in vh])
    np.testing.assert_allclose(stats, g["l14.vision_layer_stats"], rtol=1e-3, atol=1e-5)
    assert _rel(vh[-1][:, 0, :], _T(g["l14.vision_cls_last"])) < 1e-3
    # the same towers with bf16 GEMM inputs (what c5 is quoted in): MEASURED error, loose sanity bound only
    with torch.no_grad():
        img16 = m.get_image_features(pixel_values=pix, precision="bf16")
        txt16 = m.get_text_features(input_ids=ids, precision="bf16")
    cos_i = torch.nn.functional.cosine_similarity(img16.double(), img.double(), dim=1).min()
    cos_t = torch.nn.functional.cosine_similarity(txt16.double(), txt.double(), dim=1).min()
    print(f"L/14 bf16 towers vs fp32: image max rel {_rel(img16, img):.3e} min cos {float(cos_i):.6f}; "
          f"text max rel {_rel(txt16, txt):.3e} min cos {float(cos_t):.6f}")
    assert float(cos_i) > 0.999 and float(cos_t) > 0.999


def _c5_module(dev, g, tower_precision="fp32"):
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    tcfg, scfg = dcfg.vit_l14(), dcfg.vit_b32()
    tclip = from_hf_state_dict(tcfg, synth.synth_clip_state_dict(tcfg, seed=int(g["teacher_seed"]), gain=3.0), device=dev)
    for p in tclip.parameters():
        p.requires_grad = False
    E = tcfg.projection_dim
    teacher = PatchTextAggregation(embed_dim=E, num_heads=E // 64, clip_model=tclip, tower_precision=tower_precision)
    cm = synth.synth_cross_modal_state_dict(E, seed=int(g["cm_seed"]))
    teacher.load_state_dict({f"cross_modal_attention.{k}": v for k, v in cm.items()})
    student = from_hf_state_dict(scfg, synth.synth_clip_state_dict(scfg, seed=int(g["student_seed"]), gain=3.0), device=dev)
    hp = argparse.Namespace(learning_rate=1e-5, warmup_steps=0, total_steps=10, train_batch_size=2, eval_batch_size=2)
    mod = CLIPImageDistillation(hp, student, None, teacher=teacher.to(dev), freeze_mode="as_written").to(dev)
    assert mod.teacher_bridge is not None and tuple(mod.teacher_bridge.weight.shape) == (512, 768)
    assert "teacher_bridge.weight" in mod.state_dict()            # stored under its own checkpoint key
    assert not any(p is mod.teacher_bridge.weight for p in mod.parameters())
    return mod, tcfg, scfg


def test_c5_step_l14_teacher_b32_student(golden):
    """c5 end to end at B=2 against the reference-generated golden: teacher targets (768), bridged targets (512), the
    three losses ≤1e-3, and the student's gradients (as_written freeze set = every tensor the golden has a probe for
    that the rule leaves trainable)."""
    from dclip_amd.probe import probe_vector
    g = golden("step_c5.npz")
    dev = torch.device("cuda:0")
    mod, tcfg, scfg = _c5_module(dev, g)
    ids = _T(g["input_ids"])
    B = ids.shape[0]
    regions = synth.synth_regions(B, 2, tcfg.vision, seed=int(g["regions_seed"]))
    counts = _T(g["n_regions"]).to(torch.int32)
    pix = synth.synth_pixel_values(B, scfg.vision, seed=int(g["pixel_seed"]))
    with torch.no_grad():
        t_img = mod.teacher.compute_global_embedding_tensors(regions.to(dev), ids.to(dev), counts)
        t_txt = mod.teacher.last_sentence_embedding
    assert _rel(t_img, _T(g["teacher_image_768"])) < 1e-3
    assert _rel(t_txt, _T(g["teacher_text_768"])) < 1e-3
    assert _rel(mod.teacher_bridge(t_img), _T(g["bridged_image"])) < 1e-3
    loss = mod.training_step({"pixel_values": pix, "input_ids": ids, "regions": regions, "region_counts": counts})
    loss.backward()
    for k, name in (("loss_image", "loss_image"), ("loss_text", "loss_text"), ("loss_contrastive", "loss_contrastive")):
        got, want = float(mod.last_losses[name]), float(g[k])
        assert abs(got - want) <= 1e-3 * abs(want), (k, got, want)
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-3 * abs(float(g["loss"]))
    # gradient probes: (norm, <g, probe>) per HF-named tensor
    sd_grads = {}
    for n, p in mod.student.named_parameters():
        if p.grad is None:
            continue
        if n.endswith("qkv_proj.weight") or n.endswith("qkv_proj.bias"):
            D = p.shape[0] // 3
            kind = n.rsplit(".", 1)[1]
            for i, part in enumerate(("q_proj", "k_proj", "v_proj")):
                sd_grads[n.replace("qkv_proj." + kind, f"{part}.{kind}")] = p.grad[i * D:(i + 1) * D]
        else:
            sd_grads[n] = p.grad
    checked = 0
    for k, gr in sd_grads.items():
        want = g[f"gradprobe.{k}"]
        gr = gr.detach().double().cpu().reshape(-1)
        if want[0] < 1e-7:
            continue
        assert abs(float(gr.norm()) - want[0]) <= 3e-3 * want[0], (k, float(gr.norm()), want[0])
        dot = float(gr @ probe_vector(k, gr.numel()).double())
        assert abs(dot - want[1]) <= 3e-3 * max(abs(want[1]), want[0] * 0.05), (k, dot, want[1])
        checked += 1
    assert checked > 150, checked


def test_c5_step_bf16_teacher_towers_measured(golden):
    """c5 as BASELINE quotes it ("bf16 MFMA"): the frozen L/14 towers multiply in bf16.  Loss must still meet 1e-3
    against the fp32 golden; the target error is reported."""
    g = golden("step_c5.npz")
    dev = torch.device("cuda:0")
    mod, tcfg, scfg = _c5_module(dev, g, tower_precision="bf16")
    ids = _T(g["input_ids"])
    B = ids.shape[0]
    regions = synth.synth_regions(B, 2, tcfg.vision, seed=int(g["regions_seed"]))
    counts = _T(g["n_regions"]).to(torch.int32)
    pix = synth.synth_pixel_values(B, scfg.vision, seed=int(g["pixel_seed"]))
    loss = mod.training_step({"pixel_values": pix, "input_ids": ids, "regions": regions, "region_counts": counts})
    with torch.no_grad():
        t_img = mod.teacher.compute_global_embedding_tensors(regions.to(dev), ids.to(dev), counts)
    cos = torch.nn.functional.cosine_similarity(t_img.double().cpu(), _T(g["teacher_image_768"]).double(), dim=1).min()
    print(f"c5 bf16 teacher: loss {float(loss.detach()):.6f} vs fp32 golden {float(g['loss']):.6f}; "
          f"teacher target min cosine {float(cos):.6f}")
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-3 * abs(float(g["loss"]))
    assert float(cos) > 0.999


# ------------------------------------------------------------------------------------------ config c2 at the benched size

@pytest.mark.timeout(900)
def test_c2_full_size_step_vs_oracle_and_properties():
    """BASELINE config c2 at ITS size (ViT-B/32, 256 pairs, fp32): embeddings and the three losses against the CPU oracle
    (forward only there), plus size-independent properties of the step — bit-reproducibility, and the shard sum rule of the
    global-negatives loss (two half-batch row blocks against all columns add up to the full loss)."""
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.CLIP_image_distillation import distill_losses
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    cfg = dcfg.vit_b32()
    sd = synth.synth_clip_state_dict(cfg, seed=0, gain=3.0)
    m = from_hf_state_dict(cfg, sd, device=dev)
    for p in m.text_model.parameters():
        p.requires_grad = False
    m.text_projection.weight.requires_grad = False
    m.logit_scale.requires_grad = False
    B = 256
    pix = synth.synth_pixel_values(B, cfg.vision, seed=0)
    ids = synth.synth_input_ids(B, cfg.text, seed=100)
    t_img = synth.synth_embeddings(B, cfg.projection_dim, seed=1000)

    def run():
        for p in m.parameters():
            p.grad = None
        img = m.get_image_features(pixel_values=pix.to(dev))
        with torch.no_grad():
            txt = m.get_text_features(input_ids=ids.to(dev))
        out = distill_losses(img, txt, t_img.to(dev), txt.detach())
        out["loss"].backward()
        return img.detach(), txt.detach(), {k: v.detach().clone() for k, v in out.items()}, \
            m.vision_model.encoder.layers[3].mlp.fc1.weight.grad.clone()

    img, txt, out, g = run()
    img2, _txt2, out2, g2 = run()
    assert torch.equal(img, img2) and torch.equal(out["loss"], out2["loss"]) and torch.equal(g, g2)      # deterministic
    import os
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        ref = O.distill_step(sd, cfg, pix, ids, t_img)
    assert _rel(img, ref["image_emb"]) < 1e-3 and _rel(txt, ref["text_emb"]) < 1e-3
    for k in ("loss_image", "loss_text", "loss_contrastive", "loss"):
        got, want = float(out[k]), float(ref[k])
        assert abs(got - want) <= 1e-3 * max(abs(want), 1e-3), (k, got, want)
    # shard sum rule (what the N-rank run relies on): row blocks of the LSE pass against ALL columns
    ihat, _ = ops.normalize_rows_fwd(img.contiguous())
    that, _ = ops.normalize_rows_fwd(txt.contiguous())
    full_i, diag = ops.contrastive_lse(ihat, that, 0, 20.0)
    h = B // 2
    lo, dlo = ops.contrastive_lse(ihat[:h].contiguous(), that, 0, 20.0)
    hi, dhi = ops.contrastive_lse(ihat[h:].contiguous(), that, h, 20.0)
    assert torch.allclose(torch.cat([lo, hi]), full_i, rtol=1e-6, atol=1e-6)
    assert torch.allclose(torch.cat([dlo, dhi]), diag, rtol=1e-6, atol=1e-6)


@pytest.mark.timeout(900)
def test_c4_full_per_gpu_size_forward_vs_oracle():
    """BASELINE config c4's per-GPU work at ITS size (ViT-B/16, 128 pairs per GPU of the 1024 global batch): image / text
    embeddings and the local cosine + contrastive losses against the CPU oracle (forward only there)."""
    import os
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.CLIP_image_distillation import distill_losses
    dev = torch.device("cuda:0")
    cfg = dcfg.vit_b16()
    sd = synth.synth_clip_state_dict(cfg, seed=1, gain=3.0)
    m = from_hf_state_dict(cfg, sd, device=dev)
    B = 128
    pix = synth.synth_pixel_values(B, cfg.vision, seed=4)
    ids = synth.synth_input_ids(B, cfg.text, seed=104)
    t_img = synth.synth_embeddings(B, cfg.projection_dim, seed=1004)
    with torch.no_grad():
        img = m.get_image_features(pixel_values=pix.to(dev))
        txt = m.get_text_features(input_ids=ids.to(dev))
        out = distill_losses(img, txt, t_img.to(dev), txt)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        ref = O.distill_step(sd, cfg, pix, ids, t_img)
    assert _rel(img, ref["image_emb"]) < 1e-3 and _rel(txt, ref["text_emb"]) < 1e-3
    for k in ("loss_image", "loss_contrastive", "loss"):
        got, want = float(out[k]), float(ref[k])
        assert abs(got - want) <= 1e-3 * max(abs(want), 1e-3), (k, got, want)
