"""Pins oracle/dclip_oracle.py (our CPU restatement) to the golden vectors that
oracle/make_golden.py produced by running the reference's own code (CPU-only test)."""
import numpy as np
import pytest
import torch

from dclip_amd import config as dcfg, synth
from dclip_amd.probe import probe_vector
from oracle import dclip_oracle as O

torch.set_num_threads(8)


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=2e-5, atol=2e-6):
    a = a.detach().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def close_rel(a, b, tol):
    """max|a-b| <= tol * max|b| — for fp32 gradients whose entries span orders of magnitude."""
    a = a.detach().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    err, ref = np.abs(a - b).max(), max(np.abs(b).max(), 1e-30)
    assert err <= tol * ref, (err, ref)


def wsum(sd):
    return sum(float(v.double().sum()) for v in sd.values())


def check_probe(g, key, expect, rtol=2e-4, atol=2e-5):
    """(norm, <g, probe>) against the stored pair.  `atol` covers gradients that are zero in exact
    arithmetic (k_proj.bias: softmax is shift-invariant) and so hold only rounding noise."""
    g = g.detach().double().reshape(-1)
    got = np.array([float(g.norm()), float(g @ probe_vector(key, g.numel()).double())])
    scale = max(abs(expect[0]), 1e-12)
    assert abs(got[0] - expect[0]) <= rtol * scale + atol, (key, got, expect)
    assert abs(got[1] - expect[1]) <= rtol * scale * 10 + atol, (key, got, expect)


# ---------------------------------------------------------------- F1 losses

def test_survey_quoted_losses(golden):
    g = golden("losses.npz")
    # SURVEY.md §8c quotes these two values from the lifted reference functions
    assert abs(float(g["survey_con"]) - 2.1221024990081787) < 1e-6
    assert abs(float(g["survey_cos"]) - 1.0040297508239746) < 1e-6
    gen = torch.Generator().manual_seed(1234)
    img, txt, tea = (torch.randn(8, 512, generator=gen) for _ in range(3))
    close(O.contrastive_loss(img, txt), g["survey_con"])
    close(O.cosine_distillation_loss(img, tea), g["survey_cos"])


@pytest.mark.parametrize("name", ["b1_p512", "b2_p512", "b8_p512", "b8_p64", "b37_p768"])
def test_losses_small(golden, name):
    g = golden("losses.npz")
    i, t, e = (T(g[f"{name}.{k}"]).clone().requires_grad_(True) for k in ("img", "txt", "tea"))
    lc = O.contrastive_loss(i, t)
    close(lc, g[f"{name}.con"])
    gi, gt = torch.autograd.grad(lc, (i, t))
    close(gi, g[f"{name}.d_img"], rtol=1e-4, atol=1e-7)
    close(gt, g[f"{name}.d_txt"], rtol=1e-4, atol=1e-7)
    lk = O.cosine_distillation_loss(i, e)
    close(lk, g[f"{name}.cos"])
    gs, = torch.autograd.grad(lk, (i,))
    close(gs, g[f"{name}.d_stu"], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("name,B,P", [("b256_p512", 256, 512), ("b1024_p512", 1024, 512)])
def test_losses_large_seeded(golden, name, B, P):
    g = golden("losses.npz")
    gen = torch.Generator().manual_seed(int(g[f"{name}.seed"]))
    i = torch.randn(B, P, generator=gen) * 1.7
    t = torch.randn(B, P, generator=gen) + 0.1 * i
    e = torch.randn(B, P, generator=gen)
    i.requires_grad_(True), t.requires_grad_(True)
    lc = O.contrastive_loss(i, t)
    close(lc, g[f"{name}.con"])
    gi, gt = torch.autograd.grad(lc, (i, t))
    check_probe(gi, f"{name}.d_img", g[f"{name}.d_img.probe"])
    check_probe(gt, f"{name}.d_txt", g[f"{name}.d_txt.probe"])
    close(O.cosine_distillation_loss(i, e), g[f"{name}.cos"])


def test_losses_edge_cases(golden):
    g = golden("losses.npz")
    same = torch.ones(8, 512)
    close(O.contrastive_loss(same, same), g["same_b8.con"])
    assert abs(float(g["same_b8.con"]) - np.log(8.0)) < 1e-5
    z, zt = T(g["zero_row.img"]), T(g["zero_row.txt"])
    close(O.contrastive_loss(z, zt), g["zero_row.con"])
    close(O.cosine_distillation_loss(z, zt), g["zero_row.cos"])


# ---------------------------------------------------------------- F2 cross-modal block

@pytest.mark.parametrize("name", ["e128", "e512", "e512_c3"])
def test_cross_modal(golden, name):
    g = golden("cross_modal.npz")
    E, H, seed = int(g[f"{name}.E"]), int(g[f"{name}.H"]), int(g[f"{name}.seed"])
    sd = synth.synth_cross_modal_state_dict(E, seed=seed)
    assert abs(wsum(sd) - float(g[f"{name}.wsum"])) < 1e-6 * max(1.0, abs(float(g[f"{name}.wsum"])))
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    text, patches, sent = (T(g[f"{name}.{k}"]) for k in ("text", "patches", "sentence"))
    at, ai = O.cross_modal_attention(p, text, patches, H)
    close(at, g[f"{name}.attended_text"], rtol=1e-4, atol=2e-5)
    close(ai, g[f"{name}.attended_image"], rtol=1e-4, atol=2e-5)
    close(O.aggregation(at), g[f"{name}.text_global"], rtol=1e-4, atol=2e-5)
    close(O.aggregation(ai), g[f"{name}.image_global"], rtol=1e-4, atol=2e-5)
    out = O.teacher_step(p, text, patches, sent, H)
    close(out["image_emb"], g[f"{name}.global"], rtol=1e-4, atol=2e-5)
    close(out["loss"], g[f"{name}.loss"], rtol=1e-5)
    grads = torch.autograd.grad(out["loss"], list(p.values()))
    for (k, _), gr in zip(p.items(), grads):
        check_probe(gr, k, g[f"{name}.gradprobe.{k}"], rtol=5e-4)
        if E == 128:
            close_rel(gr, g[f"{name}.grad.{k}"], 2e-4)


# ---------------------------------------------------------------- F3 towers

def test_towers_tiny_forward_f32_f64(golden):
    g = golden("towers_tiny.npz")
    cfg = dcfg.tiny()
    sd = synth.synth_clip_state_dict(cfg, seed=7, gain=4.0)
    assert abs(wsum(sd) - float(g["wsum"])) < 1e-6 * abs(float(g["wsum"]))
    pix, ids = T(g["pixel_values"]), T(g["input_ids"])
    for dt, tag, tol in ((torch.float32, "f32", 3e-5), (torch.float64, "f64", 3e-6)):  # HF eager softmax is fp32 even in an f64 model
        p = O.to_dtype(sd, dt)
        img, vh = O.vision_tower(p, pix.to(dt), cfg.vision, return_hidden=True)
        txt, th = O.text_tower(p, ids, cfg.text, return_hidden=True)
        for i, h in enumerate(vh):
            close(h, g[f"{tag}.vision_hidden.{i}"], rtol=tol * 10, atol=tol * 10)
        close(th, g[f"{tag}.text_last_hidden"], rtol=tol * 10, atol=tol * 10)
        close(img, g[f"{tag}.image_emb"], rtol=tol * 10, atol=tol * 10)
        close(txt, g[f"{tag}.text_emb"], rtol=tol * 10, atol=tol * 10)


def test_towers_tiny_param_grads(golden):
    g = golden("towers_tiny.npz")
    cfg = dcfg.tiny()
    sd = synth.synth_clip_state_dict(cfg, seed=7, gain=4.0)
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    pix, ids = T(g["pixel_values"]), T(g["input_ids"])
    img = O.vision_tower(p, pix, cfg.vision)
    txt = O.text_tower(p, ids, cfg.text)
    obj = (img * T(g["obj_w_img"])).sum() + (txt * T(g["obj_w_txt"])).sum()
    keys = [k for k in p if k != "logit_scale"]
    grads = torch.autograd.grad(obj, [p[k] for k in keys], allow_unused=True)
    n = 0
    for k, gr in zip(keys, grads):
        gr = torch.zeros_like(p[k]) if gr is None else gr
        check_probe(gr, k, g[f"gradprobe.{k}"], rtol=1e-3)
        if f"grad.{k}" in g.files:
            if np.abs(g[f"grad.{k}"]).max() > 1e-4:
                close_rel(gr.reshape(g[f"grad.{k}"].shape), g[f"grad.{k}"], 2e-4)
            n += 1
    assert n > 10


def test_towers_real_b32_forward(golden):
    g = golden("towers_real.npz")
    cfg = dcfg.vit_b32()
    sd = synth.synth_clip_state_dict(cfg, seed=0, gain=3.0)
    assert abs(wsum(sd) - float(g["b32.wsum"])) < 1e-6 * abs(float(g["b32.wsum"]))
    pix = synth.synth_pixel_values(2, cfg.vision, seed=0)
    ids = T(g["b32.input_ids"])
    with torch.no_grad():
        img, vh = O.vision_tower(sd, pix, cfg.vision, return_hidden=True)
        txt = O.text_tower(sd, ids, cfg.text)
    close(img, g["b32.image_emb"], rtol=1e-3, atol=1e-4)
    close(txt, g["b32.text_emb"], rtol=1e-3, atol=1e-4)
    stats = np.array([[float(h.double().mean()), float(h.double().std()), float(h.double().abs().max())]
                      for h in vh])
    np.testing.assert_allclose(stats, g["b32.vision_layer_stats"], rtol=1e-3, atol=1e-5)


# ---------------------------------------------------------------- teacher glue (a4/a5/a8)

def test_teacher_glue(golden):
    g = golden("teacher_glue.npz")
    cfg = dcfg.tiny()
    sd = synth.synth_clip_state_dict(cfg, seed=int(g["clip_seed"]), gain=4.0)
    cm = synth.synth_cross_modal_state_dict(cfg.projection_dim, seed=int(g["cm_seed"]))
    ids, regions, n_regions = T(g["input_ids"]), T(g["regions"]), g["n_regions"]
    with torch.no_grad():
        toks, n_tok, sent = O.teacher_token_embeddings(sd, ids, cfg.text)
        np.testing.assert_array_equal(n_tok.numpy(), g["n_tok"])
        close(sent, g["sentence"], rtol=1e-4, atol=1e-5)
        for b in range(ids.shape[0]):
            close(toks[b, :int(n_tok[b])], g[f"tokens.{b}"], rtol=1e-4, atol=1e-5)
            assert float(toks[b, int(n_tok[b]):].abs().sum()) == 0.0
        embs = [O.vision_tower(sd, regions[b, :int(n_regions[b])], cfg.vision) if n_regions[b] > 0
                else torch.zeros(0, cfg.projection_dim) for b in range(ids.shape[0])]
        patches = O.pad_regions(embs, cfg.projection_dim)
        glob = O.global_embedding(cm, toks, patches, heads=cfg.projection_dim // 64)
    close(glob, g["global"], rtol=1e-4, atol=1e-5)


def test_teacher_nan_guards(golden):
    """The reference's NaN / Inf guards (run as written when the golden was made): a NaN region crop, a caption
    through a NaN token embedding, and a NaN weight inside the cross-modal block (whole batch -> zeros)."""
    g = golden("teacher_guards.npz")
    cfg = dcfg.tiny()
    sd = synth.synth_clip_state_dict(cfg, seed=int(g["clip_seed"]), gain=4.0)
    sd["text_model.embeddings.token_embedding.weight"][int(g["nan_token_id"])] = float("nan")
    ids, regions, n_regions = T(g["input_ids"]), T(g["regions"]), g["n_regions"]
    assert bool(torch.isnan(regions).any())
    with torch.no_grad():
        toks, n_tok, _ = O.teacher_token_embeddings(sd, ids, cfg.text)
        assert bool(torch.isnan(toks[1]).any())
        embs = [O.vision_tower(sd, regions[b, :int(n_regions[b])], cfg.vision) for b in range(ids.shape[0])]
        assert not bool(torch.isfinite(embs[0][1]).all()) and bool(torch.isfinite(embs[0][0]).all())
        patches = O.pad_regions(embs, cfg.projection_dim)
        cm = synth.synth_cross_modal_state_dict(cfg.projection_dim, seed=int(g["cm_seed"]))
        glob = O.global_embedding(cm, toks, patches, heads=cfg.projection_dim // 64)
        close(glob, g["global"], rtol=1e-4, atol=1e-5)
        cm["norm_text.weight"][3] = float("nan")
        glob2 = O.global_embedding(cm, toks, patches, heads=cfg.projection_dim // 64)
    assert float(glob2.abs().sum()) == 0.0 and float(np.abs(g["global_poisoned_block"]).sum()) == 0.0


# ---------------------------------------------------------------- F4 full step (config c1)

@pytest.mark.timeout(600)
def test_step_c1(golden):
    g = golden("step_c1.npz")
    cfg = dcfg.vit_b32()
    sd = synth.synth_clip_state_dict(cfg, seed=0, gain=3.0)
    assert abs(wsum(sd) - float(g["wsum"])) < 1e-6 * abs(float(g["wsum"]))
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    B = 8
    pix = synth.synth_pixel_values(B, cfg.vision, seed=0)
    ids = T(g["input_ids"])
    t_img = synth.synth_embeddings(B, cfg.projection_dim, seed=1)
    t_txt = synth.synth_embeddings(B, cfg.projection_dim, seed=5)
    out = O.distill_step(p, cfg, pix, ids, t_img, t_txt)
    close(out["image_emb"], g["image_emb"], rtol=1e-3, atol=1e-4)
    close(out["text_emb"], g["text_emb"], rtol=1e-3, atol=1e-4)
    for k in ("loss_image", "loss_text", "loss_contrastive", "loss"):
        close(out[k], g[k], rtol=1e-5)
    keys = [k for k in p if k != "logit_scale"]
    grads = torch.autograd.grad(out["loss"], [p[k] for k in keys], allow_unused=True)
    for k, gr in zip(keys, grads):
        gr = torch.zeros_like(p[k]) if gr is None else gr
        check_probe(gr, k, g[f"gradprobe.{k}"], rtol=2e-3)


# ---------------------------------------------------------------- config c5 (ViT-L/14 teacher -> ViT-B/32 student)

@pytest.mark.timeout(900)
def test_towers_l14_forward(golden):
    """The oracle's towers at ViT-L/14 size (hidden 1024 / 24 layers / 16 heads / patch 14 / proj 768; text 768/12/3072)
    against HF CLIPModel(CLIPConfig(...)) outputs — the c5 teacher."""
    g = golden("towers_l14.npz")
    cfg = dcfg.vit_l14()
    sd = synth.synth_clip_state_dict(cfg, seed=2, gain=3.0)
    assert abs(wsum(sd) - float(g["l14.wsum"])) < 1e-6 * abs(float(g["l14.wsum"]))
    pix = synth.synth_pixel_values(2, cfg.vision, seed=0)
    ids = T(g["l14.input_ids"])
    with torch.no_grad():
        img, vh = O.vision_tower(sd, pix, cfg.vision, return_hidden=True)
        txt = O.text_tower(sd, ids, cfg.text)
    close(img, g["l14.image_emb"], rtol=1e-3, atol=1e-4)
    close(txt, g["l14.text_emb"], rtol=1e-3, atol=1e-4)
    stats = np.array([[float(h.double().mean()), float(h.double().std()), float(h.double().abs().max())] for h in vh])
    np.testing.assert_allclose(stats, g["l14.vision_layer_stats"], rtol=1e-3, atol=1e-5)


@pytest.mark.timeout(900)
def test_step_c5(golden):
    """One c5 step: the golden's teacher targets come from the REFERENCE's compute_global_embedding_batch /
    aggregate_text over L/14 towers; the bridge is the build's declared rule; losses are the reference's."""
    from dclip_amd.CLIP_image_distillation import bridge_weight
    g = golden("step_c5.npz")
    tcfg, scfg = dcfg.vit_l14(), dcfg.vit_b32()
    W = O.bridge_weight(scfg.projection_dim, tcfg.projection_dim, 0)
    assert torch.equal(W, bridge_weight(scfg.projection_dim, tcfg.projection_dim, 0))        # product == restatement
    assert abs(float(W.double().sum()) - float(g["bridge_checksum"])) < 1e-9
    tsd = synth.synth_clip_state_dict(tcfg, seed=int(g["teacher_seed"]), gain=3.0)
    cm = synth.synth_cross_modal_state_dict(tcfg.projection_dim, seed=int(g["cm_seed"]))
    ids = T(g["input_ids"])
    B = ids.shape[0]
    regions = synth.synth_regions(B, 2, tcfg.vision, seed=int(g["regions_seed"]))
    with torch.no_grad():
        t_img, t_txt = O.teacher_targets(tsd, tcfg, cm, regions, g["n_regions"], ids, heads=tcfg.projection_dim // 64)
    close(t_img, g["teacher_image_768"], rtol=1e-3, atol=1e-4)
    close(t_txt, g["teacher_text_768"], rtol=1e-3, atol=1e-4)
    del tsd
    ssd = synth.synth_clip_state_dict(scfg, seed=int(g["student_seed"]), gain=3.0)
    assert abs(wsum(ssd) - float(g["wsum_student"])) < 1e-6 * abs(float(g["wsum_student"]))
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in ssd.items()}
    pix = synth.synth_pixel_values(B, scfg.vision, seed=int(g["pixel_seed"]))
    out = O.distill_step_bridged(p, scfg, pix, ids, t_img, t_txt, W)
    close(out["image_emb"], g["image_emb"], rtol=1e-3, atol=1e-4)
    close(out["text_emb"], g["text_emb"], rtol=1e-3, atol=1e-4)
    for k in ("loss_image", "loss_text", "loss_contrastive", "loss"):
        close(out[k], g[k], rtol=1e-4)
    keys = [k for k in p if k != "logit_scale"]
    grads = torch.autograd.grad(out["loss"], [p[k] for k in keys], allow_unused=True)
    for k, gr in zip(keys, grads):
        gr = torch.zeros_like(p[k]) if gr is None else gr
        check_probe(gr, k, g[f"gradprobe.{k}"], rtol=2e-3)
