"""GPU parity of the towers, the meta-teacher and the full distillation step against the golden vectors
(generated from the reference's own code) and against the CPU oracle."""
import argparse

import numpy as np
import pytest
import torch

from dclip_amd import config as dcfg, synth
from dclip_amd.probe import probe_vector
from oracle import dclip_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


def relerr(got, want):
    got = got.detach().double().cpu()
    want = want.detach().cpu().double() if isinstance(want, torch.Tensor) else torch.as_tensor(np.asarray(want)).double()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))


def hf_named_grads(model):
    """HF key -> gradient, splitting the in-memory fused qkv parameter."""
    out = {}
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        if "qkv_proj" in k:
            D = g.shape[0] // 3
            for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
                out[k.replace("qkv_proj", n)] = g[i * D:(i + 1) * D]
        else:
            out[k] = g
    return out


def check_probe(g, key, expect, rtol, atol=2e-5):
    g = g.detach().double().cpu().reshape(-1)
    got = np.array([float(g.norm()), float(g @ probe_vector(key, g.numel()).double())])
    scale = max(abs(expect[0]), 1e-12)
    assert abs(got[0] - expect[0]) <= rtol * scale + atol, (key, got, expect)
    assert abs(got[1] - expect[1]) <= rtol * scale * 10 + atol, (key, got, expect)


def make_model(cfg, sd, dev):
    from dclip_amd.clip_model import from_hf_state_dict
    return from_hf_state_dict(cfg, sd, device=dev)


# ------------------------------------------------------------------------------------------ towers (F3)

def test_towers_tiny_forward(dev, golden):
    g = golden("towers_tiny.npz")
    cfg = dcfg.tiny()
    m = make_model(cfg, synth.synth_clip_state_dict(cfg, seed=7, gain=4.0), dev)
    pix, ids = T(g["pixel_values"]).to(dev), T(g["input_ids"]).to(dev)
    vh = m.hidden_states(pixel_values=pix)
    for i, h in enumerate(vh):
        assert relerr(h, g[f"f32.vision_hidden.{i}"]) < 1e-4, i
        assert relerr(h, g[f"f64.vision_hidden.{i}"]) < 1e-4, i
    th = m.hidden_states(input_ids=ids)
    for i, h in enumerate(th):
        assert relerr(h, g[f"f32.text_hidden.{i}"]) < 1e-4, i
    with torch.no_grad():
        img = m.get_image_features(pixel_values=pix)
        txt = m.get_text_features(input_ids=ids, attention_mask=torch.ones_like(ids))
    assert relerr(img, g["f32.image_emb"]) < 1e-3 and relerr(img, g["f64.image_emb"]) < 1e-3
    assert relerr(txt, g["f32.text_emb"]) < 1e-3 and relerr(txt, g["f64.text_emb"]) < 1e-3
    sent, tokens, eos = m.text_token_level(ids)
    assert relerr(sent, g["f32.text_emb"]) < 1e-3
    lh = T(g["f32.text_last_hidden"])
    want_tok = lh @ m.text_projection.weight.detach().cpu().t()
    assert relerr(tokens, want_tok) < 1e-3


def test_towers_tiny_param_grads(dev, golden):
    g = golden("towers_tiny.npz")
    cfg = dcfg.tiny()
    m = make_model(cfg, synth.synth_clip_state_dict(cfg, seed=7, gain=4.0), dev)
    pix, ids = T(g["pixel_values"]).to(dev), T(g["input_ids"]).to(dev)
    img = m.get_image_features(pixel_values=pix)
    txt = m.get_text_features(input_ids=ids)
    obj = (img * T(g["obj_w_img"]).to(dev)).sum() + (txt * T(g["obj_w_txt"]).to(dev)).sum()
    obj.backward()
    grads = hf_named_grads(m)
    n_full = 0
    for k, gr in grads.items():
        if k == "logit_scale":
            continue
        check_probe(gr, k, g[f"gradprobe.{k}"], rtol=2e-3)
        if f"grad.{k}" in g.files and np.abs(g[f"grad.{k}"]).max() > 1e-4:
            assert relerr(gr.reshape(g[f"grad.{k}"].shape), g[f"grad.{k}"]) < 1e-3, k
            n_full += 1
    assert n_full > 10


@pytest.mark.parametrize("name,mk,seed", [("b32", dcfg.vit_b32, 0), ("b16", dcfg.vit_b16, 1)])
def test_towers_real_forward(dev, golden, name, mk, seed):
    g = golden("towers_real.npz")
    cfg = mk()
    m = make_model(cfg, synth.synth_clip_state_dict(cfg, seed=seed, gain=3.0), dev)
    pix = synth.synth_pixel_values(2, cfg.vision, seed=0).to(dev)
    ids = T(g[f"{name}.input_ids"]).to(dev)
    with torch.no_grad():
        img = m.get_image_features(pixel_values=pix)
        txt = m.get_text_features(input_ids=ids)
    assert relerr(img, g[f"{name}.image_emb"]) < 1e-3      # BASELINE parity gate: 1e-3 relative on embeddings
    assert relerr(txt, g[f"{name}.text_emb"]) < 1e-3
    vh = m.hidden_states(pixel_values=pix)
    stats = np.array([[float(h.double().mean()), float(h.double().std()), float(h.double().abs().max())] for h in vh])
    np.testing.assert_allclose(stats, g[f"{name}.vision_layer_stats"], rtol=1e-3, atol=1e-5)
    assert relerr(vh[-1][:, 0, :], g[f"{name}.vision_cls_last"]) < 1e-3


# ------------------------------------------------------------------------------------------ full step (F4, config c1)

def test_step_c1_losses_and_grads(dev, golden):
    """BASELINE config c1: B/32 + text tower, bs=8, against the reference's step arithmetic."""
    from dclip_amd.CLIP_image_distillation import distill_losses
    g = golden("step_c1.npz")
    cfg = dcfg.vit_b32()
    m = make_model(cfg, synth.synth_clip_state_dict(cfg, seed=0, gain=3.0), dev)
    B = 8
    pix = synth.synth_pixel_values(B, cfg.vision, seed=0).to(dev)
    ids = T(g["input_ids"]).to(dev)
    t_img = synth.synth_embeddings(B, cfg.projection_dim, seed=1).to(dev)
    t_txt = synth.synth_embeddings(B, cfg.projection_dim, seed=5).to(dev)
    img = m.get_image_features(pixel_values=pix)
    txt = m.get_text_features(input_ids=ids)
    assert relerr(img, g["image_emb"]) < 1e-3 and relerr(txt, g["text_emb"]) < 1e-3
    out = distill_losses(img, txt, t_img, t_txt)
    for k in ("loss_image", "loss_text", "loss_contrastive", "loss"):
        assert abs(float(out[k].detach()) - float(g[k])) <= 1e-3 * abs(float(g[k])), (k, float(out[k].detach()), float(g[k]))
    out["loss"].backward()
    grads = hf_named_grads(m)
    worst = 0.0
    for k, gr in grads.items():
        if k == "logit_scale":
            continue
        check_probe(gr, k, g[f"gradprobe.{k}"], rtol=5e-3)
    # the oracle's full gradients: cosine >= 0.9999 per tensor (SURVEY §8d parity gate)
    sd = synth.synth_clip_state_dict(cfg, seed=0, gain=3.0)
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref = O.distill_step(p, cfg, pix.cpu(), ids.cpu(), t_img.cpu(), t_txt.cpu())
    ref["loss"].backward()
    for k, gr in grads.items():
        if k == "logit_scale":
            continue
        w = p[k].grad
        if float(w.norm()) < 1e-5:
            continue
        cos = float((gr.cpu().double().reshape(-1) @ w.double().reshape(-1)) / (gr.cpu().double().norm() * w.double().norm()))
        worst = max(worst, 1 - cos)
        assert cos > 0.9999, (k, cos)
    print("worst 1-cos", worst)


def _distill_module(cfg, sd, dev, freeze_mode):
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    student = make_model(cfg, sd, dev)
    teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=cfg.projection_dim // 64, clip_model=student)
    teacher.load_state_dict({f"cross_modal_attention.{k}": v for k, v in
                             synth.synth_cross_modal_state_dict(cfg.projection_dim, seed=31).items()})
    hp = argparse.Namespace(learning_rate=1e-4, warmup_steps=0, total_steps=100, train_batch_size=4, eval_batch_size=4)
    return CLIPImageDistillation(hp, student, None, teacher=teacher.to(dev), freeze_mode=freeze_mode).to(dev)


@pytest.mark.parametrize("freeze_mode", ["north_star", "as_written"])
def test_training_step_module_freeze_modes(dev, freeze_mode):
    """training_step on the tensor batch format; the set of tensors that receive gradients follows the freeze mode,
    and every gradient that exists equals the oracle's."""
    cfg = dcfg.tiny()
    sd = synth.synth_clip_state_dict(cfg, seed=7, gain=4.0)
    mod = _distill_module(cfg, sd, dev, freeze_mode)
    B = 6
    pix = synth.synth_pixel_values(B, cfg.vision, seed=0)
    ids = synth.synth_input_ids(B, cfg.text, seed=3, ragged=True)
    t_img = synth.synth_embeddings(B, cfg.projection_dim, seed=1)
    t_txt = synth.synth_embeddings(B, cfg.projection_dim, seed=5)
    batch = {"pixel_values": pix, "input_ids": ids, "teacher_image_emb": t_img, "teacher_text_emb": t_txt}
    loss = mod.training_step(batch)
    loss.backward()
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    ref = O.distill_step(p, cfg, pix, ids, t_img, t_txt)
    ref["loss"].backward()
    assert abs(float(loss) - float(ref["loss"])) < 1e-4 * abs(float(ref["loss"]))
    assert abs(mod.logged("train_loss") - float(ref["loss"])) < 1e-4 * abs(float(ref["loss"]))
    got = {k: prm for k, prm in mod.student.named_parameters()}
    for k, prm in got.items():
        if not prm.requires_grad:
            assert prm.grad is None, k
    grads = hf_named_grads(mod.student)
    for k, prm in got.items():
        if prm.requires_grad and prm.grad is not None and k != "logit_scale":
            names = [k.replace("qkv_proj", n) for n in ("q_proj", "k_proj", "v_proj")] if "qkv_proj" in k else [k]
            for n in names:
                w = p[n].grad
                if float(w.abs().max()) > 1e-5:
                    assert relerr(grads[n], w) < 2e-3, n
    if freeze_mode == "north_star":
        assert got["text_model.final_layer_norm.weight"].grad is None
        assert got["vision_model.encoder.layers.0.mlp.fc1.weight"].grad is not None
    else:
        assert got["vision_model.encoder.layers.0.mlp.fc1.weight"].grad is None
        assert got["vision_model.encoder.layers.0.self_attn.qkv_proj.weight"].grad is not None
        assert got["text_model.embeddings.token_embedding.weight"].grad is not None


def test_shared_text_forward_regime(dev):
    """north_star regime without a given teacher text embedding: L_txt = 1 - cos(x, x) = 0 and has no gradient."""
    cfg = dcfg.tiny()
    mod = _distill_module(cfg, synth.synth_clip_state_dict(cfg, seed=7, gain=4.0), dev, "north_star")
    B = 4
    batch = {"pixel_values": synth.synth_pixel_values(B, cfg.vision), "input_ids": synth.synth_input_ids(B, cfg.text),
             "teacher_image_emb": synth.synth_embeddings(B, cfg.projection_dim)}
    mod.training_step(batch).backward()
    assert abs(float(mod.last_losses["loss_text"])) < 1e-6


# ------------------------------------------------------------------------------------------ meta-teacher (F2, a4-a8)

@pytest.mark.parametrize("name", ["e128", "e512", "e512_c3"])
def test_cross_modal_against_reference(dev, golden, name):
    from dclip_amd.patch_text_aggregation import CrossModalAttention, GlobalPoolFn
    from dclip_amd import functional
    g = golden("cross_modal.npz")
    E, H, seed = int(g[f"{name}.E"]), int(g[f"{name}.H"]), int(g[f"{name}.seed"])
    cm = CrossModalAttention(E, H)
    cm.load_state_dict(synth.synth_cross_modal_state_dict(E, seed=seed))
    cm = cm.to(dev)
    text, patches, sent = (T(g[f"{name}.{k}"]).to(dev) for k in ("text", "patches", "sentence"))
    at, ai = cm(text, patches)
    assert relerr(at, g[f"{name}.attended_text"]) < 1e-4
    assert relerr(ai, g[f"{name}.attended_image"]) < 1e-4
    glob = GlobalPoolFn.apply(at, ai, 2.0)
    assert relerr(glob, g[f"{name}.global"]) < 1e-4
    loss = functional.contrastive_loss(glob, sent, 0.05)
    assert abs(float(loss) - float(g[f"{name}.loss"])) < 1e-4 * abs(float(g[f"{name}.loss"]))
    loss.backward()
    for k, prm in cm.named_parameters():
        check_probe(prm.grad, k, g[f"{name}.gradprobe.{k}"], rtol=2e-3)
        if E == 128:
            assert relerr(prm.grad, g[f"{name}.grad.{k}"]) < 1e-3, k


def test_cross_modal_input_grads_and_aggregation(dev):
    """d/d(text, patches) of the block and the standalone `aggregation` method, against the oracle (float64)."""
    from dclip_amd.patch_text_aggregation import CrossModalAttention, AggregationFn
    E, H, B, Tn, R = 128, 2, 3, 7, 4
    sd = synth.synth_cross_modal_state_dict(E, seed=3)
    cm = CrossModalAttention(E, H)
    cm.load_state_dict(sd)
    cm = cm.to(dev)
    gen = torch.Generator().manual_seed(5)
    text, patches = torch.randn(B, Tn, E, generator=gen), torch.randn(B, R, E, generator=gen)
    wt, wi = torch.randn(B, E, generator=gen), torch.randn(B, E, generator=gen)
    td, pd = text.double().requires_grad_(True), patches.double().requires_grad_(True)
    at, ai = O.cross_modal_attention(O.to_dtype(sd, torch.float64), td, pd, H)
    obj = (O.aggregation(at) * wt.double()).sum() + (O.aggregation(ai) * wi.double()).sum()
    obj.backward()
    tg, pg = text.to(dev).requires_grad_(True), patches.to(dev).requires_grad_(True)
    at2, ai2 = cm(tg, pg)
    o2 = (AggregationFn.apply(at2, 2.0) * wt.to(dev)).sum() + (AggregationFn.apply(ai2, 2.0) * wi.to(dev)).sum()
    assert abs(float(o2) - float(obj)) < 1e-4 * abs(float(obj))
    o2.backward()
    assert relerr(tg.grad, td.grad) < 1e-3
    assert relerr(pg.grad, pd.grad) < 1e-3


def test_teacher_glue_against_reference(dev, golden):
    """a4/a5/a8: the reference's compute_global_embedding_batch (ragged regions incl. an image with no boxes, ragged
    captions incl. one without word tokens) vs compute_global_embedding_tensors."""
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    g = golden("teacher_glue.npz")
    cfg = dcfg.tiny()
    clip = make_model(cfg, synth.synth_clip_state_dict(cfg, seed=int(g["clip_seed"]), gain=4.0), dev)
    teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=cfg.projection_dim // 64, clip_model=clip)
    teacher.load_state_dict({f"cross_modal_attention.{k}": v for k, v in
                             synth.synth_cross_modal_state_dict(cfg.projection_dim, seed=int(g["cm_seed"])).items()})
    teacher = teacher.to(dev)
    ids, regions, n_regions = T(g["input_ids"]).to(dev), T(g["regions"]).to(dev), T(g["n_regions"])
    glob = teacher.compute_global_embedding_tensors(regions, ids, n_regions)
    assert relerr(glob, g["global"]) < 1e-3
    sent = teacher.text_tokenizer.aggregate_text_ids(ids)
    assert relerr(sent, g["sentence"]) < 1e-3
    for b in range(ids.shape[0]):
        toks = torch.stack(teacher.text_tokenizer.get_embeddings(ids[b]))
        assert toks.shape[0] == int(g["n_tok"][b])
        assert relerr(toks, g[f"tokens.{b}"]) < 1e-3


def test_teacher_nan_guards_against_reference(dev, golden):
    """training/patch_text_aggregation.py:497-499, :542-544, :649-651 as run by the reference itself: NaN region crop ->
    zero row, NaN token embedding -> that caption's tokens zero, NaN inside the block -> whole batch zero (and no
    gradient)."""
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    g = golden("teacher_guards.npz")
    cfg = dcfg.tiny()
    sd = synth.synth_clip_state_dict(cfg, seed=int(g["clip_seed"]), gain=4.0)
    sd["text_model.embeddings.token_embedding.weight"][int(g["nan_token_id"])] = float("nan")
    clip = make_model(cfg, sd, dev)
    ids, regions, n_regions = T(g["input_ids"]).to(dev), T(g["regions"]).to(dev), T(g["n_regions"])
    cm = synth.synth_cross_modal_state_dict(cfg.projection_dim, seed=int(g["cm_seed"]))

    def teacher_with(weights):
        t = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=cfg.projection_dim // 64, clip_model=clip)
        t.load_state_dict({f"cross_modal_attention.{k}": v for k, v in weights.items()})
        return t.to(dev)

    glob = teacher_with(cm).compute_global_embedding_tensors(regions, ids, n_regions)
    assert bool(torch.isfinite(glob).all()) and relerr(glob, g["global"]) < 1e-3
    cm["norm_text.weight"][3] = float("nan")
    t2 = teacher_with(cm)
    out = t2.compute_global_embedding_tensors(regions, ids, n_regions)
    assert float(out.abs().sum()) == 0.0
    out.sum().backward()
    for n, p in t2.cross_modal_attention.named_parameters():
        assert p.grad is None or float(p.grad.abs().sum()) == 0.0, n       # a replaced batch passes no gradient


def test_teacher_path_based_signature(dev, tmp_path):
    """compute_global_embedding_batch(paths, texts, boxes) == the tensor variant on the crops it cuts."""
    from PIL import Image
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    cfg = dcfg.tiny()
    clip = make_model(cfg, synth.synth_clip_state_dict(cfg, seed=7, gain=4.0), dev)
    teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=1, clip_model=clip).to(dev)
    rng = np.random.default_rng(0)
    paths, boxes = [], []
    for b in range(3):
        p = str(tmp_path / f"{b}.png")
        Image.fromarray(rng.integers(0, 255, (80, 96, 3), dtype=np.uint8)).save(p)
        paths.append(p)
        boxes.append([((4 * r, 2 * r, 50 + 4 * r, 40 + 2 * r), 0.9) for r in range(b)])     # 0, 1, 2 boxes
    ids = synth.synth_input_ids(3, cfg.text, seed=9, ragged=True, min_len=4)
    teacher.text_tokenizer._ids = lambda texts, keep_host=False: ids if keep_host else ids.to(dev)   # no BPE vocab offline
    got = teacher.compute_global_embedding_batch(paths, ["a", "b", "c"], boxes)
    crops = torch.zeros(3, 2, 3, cfg.vision.image_size, cfg.vision.image_size)
    for b in range(3):
        im = Image.open(paths[b]).convert("RGB")
        for r, (box, _) in enumerate(boxes[b]):
            crops[b, r] = teacher.patch_tokenizer.patch_transform(im.crop(box))
    want = teacher.compute_global_embedding_tensors(crops.to(dev), ids.to(dev), torch.tensor([0, 1, 2]))
    assert relerr(got, want.cpu()) < 1e-6
    embs = teacher.patch_tokenizer.encode_weighted_bounding_boxes(Image.open(paths[2]).convert("RGB"), boxes[2])
    assert len(embs) == 2 and embs[0][0].shape == (cfg.projection_dim,) and embs[0][1] == 0.9


def test_one_text_forward_serves_teacher_and_student(dev):
    """Meta-teacher inside the step on the student's own frozen text tower: the sentence embedding is taken from the
    teacher's token-level pass; the losses equal those of the path with a separate student text forward."""
    cfg = dcfg.tiny()
    B = 5
    batch = {"pixel_values": synth.synth_pixel_values(B, cfg.vision, seed=1),
             "input_ids": synth.synth_input_ids(B, cfg.text, seed=2, ragged=True),
             "regions": synth.synth_regions(B, 3, cfg.vision, seed=3), "region_counts": torch.tensor([3, 1, 0, 2, 3])}
    res = {}
    for shared in (True, False):
        mod = _distill_module(cfg, synth.synth_clip_state_dict(cfg, seed=7, gain=4.0), dev, "north_star")
        assert mod.teacher.shares_text_tower_with(mod.student)
        calls = []
        orig = mod.student.get_text_features
        mod.student.get_text_features = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        if not shared:
            mod.teacher.shares_text_tower_with = lambda s: False
        loss = mod.training_step(batch)
        loss.backward()
        res[shared] = (float(loss), {k: float(v) for k, v in mod.last_losses.items()}, len(calls),
                       mod.student.visual_projection.weight.grad.clone())
    assert res[True][2] == 0 and res[False][2] == 1                  # no separate student text forward when shared
    assert abs(res[True][0] - res[False][0]) < 1e-5 * abs(res[False][0])
    for k in res[True][1]:
        assert abs(res[True][1][k] - res[False][1][k]) < 1e-5
    assert relerr(res[True][3], res[False][3]) < 1e-4
