"""CPU-only host logic: module structure, HF-keyed state dicts, freeze rules, checkpoint layout."""
import argparse
import os

import pytest
import torch

from dclip_amd import config as dcfg, synth
from dclip_amd.clip_model import HipCLIPModel, from_hf_state_dict
from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
from dclip_amd.patch_text_aggregation import CrossModalAttention, PatchTextAggregation
from dclip_amd import lightning_lite


def test_state_dict_has_hf_keys_and_roundtrips():
    cfg = dcfg.tiny()
    sd = synth.synth_clip_state_dict(cfg, seed=7)
    m = from_hf_state_dict(cfg, sd)
    out = m.state_dict()
    assert sorted(out) == sorted(sd), "state_dict must expose separate q_proj/k_proj/v_proj keys (HF layout)"
    for k in sd:
        assert torch.equal(out[k], sd[k]), k
    # in memory q/k/v are fused
    a = m.vision_model.encoder.layers[0].self_attn
    assert a.qkv_proj.weight.shape == (3 * cfg.vision.hidden_size, cfg.vision.hidden_size)
    assert torch.equal(a.qkv_proj.weight[cfg.vision.hidden_size:2 * cfg.vision.hidden_size],
                       sd["vision_model.encoder.layers.0.self_attn.k_proj.weight"])


def test_b32_key_inventory_matches_survey():
    """398 student tensors for ViT-B/32 under HF key names."""
    m = HipCLIPModel(dcfg.vit_b32())
    keys = list(m.state_dict())
    assert len(keys) == 398
    assert "vision_model.pre_layrnorm.weight" in keys                      # sic
    assert m.state_dict()["vision_model.embeddings.patch_embedding.weight"].shape == (768, 3, 32, 32)
    assert m.state_dict()["text_model.embeddings.token_embedding.weight"].shape == (49408, 512)


def _module(freeze_mode):
    cfg = dcfg.tiny()
    student = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=7))
    teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=1, clip_model=student)
    hp = argparse.Namespace(learning_rate=1e-3, warmup_steps=2, total_steps=10, train_batch_size=4, eval_batch_size=4)
    return CLIPImageDistillation(hp, student, None, teacher=teacher, freeze_mode=freeze_mode)


def test_freeze_rule_as_written_matches_reference_selection():
    """training/CLIP_image_distillation.py:504-506 leaves exactly the q/k/v/out projections of the vision tower
    trainable (+ everything outside vision_model): SURVEY N1/N2."""
    mod = _module("as_written")
    vis = {n: p.requires_grad for n, p in mod.student.vision_model.named_parameters()}
    assert all(("proj" in n) == g for n, g in vis.items())
    trainable = [n for n, g in vis.items() if g]
    assert all(("qkv_proj" in n) or ("out_proj" in n) for n in trainable) and trainable
    assert all(p.requires_grad for p in mod.student.text_model.parameters())
    assert mod.student.visual_projection.weight.requires_grad and mod.student.text_projection.weight.requires_grad
    assert not any(p.requires_grad for p in mod.teacher.parameters())
    # parameter count of the trainable vision set at real size (28.35 M)
    big = HipCLIPModel(dcfg.vit_b32())
    n = sum(p.numel() for k, p in big.vision_model.named_parameters() if "proj" in k)
    assert n == 12 * (4 * 768 * 768 + 4 * 768)


def test_freeze_rule_north_star():
    mod = _module("north_star")
    assert all(p.requires_grad for p in mod.student.vision_model.parameters())
    assert not any(p.requires_grad for p in mod.student.text_model.parameters())
    assert mod.teacher.shares_text_tower_with(mod.student)


def test_teacher_checkpoint_is_exactly_the_12_tensors(tmp_path):
    mod = _module("north_star")
    sd = mod.teacher.state_dict()
    want = {f"cross_modal_attention.{d}.{k}" for d in ("text_to_image", "image_to_text")
            for k in ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias")}
    want |= {f"cross_modal_attention.{n}.{k}" for n in ("norm_text", "norm_image") for k in ("weight", "bias")}
    assert set(sd) == want
    E = mod.teacher.embed_dim
    assert sd["cross_modal_attention.text_to_image.in_proj_weight"].shape == (3 * E, E)
    path = tmp_path / "teacher.pth"
    torch.save(sd, path)
    other = _module("north_star")
    other.teacher.load_state_dict(torch.load(path, weights_only=True), strict=False)
    # torch's own nn.MultiheadAttention accepts the same keys
    ref = torch.nn.MultiheadAttention(E, 1)
    ref.load_state_dict({k.split("text_to_image.")[1]: v for k, v in sd.items() if "text_to_image" in k})


def test_student_checkpoint_layout_and_reload(tmp_path):
    mod = _module("north_star")
    path = lightning_lite.save_checkpoint(str(tmp_path / lightning_lite.checkpoint_filename(1, 3.014)), mod, epoch=1,
                                          global_step=7)
    assert os.path.basename(path) == "epoch-epoch=01-train_loss=3.01.ckpt"      # eval_scripts/flickr30k_eval.py:113
    ckpt = torch.load(path, weights_only=True)
    for k in ("state_dict", "epoch", "global_step", "hyper_parameters", "optimizer_states", "lr_schedulers",
              "callbacks", "pytorch-lightning_version"):
        assert k in ckpt
    keys = list(ckpt["state_dict"])
    assert all(k.startswith("student.") or k.startswith("teacher.cross_modal_attention.") for k in keys)
    assert sum(k.startswith("teacher.") for k in keys) == 12
    assert "student.vision_model.encoder.layers.0.self_attn.q_proj.weight" in keys
    cfg = dcfg.tiny()
    fresh = HipCLIPModel(cfg)
    re = CLIPImageDistillation.load_from_checkpoint(
        path, map_location="cpu", clip_model=fresh, clip_preprocess=None, strict=False,
        teacher=PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=1, clip_model=fresh))
    for (k, a), (_, b) in zip(mod.state_dict().items(), re.state_dict().items()):
        assert torch.equal(a, b), k


def test_optimizer_and_schedule():
    mod = _module("north_star")
    (opt,), (sched,) = mod.configure_optimizers()
    assert isinstance(opt, torch.optim.AdamW)
    lrs = []
    for _ in range(10):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    assert lrs[0] == 0.0 and abs(lrs[2] - 1e-3) < 1e-12 and lrs[-1] < lrs[3]     # linear warmup then decay


def test_argparse_surface():
    p = argparse.ArgumentParser()
    CLIPImageDistillation.add_model_specific_args(p)
    a = p.parse_args(["--train_file", "x.json"])
    assert (a.train_batch_size, a.eval_batch_size, a.learning_rate, a.warmup_steps, a.total_steps) == \
        (32, 32, 2e-5, 0, 1000)


def test_towers_fail_loudly_without_gpu():
    cfg = dcfg.tiny()
    m = HipCLIPModel(cfg)
    with pytest.raises((ValueError, RuntimeError)):
        m.get_image_features(pixel_values=torch.zeros(1, 3, cfg.vision.image_size, cfg.vision.image_size))


class _ToyModule(lightning_lite.LightningLikeModule):
    """Host-logic stand-in for the Trainer tests: loss = w * x, so each micro-batch contributes x to w.grad."""

    def __init__(self, total_steps=1000):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(()))
        self.total = total_steps

    def training_step(self, batch, batch_idx=0):
        return self.w * batch

    def configure_optimizers(self):
        opt = torch.optim.SGD([self.w], lr=1.0)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: max(0.0, (self.total - s) / self.total))
        return [opt], [sched]


def test_trainer_steps_on_trailing_micro_batches_and_schedules_per_epoch():
    """Lightning semantics the reference relies on (training/CLIP_image_distill_training.py:36-45,
    training/CLIP_image_distillation.py:679-682): accumulate 4 with a step on the LAST batch of an epoch, and a bare
    scheduler advances once per EPOCH."""
    mod = _ToyModule(total_steps=1000)
    batches = [torch.tensor(float(v)) for v in (1, 2, 3, 4, 5, 6)]          # 6 batches: groups [1..4] and [5, 6]
    tr = lightning_lite.Trainer(max_epochs=2, gradient_clip_val=None, accumulate_grad_batches=4)
    tr.fit(mod, batches, None)
    lr0, lr1 = 1.0, 0.999                                                   # epoch 0, epoch 1 (one scheduler tick)
    want = -(lr0 * (10 / 4 + 11 / 4) + lr1 * (10 / 4 + 11 / 4))
    assert abs(float(mod.w) - want) < 1e-6, (float(mod.w), want)
    assert mod.global_step == 4
    # per-step interval on request: the LR decays with every optimizer step
    mod2 = _ToyModule(total_steps=4)
    lightning_lite.Trainer(max_epochs=1, gradient_clip_val=None, accumulate_grad_batches=1,
                           lr_interval="step").fit(mod2, batches[:4], None)
    assert abs(float(mod2.w) + (1 * 1.0 + 2 * 0.75 + 3 * 0.5 + 4 * 0.25)) < 1e-6


def test_default_teacher_is_a_frozen_snapshot_of_the_student():
    """The reference teacher owns separate CLIP instances that never train (training/image_tokenizer.py:25,
    training/text_tokenizer.py:21): the default teacher must not alias the (training) student towers."""
    cfg = dcfg.tiny()
    student = HipCLIPModel(cfg)
    hp = argparse.Namespace(learning_rate=1e-3, warmup_steps=0, total_steps=10, train_batch_size=2, eval_batch_size=2)
    mod = CLIPImageDistillation(hp, student, None, freeze_mode="north_star")
    tclip = mod.teacher._clip
    assert tclip is not student
    assert not any(p.requires_grad for p in tclip.parameters())
    tw = tclip.vision_model.encoder.layers[0].mlp.fc1.weight
    sw = student.vision_model.encoder.layers[0].mlp.fc1.weight
    assert tw.data_ptr() != sw.data_ptr() and torch.equal(tw, sw)
    assert not any(k.startswith("teacher._clip") or "vision_model" in k for k in mod.teacher.state_dict())
    assert mod.teacher.shares_text_tower_with(student)          # frozen text tower, untouched since the snapshot
    with torch.no_grad():
        student.text_projection.weight.add_(1.0)                # an in-place update of the student's text weights ...
    assert not mod.teacher.shares_text_tower_with(student)      # ... ends the sharing (teacher keeps its own copy)
    mod.set_freeze_mode("as_written")
    assert not mod.teacher.shares_text_tower_with(student)
