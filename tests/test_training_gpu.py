"""End-to-end on the GPU: the two entry points train, write the reference's checkpoint files, and the files reload."""
import argparse
import glob
import os

import pytest
import torch

from dclip_amd import config as dcfg, synth

pytestmark = pytest.mark.gpu


def _clip(dev):
    from dclip_amd.clip_model import from_hf_state_dict
    cfg = dcfg.tiny()
    return cfg, from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=7, gain=4.0), device=dev)


def test_train_contrastive_teacher_entry_point(tmp_path):
    from dclip_amd import train_contrastive_teacher as T
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    dev = torch.device("cuda:0")
    cfg, clip = _clip(dev)
    teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=1, clip_model=clip).to(dev)
    before = {k: v.clone() for k, v in teacher.state_dict().items()}
    clip_before = clip.visual_projection.weight.detach().clone()

    def batches(seed0, n):
        return [{"regions": synth.synth_regions(8, 3, cfg.vision, seed=seed0 + i),
                 "input_ids": synth.synth_input_ids(8, cfg.text, seed=50 + seed0 + i, ragged=True, min_len=4),
                 "region_counts": torch.tensor([3, 2, 1, 3, 0, 2, 3, 1])} for i in range(n)]

    args = argparse.Namespace(train_file=None, val_file=None, batch_size=8, gradient_accumulation=8, learning_rate=2e-3,
                              epochs=3, output_path=str(tmp_path / "teacher" / "contrastive_teacher.pth"))
    res = T.main(args, teacher=teacher, train_batches=batches(0, 4), val_batches=batches(100, 2))
    hist = res["history"]
    assert hist[-1][0] < hist[0][0], hist                              # the 12 tensors learn
    files = sorted(os.path.basename(f) for f in glob.glob(str(tmp_path / "teacher" / "*.pth")))
    assert "contrastive_teacher.pth" in files
    assert sum(f.startswith("contrastive_teacher_epoch") and "_val" in f for f in files) == 3
    sd = torch.load(args.output_path, weights_only=True)
    assert len(sd) == 12 and all(k.startswith("cross_modal_attention.") for k in sd)
    assert any(not torch.equal(sd[k].cpu(), before[k].cpu()) for k in sd)
    assert torch.equal(clip.visual_projection.weight.detach(), clip_before)      # the CLIP towers stay frozen
    # the student module loads it the way the reference does (strict=False)
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    hp = argparse.Namespace(learning_rate=1e-4, warmup_steps=0, total_steps=10, train_batch_size=8, eval_batch_size=8)
    t2 = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=1, clip_model=clip).to(dev)
    mod = CLIPImageDistillation(hp, clip, None, teacher=t2, contrastive_teacher_path=args.output_path)
    for k, v in sd.items():
        assert torch.equal(mod.teacher.state_dict()[k].cpu(), v.cpu())


def test_student_launcher_trains_and_checkpoints(tmp_path):
    from dclip_amd import CLIP_image_distill_training as L
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    from dclip_amd.clip_model import HipCLIPModel
    dev = torch.device("cuda:0")
    cfg, clip = _clip(dev)
    teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=1, clip_model=clip).to(dev)
    batch = {"pixel_values": synth.synth_pixel_values(8, cfg.vision, seed=0),
             "input_ids": synth.synth_input_ids(8, cfg.text, seed=3, ragged=True),
             "regions": synth.synth_regions(8, 2, cfg.vision, seed=4)}
    args = argparse.Namespace(train_file="x", val_file=None, train_batch_size=8, eval_batch_size=8, learning_rate=3e-4,
                              warmup_steps=1, total_steps=50, checkpoint_dir=str(tmp_path / "ckpt"), phase1_epochs=2)
    model, trainer = L.main(args, clip_model=clip, train_batches=[batch] * 16, val_batches=[batch], teacher=teacher)
    first = float(trainer.saved[-1][0]) if len(trainer.saved) > 1 else None
    files = sorted(glob.glob(str(tmp_path / "ckpt" / "*.ckpt")))
    assert len(files) == 2 and all("epoch-epoch=0" in os.path.basename(f) and "train_loss=" in f for f in files)
    losses = sorted(l for l, _ in trainer.saved)
    assert losses[0] < 3.0                                         # it learns to imitate the (fixed) teacher batch
    fresh = HipCLIPModel(cfg).to(dev)
    re = CLIPImageDistillation.load_from_checkpoint(
        files[-1], map_location="cpu", clip_model=fresh, clip_preprocess=None, strict=False,
        teacher=PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=1, clip_model=fresh))
    a = model.student.state_dict()["visual_projection.weight"].cpu()
    assert torch.equal(re.student.state_dict()["visual_projection.weight"].cpu(), a)
    assert model.logged("val_loss") > 0


def test_graphed_step_replays_the_eager_step():
    """hipGraph replay of forward + backward: same loss and gradients as the eager step, on new data each replay."""
    import argparse
    from dclip_amd import config as dcfg, synth
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.graph import GraphedStep
    dev = torch.device("cuda:0")
    cfg = dcfg.tiny()
    B = 6

    def make():
        student = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0), device=dev)
        hp = argparse.Namespace(learning_rate=1e-4, warmup_steps=0, total_steps=100, train_batch_size=B, eval_batch_size=B)
        return CLIPImageDistillation(hp, student, None, freeze_mode="north_star").to(dev)

    def batch(seed):
        return {"pixel_values": synth.synth_pixel_values(B, cfg.vision, seed=seed).to(dev),
                "input_ids": synth.synth_input_ids(B, cfg.text, seed=seed + 1, ragged=True).to(dev),
                "teacher_image_emb": synth.synth_embeddings(B, cfg.projection_dim, seed=seed + 2).to(dev)}

    eager, graphed = make(), make()
    g = GraphedStep(graphed, batch(10))
    names = [n for n, p in eager.named_parameters() if p.requires_grad]
    for seed in (20, 30, 40):
        for p in eager.parameters():
            p.grad = None
        le = eager.training_step(batch(seed))
        le.backward()
        lg = g.step(batch(seed))
        assert torch.equal(le.detach(), lg.detach()), (float(le), float(lg))
        ge = dict(eager.named_parameters())
        gg = dict(graphed.named_parameters())
        for n in names:
            assert torch.equal(ge[n].grad, gg[n].grad), n
    with pytest.raises(ValueError):
        g.step({"pixel_values": torch.zeros(2, 3, cfg.vision.image_size, cfg.vision.image_size, device=dev)})


def test_trainer_with_hip_graph_matches_eager_trainer():
    """Trainer(use_hip_graph=True) with gradient accumulation 2 and clipping: same parameters after 4 optimizer steps
    as the eager Trainer (the only difference is the association of the accumulated sum: g1/2 + g2/2 either way)."""
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.lightning_lite import Trainer
    dev = torch.device("cuda:0")
    cfg = dcfg.tiny()
    B = 4

    def run(use_graph):
        student = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0), device=dev)
        hp = argparse.Namespace(learning_rate=1e-3, warmup_steps=0, total_steps=100, train_batch_size=B, eval_batch_size=B)
        m = CLIPImageDistillation(hp, student, None, freeze_mode="north_star").to(dev)
        batches = [{"pixel_values": synth.synth_pixel_values(B, cfg.vision, seed=s),
                    "input_ids": synth.synth_input_ids(B, cfg.text, seed=s + 1, ragged=True),
                    "teacher_image_emb": synth.synth_embeddings(B, cfg.projection_dim, seed=s + 2)} for s in range(0, 80, 10)]
        Trainer(max_epochs=1, gradient_clip_val=0.5, accumulate_grad_batches=2, use_hip_graph=use_graph).fit(m, batches)
        return {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}

    a, b = run(False), run(True)
    assert a.keys() == b.keys()
    for n in a:
        assert torch.allclose(a[n], b[n], rtol=1e-5, atol=1e-7), (n, float((a[n] - b[n]).abs().max()))
