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


def test_reference_batch_format_end_to_end(tmp_path):
    """The reference's own data path: JSON + image files + pickled box cache -> MultiModalDataset -> DataLoader with
    custom_collate_fn -> (images, captions, paths, boxes) tuples -> training_step -> Trainer.fit; and the GpuCollate
    variant of the same batch gives the same loss."""
    import json, pickle
    from PIL import Image
    from dclip_amd import data
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.lightning_lite import Trainer
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    dev = torch.device("cuda:0")
    cfg = dcfg.tiny(image_size=64, patch_size=16)
    T = cfg.text.max_position_embeddings

    class ToyTok:                                                   # no BPE vocabulary offline
        def __call__(self, text, return_tensors="pt", padding=True, truncation=True, max_length=77, **kw):
            caps = [text] if isinstance(text, str) else list(text)
            rows = [[cfg.text.bos_token_id] + [1 + (sum(map(ord, w)) % (cfg.text.bos_token_id - 2)) for w in c.split()][:T - 2]
                    + [cfg.text.eos_token_id] for c in caps]
            L = max(len(r) for r in rows)                           # padding=True: to the longest caption of the batch
            ids = torch.full((len(rows), L), cfg.text.eos_token_id, dtype=torch.int64)
            for b, r in enumerate(rows):
                ids[b, :len(r)] = torch.tensor(r)
            return type("Enc", (dict,), {"input_ids": property(lambda s: s["input_ids"])})(input_ids=ids)

    tok = ToyTok()
    pre = data.ClipImagePreprocess(size=64, tokenizer=tok)
    recs, cache = [], {}
    rs = __import__("numpy").random.RandomState(0)
    for i in range(8):
        p = tmp_path / f"im{i}.png"
        h, w = 70 + 5 * i, 90 + 3 * i
        Image.fromarray(synth.synth_photo(h, w, seed=40 + i)).save(p)
        recs.append({"image_path": str(p), "captions": [f"photo of thing number {i} in a field", f"thing {i}"]})
        cache[str(p)] = [((int(rs.randint(0, 30)), int(rs.randint(0, 20)), int(rs.randint(40, w)), int(rs.randint(30, h))), 0.9)
                         for _ in range(i % 3)]
    (tmp_path / "train.json").write_text(json.dumps(recs))
    (tmp_path / "cache").mkdir()
    with open(tmp_path / "cache" / "train_precache.pkl", "wb") as f:
        pickle.dump(cache, f, protocol=4)

    def module():
        clip = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=7, gain=3.0), device=dev)
        teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=1, clip_model=clip, tokenizer=tok).to(dev)
        hp = argparse.Namespace(train_file=str(tmp_path / "train.json"), val_file=None, train_batch_size=4, eval_batch_size=4,
                                learning_rate=1e-3, warmup_steps=0, total_steps=100, cache_dir=str(tmp_path / "cache"))
        return CLIPImageDistillation(hp, clip, pre, teacher=teacher).to(dev)

    import random
    m = module()
    random.seed(5)
    loader = m.train_dataloader()
    batch = next(iter(loader))
    assert isinstance(batch, tuple) and batch[0].shape == (4, 3, 64, 64) and len(batch[3]) == 4
    loss_host = m.training_step(batch)
    assert bool(torch.isfinite(loss_host))
    # same four samples through the GPU collate: decode only on the host, preprocessing + crops from one upload
    ds = data.MultiModalDataset(str(tmp_path / "train.json"), pre, cache_dir=str(tmp_path / "cache"), decode_only=True)
    idx = [next(i for i, r in enumerate(recs) if r["image_path"] == p) for p in batch[2]]
    items = []
    for i, cap in zip(idx, batch[1]):
        arr, _, path, boxes = ds[i]
        items.append((arr, cap, path, boxes))                      # keep the caption the host loader drew
    gb = data.GpuCollate(dev, size=64)(items)
    loss_gpu = m.training_step(gb)
    assert torch.equal(loss_host.detach(), loss_gpu.detach())
    # and the loop itself: two optimiser steps over the loader change the trainable parameters
    before = m.student.visual_projection.weight.detach().clone()
    Trainer(max_epochs=1, gradient_clip_val=0.5, accumulate_grad_batches=1).fit(m)
    assert not torch.equal(before, m.student.visual_projection.weight.detach())
    # decode in worker processes, everything else on the GPU
    m.hparams.gpu_preprocess, m.hparams.num_workers = True, 2
    gl = m.train_dataloader()
    assert len(gl) == 2
    seen = 0
    for gbatch in gl:
        assert gbatch["pixel_values"].is_cuda and gbatch["pixel_values"].shape[1:] == (3, 64, 64)
        assert bool(torch.isfinite(m.training_step(gbatch)))
        seen += gbatch["pixel_values"].shape[0]
    assert seen == 8


def test_teacher_trainer_on_reference_batches(tmp_path):
    """train_contrastive_teacher.main over (images, captions, paths, boxes) tuples built from a JSON file with image
    files: the path-based teacher + the shared text pass; checkpoints written, loss finite."""
    import json
    from PIL import Image
    from dclip_amd import train_contrastive_teacher as T
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    dev = torch.device("cuda:0")
    cfg = dcfg.tiny(image_size=64, patch_size=16)
    from dclip_amd.clip_model import from_hf_state_dict
    clip = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=7, gain=4.0), device=dev)
    Tm = cfg.text.max_position_embeddings

    class Tok:
        def __call__(self, text, **kw):
            caps = [text] if isinstance(text, str) else list(text)
            rows = [[cfg.text.bos_token_id] + [1 + (sum(map(ord, w)) % (cfg.text.bos_token_id - 2)) for w in c.split()][:Tm - 2]
                    + [cfg.text.eos_token_id] for c in caps]
            L = max(len(r) for r in rows)
            ids = torch.full((len(rows), L), cfg.text.eos_token_id, dtype=torch.int64)
            for b, r in enumerate(rows):
                ids[b, :len(r)] = torch.tensor(r)
            return type("Enc", (), {"input_ids": ids})()

    recs = []
    for i in range(6):
        p = tmp_path / f"t{i}.png"
        Image.fromarray(synth.synth_photo(72 + 4 * i, 96, seed=60 + i)).save(p)
        recs.append({"image_path": str(p), "captions": [f"a picture of item {i} on a table"],
                     "boxes": [[[2 * i, 3, 50 + i, 40 + i], 0.8]] * (i % 3)})
    (tmp_path / "train.json").write_text(json.dumps(recs))
    teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=1, clip_model=clip, tokenizer=Tok()).to(dev)
    args = argparse.Namespace(train_file=str(tmp_path / "train.json"), val_file=str(tmp_path / "train.json"), batch_size=3,
                              gradient_accumulation=8, learning_rate=1e-3, epochs=2,
                              output_path=str(tmp_path / "out" / "teacher.pth"))
    res = T.main(args, teacher=teacher)
    assert all(torch.isfinite(torch.tensor(h)).all() for h in res["history"])
    assert os.path.exists(args.output_path)
    sd = torch.load(args.output_path, weights_only=True)
    assert len(sd) == 12


def test_default_teacher_targets_do_not_move_with_the_student():
    """with teacher=None the meta-teacher must not encode regions with the TRAINING student tower.
    The reference teacher owns separate, never-updated CLIP instances (training/image_tokenizer.py:25,
    training/text_tokenizer.py:21): the teacher embedding of a fixed batch is bit-identical before and after updates."""
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    cfg, clip = _clip(dev)
    hp = argparse.Namespace(learning_rate=5e-3, warmup_steps=0, total_steps=100, train_batch_size=6, eval_batch_size=6)
    for mode in ("north_star", "as_written"):
        cfg, clip = _clip(dev)
        mod = CLIPImageDistillation(hp, clip, None, freeze_mode=mode).to(dev)
        assert mod.teacher._clip is not mod.student
        batch = {"pixel_values": synth.synth_pixel_values(6, cfg.vision, seed=0).to(dev),
                 "input_ids": synth.synth_input_ids(6, cfg.text, seed=3, ragged=True).to(dev),
                 "regions": synth.synth_regions(6, 2, cfg.vision, seed=4).to(dev)}
        with torch.no_grad():
            before = mod.teacher.compute_global_embedding_tensors(batch["regions"], batch["input_ids"]).clone()
            sent_before = mod.teacher.last_sentence_embedding.clone()
        w0 = mod.student.vision_model.encoder.layers[0].self_attn.qkv_proj.weight.detach().clone()
        opt = FusedAdamW([p for p in mod.parameters() if p.requires_grad], lr=5e-3, max_grad_norm=0.5)
        for _ in range(3):
            mod.training_step(batch).backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
        assert not torch.equal(mod.student.vision_model.encoder.layers[0].self_attn.qkv_proj.weight.detach(), w0)
        with torch.no_grad():
            after = mod.teacher.compute_global_embedding_tensors(batch["regions"], batch["input_ids"])
        assert torch.equal(before, after), mode
        assert torch.equal(sent_before, mod.teacher.last_sentence_embedding), mode
