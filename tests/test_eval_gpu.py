"""Retrieval / zero-shot consumers on the GPU vs the reference's own metric function (golden) and torch.topk."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_retrieval_metrics_match_reference(golden):
    from dclip_amd import eval as E
    g = golden("eval_retrieval.npz")
    dev = torch.device("cuda:0")
    img, cap = torch.from_numpy(g["image_emb"]).to(dev), torch.from_numpy(g["caption_emb"]).to(dev)
    Ni, per = img.shape[0], int(g["per"])
    image_ids = [f"img{i}" for i in range(Ni)]
    caption_image_ids = [f"img{i}" for i in range(Ni) for _ in range(per)]
    m = E.calculate_retrieval_metrics(img, cap, image_ids, caption_image_ids)
    for d in ("t2i", "i2t"):
        got = np.array([m[d]["R@1"], m[d]["R@5"], m[d]["R@10"], m[d]["MAP"]])
        np.testing.assert_allclose(got, g[d], rtol=1e-9, atol=1e-9)
    assert 0.05 < m["t2i"]["R@1"] < 0.99                        # a non-trivial case


@pytest.mark.parametrize("Bq,Bk,P", [(37, 1000, 512), (300, 77, 64), (5, 5, 128)])
def test_rank_count_against_argsort(Bq, Bk, P):
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(Bq)
    q = torch.nn.functional.normalize(torch.randn(Bq, P, generator=gen), dim=1)
    c = torch.nn.functional.normalize(torch.randn(Bk, P, generator=gen), dim=1)
    gt = torch.randint(0, Bk, (Bq,), generator=gen)
    s_gt = ops.rowdot_gather(q.to(dev), c.to(dev), gt.to(torch.int32).to(dev))
    got = ops.rank_count(q.to(dev), c.to(dev), s_gt, gt.to(torch.int32).to(dev)).cpu()
    sim = q.double() @ c.double().t()
    want = (sim > sim[torch.arange(Bq), gt][:, None]).sum(1)
    # a candidate within float rounding of the threshold may fall on either side
    margin = (sim - sim[torch.arange(Bq), gt][:, None]).abs()
    margin[torch.arange(Bq), gt] = 1.0
    ok = margin.min(1).values > 1e-6
    assert torch.equal(got[ok].long(), want[ok])
    assert ok.float().mean() > 0.9


def test_zero_shot_matches_topk():
    from dclip_amd import eval as E
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    img = torch.randn(200, 512, generator=gen)
    txt = torch.randn(100, 512, generator=gen)
    labels = torch.randint(0, 100, (200,), generator=gen)
    img[:60] += 3.0 * txt[labels[:60]]
    ranks = E.zero_shot_ranks(img.to(dev), txt.to(dev), labels).cpu()
    sim = 100.0 * torch.nn.functional.normalize(img, dim=1) @ torch.nn.functional.normalize(txt, dim=1).t()
    top5 = sim.topk(5, dim=1).indices
    assert int((ranks == 0).sum()) == int((top5[:, 0] == labels).sum())
    assert int((ranks < 5).sum()) == int((top5 == labels[:, None]).any(1).sum())


def test_evaluate_zero_shot_end_to_end():
    from dclip_amd import eval as E, config as dcfg, synth
    from dclip_amd.clip_model import from_hf_state_dict
    dev = torch.device("cuda:0")
    cfg = dcfg.tiny()
    m = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=7, gain=4.0), device=dev)
    imgs = [torch.rand(6, 3, cfg.vision.image_size, cfg.vision.image_size) for _ in range(2)]
    labels = [torch.randint(0, 10, (6,)) for _ in range(2)]
    res = E.evaluate_zero_shot(m, imgs, labels, synth.synth_input_ids(10, cfg.text, seed=2, ragged=True))
    assert 0.0 <= res["top1"] <= res["top5"] <= 1.0


def test_flickr_eval_flow_base_and_checkpoint(tmp_path):
    """Script-level flow: JSON dataset -> decode -> GPU preprocessing -> embeddings -> metrics, for the base model and
    for a Lightning-layout checkpoint; checked against the metrics computed from a plain similarity matrix."""
    import argparse, json
    import numpy as np
    from PIL import Image
    from dclip_amd import config as dcfg, synth, flickr30k_eval as F, data
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.lightning_lite import save_checkpoint
    dev = torch.device("cuda:0")
    cfg = dcfg.tiny(image_size=64, patch_size=16)
    clip = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=3, gain=3.0), device=dev)
    recs = []
    for i in range(7):
        p = tmp_path / f"im{i}.png"
        Image.fromarray(synth.synth_photo(80 + 7 * i, 100 + 3 * i, seed=i)).save(p)
        recs.append({"image_id": f"id{i}", "image_path": str(p), "captions": [f"w{i} a{j} photo" for j in range(1 + i % 3)]})
    recs.append({"image_id": "nocap", "image_path": str(tmp_path / "im0.png"), "captions": []})
    recs.append({"image_id": "broken", "image_path": str(tmp_path / "missing.png"), "captions": ["x y"]})
    (tmp_path / "test.json").write_text(json.dumps(recs))
    T = cfg.text.max_position_embeddings

    def toy_tokenizer(caps):
        ids = torch.full((len(caps), T), cfg.text.eos_token_id, dtype=torch.int64)
        ids[:, 0] = cfg.text.bos_token_id
        for b, c in enumerate(caps):
            for j, w in enumerate(c.split()[:T - 2]):
                ids[b, 1 + j] = 1 + (sum(ord(ch) * (k + 1) for k, ch in enumerate(w)) % (cfg.text.bos_token_id - 2))
        return ids

    m = F.evaluate_model("base", dev, max_images=100, dataset_json=str(tmp_path / "test.json"), clip_model=clip,
                         tokenizer=toy_tokenizer, batch_size=3)
    # independent check: host preprocessing, one by one, dense similarity matrix + argsort
    pre = data.ClipImagePreprocess(64)
    kept = recs[:7]
    with torch.no_grad():
        img = torch.cat([clip.get_image_features(pixel_values=pre(images=Image.open(r["image_path"]))["pixel_values"].to(dev))
                         for r in kept])
        caps = [c for r in kept for c in r["captions"]]
        cap = clip.get_text_features(input_ids=toy_tokenizer(caps).to(dev))
    owner = [i for i, r in enumerate(kept) for _ in r["captions"]]
    sim = torch.nn.functional.normalize(cap.double(), dim=1) @ torch.nn.functional.normalize(img.double(), dim=1).t()
    t2i_rank = [(sim[c] > sim[c, owner[c]]).sum().item() for c in range(len(caps))]
    assert abs(m["t2i"]["R@1"] - np.mean([r < 1 for r in t2i_rank])) < 1e-12
    assert abs(m["t2i"]["MAP"] - np.mean([1.0 / (r + 1) for r in t2i_rank])) < 1e-9
    # checkpoint flow
    hp = argparse.Namespace(learning_rate=1e-4, warmup_steps=0, total_steps=10, train_batch_size=4, eval_batch_size=4)
    mod = CLIPImageDistillation(hp, clip, None).to(dev)
    with torch.no_grad():
        mod.student.visual_projection.weight.mul_(-1.0)          # make the checkpointed model differ from the base
    ck = save_checkpoint(str(tmp_path / "ck" / "epoch-epoch=00-train_loss=1.00.ckpt"), mod)
    base2 = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=3, gain=3.0), device=dev)
    m2 = F.evaluate_model("custom", dev, max_images=100, dataset_json=str(tmp_path / "test.json"), clip_model=base2,
                          checkpoint=ck, tokenizer=toy_tokenizer, batch_size=4)
    assert set(m2) == {"t2i", "i2t"} and set(m2["t2i"]) == {"R@1", "R@5", "R@10", "MAP"}
    sim2 = -sim                                                   # negated image embeddings flip every similarity
    t2i_rank2 = [(sim2[c] > sim2[c, owner[c]]).sum().item() for c in range(len(caps))]
    assert abs(m2["t2i"]["MAP"] - np.mean([1.0 / (r + 1) for r in t2i_rank2])) < 1e-9


def test_zero_shot_script_flow_base_and_checkpoint(tmp_path):
    """eval_scripts/test_zero_shot_ImageNet.py / CIFAR_zeroshot.py as a flow: class folders -> Resize + CenterCrop +
    ToTensor -> mean/std -> embeddings -> top-1 / top-5, for the base model and a Lightning-layout checkpoint; checked
    against a dense similarity matrix + topk."""
    import argparse
    from PIL import Image
    from dclip_amd import config as dcfg, synth, zero_shot_eval as Z
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.lightning_lite import save_checkpoint
    dev = torch.device("cuda:0")
    cfg = dcfg.tiny(image_size=64, patch_size=16)
    clip = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=3, gain=3.0), device=dev)
    names = ["cat", "dog", "tree", "car", "bird", "boat", "cup"]
    k = 0
    for c in names:
        (tmp_path / "val" / c).mkdir(parents=True)
        for j in range(3):
            Image.fromarray(synth.synth_photo(70 + 5 * k, 90 + 3 * k, seed=k)).save(tmp_path / "val" / c / f"{j}.png")
            k += 1
    (tmp_path / "classes.txt").write_text("\n".join(sorted(names)) + "\n")
    T = cfg.text.max_position_embeddings

    def toy_processor(text=None, return_tensors="pt", padding=True):
        ids = torch.full((len(text), T), cfg.text.eos_token_id, dtype=torch.int64)
        ids[:, 0] = cfg.text.bos_token_id
        for b, c in enumerate(text):
            for j, w in enumerate(c.replace(",", " ").split()[:T - 2]):
                ids[b, 1 + j] = 1 + (sum(ord(ch) * (q + 1) for q, ch in enumerate(w)) % (cfg.text.bos_token_id - 2))
        return {"input_ids": ids}

    for dataset, suffix in (("imagenet", ""), ("cifar10", ", a type of object")):
        res = Z.main(["--data_root", str(tmp_path / "val"), "--classnames", str(tmp_path / "classes.txt"), "--dataset", dataset,
                      "--batch_size", "4", "--results", str(tmp_path / f"{dataset}.txt")], clip_model=clip, processor=toy_processor)
        ds = Z.ImageFolderDataset(str(tmp_path / "val"), 64)
        assert ds.classes == sorted(names) and len(ds) == 21
        x = torch.stack([ds[i][0] for i in range(len(ds))]).to(dev)
        y = torch.tensor([ds[i][1] for i in range(len(ds))])
        mean = torch.tensor(Z.E.CLIP_MEAN, device=dev).view(1, 3, 1, 1)
        std = torch.tensor(Z.E.CLIP_STD, device=dev).view(1, 3, 1, 1)
        with torch.no_grad():
            img = clip.get_image_features(pixel_values=(x - mean) / std).double().cpu()
            txt = clip.get_text_features(input_ids=toy_processor([f"a photo of a {n}{suffix}" for n in sorted(names)])["input_ids"]
                                         .to(dev)).double().cpu()
        sim = 100.0 * torch.nn.functional.normalize(img, dim=1) @ torch.nn.functional.normalize(txt, dim=1).t()
        top5 = sim.topk(5, dim=1).indices
        assert abs(res["base"]["top1"] - float((top5[:, 0] == y).double().mean())) < 1e-12
        assert abs(res["base"]["top5"] - float((top5 == y[:, None]).any(1).double().mean())) < 1e-12
        assert "base Top-1" in (tmp_path / f"{dataset}.txt").read_text()
    hp = argparse.Namespace(learning_rate=1e-4, warmup_steps=0, total_steps=10, train_batch_size=4, eval_batch_size=4)
    mod = CLIPImageDistillation(hp, clip, None).to(dev)
    with torch.no_grad():
        mod.student.visual_projection.weight.mul_(-1.0)
    ck = save_checkpoint(str(tmp_path / "ck" / "epoch-epoch=00-train_loss=1.00.ckpt"), mod)
    base2 = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=3, gain=3.0), device=dev)
    res2 = Z.main(["--data_root", str(tmp_path / "val"), "--checkpoint", ck, "--batch_size", "8", "--results",
                   str(tmp_path / "r2.txt")], clip_model=base2, processor=toy_processor)
    assert set(res2) == {"base", "custom"} and 0.0 <= res2["custom"]["top1"] <= res2["custom"]["top5"] <= 1.0
    assert res2["custom"]["top5"] != res2["base"]["top5"] or res2["custom"]["top1"] != res2["base"]["top1"]
