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
