"""GPU crop + resize + ToTensor is bit-exact with the PIL path the reference uses (training/image_tokenizer.py:28-32,
:100-110): upscaling, downscaling, extreme aspect ratios, boxes reaching outside the image."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def pil_path(img, box, s):
    from PIL import Image
    arr = np.asarray(img.crop(box).resize((s, s), Image.BILINEAR), dtype=np.uint8)
    return torch.from_numpy(arr).permute(2, 0, 1).float().div(255)          # T.ToTensor()


@pytest.mark.parametrize("S", [224, 64])
def test_crop_resize_bit_exact_with_pillow(S):
    from PIL import Image
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    sizes = [(480, 640), (333, 500), (37, 53), (1200, 900)]
    imgs = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in sizes]
    boxes = []
    for b, (h, w) in enumerate(sizes):
        boxes += [(b, 0, 0, w, h), (b, 3, 5, min(w, 3 + 17), min(h, 5 + 11)), (b, w // 4, h // 5, w - 1, h - 2),
                  (b, w // 2, 0, w // 2 + 1, h), (b, 0, h // 2, w, h // 2 + 2), (b, -7, -3, w // 2, h // 2),
                  (b, w - 20, h - 10, w + 15, h + 9)]
    for _ in range(20):
        b = int(rng.integers(0, len(sizes)))
        h, w = sizes[b]
        x1, y1 = int(rng.integers(0, w - 2)), int(rng.integers(0, h - 2))
        boxes.append((b, x1, y1, int(rng.integers(x1 + 1, w + 1)), int(rng.integers(y1 + 1, h + 1))))
    hmax, wmax = max(h for h, _ in sizes), max(w for _, w in sizes)
    batch = np.zeros((len(sizes), hmax, wmax, 3), dtype=np.uint8)
    for b, im in enumerate(imgs):
        a = np.asarray(im)
        batch[b, :a.shape[0], :a.shape[1]] = a
    bx = torch.tensor(boxes, dtype=torch.int32)
    out = ops.crop_resize(torch.from_numpy(batch).to(dev), torch.tensor(sizes, dtype=torch.int32).to(dev), bx.to(dev), S,
                          int((bx[:, 4] - bx[:, 2]).max()), int((bx[:, 3] - bx[:, 1]).max())).cpu()
    for r, (b, x1, y1, x2, y2) in enumerate(boxes):
        want = pil_path(imgs[b], (x1, y1, x2, y2), S)
        assert torch.equal(out[r], want), (r, boxes[r], float((out[r] - want).abs().max()))


def test_tokenizer_gpu_crops_equal_host_crops():
    from PIL import Image
    from dclip_amd import config as dcfg, synth
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.patch_text_aggregation import CLIPPatchTokenizer
    dev = torch.device("cuda:0")
    cfg = dcfg.tiny()
    tok = CLIPPatchTokenizer(from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=7), device=dev))
    rng = np.random.default_rng(1)
    imgs = [Image.fromarray(rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)) for _ in range(3)]
    boxes = [[((4, 2, 60, 50), 0.9), ((10, 20, 110, 85), 0.8)], [], [((0, 0, 120, 90), 0.7)]]
    regions, counts = tok.crop_boxes_gpu(imgs, boxes)
    assert counts.tolist() == [2, 0, 1] and regions.shape[:2] == (3, 2)
    for b, bl in enumerate(boxes):
        for r, (box, _) in enumerate(bl):
            assert torch.equal(regions[b, r].cpu(), tok.patch_transform(imgs[b].crop(box)))
    assert float(regions[1].abs().sum()) == 0.0


def test_degenerate_box_takes_the_reference_fallback():
    """A zero-width / zero-height box (YOLO coordinates are int-truncated) makes PIL raise inside the reference's
    encode_weighted_bounding_boxes; the reference catches it for the WHOLE image and uses the single zero patch row
    (training/patch_text_aggregation.py:479-491).  No exception here either: that image gets count 0."""
    from PIL import Image
    from dclip_amd import config as dcfg, synth
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    dev = torch.device("cuda:0")
    cfg = dcfg.tiny()
    clip = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=7, gain=4.0), device=dev)
    teacher = PatchTextAggregation(embed_dim=cfg.projection_dim, num_heads=1, clip_model=clip).to(dev)
    rng = np.random.default_rng(2)
    imgs = [Image.fromarray(rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)) for _ in range(3)]
    good = ((4, 2, 60, 50), 0.9)
    bad_boxes = [[good, ((30, 10, 30, 40), 0.8)], [good], [((5, 9, 50, 9), 0.5)]]       # zero width; fine; zero height
    ref_boxes = [[], [good], []]                                                        # what the fallback amounts to
    regions, counts = teacher.patch_tokenizer.crop_boxes_gpu(imgs, bad_boxes)
    assert counts.tolist() == [0, 1, 0]
    regions2, counts2 = teacher.patch_tokenizer.crop_boxes_gpu(imgs, ref_boxes)
    assert counts2.tolist() == [0, 1, 0] and torch.equal(regions[:, :1], regions2[:, :1])
    ids = synth.synth_input_ids(3, cfg.text, seed=3, ragged=True, min_len=4).to(dev)
    with torch.no_grad():
        a = teacher.compute_global_embedding_tensors(regions, ids, counts)
        b = teacher.compute_global_embedding_tensors(regions2, ids, counts2)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
