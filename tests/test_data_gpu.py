"""Student preprocessing on the GPU (bit-exact with HF CLIPImageProcessor / PIL) and the GpuCollate batch path."""
import hashlib
import json
import os
import pickle

import numpy as np
import pytest
import torch

from dclip_amd import config as dcfg, data, synth

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_front.npz"))


def _batch(sizes, seed0=100):
    arrs = [synth.synth_photo(int(h), int(w), seed=seed0 + i) for i, (h, w) in enumerate(sizes)]
    hmax, wmax = max(a.shape[0] for a in arrs), max(a.shape[1] for a in arrs)
    host = torch.zeros((len(arrs), hmax, wmax, 3), dtype=torch.uint8)
    for b, a in enumerate(arrs):
        host[b, :a.shape[0], :a.shape[1]] = torch.from_numpy(a)
    return arrs, host, torch.tensor([a.shape[:2] for a in arrs], dtype=torch.int32)


def test_clip_preprocess_bit_exact_with_golden_and_host():
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    arrs, host, dims = _batch(G["sizes"])
    out = ops.clip_preprocess(host.to(dev), dims.to(dev)).cpu()
    pre = data.ClipImagePreprocess()
    for i, a in enumerate(arrs):
        sha = hashlib.sha256(np.ascontiguousarray(out[i].numpy()).tobytes()).hexdigest()
        assert torch.equal(out[i], pre.image(a)), f"image {i} {a.shape}: max diff {(out[i] - pre.image(a)).abs().max()}"
        assert sha == str(G[f"sha_{i}"]), i


def test_clip_preprocess_large_downscale_and_other_size():
    from dclip_amd import ops
    dev = torch.device("cuda:0")
    arrs, host, dims = _batch([(1500, 2000), (2048, 700), (31, 999)], seed0=300)
    pre = data.ClipImagePreprocess(size=96)
    out = ops.clip_preprocess(host.to(dev), dims.to(dev), 96).cpu()
    for i, a in enumerate(arrs):
        assert torch.equal(out[i], pre.image(a)), i


def test_gpu_collate_step_matches_host_path(tmp_path):
    """decode_only dataset + GpuCollate -> dict batch; the teacher embedding cut from the uploaded images equals the
    path-based one, and pixel_values equal the host dataset's."""
    from PIL import Image
    from dclip_amd.clip_model import from_hf_state_dict
    from dclip_amd.patch_text_aggregation import PatchTextAggregation
    dev = torch.device("cuda:0")
    recs, cache = [], {}
    rs = np.random.RandomState(3)
    for i, (h, w) in enumerate([(300, 400), (256, 256), (180, 333), (401, 299)]):
        p = tmp_path / f"i{i}.png"
        Image.fromarray(synth.synth_photo(h, w, seed=500 + i)).save(p)
        recs.append({"image_path": str(p), "caption": f"c{i}"})
        boxes = []
        for _ in range(int(rs.randint(0, 4))):
            x1, y1 = int(rs.randint(0, w - 20)), int(rs.randint(0, h - 20))
            boxes.append(((x1, y1, int(rs.randint(x1 + 8, w + 1)), int(rs.randint(y1 + 8, h + 1))), float(rs.rand())))
        cache[str(p)] = boxes
    (tmp_path / "d.json").write_text(json.dumps(recs))
    (tmp_path / "c").mkdir()
    with open(tmp_path / "c" / "train_precache.pkl", "wb") as f:
        pickle.dump(cache, f, protocol=4)
    ds_host = data.MultiModalDataset(str(tmp_path / "d.json"), None, cache_dir=str(tmp_path / "c"))
    ds_dev = data.MultiModalDataset(str(tmp_path / "d.json"), None, cache_dir=str(tmp_path / "c"), decode_only=True)
    hb = data.MultiModalDataset.custom_collate_fn([ds_host[i] for i in range(4)])
    gb = data.GpuCollate(dev)([ds_dev[i] for i in range(4)])
    assert torch.equal(gb["pixel_values"].cpu(), hb[0])
    assert gb["captions"] == hb[1] and gb["image_paths"] == hb[2] and gb["weighted_boxes"] == hb[3]

    cfg = dcfg.tiny(image_size=64, patch_size=16)
    clip = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0), device=dev)
    E = cfg.projection_dim
    t = PatchTextAggregation(embed_dim=E, num_heads=max(1, E // 64), clip_model=clip).to(dev)
    t.cross_modal_attention.load_state_dict(synth.synth_cross_modal_state_dict(E, seed=5))
    ids = synth.synth_input_ids(4, cfg.text, seed=3, ragged=True).to(dev)
    with torch.no_grad():
        a = t.compute_global_embedding_batch(hb[2], ids, hb[3])
        b = t.compute_global_embedding_batch(gb["image_paths"], ids, gb["weighted_boxes"], gb["images_u8"], gb["dims"])
    assert torch.equal(a, b)
