"""Input side of the step: student preprocessing, box caches, MultiModalDataset.
Golden: tests/golden/data_front.npz, written by oracle/make_golden.py from HF's CLIPImageProcessor and from the
reference's own MultiModalDataset / load_or_compute_yolo (lifted by ast)."""
import hashlib
import json
import os
import pickle
import random

import numpy as np
import pytest
import torch

from dclip_amd import data, synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_front.npz"))


def sha(t):
    a = t.numpy() if isinstance(t, torch.Tensor) else t
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_preprocess_matches_hf_processor_bit_for_bit():
    pre = data.ClipImagePreprocess()
    for i, (h, w) in enumerate(G["sizes"]):
        out = pre(images=synth.synth_photo(int(h), int(w), seed=100 + i), text="", return_tensors="pt")["pixel_values"]
        assert out.shape == (1, 3, 224, 224) and out.dtype == torch.float32
        assert sha(out[0]) == str(G[f"sha_{i}"]), (i, h, w)
        probe = G[f"probe_{i}"]
        assert abs(float(out.double().sum()) - probe[0]) < 1e-6


def test_shortest_edge_size():
    assert data.shortest_edge_size(300, 400, 224) == (224, 298)
    assert data.shortest_edge_size(400, 300, 224) == (298, 224)
    assert data.shortest_edge_size(224, 1000, 224) == (224, 1000)
    assert data.shortest_edge_size(60, 60, 224) == (224, 224)


def _materialise(tmp_path):
    recs = json.loads(str(G["records_json"]))
    cache = json.loads(str(G["cache_json"]))
    from PIL import Image
    for i, (h, w) in enumerate(G["sizes"][:6]):
        Image.fromarray(synth.synth_photo(int(h), int(w), seed=100 + i)).save(tmp_path / f"img_{i}.png")
    for r in recs:
        r["image_path"] = str(tmp_path / r["image_path"])
    jf = tmp_path / "train.json"
    jf.write_text(json.dumps(recs))
    cdir = tmp_path / "cache"
    cdir.mkdir()
    full = {str(tmp_path / k): [(tuple(b[0]), b[1]) for b in v] for k, v in cache.items()}
    with open(cdir / "train_precache.pkl", "wb") as f:
        pickle.dump(full, f, protocol=4)
    return str(jf), str(cdir), full


def test_dataset_matches_reference_dataset(tmp_path):
    jf, cdir, full = _materialise(tmp_path)
    ds = data.MultiModalDataset(jf, data.ClipImagePreprocess(), cache_dir=cdir, cache_filename="train_precache.pkl")
    assert len(ds) == int(G["n_items"])
    random.seed(1234)
    items = [ds[i] for i in range(len(ds))]
    for i, (pv, cap, path, boxes) in enumerate(items):
        assert sha(pv) == str(G[f"item_sha_{i}"]), i
        assert cap == str(G[f"item_caption_{i}"])
        assert os.path.basename(path) == str(G[f"item_path_{i}"])
        assert json.loads(json.dumps(boxes)) == json.loads(str(G[f"item_boxes_{i}"]))
    pvs, caps, paths, bx = data.MultiModalDataset.custom_collate_fn(items[:3])
    assert tuple(pvs.shape) == tuple(G["collate_shape"]) and len(caps) == len(paths) == len(bx) == 3


def test_dataset_all_broken_gives_blank_item(tmp_path):
    jf = tmp_path / "d.json"
    jf.write_text(json.dumps([{"image_path": str(tmp_path / f"no{i}.png"), "caption": "x"} for i in range(4)]))
    ds = data.MultiModalDataset(str(jf), None, cache_dir=str(tmp_path / "c"))
    pv, cap, path, boxes = ds[1]
    assert pv.shape == (3, 224, 224) and float(pv.abs().sum()) == 0 and cap == "" and path == "" and boxes == []


def test_missing_boxes_without_detector_is_an_error_and_detector_fills_cache(tmp_path):
    from PIL import Image
    p = tmp_path / "a.png"
    Image.fromarray(synth.synth_photo(64, 80, 1)).save(p)
    jf = tmp_path / "d.json"
    jf.write_text(json.dumps([{"image_path": str(p), "caption": "x"}]))
    ds = data.MultiModalDataset(str(jf), None, cache_dir=str(tmp_path / "c1"))
    with pytest.raises(RuntimeError, match="no cached boxes"):
        ds[0]
    calls = []

    def det(path):
        calls.append(path)
        return [((1, 2, 30, 40), np.float32(0.5))]

    ds = data.MultiModalDataset(str(jf), None, cache_dir=str(tmp_path / "c2"), detector=det)
    assert ds[0][3] == [((1, 2, 30, 40), 0.5)] and calls == [str(p)]
    # strategy 2 wrote the batch cache; a second dataset needs no detector
    ds2 = data.MultiModalDataset(str(jf), None, cache_dir=str(tmp_path / "c2"))
    assert ds2[0][3] == [((1, 2, 30, 40), 0.5)]
    # per-image cache layout
    ds3 = data.MultiModalDataset(str(jf), None, cache_dir=str(tmp_path / "c3"), use_batch_cache=False, detector=det)
    assert ds3[0][3] == [((1, 2, 30, 40), 0.5)]
    assert os.path.exists(tmp_path / "c3" / "a.png.pkl")
    assert data.load_or_compute_yolo(str(p), None, str(tmp_path / "c3")) == [((1, 2, 30, 40), 0.5)]


def test_big_cache_goes_through_dbm(tmp_path):
    cache = {f"/img/{i}.jpg": [((i, i + 1, i + 50, i + 60), 0.25 + i / 100)] for i in range(40)}
    with open(tmp_path / "big.pkl", "wb") as f:
        pickle.dump(cache, f, protocol=4)
    c = data.open_box_cache(str(tmp_path), "big.pkl", big_bytes=16)           # forces the conversion branch
    assert isinstance(c, data.DiskCache) and len(c) == 40
    assert c.get("/img/7.jpg") == cache["/img/7.jpg"] and "/img/7.jpg" in c
    assert c.get("/img/nope.jpg") is None and c.get("/img/nope.jpg", []) == []
    assert os.path.exists(tmp_path / "big_keys.pkl")
    c["/img/new.jpg"] = []
    assert c.get("/img/new.jpg", None) == [] and len(c) == 41
    c2 = data.open_box_cache(str(tmp_path), "big.pkl", big_bytes=16)          # second open: reuses the database
    assert c2.get("/img/39.jpg") == cache["/img/39.jpg"]


class _Boom:
    def __reduce__(self):
        return (os.system, ("echo pwned > /tmp/dclip_pwned",))


def test_cache_that_names_globals_is_rejected_not_executed(tmp_path):
    with open(tmp_path / "evil.pkl", "wb") as f:
        pickle.dump({"a": _Boom()}, f)
    if os.path.exists("/tmp/dclip_pwned"):
        os.remove("/tmp/dclip_pwned")
    assert data.open_box_cache(str(tmp_path), "evil.pkl") == {}
    assert not os.path.exists("/tmp/dclip_pwned")
    with pytest.raises(pickle.UnpicklingError):
        data.plain_load(str(tmp_path / "evil.pkl"))


def test_module_dataloaders(tmp_path):
    import argparse
    from dclip_amd import config as dcfg
    from dclip_amd.CLIP_image_distillation import CLIPImageDistillation
    from dclip_amd.clip_model import HipCLIPModel
    jf, cdir, full = _materialise(tmp_path)
    hp = argparse.Namespace(train_file=jf, val_file=jf, eval_batch_size=3, train_batch_size=2, learning_rate=1e-5,
                            warmup_steps=0, total_steps=10, cache_dir=cdir, val_cache_filename="train_precache.pkl")
    m = CLIPImageDistillation(hp, HipCLIPModel(dcfg.tiny()), data.ClipImagePreprocess())
    pv, caps, paths, boxes = next(iter(m.val_dataloader()))
    assert pv.shape == (3, 3, 224, 224) and len(caps) == 3 and boxes[0] == full[paths[0]]
    assert len(list(m.train_dataloader())) == 3                                # 7 items, batch 3 (eval_batch_size, N3)


def test_zero_shot_image_folder_transform(tmp_path):
    """ImageFolderDataset = torchvision ImageFolder + Resize(S) + CenterCrop(S) + ToTensor()
    (eval_scripts/test_zero_shot_ImageNet.py:141-151): shortest edge to S (long edge int(S*long/short)), bilinear,
    centred window, [0,1] floats, sorted class folders."""
    import numpy as np
    from PIL import Image
    from dclip_amd import synth
    from dclip_amd.zero_shot_eval import ImageFolderDataset, make_prompts
    for c, (h, w) in (("b_dog", (50, 80)), ("a_cat", (90, 60))):
        (tmp_path / c).mkdir()
        Image.fromarray(synth.synth_photo(h, w, seed=h)).save(tmp_path / c / "0.png")
    ds = ImageFolderDataset(str(tmp_path), 32)
    assert ds.classes == ["a_cat", "b_dog"] and len(ds) == 2
    x, y = ds[0]
    assert y == 0 and tuple(x.shape) == (3, 32, 32) and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    im = Image.open(tmp_path / "a_cat" / "0.png").convert("RGB")            # 60 wide, 90 high -> 32 x 48 -> rows 8..40
    want = np.asarray(im.resize((32, 48), Image.BILINEAR).crop((0, 8, 32, 40)), dtype=np.float32) / 255.0
    assert np.array_equal(x.permute(1, 2, 0).numpy(), want)
    assert make_prompts(["cat"]) == ["a photo of a cat"]
    assert make_prompts(["cat"], "CIFAR-10") == ["a photo of a cat, a type of object"]
