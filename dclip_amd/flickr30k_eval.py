"""`flickr30k_eval` on the HIP kernels: the script-level flow of eval_scripts/flickr30k_eval.py:90-330.

    evaluate_model(model_name, device, max_images=1000, dataset_json=..., clip_path=... | clip_model=...,
                   checkpoint=..., tokenizer=..., batch_size=...)
        dataset JSON: [{image_id, image_path, captions: [...]}, ...]; items without captions dropped, first
        `max_images` kept (:95-106); images that fail to open are skipped (:160-168)
        "base": the CLIP model itself; "custom": CLIPImageDistillation.load_from_checkpoint(checkpoint, map_location=,
        clip_model=, clip_preprocess=, strict=False) (:126-132) and its forward(image=) / forward(text=)
        -> {"t2i": {"R@1","R@5","R@10","MAP"}, "i2t": {...}}  (same dict as the reference)
    main(argv): --max_images --model {base,custom,both} --checkpoint  (+ --dataset_json --clip_path, which the
        reference hard-codes / downloads by name)

What differs: images are decoded on the host, preprocessed in batches on the GPU (`data.GpuCollate`'s kernel,
bit-exact with the HF processor) and encoded in batches instead of one by one; the caption x image matrix is never
built (eval.calculate_retrieval_metrics counts ranks on MFMA tiles).  Models and tokenizers come from LOCAL paths.
"""
from __future__ import annotations

import argparse
import json
from typing import Callable, Optional

import numpy as np
import torch

from . import eval as E
from . import ops
from .data import ClipImagePreprocess


def _load_dataset(dataset_json: str, max_images: int):
    with open(dataset_json, "r") as f:
        dataset = json.load(f)
    dataset = [it for it in dataset if it.get("captions") and len(it.get("captions")) > 0]
    if 0 < max_images < len(dataset):
        print(f"Limiting evaluation to first {max_images} images")
        dataset = dataset[:max_images]
    print(f"Testing on {len(dataset)} images with captions")
    return dataset


def _pixel_batch(arrs, device, size):
    hmax, wmax = max(a.shape[0] for a in arrs), max(a.shape[1] for a in arrs)
    host = torch.zeros((len(arrs), hmax, wmax, 3), dtype=torch.uint8)
    for b, a in enumerate(arrs):
        host[b, :a.shape[0], :a.shape[1]] = torch.from_numpy(np.ascontiguousarray(a))
    dims = torch.tensor([a.shape[:2] for a in arrs], dtype=torch.int32)
    return ops.clip_preprocess(host.to(device), dims.to(device), size)


@torch.no_grad()
def evaluate_model(model_name: str, device, max_images: int = 1000, dataset_json: Optional[str] = None,
                   clip_model=None, clip_path: Optional[str] = None, checkpoint: Optional[str] = None,
                   tokenizer: Optional[Callable] = None, batch_size: int = 64):
    from PIL import Image
    from .CLIP_image_distillation import CLIPImageDistillation, _as_hip_model
    print(f"\n=== Evaluating {model_name} Model ===")
    if dataset_json is None:
        raise ValueError("dataset_json: path of the Karpathy-format test JSON (the reference hard-codes it, :95)")
    dataset = _load_dataset(dataset_json, max_images)
    if clip_model is None:
        if not clip_path:
            raise ValueError("pass clip_model or clip_path (a LOCAL directory with HF CLIP weights; nothing is fetched)")
        from transformers import CLIPModel, CLIPTokenizer
        clip_model = CLIPModel.from_pretrained(clip_path, local_files_only=True)
        if tokenizer is None:
            tok = CLIPTokenizer.from_pretrained(clip_path, local_files_only=True)
            tokenizer = lambda caps: tok(caps, return_tensors="pt", padding=True, truncation=True, max_length=77)["input_ids"]
    if tokenizer is None:
        raise ValueError("caption strings need a tokenizer: callable list[str] -> LongTensor[B,T]")
    base = _as_hip_model(clip_model).to(device)
    size = base.config.vision.image_size
    if model_name == "custom":
        if not checkpoint:
            raise ValueError("--model custom needs --checkpoint")
        model = CLIPImageDistillation.load_from_checkpoint(checkpoint, map_location=device, clip_model=base,
                                                           clip_preprocess=ClipImagePreprocess(size), strict=False).to(device)
        enc_i, enc_t = (lambda x: model(image=x)), (lambda x: model(text=x))
    else:
        model = base
        enc_i = lambda x: model.get_image_features(pixel_values=x)
        enc_t = lambda x: model.get_text_features(input_ids=x)
    model.eval()

    print("Processing images...")
    image_emb, image_ids, pre = [], [], ClipImagePreprocess(size)
    for i in range(0, len(dataset), batch_size):
        arrs, ids = [], []
        for item in dataset[i:i + batch_size]:
            try:
                with Image.open(item["image_path"]) as im:
                    arrs.append(pre.decode(im))
                ids.append(item["image_id"])
            except Exception as e:
                print(f"Error loading image {item['image_path']}: {e}")
        if arrs:
            image_emb.append(enc_i(_pixel_batch(arrs, device, size)).float())
            image_ids.extend(ids)
    print("Processing captions...")
    caption_emb, caption_image_ids, caps, cap_ids = [], [], [], []
    have = set(image_ids)

    def flush():
        if caps:
            caption_emb.append(enc_t(tokenizer(list(caps)).to(device)).float())
            caption_image_ids.extend(cap_ids)
            caps.clear()
            cap_ids.clear()

    for item in dataset:
        if item["image_id"] not in have:
            continue                      # its image failed to load: a caption without a candidate cannot be ranked
        for caption in item["captions"]:
            caps.append(caption)
            cap_ids.append(item["image_id"])
            if len(caps) >= batch_size:
                flush()
    flush()
    img, cap = torch.cat(image_emb), torch.cat(caption_emb)
    print(f"Computing metrics for {img.shape[0]} images and {cap.shape[0]} captions")
    metrics = E.calculate_retrieval_metrics(img, cap, image_ids, caption_image_ids)
    for title, d in (("Text-to-Image Retrieval", "t2i"), ("Image-to-Text Retrieval", "i2t")):
        print(f"\n--- {title} ---")
        for k in ("R@1", "R@5", "R@10"):
            print(f"Recall@{k[2:]}: {metrics[d][k]:.4f}")
        print(f"MAP: {metrics[d]['MAP']:.4f}")
    return metrics


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Evaluate models on Flickr30K")
    parser.add_argument("--max_images", type=int, default=1000, help="Maximum number of images to evaluate (default: 1000)")
    parser.add_argument("--model", type=str, default="both", choices=["base", "custom", "both"],
                        help="Which model(s) to evaluate (default: both)")
    parser.add_argument("--checkpoint", type=str, default=None, help="Path to custom model checkpoint")
    parser.add_argument("--dataset_json", type=str, required=True, help="flickr30k_test_karpathy.json (local path)")
    parser.add_argument("--clip_path", type=str, required=True, help="local directory with HF CLIP weights + tokenizer")
    return parser


def main(argv=None, **kw):
    args = build_parser().parse_args(argv)
    device = torch.device("cuda")
    print(f"Using device: {device}")
    res = {}
    if args.model in ("base", "both"):
        res["base"] = evaluate_model("base", device, args.max_images, args.dataset_json, clip_path=args.clip_path, **kw)
    if args.model in ("custom", "both"):
        res["custom"] = evaluate_model("custom", device, args.max_images, args.dataset_json, clip_path=args.clip_path,
                                       checkpoint=args.checkpoint, **kw)
    if len(res) == 2:
        b, c = res["base"], res["custom"]
        print("\n=== Model Comparison ===")
        print("                Base CLIP    Custom Model")
        for d, name in (("t2i", "T→I"), ("i2t", "I→T")):
            for k, label in (("R@1", "Recall@1: "), ("R@5", "Recall@5: "), ("R@10", "Recall@10:"), ("MAP", "MAP:      ")):
                print(f"{name} {label}   {b[d][k]:.4f}        {c[d][k]:.4f}")
    return res


if __name__ == "__main__":
    main()
